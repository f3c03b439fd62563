"""CPU tests of the sequence-parallel plumbing (pyflow_hip/sp.py): partition tables and the uneven all-to-all over
gloo (world size 2 and 3), against a single-process restatement of what each exchange must deliver."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_layout_tables():
    from pyflow_hip.sp import SPLayout, even_split
    assert even_split(30, 8) == [4, 4, 4, 4, 4, 4, 3, 3]
    assert even_split(30, 4) == [8, 8, 7, 7]
    L, Lt, H, P = 15488, 128, 30, 8
    lays = [SPLayout(L, Lt, H, P, r) for r in range(P)]
    assert sum(l.nloc for l in lays) == L and sum(l.my_heads for l in lays) == H
    assert [l.r0 for l in lays] == [1936 * r for r in range(P)]
    assert lays[0].n_txt == 128 and lays[0].n_img == 1936 - 128 and lays[1].n_txt == 0
    assert lays[1].img0 == 1936 - 128
    B = 2
    for r in range(P):
        s1, r1 = lays[r].a2a1_splits(B)
        for p in range(P):       # what r sends to p is what p expects from r
            assert s1[p] == lays[p].a2a1_splits(B)[1][r]
        s2, r2 = lays[r].a2a2_splits(B)
        for p in range(P):
            assert s2[p] == lays[p].a2a2_splits(B)[1][r]
    # text rows spread over several ranks when the chunk is shorter than the text
    small = [SPLayout(300, 128, 4, 3, r) for r in range(3)]
    assert [l.n_txt for l in small] == [100, 28, 0] and [l.n_img for l in small] == [0, 72, 100]


def _worker(rank, world, port, H, L, Lt, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyflow_hip.sp import SPLayout, init_sequence_parallel_group
    comm = init_sequence_parallel_group(sp_group_size=world)
    lay = SPLayout(L, Lt, H, world, rank)
    # global "qkv" value of (row, b, head, col) -- every rank can evaluate any entry
    def val(rows, heads, cols):
        r = torch.arange(rows[0], rows[1])[:, None, None, None].float()
        b = torch.arange(B)[None, :, None, None].float()
        h = torch.arange(heads[0], heads[1])[None, None, :, None].float()
        c = torch.arange(cols)[None, None, None, :].float()
        return r * 1000 + b * 500 + h * 10 + c * 0.01
    # exchange 1: send chunks [dest][my rows][b][dest heads x 192]
    send = torch.cat([val((lay.r0, lay.r1), (lay.head0[p], lay.head0[p] + lay.heads[p]), 192).reshape(-1) for p in range(world)])
    s_spl, r_spl = lay.a2a1_splits(B)
    recv = torch.empty(sum(r_spl))
    comm.all_to_all(recv, send, r_spl, s_spl)
    exp = val((0, L), (lay.head0[rank], lay.head0[rank] + lay.my_heads), 192).reshape(-1)
    ok1 = torch.equal(recv, exp)
    # exchange 2: send [dest rows][b][my heads x 64]; receive [src][my rows][b][src heads x 64]
    send2 = val((0, L), (lay.head0[rank], lay.head0[rank] + lay.my_heads), 64).reshape(-1)
    s_spl, r_spl = lay.a2a2_splits(B)
    recv2 = torch.empty(sum(r_spl))
    comm.all_to_all(recv2, send2, r_spl, s_spl)
    exp2 = torch.cat([val((lay.r0, lay.r1), (lay.head0[p], lay.head0[p] + lay.heads[p]), 64).reshape(-1) for p in range(world)])
    ok2 = torch.equal(recv2, exp2)
    t = torch.full((4,), float(rank + 1))
    comm.all_reduce(t)
    ok3 = bool((t == sum(range(1, world + 1))).all())
    bc = torch.full((3,), float(rank))
    comm.broadcast(bc, 0)
    ok4 = bool((bc == 0).all())
    comm.selftest("cpu")           # what bench.py runs first at N > 1: every collective once with known values + p2p set-up
    q.put((rank, ok1, ok2, ok3, ok4))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,L", [(2, 3, 37), (3, 4, 50)])
def test_uneven_all_to_all_over_gloo(world, H, L):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, L, 16, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(all(r[1:]) for r in res), res


def _cp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyflow_hip import cp
    cp.initialize_context_parallel(world)
    T, k = 1 + 4 * world, 1
    full = torch.arange(2 * 3 * T * 2 * 2, dtype=torch.float32).reshape(2, 3, T, 2, 2)
    loc = cp.conv_scatter_to_context_parallel_region(full, 2, k)
    n = (T - k) // world
    exp = full[:, :, : n + k] if rank == 0 else full[:, :, rank * n + k:(rank + 1) * n + k]
    ok1 = torch.equal(loc, exp)
    halo = cp.cp_pass_from_previous_rank(loc, 2, 3)
    if rank == 0:
        exp_h = torch.cat([torch.zeros_like(loc[:, :, :2]), loc], dim=2)
    else:
        lo = rank * n + k
        exp_h = full[:, :, lo - 2:(rank + 1) * n + k]
    ok2 = torch.equal(halo, exp_h)
    back = cp.conv_gather_from_context_parallel_region(loc, 2, k)
    ok3 = torch.equal(back, full)
    q.put((rank, ok1, ok2, ok3))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_context_parallel_helpers_over_gloo(world):
    """split / halo pass / gather of video_vae/context_parallel_ops.py (rank 0 sees [0,0,x...],
    rank r sees [prev[-2:], x...])."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(all(r[1:]) for r in res), res


def _pair_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyflow_hip.sp import init_sequence_parallel_group, is_guidance_parallel, pair_gather
    comm = init_sequence_parallel_group(sp_group_size=world, guidance_parallel=True)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(1, 37, 64, generator=g)
    mine[0, 0, 0] = -0.0                                   # the sum with the other rank's zero must keep every value's bits ...
    mine[0, 0, 1] = 1e-42                                  # ... (a denormal; -0.0 + 0.0 = +0.0 compares equal)
    out = torch.full((2 * mine.numel(),), float("nan"))    # stale contents of the reused buffer
    both = pair_gather(comm, mine, out).view(2, 37, 64)
    exp = torch.stack([torch.randn(1, 37, 64, generator=torch.Generator().manual_seed(100 + r))[0] for r in range(2)])
    exp[:, 0, 0] = 0.0
    exp[:, 0, 1] = 1e-42
    q.put((rank, is_guidance_parallel(), torch.equal(both, exp), both.data_ptr() == out.data_ptr()))
    dist.barrier()
    dist.destroy_process_group()


def test_guidance_pair_gather_over_gloo():
    """the one exchange of the guidance-parallel engine (flux_cfg.py: FluxEngineCFG.forward_tokens): each rank's velocity
    tokens into a replicated [2, n_cur, 64] buffer -- exact, identical on both ranks, in place in the caller's buffer"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, True, True), (1, True, True, True)], res


def test_segmented_program_structure_on_cpu():
    """flux_sp._SegmentedProgram (round 5): what the sequence-parallel engine records when its communicator's calls cannot go
    into a launch list (torch.distributed) -- launch-list segments between the collectives, the collectives and the waits on
    their handles as closures in issue order.  Recording launches nothing, so the structure is checkable without a GPU:
    the library's list entries are appended from CPU tensors' addresses; replay is driven with the lists' run() stubbed."""
    from pyflow_hip import ops
    from pyflow_hip.flux_sp import _RecordingComm, _SegmentedProgram

    class FakeHandle:
        def __init__(self, log, i):
            self.log, self.i = log, i

        def wait(self):
            self.log.append(("waited", self.i))

    class FakeComm:
        rank, world = 1, 3

        def __init__(self):
            self.log = []

        def all_to_all(self, recv, send, recv_splits, send_splits, async_op=False):
            self.log.append(("a2a", tuple(recv_splits), tuple(send_splits), async_op))
            return FakeHandle(self.log, len(self.log)) if async_op else None

    real = FakeComm()
    prog = _SegmentedProgram()
    comm = _RecordingComm(real, prog)
    assert (comm.rank, comm.world) == (1, 3) and not comm.recordable
    a = torch.zeros(64, dtype=torch.bfloat16)
    b = torch.zeros(64, dtype=torch.bfloat16)
    prog.begin()
    try:
        assert ops.RECORDER is prog.cur
        ops.copy_rows(a, b, 1, 64, 64, 64, 0, 0, 1)                   # segment 0: one entry
        h1 = comm.all_to_all(b, a, [8, 8, 8], [8, 8, 8], async_op=True)      # nothing is issued while recording
        ops.copy_rows(a, b, 1, 64, 64, 64, 0, 0, 1)                   # segment 1: work that overlaps the exchange
        ops.copy_rows(a, b, 1, 64, 64, 64, 0, 0, 1)
        h1.wait()
        h2 = comm.all_to_all(b, a, [4, 4, 4], [4, 4, 4], async_op=True)      # two collectives back to back: no empty list between
        h2.wait()
        ops.copy_rows(a, b, 1, 64, 64, 64, 0, 0, 1)                   # segment 2
    finally:
        prog.end()
    assert ops.RECORDER is None and real.log == []
    kinds = [e[0] for e in prog.entries]
    assert kinds == ["list", "call", "list", "wait", "call", "wait", "list"], kinds
    assert [len(e[1]) for e in prog.entries if e[0] == "list"] == [1, 2, 1]
    ran = []
    for e in prog.entries:
        if e[0] == "list":
            e[1].run = (lambda main=None, side=None, n=len(e[1]): ran.append(("list", n)))
    for _ in range(2):                                                # replays issue the same sequence through the real communicator
        ran.clear()
        real.log.clear()
        prog.run(None)
        assert ran == [("list", 1), ("list", 2), ("list", 1)]
        assert [x[0] for x in real.log] == ["a2a", "waited", "a2a", "waited"]
        assert real.log[0][1:] == ((8, 8, 8), (8, 8, 8), True) and real.log[2][1] == (4, 4, 4)


def _uid_worker(rank, world, port, q, fail):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyflow_hip import comm_native
    from pyflow_hip import lib as L
    if fail and rank == 0:        # rank 0 cannot create the id (what a missing librccl gives): the library call is made to fail
        lib = L.load()

        class _Failing:
            def __getattr__(self, name):
                return getattr(lib, name)

            @staticmethod
            def pf_comm_unique_id(buf):
                return 1
        comm_native.L = type("_L", (), {"load": staticmethod(lambda: _Failing())})
    try:
        uid = comm_native.exchange_unique_id(rank, world)
        q.put((rank, "ok", uid))
    except Exception as e:              # noqa: BLE001
        q.put((rank, "raised", type(e).__name__))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
def test_unique_id_exchange_over_gloo(fail):
    """comm_native.exchange_unique_id (the one use of torch.distributed by the C-ABI communicator): every rank ends with
    rank 0's 128 bytes -- and when rank 0 cannot create an id, EVERY rank raises (bench.py then agrees on the torch.distributed
    fallback) instead of the others waiting in the broadcast for the process group's timeout"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uid_worker, args=(r, 2, port, q, fail)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    if fail:
        assert [(r, what) for r, what, _ in res] == [(0, "raised"), (1, "raised")], res
    else:
        assert [what for _, what, _ in res] == ["ok", "ok"] and res[0][2] == res[1][2] and len(res[0][2]) == 128, res
        assert any(res[0][2])


def _selftest_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyflow_hip import sp
    comm = sp.init_sequence_parallel_group(sp_group_size=world)
    issued = []
    for name in ("all_to_all", "all_reduce", "broadcast", "warm_p2p"):
        real = getattr(comm, name)

        def spy(*a, _real=real, _name=name, **k):
            issued.append(_name)
            out = _real(*a, **k)
            if _name == "all_to_all" and rank == 1:
                a[0].zero_()                    # rank 1's transport "delivers" wrong data
            return out
        setattr(comm, name, spy)
    try:
        comm.selftest("cpu")
        q.put((rank, "ok", issued))
    except RuntimeError as e:
        q.put((rank, str(e), issued))
    dist.barrier()
    dist.destroy_process_group()


def test_selftest_issues_every_collective_before_raising():
    """SPComm.selftest (bench.py's first call at N > 1): a rank that sees wrong data still issues EVERY collective of the
    round before it raises -- leaving at the first mismatch would park its peers inside the next collective until the
    process group's timeout (round-5 advisor finding: a duplicated class definition had shadowed this behaviour)."""
    import inspect
    from pyflow_hip import sp
    assert inspect.getsource(sp).count("class SPComm") == 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_selftest_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    order = ["all_to_all", "all_reduce", "broadcast", "warm_p2p"]
    assert res[0] == (0, "ok", order), res
    assert res[1][0] == 1 and "all_to_all" in res[1][1] and res[1][2] == order, res
