"""Sequence parallelism for the DiT (Ulysses-style): process-group plumbing and the partition tables.

Replaces trainer_misc/sp_utils.py:14-98 (group bookkeeping) and trainer_misc/communicate.py:7-66 (the list-form
all_to_all built from tensor_split + contiguous copies + cat) of the reference.  Differences by design
(SURVEY 2.3 / 8e): ONE token layout for all 24 blocks (contiguous row chunks of the merged [text | image]
sequence per rank -- the reference's batch-scatter re-partition between double and single blocks, which limits it
to P = 2, does not exist), an UNEVEN head map (H = 30 heads over 4 or 8 ranks), exchange buffers laid out so that
each all-to-all is a single `all_to_all_single` with split sizes and needs no cat / split copies on the receiving
side, and no replicated-table exchanges (C4-C6: every rank derives RoPE tables and the implicit mask locally).

torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the GPU box.  For tests the same code runs
over "gloo" (CPU tensors directly, device tensors staged through the host) because gloo has no all_to_all.
"""
import torch
import torch.distributed as dist


def even_split(n, parts):
    """n items over `parts` ranks, sizes differing by at most one (first n % parts ranks get the extra item)."""
    base, rem = divmod(n, parts)
    return [base + (1 if p < rem else 0) for p in range(parts)]


def starts_of(counts):
    out, s = [], 0
    for c in counts:
        out.append(s)
        s += c
    return out


class SPLayout:
    """Partition of one stage sequence (L rows, first Lt text) and of the H heads over P ranks."""

    HEAD_COLS = 192          # head-major exchange layout: per head [k(64) | v(64) | q(64)]

    def __init__(self, L, Lt, H, P, rank):
        self.L, self.Lt, self.H, self.P, self.rank = L, Lt, H, P, rank
        self.rows = even_split(L, P)
        self.row0 = starts_of(self.rows)
        self.heads = even_split(H, P)
        self.head0 = starts_of(self.heads)
        self.r0 = self.row0[rank]
        self.nloc = self.rows[rank]
        self.r1 = self.r0 + self.nloc
        self.n_txt = min(max(Lt - self.r0, 0), self.nloc)      # local text rows come first
        self.n_img = self.nloc - self.n_txt
        self.img0 = max(self.r0, Lt) - Lt                       # first local image token (index into the clip-concatenated tokens)
        self.my_heads = self.heads[rank]
        self.my_cols = self.my_heads * self.HEAD_COLS

    # element counts of the two exchanges, for a CFG batch of B
    def a2a1_splits(self, B):
        """qkv exchange: I send my rows x the peer's heads; I receive the peer's rows x my heads."""
        send = [self.nloc * B * h * self.HEAD_COLS for h in self.heads]
        recv = [r * B * self.my_cols for r in self.rows]
        return send, recv

    def a2a2_splits(self, B):
        """attention-output exchange: I send the peer's rows x my heads; I receive my rows x the peer's heads."""
        send = [r * B * self.my_heads * 64 for r in self.rows]
        recv = [self.nloc * B * h * 64 for h in self.heads]
        return send, recv


class LocalComm:
    """world of one: the exchanges are plain device copies (used to test the SP code path on a single GPU)."""
    rank, world = 0, 1
    recordable = True          # its one collective is a recordable copy kernel (launch lists, pyflow_hip/cmdlist.py)

    def all_to_all(self, recv, send, recv_splits, send_splits, async_op=False):
        n = sum(send_splits)
        if n == 0:
            return None
        if send.is_cuda and send.dtype == torch.bfloat16 and n % 8 == 0:
            from . import ops
            ops.copy_rows(send, recv, 1, n, n, n, 0, 0, 1)          # pf_copy_rows: recordable, stream-ordered
        else:
            recv[:n].copy_(send[:n])
        return None

    def all_reduce(self, t):
        return t

    def broadcast(self, t, src=0):
        return t

    def barrier(self):
        pass

    def send(self, t, dst):
        raise RuntimeError("LocalComm has no peers")

    def shift(self, send_t, recv_t):
        """pass `send_t` to rank+1, receive rank-1's into `recv_t` (no wrap-around): nothing to do on one rank"""
        return

    def recv(self, t, src):
        raise RuntimeError("LocalComm has no peers")


class SPComm:
    """all-to-all / all-reduce / broadcast over one torch.distributed group (the SP group = consecutive ranks,
    sp_utils.py:42-47; inference uses world_size == sp_group_size, inference_multigpu.py:36)."""

    def __init__(self, group=None):
        assert dist.is_initialized(), "init the process group first (trainer_misc/utils.py:71-106 contract: env://)"
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.native = self.backend == "nccl"          # RCCL: device-side all_to_all_single with split sizes
        self.recordable = False                       # torch.distributed calls cannot be recorded into a launch list

    def _global(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def all_to_all(self, recv, send, recv_splits, send_splits, async_op=False):
        """recv / send: flat 1-D tensors; splits in elements, indexed by group rank.
        async_op (RCCL only): the exchange runs on the communicator's own HIP stream, ordered after the work already
        queued on the current stream; kernels launched next overlap with it until `.wait()` of the returned handle
        (which makes the current stream wait, not the host).  Returns None when the exchange completed inline."""
        if self.native:
            return dist.all_to_all_single(recv[:sum(recv_splits)], send[:sum(send_splits)], recv_splits, send_splits,
                                          group=self.group, async_op=async_op) if async_op else \
                dist.all_to_all_single(recv[:sum(recv_splits)], send[:sum(send_splits)], recv_splits, send_splits,
                                       group=self.group)
        # gloo: no all_to_all; emulate with point-to-point (device tensors staged through the host)
        dev = send.device
        hs = send[:sum(send_splits)].cpu() if dev.type != "cpu" else send
        hr = torch.empty(sum(recv_splits), dtype=recv.dtype)
        so, ro = starts_of(send_splits), starts_of(recv_splits)
        ops = []
        for p in range(self.world):
            if p == self.rank:
                hr[ro[p]:ro[p] + recv_splits[p]].copy_(hs[so[p]:so[p] + send_splits[p]])
                continue
            if send_splits[p]:
                ops.append(dist.P2POp(dist.isend, hs[so[p]:so[p] + send_splits[p]], self._global(p), self.group))
            if recv_splits[p]:
                ops.append(dist.P2POp(dist.irecv, hr[ro[p]:ro[p] + recv_splits[p]], self._global(p), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        recv[:hr.numel()].copy_(hr)

    def all_reduce(self, t):
        if self.native or t.device.type == "cpu":
            dist.all_reduce(t, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        return t

    def broadcast(self, t, src=0):
        if self.native or t.device.type == "cpu":
            dist.broadcast(t, self._global(src), group=self.group)
        else:
            h = t.cpu()
            dist.broadcast(h, self._global(src), group=self.group)
            t.copy_(h)
        return t

    def barrier(self):
        dist.barrier(group=self.group)

    def send(self, t, dst):
        """point-to-point (halo / strip exchange); stream-ordered on RCCL, blocking on gloo"""
        if self.native or t.device.type == "cpu":
            dist.send(t.contiguous(), self._global(dst), group=self.group)
        else:
            dist.send(t.contiguous().cpu(), self._global(dst), group=self.group)

    def shift(self, send_t, recv_t):
        """halo pass of the temporal context parallelism (video_vae/context_parallel_ops.py:76-114): send `send_t` to
        rank+1 and receive rank-1's tensor into `recv_t`; no wrap-around (rank 0 receives nothing, the last rank sends
        nothing).  One grouped isend/irecv pair, so neither side can block the other."""
        staged = not (self.native or send_t.device.type == "cpu")
        ops, hr = [], None
        if self.rank + 1 < self.world:
            s_ = send_t.contiguous()
            ops.append(dist.P2POp(dist.isend, s_.cpu() if staged else s_, self._global(self.rank + 1), self.group))
        if self.rank > 0:
            hr = torch.empty(recv_t.shape, dtype=recv_t.dtype) if staged else recv_t
            ops.append(dist.P2POp(dist.irecv, hr, self._global(self.rank - 1), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged and hr is not None:
            recv_t.copy_(hr)

    def shift_start(self, send_t, recv_t):
        """`shift` without waiting (RCCL only): returns a handle whose wait() makes the CURRENT stream wait for the received
        halo -- the host does not block, kernels queued meanwhile overlap with the transfer.  None = completed inline."""
        if not self.native:
            self.shift(send_t, recv_t)
            return None
        ops = []
        if self.rank + 1 < self.world:
            ops.append(dist.P2POp(dist.isend, send_t.contiguous(), self._global(self.rank + 1), self.group))
        if self.rank > 0:
            ops.append(dist.P2POp(dist.irecv, recv_t, self._global(self.rank - 1), self.group))
        if not ops:
            return None
        reqs = dist.batch_isend_irecv(ops)

        class _H:
            def wait(self_inner):
                for r_ in reqs:
                    r_.wait()
        return _H()

    def warm_p2p(self, device):
        """Create the point-to-point channels the tile-parallel decode uses (neighbour strips r -> r+1, column blocks
        r -> 0) before anything is timed: RCCL sets a pair's channel up lazily at its first send / recv."""
        if self.world == 1:
            return
        t = torch.zeros(8, dtype=torch.float32, device=device)
        r = torch.empty_like(t)
        self.shift(t, r)
        if self.rank == 0:
            for src in range(1, self.world):
                self.recv(r, src)
        else:
            self.send(t, 0)

    def selftest(self, device):
        """One round of every collective the sampling path uses, with known values: uneven all_to_all (rank r sends
        p + 1 elements of value 100 r + p to rank p), all_reduce, broadcast, and the point-to-point channels.  Raises on a
        wrong value -- AFTER every collective has been issued: a rank that left at its first mismatch would leave the others
        waiting inside the next collective; transport errors propagate."""
        P, r = self.world, self.rank
        wrong = []
        send_spl = [p + 1 for p in range(P)]
        recv_spl = [r + 1] * P
        send = torch.cat([torch.full((p + 1,), float(100 * r + p)) for p in range(P)]).to(device=device, dtype=torch.bfloat16)
        recv = torch.empty(sum(recv_spl), dtype=torch.bfloat16, device=device)
        h = self.all_to_all(recv, send, recv_spl, send_spl, async_op=self.native)
        if h is not None:
            h.wait()
        exp = torch.cat([torch.full((r + 1,), float(100 * p + r)) for p in range(P)]).to(torch.bfloat16)
        if not torch.equal(recv.cpu(), exp):
            wrong.append("all_to_all")
        t = torch.full((4,), float(r + 1), device=device)
        self.all_reduce(t)
        if not bool((t.cpu() == P * (P + 1) / 2).all()):
            wrong.append("all_reduce")
        b = torch.full((3,), float(r), device=device)
        self.broadcast(b, 0)
        if not bool((b.cpu() == 0).all()):
            wrong.append("broadcast")
        self.warm_p2p(device)
        if wrong:
            raise RuntimeError("sequence-parallel self-test: wrong data delivered by " + ", ".join(wrong))

    def recv(self, t, src):
        if self.native or t.device.type == "cpu":
            dist.recv(t, self._global(src), group=self.group)
        else:
            h = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(h, self._global(src), group=self.group)
            t.copy_(h)
        return t


# ---- reference-named helpers (trainer_misc/sp_utils.py) ---------------------------------------------------------
_SP = None
_SP_PROC_NUM = None
_GUIDANCE_PARALLEL = False


def pair_gather(comm, mine, out):
    """Guidance parallelism (flux_cfg.py): both ranks of a world of 2 end with `out` = [rank 0's `mine` | rank 1's `mine`].
    `out` (2 x mine.numel() elements, same dtype / device) is zeroed, this rank's slot filled and the buffer summed over the
    ranks: every element is ONE value + one zero, so the result is exact and identical on both ranks in any reduction order
    (an all-gather in effect, through the all-reduce every communicator backend of SPComm already has)."""
    n = mine.numel()
    assert comm.world == 2 and out.numel() == 2 * n and out.dtype == mine.dtype
    flat = out.view(-1)
    flat.zero_()
    flat[comm.rank * n:(comm.rank + 1) * n].copy_(mine.reshape(-1))
    comm.all_reduce(flat)
    return out


def init_sequence_parallel_group(args=None, sp_group_size=None, native=False, guidance_parallel=False, window_mib=0):
    """trainer_misc/sp_utils.py:21-47: consecutive-rank groups of `sp_group_size` (default: the whole world) over the
    first `args.sp_proc_num` processes (-1 / absent = all).  A process outside every group stays un-initialised.
    native=True (whole-world group only): the collectives go through the C-ABI communicator (pf_comm_*, RCCL driven
    directly on its own HIP stream) instead of torch.distributed; torch.distributed is used once, to ship the id.
    window_mib > 0 (native only): attach exchange windows of that slot size -- the copy-engine transport."""
    global _SP, _SP_PROC_NUM, _GUIDANCE_PARALLEL
    world = dist.get_world_size()
    # guidance_parallel (a world of two only; not a reference option): the two ranks split the classifier-free-guidance pair
    # instead of the sequence (pyflow_hip/flux_cfg.py) -- the group and its collectives are the same, the engine differs
    _GUIDANCE_PARALLEL = bool(guidance_parallel) and world == 2
    if native:
        from .comm_native import NativeComm, exchange_unique_id
        rank = dist.get_rank()
        _SP_PROC_NUM = world
        _SP = NativeComm(rank, world, exchange_unique_id(rank, world))
        if window_mib > 0:
            # the copy-engine transport for chunks of at most window_mib MiB (csrc/comm.hip: IPC-mapped exchange windows, no
            # communication kernel beside the persistent GEMM); torch.distributed ships the 64-byte IPC handles, once
            def gather(mine):
                out = [None] * world
                dist.all_gather_object(out, mine)
                return out
            _SP.attach_windows(window_mib << 20, gather)
        return _SP
    size = sp_group_size or getattr(args, "sp_group_size", None) or world
    proc = getattr(args, "sp_proc_num", -1) if args is not None else -1
    proc = world if proc in (-1, None) else proc
    assert proc % size == 0, "The process needs to be evenly divided"
    _SP_PROC_NUM = proc
    rank = dist.get_rank()
    group, member = None, False
    if size == world and proc == world:
        member = True                                  # the default group
    else:
        for g0 in range(0, proc, size):
            grp = dist.new_group(list(range(g0, g0 + size)))
            if g0 <= rank < g0 + size:
                group, member = grp, True
    if member:
        _SP = SPComm(group)
    return _SP


def is_sequence_parallel_initialized():
    return _SP is not None


def is_guidance_parallel():
    return _SP is not None and _GUIDANCE_PARALLEL


def get_sequence_parallel_comm():
    return _SP


def get_sequence_parallel_world_size():
    return _SP.world if _SP else 1


def get_sequence_parallel_rank():
    return _SP.rank if _SP else 0


def get_sequence_parallel_group():
    """sp_utils.py:69-71"""
    assert _SP is not None, "sequence parallel group is not initialized"
    return _SP.group if _SP.group is not None else dist.group.WORLD


def get_sequence_parallel_group_rank():
    """sp_utils.py:89-93: index of this process's group = global rank // group size"""
    assert _SP is not None, "sequence parallel size is not initialized"
    return dist.get_rank() // _SP.world


def get_sequence_parallel_proc_num():
    return _SP_PROC_NUM


def reset_sequence_parallel():
    """test hook: forget the group (the reference has no teardown; a process normally initialises once)"""
    global _SP, _SP_PROC_NUM, _GUIDANCE_PARALLEL
    _SP, _SP_PROC_NUM, _GUIDANCE_PARALLEL = None, None, False
