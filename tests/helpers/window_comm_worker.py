"""worker of tests/test_sp_gpu.py::test_copy_engine_window_transport: P rank processes (sharing cuda:0) exchange through the
C-ABI communicator's COPY-ENGINE transport (pf_comm_create_window / pf_comm_attach_windows: IPC-mapped windows, device-to-device
copies, hipStreamWriteValue32 / hipStreamWaitValue32 flags) -- no RCCL, no kernel.  gloo carries the 64-byte IPC handles only."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from pyflow_hip.comm_native import NativeComm

    def gather(mine):
        out = [None] * world
        dist.all_gather_object(out, mine)
        return out
    c = NativeComm(rank, world, None).attach_windows(1 << 20, gather)
    assert c._lib.pf_comm_transport(c._h) == 1 and c.transport == "windows"
    ok = True
    print(f"rank {rank}: windows attached", flush=True)
    for it in range(5):                       # repeated exchanges reuse the slots: the acknowledgement protocol
        # uneven all-to-all: rank r sends (p + 1) * 100 + it elements of value 1000 r + p + it / 16 to rank p
        send_spl = [(p + 1) * 100 + it for p in range(world)]
        recv_spl = [(rank + 1) * 100 + it] * world
        send = torch.cat([torch.full((n,), 1000.0 * rank + p + it / 16.0) for p, n in enumerate(send_spl)]).to("cuda", torch.bfloat16)
        recv = torch.full((sum(recv_spl),), -1.0, dtype=torch.bfloat16, device="cuda")
        h = c.all_to_all(recv, send, recv_spl, send_spl, async_op=True)
        filler = torch.ones(1 << 18, device="cuda") * 3          # compute-stream work while the chunks travel
        h.wait()
        exp = torch.cat([torch.full((recv_spl[p],), 1000.0 * p + rank + it / 16.0) for p in range(world)]).to(torch.bfloat16)
        ok = ok and torch.equal(recv.cpu(), exp) and float(filler[0]) == 3.0
        # the halo pass in between (only neighbour pairs take part: per-pair sequence numbers)
        keep = torch.full((64,), -7.0, device="cuda")
        c.shift(torch.full((64,), float(10 * rank + it), device="cuda"), keep)
        want = -7.0 if rank == 0 else float(10 * (rank - 1) + it)
        ok = ok and bool((keep.cpu() == want).all())
        print(f"rank {rank}: round {it} ok={ok}", flush=True)
    counts = [10 + 3 * p for p in range(world)]
    g = torch.zeros(sum(counts), device="cuda")
    c.all_gather_v(torch.full((counts[rank],), float(rank + 1), device="cuda"), g, counts)
    ok = ok and torch.equal(g.cpu(), torch.cat([torch.full((n,), float(p + 1)) for p, n in enumerate(counts)]))
    if rank == 0:                             # point-to-point (tile-parallel decode's column blocks): r -> 0
        for src in range(1, world):
            t = torch.zeros(33, device="cuda")
            c.recv(t, src)
            ok = ok and bool((t.cpu() == float(src)).all())
    else:
        c.send(torch.full((33,), float(rank), device="cuda"), 0)
    torch.cuda.synchronize()
    # a chunk larger than a slot has no route on a communicator without RCCL: BOTH ends of that pair (0 -> 1) get a clean
    # error, the pairs that fit still complete (nobody is left waiting), and the next exchange works
    n_big = (1 << 20) + 512
    big = torch.zeros(n_big, dtype=torch.uint8, device="cuda")
    ss = [0] * world
    rs = [0] * world
    if rank == 0:
        ss[1] = n_big
    if rank == 1:
        rs[0] = n_big
    try:
        c.all_to_all(big.clone(), big, rs, ss)
        ok = ok and rank not in (0, 1)
    except RuntimeError as e:
        ok = ok and rank in (0, 1) and "window slot" in str(e)
    t_in = torch.full((8 * world,), float(rank), device="cuda")
    t_out = torch.zeros(8 * world, device="cuda")
    c.all_to_all(t_out, t_in, [8] * world, [8] * world)
    ok = ok and torch.equal(t_out.cpu(), torch.arange(world, dtype=torch.float32).repeat_interleave(8))
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.barrier()
    c.close()
    dist.destroy_process_group()
    if rank == 0:
        with open(sys.argv[1], "w") as f:
            f.write(f"copy-engine window transport, {world} ranks on one GPU: {flags}\n")
    sys.exit(0 if all(flags) else 1)


if __name__ == "__main__":
    main()
