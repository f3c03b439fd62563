"""The C-ABI communicator (pf_comm_*, csrc/comm.hip) against the real RCCL on one rank: bootstrap (unique id ->
ncclCommInitRank), all-to-all with uneven counts to self, all-reduce, broadcast, all-gather, halo pass (no neighbours on
one rank: must leave the receive buffer alone), ordering against the compute stream by events only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))


def main():
    from pyflow_hip.comm_native import NativeComm, exchange_unique_id
    torch.cuda.set_device(0)
    c = NativeComm(0, 1, exchange_unique_id(0, 1))
    send = torch.arange(1000, dtype=torch.bfloat16, device="cuda")
    recv = torch.zeros(1000, dtype=torch.bfloat16, device="cuda")
    h = c.all_to_all(recv, send, [1000], [1000], async_op=True)
    filler = torch.ones(1 << 20, device="cuda") * 2          # work queued while the exchange is in flight
    h.wait()
    assert torch.equal(recv, send) and float(filler[0]) == 2.0
    t = torch.full((5,), 3.0, device="cuda")
    c.all_reduce(t)
    assert bool((t == 3.0).all())
    b = torch.arange(7, dtype=torch.float32, device="cuda")
    c.broadcast(b, 0)
    assert torch.equal(b.cpu(), torch.arange(7, dtype=torch.float32))
    g = torch.zeros(6, dtype=torch.float32, device="cuda")
    c.all_gather_v(torch.arange(6, dtype=torch.float32, device="cuda"), g, [6])
    assert torch.equal(g.cpu(), torch.arange(6, dtype=torch.float32))
    keep = torch.full((4,), 9.0, device="cuda")
    c.shift(torch.ones(4, device="cuda"), keep)
    assert bool((keep == 9.0).all())
    c.barrier()
    c.close()
    print("native communicator on one rank: ok")


if __name__ == "__main__":
    main()
