"""RCCL API smoke test on ONE GPU: world_size-1 process group with backend nccl, so that the exact torch.distributed
calls of the N > 1 path (all_to_all_single with split sizes, all_reduce, broadcast, barrier) run against the real RCCL
library; the exchanged data must leave the sequence-parallel forward identical to the plain engine."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    from pyflow_hip.flux_sp import FluxEngineSP
    from pyflow_hip.sp import init_sequence_parallel_group
    from util import rel_l2, round_sd
    comm = init_sequence_parallel_group(sp_group_size=1)
    assert comm.native
    cfg = synth.TINY_FLUX
    sd = round_sd(synth.random_state_dict(synth.flux_param_shapes(cfg), seed=3, std=0.05, lively=True))
    g = torch.Generator().manual_seed(0)
    shapes = [(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)]
    clips = [torch.randn(2, 16, *s, generator=g).to(torch.bfloat16).float().cuda() for s in shapes]
    enc = torch.randn(2, 16, 32, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, 16, generator=g)
    a = FluxEngine(sd, cfg, "cuda")
    b = FluxEngineSP(sd, cfg, "cuda", comm=comm)
    plan = a.make_plan(shapes, mask)
    a.encode_context(enc)
    b.encode_context(enc)
    va = a.forward_tokens(plan, clips, [704.0, 704.0], pooled).clone()
    vb = b.forward_tokens(plan, clips, [704.0, 704.0], pooled).clone()
    x = torch.arange(8, dtype=torch.float32, device="cuda")
    comm.all_reduce(x)
    comm.broadcast(x, 0)
    comm.barrier()
    torch.cuda.synchronize()
    err = rel_l2(vb.cpu(), va.cpu())
    print(f"nccl world-1 SP forward rel_l2 vs plain engine = {err:.3e}")
    dist.destroy_process_group()
    sys.exit(0 if err < 2e-3 else 1)


if __name__ == "__main__":
    main()
