"""Where the 8-wave GEMM's K-tile time goes: pf_gemm_set_variant(11) stamps s_memtime around the four slots (L0 = fragment
reads + B pieces, M0 = 16 MFMAs, L1 = fragment reads + A pieces, M1) and the two barrier waits of workgroup 0 and returns
per-wave sums.  Cycles are s_memtime ticks (100 MHz constant clock on gfx9 parts unless it reports shader cycles -- the
PROPORTIONS are what matters); the stamps themselves add a scalar-memory round trip per slot."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops                                                                    # noqa: E402

lib = ops.L.load()
g = torch.Generator(device="cuda").manual_seed(0)
for bn, (M, N, K) in ((192, (30976, 1920, 7680)), (192, (30976, 1920, 1920)), (256, (30976, 7680, 1920)), (256, (16384, 2048, 13824))):
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(64, device="cuda")
    ops.gemm_set_policy(bn)
    lib.pf_gemm_set_variant(1)
    for _ in range(20):
        ops.gemm(A, W, C, M, N, K, K, K, N)
    lib.pf_gemm_set_variant(11)
    for _ in range(3):
        ops.gemm(A, W, C, M, N, K, K, K, N, gate=dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().view(8, 8)
    nk = d[0, 6].item()
    print(f"--- 256x{bn} M={M} N={N} K={K}  ({int(nk)} K-tiles; per-K-tile ticks of workgroup 0)")
    print("wave grp    L0      M0      L1   wait1      M1   wait0   total")
    for w in range(8):
        v = (d[w, :6] / nk).tolist()
        print(f"  {w}   {w >> 2}  " + " ".join(f"{x:7.1f}" for x in v) + f" {sum(v):7.1f}")
    lib.pf_gemm_set_variant(1)
    ops.gemm_set_policy(0)
