"""The diagnostic the round-4 review asked for: what does the vendor GEMM (hipBLASLt behind torch.matmul; measurement only)
move between L2 and the fabric on the DiT's shapes, next to gemm8p?  Run under `rocprofv3 --kernel-trace --pmc <counter>`
(one counter set per pass: tools/gpu_r5_blas_pmc.sh); few launches of each kernel so that a pass stays short.
usage: python tools/blas_pmc.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops                                                                    # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in [(30976, 5760, 1920), (30976, 13440, 1920), (30976, 1920, 1920), (30976, 1920, 7680)]:
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(iters):
            ops.gemm(A, W, out, M, N, K, K, K, N)
        for _ in range(iters):
            torch.matmul(A, W.t(), out=out)
        torch.cuda.synchronize()
        print(f"done M={M} N={N} K={K}", flush=True)


if __name__ == "__main__":
    main()
