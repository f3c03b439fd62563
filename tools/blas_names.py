"""Which vendor GEMM kernels torch picks for the DiT shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in [(30976, 7680, 1920), (30976, 1920, 7680), (30976, 5760, 1920), (16384, 2048, 13824)]:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10):
        torch.matmul(A, W.t(), out=out)
    torch.cuda.synchronize()
