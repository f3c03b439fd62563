"""one DiT GEMM shape repeated (target of tools/gpu_pmc.sh): M=30976 N=1920 K=7680, gemm256<192>, variant from argv[2]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops
M, N, K = 30976, 1920, 7680
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ops.gemm_set_policy(192)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.gemm(A, W, C, M, N, K, K, K, N)
torch.cuda.synchronize()
