"""DVFS / power diagnostic: the same kernels on zero-filled vs random operands (identical instruction streams)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops
from pyflow_hip.plan import SequencePlan


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B, H, Lt, d = 2, 30, 128, 1920
clips = [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)]
mask = torch.zeros(B, Lt, dtype=torch.long); mask[0, :40] = 1; mask[1, :96] = 1
plan = SequencePlan(clips, mask, [16, 24, 24], "cuda")
L, Lp = plan.L, plan.Lp
useful = 4 * plan.useful_pairs() * 64 * H
M, N, K = 30976, 1920, 7680
# warm the clocks
Aw = torch.randn(8192, 4096, device="cuda").to(torch.bfloat16); Ww = torch.randn(4096, 4096, device="cuda").to(torch.bfloat16)
Cw = torch.empty(8192, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(200):
    ops.gemm(Aw, Ww, Cw, 8192, 4096, 4096, 4096, 4096, 4096)
for fill in ("random", "zeros", "random", "zeros"):
    if fill == "random":
        qkv = torch.randn(B, L, 3 * d, device="cuda"); qkv[..., 2 * d:] *= 0.5; qkv = qkv.to(torch.bfloat16)
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    else:
        qkv = torch.zeros(B, L, 3 * d, device="cuda", dtype=torch.bfloat16)
        A = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16); W = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device="cuda")
    ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp)
    out = torch.empty_like(qkv)
    ms = min(timeit(lambda: ops.attention(qkv, qkv, vT, out, 2 * d, 0, 2 * d, 3 * d, L * 3 * d, B, H, L, Lp, Lt, plan, 0.125, q_prescaled=True)) for _ in range(2))
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_set_policy(192)
    mg = min(timeit(lambda: ops.gemm(A, W, C, M, N, K, K, K, N)) for _ in range(2))
    print(f"{fill}: attention {ms:.3f} ms ({useful / ms / 1e9:.0f} TF useful)   gemm256<192> {mg:.3f} ms ({2 * M * N * K / mg / 1e9:.0f} TF)", flush=True)
