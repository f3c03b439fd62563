"""Kernel microbenchmarks at the C3 (768p, 241-frame) worst-case shapes. Usage: python tools/microbench.py [gemm|attn|all]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops  # noqa: E402
from pyflow_hip.plan import SequencePlan  # noqa: E402


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


def bench_gemm():
    # clock ramp: ~0.3 s of GEMM work before anything is timed
    Aw = torch.randn(8192, 4096, device="cuda").to(torch.bfloat16)
    Ww = torch.randn(4096, 4096, device="cuda").to(torch.bfloat16)
    Cw = torch.empty(8192, 4096, device="cuda", dtype=torch.bfloat16)
    for _ in range(300):
        ops.gemm(Aw, Ww, Cw, 8192, 4096, 4096, 4096, 4096, 4096)
    torch.cuda.synchronize()
    d = 1920
    M = 2 * 15488
    shapes = [(M, d, d, -1), (M, 3 * d, d, -1), (M, 4 * d, d, 0), (M, d, 4 * d, -1), (M, 7 * d, d, 3 * d), (M, d, 5 * d, -1),
              (2 * 3968, 7 * d, d, 3 * d), (2 * 1088, 7 * d, d, 3 * d),
              # VAE conv-like GEMM shapes (plain GEMM addressing): 8 frames x 256 x 256 px, Cin*27 x Cout
              (8 * 256 * 256, 128, 27 * 128, -1), (8 * 128 * 128, 256, 27 * 256, -1), (4 * 64 * 64, 2048, 27 * 512, -1)]
    for (Mx, N, K, gelu) in shapes:
        A = torch.randn(Mx, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        C = torch.empty(Mx, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.zeros(N, device="cuda")
        line = f"gemm M={Mx} N={N} K={K} gelu_from={gelu}:"
        lib = ops.L.load()
        for pol, var in ((-1, 0), (128, 1), (192, 1), (192, 3), (192, 10), (256, 1), (256, 3), (256, 10)):
            if pol > 0 and N % pol:
                continue
            ops.gemm_set_policy(pol)
            lib.pf_gemm_set_variant(var)
            ms = min(timeit(lambda: ops.gemm(A, W, C, Mx, N, K, K, K, N, bias=bias, gelu_from=gelu)) for _ in range(2))
            line += f"  [{'128x128' if pol < 0 else '256x%d/v%d' % (pol, var)}] {ms:.3f} ms {2 * Mx * N * K / ms / 1e9:.0f} TF"
        ops.gemm_set_policy(0)
        lib.pf_gemm_set_variant(1)
        print(line, flush=True)


def bench_attn():
    B, H, Lt, d = 2, 30, 128, 1920
    for name, clips in [("unit30_s2", [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)]),
                        ("unit30_s0", [(29, 24, 40), (1, 24, 40), (1, 24, 40)]),
                        ("unit0_s2", [(1, 96, 160)])]:
        mask = torch.zeros(B, Lt, dtype=torch.long)
        mask[0, :40] = 1
        mask[1, :96] = 1
        plan = SequencePlan(clips, mask, [16, 24, 24], "cuda")
        L, Lp = plan.L, plan.Lp
        qkv = torch.randn(B, L, 3 * d, device="cuda")
        qkv[..., 2 * d:] *= 0.5       # q ~ scores of a few units in the exponent, like normalised q.k/8
        qkv = qkv.to(torch.bfloat16)
        vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device="cuda")
        ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp)
        out = torch.empty_like(qkv)
        useful = 4 * plan.useful_pairs() * 64 * H
        dense = 4 * B * L * L * 64 * H
        for pre in (False, True):
            ms = min(timeit(lambda: ops.attention(qkv, qkv, vT, out, 2 * d, 0, 2 * d, 3 * d, L * 3 * d, B, H, L, Lp, Lt, plan,
                                                  0.125, q_prescaled=pre)) for _ in range(2))
            print(f"attn {name} L={L} prescaled={pre}: {ms:.3f} ms  useful {useful / ms / 1e9:.1f} TFLOP/s "
                  f"(dense-equivalent {dense / ms / 1e9:.1f})", flush=True)
        ms = timeit(lambda: ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp))
        print(f"v_transpose L={L}: {ms:.3f} ms  {2 * B * L * d * 2 / ms / 1e6:.1f} GB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("attn", "all"):
        bench_attn()
