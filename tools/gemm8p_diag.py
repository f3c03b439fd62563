"""Where a gemm8p tile's time goes: the same launch with (0) the full epilogue, (1) the epilogue without its global
stores, (2) no epilogue at all -- interleaved, medians.  Measurement only (diag modes produce no output)."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops                                                                    # noqa: E402

lib = ops.L.load()


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device="cuda").manual_seed(0)
for M, B, N, K, gf, gr in [(15488, 2, 5760, 1920, -1, False), (15488, 2, 7680, 1920, 0, False),
                           (15488, 2, 1920, 9600, -1, True), (8192, 1, 8192, 8192, -1, False)]:
    A = torch.randn(B, M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(B, N, device="cuda", generator=g)
    res = torch.randn(B, M, N, device="cuda", generator=g).bfloat16()
    out = torch.empty(B, M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(bias=bias, batch=B, strideA=M * K, strideC=M * N, gelu_from=gf)
    if gr:
        kw.update(res=res, gate=gate, ldr=N, strideR=M * N, gate_stride=N, flags=ops.GEMM_GATE_RES)
    ops.gemm_set_policy(8)
    times = {0: [], 1: [], 2: []}
    iters = max(3, int(2e12 / (2.0 * M * B * N * K)))
    for _ in range(5):
        for d in (0, 1, 2):
            lib.pf_gemm8p_diag(C.c_int(d))
            ops.gemm(A, W, out, M, N, K, K, K, N, **kw)
            times[d].append(timed(lambda: ops.gemm(A, W, out, M, N, K, K, K, N, **kw), iters))
    lib.pf_gemm8p_diag(C.c_int(0))
    fl = 2.0 * M * B * N * K / 1e9
    tiles = ((M + 255) // 256) * B * ((N + 255) // 256)
    msg = f"M={M}x{B} N={N} K={K} gelu={gf} res={int(gr)} tiles={tiles} ({tiles / 256:.2f} rounds):"
    for d, nm in ((0, "full"), (1, "no-store"), (2, "no-epilogue")):
        med = statistics.median(times[d])
        msg += f"  [{nm}] {med:.3f} ms {fl / med:.0f} TF"
    print(msg, flush=True)
ops.gemm_set_policy(0)
