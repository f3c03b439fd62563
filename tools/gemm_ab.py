"""Within-process interleaved A/B of the GEMM kernels on the DiT / VAE shapes (random data, several rounds, median):
policy -8 (gemm256 / 128x128 auto), policy 8 (gemm8p), and the vendor library through torch.matmul as the reference point
(measurement only).  Usage: python tools/gemm_ab.py [rounds]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as _lib                                                            # noqa: E402
if os.environ.get("PF_BENCH_LIB"):          # measurement only (a switch of this TOOL): A/B against another build of the library
    _lib.LIB_PATH = os.path.join(ROOT, "pyramid-flow_amd", os.environ["PF_BENCH_LIB"])
from pyflow_hip import ops                                                                    # noqa: E402

SHAPES = [  # (M per batch, batch, N, K, gelu_from, gate_res)
    (15488, 2, 1920, 1920, -1, True), (15488, 2, 5760, 1920, -1, False), (15488, 2, 7680, 1920, 0, False),
    (15488, 2, 1920, 7680, -1, True), (15488, 2, 13440, 1920, 5760, False), (15488, 2, 1920, 9600, -1, True),
    (7808, 2, 13440, 1920, 5760, False), (7808, 2, 1920, 9600, -1, True), (3008, 2, 13440, 1920, 5760, False),
    (3008, 2, 1920, 9600, -1, True), (16384, 1, 2048, 13824, -1, False), (8192, 1, 8192, 8192, -1, False),
    (4096, 1, 4096, 4096, -1, False),
]


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = SHAPES
    if os.environ.get("GEMM_AB_SHAPES"):
        shapes = [SHAPES[int(i)] for i in os.environ["GEMM_AB_SHAPES"].split(",")]
    for M, B, N, K, gf, gr in shapes:
        A = torch.randn(B, M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
        bias = torch.randn(N, device="cuda", generator=g)
        gate = torch.randn(B, N, device="cuda", generator=g)
        res = torch.randn(B, M, N, device="cuda", generator=g).bfloat16()
        out = torch.empty(B, M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(bias=bias, batch=B, strideA=M * K, strideC=M * N, gelu_from=gf)
        if gr:
            kw.update(res=res, gate=gate, ldr=N, strideR=M * N, gate_stride=N, flags=ops.GEMM_GATE_RES)

        def run(pol):
            ops.gemm_set_policy(pol)
            ops.gemm(A, W, out, M, N, K, K, K, N, **kw)

        def lib():
            torch.matmul(A.view(B * M, K), W.t(), out=out.view(B * M, N))
        arms = {"auto": lambda: run(-8), "8p": lambda: run(8), "lib": lib}
        iters = max(3, int(2e12 / (2.0 * M * B * N * K)))
        times = {k: [] for k in arms}
        for fn in arms.values():
            fn()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for k, fn in arms.items():
                times[k].append(timed(fn, iters))
        fl = 2.0 * M * B * N * K / 1e9
        ops.gemm_set_policy(-8)
        which = ops.L.load().pf_gemm_which(M, B, N, K)
        ops.gemm_set_policy(0)
        msg = f"M={M}x{B} N={N} K={K} gelu={gf} res={int(gr)} (auto={which}):"
        for k in arms:
            med = statistics.median(times[k])
            msg += f"  [{k}] {med:.3f} ms {fl / med:.0f} TF (min {fl / max(times[k]):.0f} max {fl / min(times[k]):.0f})"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
