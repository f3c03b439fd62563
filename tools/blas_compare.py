"""Reference points for the two MFMA kernels on the same MI355X: the vendor libraries torch ships (hipBLASLt / rocBLAS
behind torch.matmul, the flash backend behind F.scaled_dot_product_attention) on the DiT's shapes, next to
pf_gemm_bf16 / pf_attention_bf16.  Measurement only -- nothing in the product path calls torch compute."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops                                                                    # noqa: E402


def bench(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in [(30976, 1920, 1920), (30976, 5760, 1920), (30976, 7680, 1920), (30976, 1920, 7680),
                    (30976, 13440, 1920), (30976, 1920, 9600), (16384, 2048, 13824), (8192, 8192, 8192)]:
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t_pf = bench(lambda: ops.gemm(A, W, out, M, N, K, K, K, N))
        t_lib = bench(lambda: torch.matmul(A, W.t(), out=out))
        fl = 2.0 * M * N * K / 1e9
        print(f"gemm M={M} N={N} K={K}: pf_gemm_bf16 {t_pf:.3f} ms {fl / t_pf:.0f} TF | torch.matmul {t_lib:.3f} ms "
              f"{fl / t_lib:.0f} TF | ratio {t_lib / t_pf:.3f}", flush=True)
    # dense (unmasked) attention, the library's best case: B=2, H=30, d=64
    for L in (4096, 15488):
        q, k, v = [torch.randn(2, 30, L, 64, device="cuda", generator=g).bfloat16() for _ in range(3)]
        t = bench(lambda: F.scaled_dot_product_attention(q, k, v), iters=5, warm=2)
        fl = 4.0 * 2 * 30 * L * L * 64 / 1e9
        print(f"sdpa dense L={L}: {t:.3f} ms {fl / t:.0f} TF (library flash backend, no mask)", flush=True)


if __name__ == "__main__":
    main()
