"""same-box A/B of pf_gemm_set_policy(9) (desynchronised start of gemm8p's workgroups) on the DiT's projections at L = 15 488, batch 2:
ABBA order, many short samples (box noise between consecutive samples is +-2...4 %), bitwise equality of the results."""
import ctypes as C, statistics, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as _labsel; _labsel.use_lab_library()      # needs `make -C pyramid-flow_amd/csrc lab`
from pyflow_hip import ops, lib as L
lib = L.load()
dev = "cuda"
torch.manual_seed(0)
M = 30976
ws = torch.empty(int(lib.pf_gemm_workspace_bytes(C.c_int(M), C.c_int(1), C.c_int(1920), C.c_int(7680))) // 4 + 16, device=dev, dtype=torch.float32)
cases = [("attn out  N=1920 K=1920 (residual)", 1920, 1920, True), ("MLP down  N=1920 K=7680 (residual)", 1920, 7680, True),
         ("K|V|Q     N=5760 K=1920 (plain)", 5760, 1920, False), ("MLP up    N=7680 K=1920 (GELU)", 7680, 1920, False)]
for name, N, K, res in cases:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    gate = torch.randn(1, N, device=dev) if res else None
    out = {}
    def run(c):
        if res:
            ops.gemm(a, w, c, M, N, K, K, K, N, res=r, gate=gate, ldr=N, gate_stride=N, flags=ops.GEMM_GATE_RES, workspace=ws)
        else:
            ops.gemm(a, w, c, M, N, K, K, K, N, gelu_from=0 if "GELU" in name else -1, workspace=ws)
    times = {0: [], 1: []}
    for rnd in range(12):
        for mode in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            ops.gemm_set_policy(9 if mode else -9)
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                run(c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(c)
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / 20)
            out[mode] = c
    ops.gemm_set_policy(-9)
    t0, t1 = statistics.median(times[0]), statistics.median(times[1])
    fl = 2.0 * M * N * K
    print(f"{name}: together {t0:.4f} ms ({fl / t0 / 1e9:6.0f} TFLOP/s)   desynchronised {t1:.4f} ms ({fl / t1 / 1e9:6.0f} TFLOP/s)   "
          f"{(t0 / t1 - 1) * 100:+.1f} %   same bits: {torch.equal(out[0], out[1])}", flush=True)
