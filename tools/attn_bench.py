"""Times pf_attention_bf16 on the masked C3 sequences (B=2, H=30, hd=64), q pre-scaled as the DiT leaves it (scores
in the +-10 log2 range of real activations after QK-norm), and checks 14 sampled rows against fp32; prints useful TF."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops
from pyflow_hip.plan import SequencePlan
B, H, Lt, d = 2, 30, 128, 1920
mask = torch.zeros(B, Lt, dtype=torch.long); mask[0, :40] = 1; mask[1, :96] = 1
for name, clips in {"u30s2_L15488": [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)],
                    "u15s2_L11888": [(13, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)],
                    "u30s0_L7568": [(29, 24, 40), (1, 24, 40), (1, 24, 40)], "u1s2_L7808": [(1, 96, 160), (1, 96, 160)],
                    "u5s1_L3008": [(4, 24, 40), (1, 48, 80), (1, 48, 80)]}.items():
    plan = SequencePlan(clips, mask, [16, 24, 24], "cuda")
    L, Lp = plan.L, plan.Lp
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(B, L, 3 * d, device="cuda", generator=g)
    qkv[..., 2 * d:] *= 0.125 * ops.LOG2E
    qkv = qkv.to(torch.bfloat16)
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device="cuda")
    ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp)
    out = torch.empty(B, L, d, dtype=torch.bfloat16, device="cuda")
    run = lambda: ops.attention(qkv, qkv, vT, out, 2 * d, 0, 0, 3 * d, L * 3 * d, B, H, L, Lp, Lt, plan, 0.125,
                                q_prescaled=True, ldo=d, o_bstride=L * d)
    import ctypes
    lib = ops.L.load()
    fl = 4.0 * plan.useful_pairs() * 64 * H
    res = {}
    for _ in range(4):
        for ab in (9, 0):
            if not hasattr(lib, "pf_attn_ab") and ab:
                continue
            if hasattr(lib, "pf_attn_ab"):
                lib.pf_attn_ab(ctypes.c_int(ab))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(ab, []).append(e0.elapsed_time(e1) / 5)
    ms = statistics.median(res[0])
    if 9 in res:
        m9 = statistics.median(res[9])
        print(f"   round-1 kernel: {m9:.3f} ms {fl / m9 / 1e9:.0f} TF", flush=True)
    rows = [0, 39, 40, 127, 128, 367, 368, L // 2, L - 1]
    dm = torch.from_numpy(plan.dense_mask()[:, rows]).to("cuda")
    q = qkv[:, rows, 2 * d:].float().view(B, len(rows), H, 64).transpose(1, 2)
    k = qkv[..., :d].float().view(B, L, H, 64).transpose(1, 2)
    v = qkv[..., d:2 * d].float().view(B, L, H, 64).transpose(1, 2)
    sc = torch.einsum("bhrd,bhld->bhrl", q, k) * 0.6931471805599453
    sc = sc.masked_fill(~dm[:, None], float("-inf"))
    ref = torch.einsum("bhrl,bhld->bhrd", torch.softmax(sc, -1), v).transpose(1, 2).reshape(B, len(rows), d)
    err = ((out[:, rows].float() - ref).norm() / ref.norm()).item()
    print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF useful  (sampled-row rel-L2 {err:.2e})", flush=True)
