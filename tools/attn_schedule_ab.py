"""Attention seconds per C3 video: times pf_attention_bf16 on EVERY (unit, stage) sequence of the headline schedule (31 units x
3 stages, L = 368 ... 15 488; B = 2, H = 30, the DiT's K | V | Q column layout with V read token-major) and weights each with
its forwards (20 / 10 steps) x 24 blocks.  Usage: python tools/attn_schedule_ab.py [variant]   (a library build under
pyramid-flow_amd/variants/, default: the shipping library).  Prints per length: kernel choice, ms, useful TFLOP/s; then the
total.  Run one process per variant, alternating, for a same-box A/B."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as L                                                                  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "ship":
    L.use_lab_library(sys.argv[1])
from pyflow_hip import ops                                                                       # noqa: E402
from pyflow_hip.plan import SequencePlan                                                         # noqa: E402

B, H, Lt, d = 2, 30, 128, 1920
mask = torch.zeros(B, Lt, dtype=torch.long)
mask[0, :40] = 1
mask[1, :96] = 1
res_of = [(24, 40), (48, 80), (96, 160)]


def clips_of(u, s):
    out = []
    if u >= 3:
        out.append((u - 2, *res_of[0]))
    if u >= 2:
        out.append((1, *res_of[max(s - 1, 0)]))
    if u >= 1:
        out.append((1, *res_of[s]))
    out.append((1, *res_of[s]))
    return out


so = L.load()
tot_ms, tot_fl, rows = 0.0, 0.0, []
quick = os.environ.get("ATTN_AB_QUICK")          # only every 3rd unit
for u in range(31):
    if quick and u % 3 and u != 30:
        continue
    for s in range(3):
        plan = SequencePlan(clips_of(u, s), mask, [16, 24, 24], "cuda")
        Ls, Lp = plan.L, plan.Lp
        g = torch.Generator(device="cuda").manual_seed(3)
        qkv = torch.randn(B, Ls, 3 * d, device="cuda", generator=g)
        qkv[..., 2 * d:] *= 0.125 * ops.LOG2E
        qkv = qkv.to(torch.bfloat16)
        out = torch.empty(B, Ls, d, dtype=torch.bfloat16, device="cuda")

        def run():
            ops.attention(qkv, qkv, None, out, 2 * d, 0, 0, 3 * d, Ls * 3 * d, B, H, Ls, Lp, Lt, plan, 0.125, q_prescaled=True,
                          ldo=d, o_bstride=Ls * d, v_off=d)
        run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 4 if Ls > 4000 else 12
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n)
        ms = statistics.median(ts)
        fl = 4.0 * plan.useful_pairs() * 64 * H
        w = (20 if u == 0 else 10) * 24 * (3 if quick and u % 3 == 0 and u not in (0, 30) else 1)
        tot_ms += ms * w
        tot_fl += fl * w
        nq = (Ls + 255) // 256 * H * B
        rows.append((u, s, Ls, nq, ms, fl / ms / 1e9))
for u, s, Ls, nq, ms, tf in rows:
    if s == 2 or u in (0, 1, 2):
        print(f"u{u:02d}s{s} L={Ls:5d} workgroups {nq:5d} ({nq / 512:.2f} rounds): {ms:7.3f} ms {tf:6.0f} TF useful")
print(f"TOTAL attention per video ({sys.argv[1] if len(sys.argv) > 1 else 'ship'}): {tot_ms / 1e3:.3f} s, mean {tot_fl / tot_ms / 1e9:.0f} TFLOP/s useful")
