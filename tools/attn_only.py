"""runs the L=15488 joint attention a few times (target for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops
from pyflow_hip.plan import SequencePlan
B, H, Lt, d = 2, 30, 128, 1920
clips = [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)]
mask = torch.zeros(B, Lt, dtype=torch.long); mask[0, :40] = 1; mask[1, :96] = 1
plan = SequencePlan(clips, mask, [16, 24, 24], "cuda")
L, Lp = plan.L, plan.Lp
qkv = torch.randn(B, L, 3 * d, device="cuda"); qkv[..., 2 * d:] *= 0.5; qkv = qkv.to(torch.bfloat16)
vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device="cuda")
ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp)
out = torch.empty_like(qkv)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.attention(qkv, qkv, vT, out, 2 * d, 0, 2 * d, 3 * d, L * 3 * d, B, H, L, Lp, Lt, plan, 0.125, q_prescaled=True)
torch.cuda.synchronize()
