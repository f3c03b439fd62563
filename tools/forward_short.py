"""N production (graph-replayed) miniFLUX forwards at ONE short (unit, stage) sequence of the C3 schedule -- the target of a
rocprofv3 --kernel-trace --stats pass that shows what a forward's fixed cost (~3 ms at any length: DESIGN.md section 3) is made of.
argv: case (u0s0 | u3s0 | u8s0 | u0s1 | u5s1 | u0s2) [forwards = 20]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import synth
from pyflow_hip.flux import FluxEngine
dev = "cuda"
cfg = synth.MINIFLUX
g = torch.Generator(device=dev).manual_seed(1234)
sd = {}
for k, shp in synth.flux_param_shapes(cfg).items():
    sd[k] = (torch.ones(shp, device=dev) if k.endswith(".weight") else torch.zeros(shp, device=dev)) if len(shp) == 1 \
        else torch.randn(shp, generator=g, device=dev) * 0.02
eng = FluxEngine(sd, cfg, dev)
del sd
mask = torch.zeros(2, 128, dtype=torch.long); mask[0, :40] = 1; mask[1, :96] = 1
enc = torch.randn(2, 128, 4096).to(torch.bfloat16)
pooled = torch.randn(2, 768)
eng.encode_context(enc)
cases = {"u0s0": [(1, 24, 40)], "u3s0": [(2, 24, 40), (1, 24, 40), (1, 24, 40)], "u8s0": [(7, 24, 40), (1, 24, 40), (1, 24, 40)],
         "u0s1": [(1, 48, 80)], "u5s1": [(4, 24, 40), (1, 48, 80), (1, 48, 80)], "u0s2": [(1, 96, 160)],
         "u30s0": [(29, 24, 40), (1, 24, 40), (1, 24, 40)]}
name = sys.argv[1] if len(sys.argv) > 1 else "u0s0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = cases[name]
clips = [torch.randn(1, 16, *s, device=dev) for s in shapes]
plan = eng.make_plan(shapes, mask)
for _ in range(3):
    eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
torch.cuda.synchronize()
print(f"{name} L={plan.L}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per forward ({n} forwards, launch_mode={eng.launch_mode})", flush=True)
