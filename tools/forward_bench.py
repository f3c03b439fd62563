"""times one full-width miniFLUX forward at a few (unit, stage) sequences; options via env: OVERLAP_TEXT=0/1"""
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
cases = {"u30s2": [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)], "u30s0": [(29, 24, 40), (1, 24, 40), (1, 24, 40)],
         "u5s1": [(4, 24, 40), (1, 48, 80), (1, 48, 80)], "u0s2": [(1, 96, 160)]}
for name, shapes in cases.items():
    clips = [torch.randn(1, 16, *s, device=dev) for s in shapes]
    plan = eng.make_plan(shapes, mask)
    line = f"{name} L={plan.L}:"
    for ov in (False, True, False, True):
        eng.overlap_text = ov
        for _ in range(2):
            eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
        torch.cuda.synchronize()
        line += f"  overlap={int(ov)} {(time.perf_counter() - t0) / n * 1e3:.2f} ms"
    print(line, flush=True)
