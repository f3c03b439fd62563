"""one full-width miniFLUX denoise forward at the C3 worst-case sequence (unit 30, stage 2: L = 15 488, B = 2),
repeated argv[1] times -- the target of the rocprofv3 --pmc passes (tools/gpu_pmc.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import synth
from pyflow_hip.flux import FluxEngine
dev = "cuda"
cfg = synth.MINIFLUX
g = torch.Generator(device=dev).manual_seed(1234)
sd = {}
for k, shp in synth.flux_param_shapes(cfg).items():
    if len(shp) == 1:
        sd[k] = torch.ones(shp, device=dev) if k.endswith(".weight") else torch.zeros(shp, device=dev)
    else:
        sd[k] = torch.randn(shp, generator=g, device=dev) * 0.02
eng = FluxEngine(sd, cfg, dev)
del sd
shapes = [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)]
clips = [torch.randn(1, 16, *s, device=dev) for s in shapes]
mask = torch.zeros(2, 128, dtype=torch.long); mask[0, :40] = 1; mask[1, :96] = 1
enc = torch.randn(2, 128, 4096).to(torch.bfloat16)
pooled = torch.randn(2, 768)
plan = eng.make_plan(shapes, mask)
eng.encode_context(enc)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    eng._mod_cache = {}
    eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
torch.cuda.synchronize()
print("forward done")
