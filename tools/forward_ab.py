"""same-box A/B of the production engine's round-4 switches on one forward: fuse_qk (QK-RMSNorm + RoPE in the projection epilogue
vs the separate pass) x v_rowmajor (V token-major via ds_read_b64_tr_b16 vs pf_v_transpose + V^T image), at the headline
sequence (unit 30, stage 2) and a short one (unit 5, stage 1); ABBA order, median of the runs."""
import os, sys, time, statistics, torch
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
cases = {"u30s2 L=15488": [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)], "u5s1 L=3008": [(4, 24, 40), (1, 48, 80), (1, 48, 80)]}
modes = [(True, True), (False, True), (True, False), (False, False)]
for name, shapes in cases.items():
    clips = [torch.randn(1, 16, *s, device=dev) for s in shapes]
    plan = eng.make_plan(shapes, mask)
    res = {m: [] for m in modes}
    for rnd in range(4):
        for m in (modes if rnd % 2 == 0 else modes[::-1]):
            eng.fuse_qk, eng.v_rowmajor = m
            for _ in range(3):
                eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 6
            for _ in range(n):
                eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
            torch.cuda.synchronize()
            res[m].append((time.perf_counter() - t0) / n * 1e3)
    print(name + ": " + "  ".join(f"fuse_qk={int(a)} v_rowmajor={int(b)}: {statistics.median(v):7.2f} ms" for (a, b), v in res.items()), flush=True)
