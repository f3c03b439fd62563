"""host-side cost of one forward (launch loop only, no sync) vs its device time, at a small sequence"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import synth
from pyflow_hip.flux import FluxEngine
from pyflow_hip.flux_sp import FluxEngineSP
dev = "cuda"
cfg = synth.MINIFLUX
g = torch.Generator(device=dev).manual_seed(1234)
sd = {}
for k, shp in synth.flux_param_shapes(cfg).items():
    sd[k] = (torch.ones(shp, device=dev) if k.endswith(".weight") else torch.zeros(shp, device=dev)) if len(shp) == 1 \
        else torch.randn(shp, generator=g, device=dev) * 0.02
mask = torch.zeros(2, 128, dtype=torch.long); mask[0, :40] = 1; mask[1, :96] = 1
enc = torch.randn(2, 128, 4096).to(torch.bfloat16)
pooled = torch.randn(2, 768)
for cls, mode in ((FluxEngine, "eager"), (FluxEngine, "list"), (FluxEngine, "graph"), (FluxEngineSP, "eager"), (FluxEngineSP, "list")):
    eng = cls(sd, cfg, dev)
    eng.launch_mode = mode
    eng.encode_context(enc)
    for name, shapes in {"u1s0": [(1, 24, 40), (1, 24, 40)], "u5s1": [(4, 24, 40), (1, 48, 80), (1, 48, 80)]}.items():
        clips = [torch.randn(1, 16, *s, device=dev) for s in shapes]
        plan = eng.make_plan(shapes, mask)
        for _ in range(3):
            eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
        print(f"{cls.__name__} [{mode}] {name} L={plan.L}: host launch loop {t_host * 1e3:.2f} ms / forward, wall {t_all * 1e3:.2f} ms / forward", flush=True)
        if cls is FluxEngineSP and mode == "list":
            # where the host time of the recorded sequence-parallel forward goes (one forward at a time, device idle before)
            import time as _t
            st = eng._sp_state(plan)
            mod, _ = eng.conditioning([500.0, 500.0], pooled)
            torch.cuda.synchronize()
            t0 = _t.perf_counter(); eng._embed_local(plan, clips, eng._ctx, True, st); t1 = _t.perf_counter()
            eng._run_sp_list(plan, mod, st); t2 = _t.perf_counter()
            torch.cuda.synchronize(); t3 = _t.perf_counter()
            print(f"    one forward from an idle device: embed {1e3 * (t1 - t0):.2f} ms, list replay call {1e3 * (t2 - t1):.2f} ms "
                  f"({len(plan._sp_list[1])} entries), device drain {1e3 * (t3 - t2):.2f} ms", flush=True)
    del eng
    torch.cuda.empty_cache()
