"""GPU parity: whole miniFLUX DiT forward (HIP path, bf16) vs the CPU fp32 oracle, same bf16-rounded weights.
Tolerance: one DiT forward rel-L2 <= 2e-2."""
import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu


def _inputs(clip_shapes, B=2, Lt=16, C=32, Cp=16, seed=7):
    g = torch.Generator().manual_seed(seed)
    clips = [torch.randn(B, 16, *s, generator=g) for s in clip_shapes]
    enc = torch.randn(B, Lt, C, generator=g)
    mask = torch.zeros(B, Lt, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(B, Cp, generator=g)
    return clips, enc, mask, pooled


@pytest.mark.parametrize("clip_shapes", [
    [(1, 16, 32)],
    [(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)],
])
def test_tiny_forward_vs_oracle(clip_shapes):
    from pyflow_hip.flux import FluxEngine
    from pyflow_hip import synth
    from oracle.flux_oracle import flux_forward
    cfg = synth.TINY_FLUX
    sd = round_sd(synth.random_state_dict(synth.flux_param_shapes(cfg), seed=3, std=0.05, lively=True))
    clips, enc, mask, pooled = _inputs(clip_shapes)
    clips = [c.to(torch.bfloat16).float() for c in clips]
    enc = enc.to(torch.bfloat16).float()
    t = torch.tensor([704.0, 704.0])
    ref, inter = flux_forward(sd, cfg, clips, enc, mask, pooled, t, return_intermediates=True)
    eng = FluxEngine(sd, cfg, "cuda")
    dbg = {}
    clips_d = [c.cuda() for c in clips]
    plan = eng.make_plan(clip_shapes, mask)
    ctx = eng.encode_context(enc)
    eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx, debug=dbg)
    Lt = mask.shape[1]
    h0 = dbg["hidden0"].float().cpu()
    assert rel_l2(h0[:, :Lt], inter["c0"]) < 1e-2
    assert rel_l2(h0[:, Lt:], inter["x0"]) < 1e-2
    hd0 = dbg["hidden_d0"].float().cpu()
    assert rel_l2(hd0[:, Lt:], inter["x_after_double0"]) < 1.5e-2
    assert rel_l2(hd0[:, :Lt], inter["c_after_double0"]) < 1.5e-2
    # the last block only updates the current frame's rows (the only ones that reach the output)
    n_cur = plan.n_cur
    assert rel_l2(dbg["hidden_final"].float().cpu()[:, -n_cur:], inter["x_final"][:, -n_cur:]) < 2e-2
    out = eng.forward(clips_d, enc, mask, pooled, t).cpu()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-2
    # ... and that restriction changes nothing: same velocity tokens with the full last block
    v_skip = eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx).clone()
    eng.skip_dead_rows = False
    dbg2 = {}
    v_full = eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx, debug=dbg2).clone()
    assert torch.equal(v_skip, v_full)
    assert rel_l2(dbg2["hidden_final"].float().cpu()[:, Lt:], inter["x_final"]) < 2e-2


def test_golden_fixture_forward():
    """Committed fixture produced by the UNMODIFIED reference (oracle/gen_golden.py)."""
    import os
    from pyflow_hip.flux import FluxEngine
    path = os.path.join(os.path.dirname(__file__), "golden", "flux_tiny_forward.pt")
    from pyflow_hip import synth
    g = torch.load(path)
    sd = round_sd(synth.random_state_dict(synth.flux_param_shapes(g["cfg"]), seed=g["weight_seed"], std=0.05, lively=True))
    eng = FluxEngine(sd, g["cfg"], "cuda")
    out = eng.forward([c.cuda() for c in g["clips"]], g["enc"], g["mask"], g["pooled"], g["timestep"]).cpu()
    assert rel_l2(out, g["out"]) < 2e-2


def _mmdit_sd(cfg, seed):
    from pyflow_hip import synth
    sd = round_sd(synth.mmdit_state_dict(cfg, seed=seed, std=0.05, lively=True))
    sd["pos_embed.pos_embed"] = synth.mmdit_state_dict(cfg, seed=seed)["pos_embed.pos_embed"]       # fp32 table
    return sd


@pytest.mark.parametrize("clip_shapes", [
    [(1, 16, 32)],
    [(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)],
])
def test_mmdit_forward_vs_oracle(clip_shapes):
    """SD3-style variant (config C4): PatchEmbed3D conv + cropped / interpolated sincos rows, temporal RoPE,
    context_pre_only last block, QK-norm eps 1e-5."""
    from pyflow_hip.flux import FluxEngine
    from pyflow_hip import synth
    from oracle.mmdit_oracle import mmdit_forward
    cfg = synth.tiny_mmdit_cfg()
    sd = _mmdit_sd(cfg, 9)
    clips, enc, mask, pooled = _inputs(clip_shapes)
    clips = [c.to(torch.bfloat16).float() for c in clips]
    enc = enc.to(torch.bfloat16).float()
    t = torch.tensor([704.0, 704.0])
    ref, inter = mmdit_forward(sd, cfg, clips, enc, mask, pooled, t, return_intermediates=True)
    eng = FluxEngine(sd, cfg, "cuda")
    assert eng.w.mmdit and len(eng.w.sgl) == 0 and eng.w.dbl[-1]["pre_only"]
    dbg = {}
    clips_d = [c.cuda() for c in clips]
    plan = eng.make_plan(clip_shapes, mask)
    ctx = eng.encode_context(enc)
    eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx, debug=dbg)
    Lt = mask.shape[1]
    h0 = dbg["hidden0"].float().cpu()
    assert rel_l2(h0[:, Lt:], inter["x0"]) < 1e-2
    assert rel_l2(dbg["hidden_d0"].float().cpu()[:, Lt:], inter["x_after_block0"]) < 1.5e-2
    n_cur = plan.n_cur
    assert rel_l2(dbg["hidden_final"].float().cpu()[:, -n_cur:], inter["x_final"][:, -n_cur:]) < 2e-2
    out = eng.forward(clips_d, enc, mask, pooled, t).cpu()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-2
    v_skip = eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx).clone()
    eng.skip_dead_rows = False
    dbg2 = {}
    v_full = eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx, debug=dbg2).clone()
    assert torch.equal(v_skip, v_full)
    assert rel_l2(dbg2["hidden_final"].float().cpu()[:, Lt:], inter["x_final"]) < 2e-2


def test_mmdit_golden_fixture_forward():
    """Committed fixture produced by the UNMODIFIED reference PyramidDiffusionMMDiT (oracle/gen_golden.py)."""
    import os
    from pyflow_hip.flux import FluxEngine
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "mmdit_tiny_forward.pt"))
    eng = FluxEngine(_mmdit_sd(g["cfg"], g["weight_seed"]), g["cfg"], "cuda")
    out = eng.forward([c.cuda() for c in g["clips"]], g["enc"], g["mask"], g["pooled"], g["timestep"]).cpu()
    assert rel_l2(out, g["out"]) < 2e-2
