"""Parity of the EXACT headline configuration against the CPU oracle: full DEPTH as well as full width.

tests/test_fullwidth_oracle_gpu.py cuts the models to one block of each kind; bench.py runs 8 double + 16 single
miniFLUX blocks at d = 1920 / 30 heads, so the accumulated bf16 error of 24 blocks is what these tests measure:
  * miniFLUX 8 + 16 blocks at the (unit 1, stage 0) L = 608 and (unit 5, stage 1) L = 3 008 sequences -- history clips,
    padded text, temporal-causal mask -- with the hidden state after EVERY block compared (per-block rel-L2 printed),
  * SD3 MMDiT at its full depth of 24 joint blocks (d = 1536 / 24 heads) at L = 608,
  * the un-tiled 768p decode (the shapes the context-parallel decode of config C5 launches: 96 x 160 latent, the
    15 360-token mid-block attention, full-frame GroupNorm) of one latent frame against the oracle, and of three
    latent frames (17 output frames) chunked vs un-chunked.
Tolerance (bf16 HIP vs fp32 oracle, weights rounded to bf16 on both sides): one forward / decode rel-L2 <= 2e-2.
The fp32 oracle needs ~5 s (L = 608) / ~25 s (L = 3 008) per forward on the GPU box's host cores.
"""
import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu

SEQS = {
    "u1s0_L608": [(1, 24, 40), (1, 24, 40)],
    "u5s1_L3008": [(3, 24, 40), (1, 24, 40), (1, 48, 80), (1, 48, 80)],
}


def _inputs(clip_shapes, Cenc, Cpool, seed=11, Lt=128):
    g = torch.Generator().manual_seed(seed)
    clips = [torch.randn(2, 16, *s, generator=g).to(torch.bfloat16).float() for s in clip_shapes]
    enc = torch.randn(2, Lt, Cenc, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(2, Lt, dtype=torch.long)
    mask[0, :40] = 1            # negative prompt: 40 valid tokens, positive: 96 (SURVEY 8d synthetic prompts)
    mask[1, :96] = 1
    pooled = torch.randn(2, Cpool, generator=g)
    return clips, enc, mask, pooled


@pytest.fixture(scope="module")
def miniflux():
    """full-size miniFLUX (1.97 B parameters), weights as bench.py draws them but `lively` (gains / biases perturbed so
    every term of every block matters), rounded to bf16 on both sides; the engine is shared by the two sequences"""
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    cfg = dict(synth.MINIFLUX)
    assert (cfg["num_layers"], cfg["num_single_layers"], cfg["num_attention_heads"], cfg["attention_head_dim"]) == (8, 16, 30, 64)
    sd = round_sd(synth.random_state_dict(synth.flux_param_shapes(cfg), seed=31, std=0.02, lively=True))
    eng = FluxEngine(sd, cfg, "cuda")
    return cfg, sd, eng


@pytest.mark.parametrize("seq", list(SEQS))
def test_miniflux_full_depth_forward_vs_oracle(miniflux, seq):
    from oracle.flux_oracle import flux_forward
    cfg, sd, eng = miniflux
    shapes = SEQS[seq]
    clips, enc, mask, pooled = _inputs(shapes, 4096, 768)
    t = torch.tensor([704.0, 704.0])
    with torch.no_grad():
        ref, inter = flux_forward(sd, cfg, clips, enc, mask, pooled, t, return_intermediates=True)
    plan = eng.make_plan(shapes, mask)
    assert plan.L == int(seq.split("L")[1])
    clips_d = [c.cuda() for c in clips]
    ctx = eng.encode_context(enc)
    dbg = {"blocks": []}
    eng.skip_dead_rows = False          # every row of every block is compared
    eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx, debug=dbg)
    eng.skip_dead_rows = True
    assert len(dbg["blocks"]) == len(inter["blocks"]) == 24
    errs = [rel_l2(h.float().cpu(), r) for h, r in zip(dbg["blocks"], inter["blocks"])]
    print(f"miniFLUX 8+16 blocks d=1920 H=30 {seq}: hidden-state rel-L2 after each block:",
          " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) < 2e-2
    # production path: launch list / hipGraph, dead rows of the last block skipped, text stream on the side stream
    out = eng.forward(clips_d, enc, mask, pooled, t).cpu()
    err = rel_l2(out, ref)
    print(f"miniFLUX 8+16 blocks {seq}: forward rel-L2 vs oracle {err:.3e}")
    assert out.shape == ref.shape and err < 2e-2


def test_mmdit_full_depth_forward_vs_oracle():
    """SD3-style MMDiT, 24 joint blocks (the last one context_pre_only), d = 1536 / 24 heads, at L = 608."""
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    from oracle.mmdit_oracle import mmdit_forward
    cfg = dict(synth.SD3_MMDIT)
    assert cfg["num_layers"] == 24
    sd = round_sd(synth.mmdit_state_dict(cfg, seed=32, std=0.02, lively=True))
    sd["pos_embed.pos_embed"] = synth.mmdit_state_dict(dict(cfg, num_layers=1), seed=32)["pos_embed.pos_embed"]    # fp32 sincos table
    shapes = SEQS["u1s0_L608"]
    clips, enc, mask, pooled = _inputs(shapes, 4096, 2048, seed=12)
    t = torch.tensor([386.0, 386.0])
    with torch.no_grad():
        ref = mmdit_forward(sd, cfg, clips, enc, mask, pooled, t)
    eng = FluxEngine(sd, cfg, "cuda")
    assert eng.w.mmdit and len(eng.w.dbl) == 24 and eng.w.dbl[-1]["pre_only"]
    out = eng.forward([c.cuda() for c in clips], enc, mask, pooled, t).cpu()
    err = rel_l2(out, ref)
    print(f"MMDiT 24 blocks d=1536 H=24 L=608: forward rel-L2 vs oracle {err:.3e}")
    assert out.shape == ref.shape and err < 2e-2


def _vae(seed=33):
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    cfg = synth.VAE_DEFAULT
    sd = round_sd(synth.random_state_dict(synth.vae_decoder_param_shapes(cfg), seed=seed, std=0.02, lively=True))
    ocfg = dict(decoder_block_out_channels=cfg["block_out_channels"], decoder_layers_per_block=cfg["layers_per_block"],
                decoder_spatial_up_sample=cfg["spatial_up_sample"], decoder_temporal_up_sample=cfg["temporal_up_sample"])
    return CausalVideoVAE(sd, cfg, "cuda"), sd, ocfg


def test_vae_untiled_768p_decode_vs_oracle():
    """The un-tiled decode at the headline resolution (what a context-parallel rank runs): latent 96 x 160,
    released channel widths.  One latent frame -> one 768 x 1280 frame against the fp32 oracle (every conv at its
    full-frame M, GroupNorm over the whole frame, the mid-block attention over 15 360 tokens)."""
    from oracle.vae_oracle import vae_decode
    vae, sd, ocfg = _vae()
    z = torch.randn(1, 16, 1, 96, 160, generator=torch.Generator().manual_seed(34)).to(torch.bfloat16).float()
    with torch.no_grad():
        ref = vae_decode(sd, ocfg, z)
    assert ref.shape == (1, 3, 1, 768, 1280)
    out = vae.decode(z.cuda(), temporal_chunk=False).sample.float().cpu()
    err = rel_l2(out, ref)
    print(f"VAE un-tiled 768p, 1 latent frame: decode rel-L2 vs oracle {err:.3e}")
    assert out.shape == ref.shape and err < 2e-2


def _per_frame(a, b):
    return [rel_l2(a[:, :, f], b[:, :, f]) for f in range(a.shape[2])]


def test_vae_untiled_768p_chunked_vs_unchunked():
    """three latent frames -> 17 frames of 768 x 1280, un-tiled: the chunked schedule ([2] + [1] latent frames, cache slots
    between the chunks) against the single pass, frame by frame.  The two are the same arithmetic in exact numbers
    (5e-6 in fp32) but not bit-identical in bf16: the launch shapes differ (split-K of the skinny
    mid-block GEMMs, atomics order of the GroupNorm sums), fp32-ulp differences flip bf16 roundings and the flips
    de-correlate over ~60 layers up to the bf16 noise floor (the distance of either to the fp32 oracle, 1.3e-2).  What
    the test pins is that NO frame is worse than that floor -- a wrong cache frame at the chunk boundary is an O(1)
    error in frames 9..16 -- and that frame 0 is the one-latent decode's frame (causality; pinned to the oracle above)."""
    vae, _, _ = _vae()
    vae.chunk_coalesce = 1               # the reference's own schedule: window_size latent frames per chunk
    z = torch.randn(1, 16, 3, 96, 160, generator=torch.Generator().manual_seed(34)).to(torch.bfloat16).float().cuda()
    full = vae.decode(z, temporal_chunk=False).sample.float().cpu()
    assert full.shape == (1, 3, 17, 768, 1280)
    assert torch.isfinite(full).all()
    chunked = vae.decode(z, temporal_chunk=True, window_size=1).sample.float().cpu()
    pf = _per_frame(chunked, full)
    print("VAE un-tiled 768p, 3 latent frames, chunked vs single pass, rel-L2 per frame:", " ".join(f"{e:.2e}" for e in pf))
    assert max(pf) < 2e-2
    one = vae.decode(z[:, :, :1].contiguous(), temporal_chunk=False).sample.float().cpu()
    assert rel_l2(full[:, :, :1], one) < 2e-2 and rel_l2(chunked[:, :, :1], one) < 2e-2


def test_vae_released_width_three_latents_chunked_and_unchunked_vs_oracle():
    """the same comparison where the fp32 oracle is affordable for all 17 frames (32 x 32 latent, released widths):
    chunked and single-pass decodes are each within the bf16 tolerance of the oracle in EVERY frame, i.e. the chunk
    boundary (frames 9..16 come from the second chunk's cache slots) is as exact as the interior."""
    from oracle.vae_oracle import vae_decode
    vae, sd, ocfg = _vae()
    vae.chunk_coalesce = 1
    z = torch.randn(1, 16, 3, 32, 32, generator=torch.Generator().manual_seed(35)).to(torch.bfloat16).float()
    with torch.no_grad():
        ref = vae_decode(sd, ocfg, z)
    assert ref.shape == (1, 3, 17, 256, 256)
    full = vae.decode(z.cuda(), temporal_chunk=False).sample.float().cpu()
    chunked = vae.decode(z.cuda(), temporal_chunk=True, window_size=1).sample.float().cpu()
    e_full, e_chunk, e_between = _per_frame(full, ref), _per_frame(chunked, ref), _per_frame(chunked, full)
    print("single pass vs oracle per frame:", " ".join(f"{e:.2e}" for e in e_full))
    print("chunked     vs oracle per frame:", " ".join(f"{e:.2e}" for e in e_chunk))
    print("chunked vs single pass        :", " ".join(f"{e:.2e}" for e in e_between))
    assert max(e_full) < 2e-2 and max(e_chunk) < 2e-2
