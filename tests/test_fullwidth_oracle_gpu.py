"""Parity at the BASELINE widths against the CPU oracle.

The tiny-config tests never reach the kernels the benchmark spends its time in (the 256-row MFMA GEMMs, H = 30
attention with a non-power-of-two head count, the 512/256/128-channel implicit-GEMM convs).  Here the models have the
RELEASED widths (miniFLUX d = 1920 / 30 heads, SD3 MMDiT d = 1536 / 24 heads, VAE 512/256/128 channels) with the
depth cut to one block of each kind, at sequence lengths the fp32 oracle finishes in seconds:
  * miniFLUX (unit 1, stage 0)  L = 608   and   (unit 5, stage 1)  L = 3 008  -- history clips + padded text,
  * SD3 MMDiT (one joint block + the context_pre_only last block) at the same two sequences,
  * VAE decode of one 32 x 32-latent tile, 2 latent frames (-> 9 frames of 256 x 256) at VAE_DEFAULT widths.
Tolerance (SURVEY 8c, bf16 HIP vs fp32 oracle on the same bf16-rounded weights): one forward / decode rel-L2 <= 2e-2.
Each test also asserts, through pf_gemm_which, that the shapes it ran dispatch to the large-tile MFMA kernels the
benchmark's profile is made of (non-zero = not the 128 x 128 fallback kernel).
"""
import ctypes as C

import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu

SEQS = {
    "u1s0_L608": [(1, 24, 40), (1, 24, 40)],
    "u5s1_L3008": [(3, 24, 40), (1, 24, 40), (1, 48, 80), (1, 48, 80)],
}


def _inputs(clip_shapes, C_lat, Cenc, Cpool, seed=11, Lt=128):
    g = torch.Generator().manual_seed(seed)
    clips = [torch.randn(2, C_lat, *s, generator=g).to(torch.bfloat16).float() for s in clip_shapes]
    enc = torch.randn(2, Lt, Cenc, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(2, Lt, dtype=torch.long)
    mask[0, :40] = 1            # negative prompt: 40 valid tokens, positive: 96 (SURVEY 8d synthetic prompts)
    mask[1, :96] = 1
    pooled = torch.randn(2, Cpool, generator=g)
    return clips, enc, mask, pooled


def _which(M, batch, N, K):
    from pyflow_hip import lib
    return lib.load().pf_gemm_which(C.c_int(M), C.c_int(batch), C.c_int(N), C.c_int(K))


@pytest.mark.parametrize("seq", list(SEQS))
def test_miniflux_full_width_forward_vs_oracle(seq):
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    from oracle.flux_oracle import flux_forward
    cfg = dict(synth.MINIFLUX, num_layers=1, num_single_layers=1)
    assert cfg["num_attention_heads"] == 30 and cfg["attention_head_dim"] == 64
    sd = round_sd(synth.random_state_dict(synth.flux_param_shapes(cfg), seed=21, std=0.02, lively=True))
    shapes = SEQS[seq]
    clips, enc, mask, pooled = _inputs(shapes, 16, 4096, 768)
    t = torch.tensor([704.0, 704.0])
    ref, inter = flux_forward(sd, cfg, clips, enc, mask, pooled, t, return_intermediates=True)
    eng = FluxEngine(sd, cfg, "cuda")
    plan = eng.make_plan(shapes, mask)
    L = plan.L
    assert L == int(seq.split("L")[1])
    d = 1920
    dbg = {}
    clips_d = [c.cuda() for c in clips]
    ctx = eng.encode_context(enc)
    eng.skip_dead_rows = False          # every row of every block is compared below
    eng.forward_tokens(plan, clips_d, [704.0, 704.0], pooled, ctx, debug=dbg)
    Lt = 128
    assert rel_l2(dbg["hidden_d0"].float().cpu()[:, Lt:], inter["x_after_double0"]) < 1.5e-2
    assert rel_l2(dbg["hidden_d0"].float().cpu()[:, :Lt], inter["c_after_double0"]) < 1.5e-2
    assert rel_l2(dbg["hidden_final"].float().cpu()[:, Lt:], inter["x_final"]) < 2e-2
    eng.skip_dead_rows = True
    out = eng.forward(clips_d, enc, mask, pooled, t).cpu()
    assert out.shape == ref.shape
    err = rel_l2(out, ref)
    print(f"miniFLUX d=1920 H=30 {seq}: forward rel-L2 vs oracle {err:.3e}")
    assert err < 2e-2
    # the kernels behind these shapes: fused K|V|Q|MLP projection (N = 7d) and the MLP of the double blocks run the
    # 256-row MFMA kernel at both lengths; the d-wide projections join at L = 3 008
    assert _which(L, 2, 7 * d, d) != 0
    if L >= 3008:
        assert _which(L - Lt, 2, 4 * d, d) != 0
        assert _which(L, 2, d, 5 * d) != 0
        assert _which(L - Lt, 2, d, 4 * d) != 0


@pytest.mark.parametrize("seq", list(SEQS))
def test_mmdit_full_width_forward_vs_oracle(seq):
    """SD3-style MMDiT at d = 1536 / 24 heads: one full joint block + the context_pre_only last block."""
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    from oracle.mmdit_oracle import mmdit_forward
    cfg = dict(synth.SD3_MMDIT, num_layers=2)
    sd = round_sd(synth.mmdit_state_dict(cfg, seed=22, std=0.02, lively=True))
    sd["pos_embed.pos_embed"] = synth.mmdit_state_dict(cfg, seed=22)["pos_embed.pos_embed"]      # fp32 sincos table
    shapes = SEQS[seq]
    clips, enc, mask, pooled = _inputs(shapes, 16, 4096, 2048, seed=12)
    t = torch.tensor([386.0, 386.0])
    ref = mmdit_forward(sd, cfg, clips, enc, mask, pooled, t)
    eng = FluxEngine(sd, cfg, "cuda")
    assert eng.w.mmdit and eng.w.d == 1536 and eng.w.H == 24 and eng.w.dbl[-1]["pre_only"]
    out = eng.forward([c.cuda() for c in clips], enc, mask, pooled, t).cpu()
    err = rel_l2(out, ref)
    print(f"MMDiT d=1536 H=24 {seq}: forward rel-L2 vs oracle {err:.3e}")
    assert out.shape == ref.shape and err < 2e-2


def test_vae_default_width_tile_decode_vs_oracle():
    """One 32 x 32-latent tile, 2 latent frames, at the released channel widths (512/512/256/128): every conv of the
    decoder at its real K and N, the 512-channel mid attention, GroupNorm over 16/8/4-channel groups."""
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_decode
    cfg = synth.VAE_DEFAULT
    sd = round_sd(synth.random_state_dict(synth.vae_decoder_param_shapes(cfg), seed=23, std=0.02, lively=True))
    ocfg = dict(decoder_block_out_channels=cfg["block_out_channels"], decoder_layers_per_block=cfg["layers_per_block"],
                decoder_spatial_up_sample=cfg["spatial_up_sample"], decoder_temporal_up_sample=cfg["temporal_up_sample"])
    z = torch.randn(1, 16, 2, 32, 32, generator=torch.Generator().manual_seed(24)).to(torch.bfloat16).float()
    ref = vae_decode(sd, ocfg, z)
    assert ref.shape == (1, 3, 9, 256, 256)
    vae = CausalVideoVAE(sd, cfg, "cuda")
    out = vae.decode(z.cuda(), temporal_chunk=True, window_size=1).sample.float().cpu()
    err = rel_l2(out, ref)
    print(f"VAE default widths, 32x32x2 latent tile: decode rel-L2 vs oracle {err:.3e}")
    assert out.shape == ref.shape and err < 2e-2
    out2 = vae.decode(z.cuda(), temporal_chunk=False).sample.float().cpu()
    assert rel_l2(out2, ref) < 2e-2
    # conv_out (128 -> 3 channels) runs the narrow-N kernel (convnarrow.hip) by default: same result as the padded
    # implicit GEMM up to the fp32 summation tree (a bf16 ulp here and there), chunked (T = 1 and 8) and un-chunked (T = 9)
    from pyflow_hip import ops
    ops.gemm_set_policy(-3)
    try:
        wide = vae.decode(z.cuda(), temporal_chunk=True, window_size=1).sample.float().cpu()
        wide2 = vae.decode(z.cuda(), temporal_chunk=False).sample.float().cpu()
    finally:
        ops.gemm_set_policy(3)
    assert rel_l2(wide, ref) < 2e-2
    for a, b in ((out, wide), (out2, wide2)):
        d = (a - b).abs()
        assert d.max() <= 2 ** -6 * max(b.abs().max().item(), 1.0), d.max()
        assert rel_l2(a, b) < 2e-3
        assert not torch.equal(a, torch.zeros_like(a))
    # launch shapes of this decode: the full-resolution 128-filter convs (M = frames x 256 x 256 pixels) and the
    # 256/512-filter convs below them run the 256-row MFMA conv kernels
    assert _which(8 * 256 * 256, 1, 128, 27 * 128) != 0
    assert _which(4 * 128 * 128, 1, 512, 27 * 256) != 0
