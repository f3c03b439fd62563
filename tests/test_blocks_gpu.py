"""Block-level operators (pyramid_dit.FluxTransformerBlock / FluxSingleTransformerBlock / JointTransformerBlock) vs the
oracle's block functions at the RELEASED widths (d = 1920 / 30 heads, d = 1536 / 24 heads) with L ~ 0.6-2 k
(per-block goldens at d = 1920 / 1536).  Inputs are what the reference's transformer hands its
blocks: embedded states, temb, the [B,1,L,L] bool mask and the rotary table -- built here with the oracle's helpers.
Tolerance: one block rel-L2 <= 1e-2 (bf16 HIP vs fp32 oracle, same bf16-rounded weights)."""
import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu

SEQS = {"L608": [(1, 24, 40), (1, 24, 40)], "L2048": [(1, 48, 80), (1, 48, 80)]}


def _block_sd(shapes, prefix, seed):
    from pyflow_hip import synth
    full = synth.random_state_dict(shapes, seed=seed, std=0.02, lively=True)
    return round_sd({k[len(prefix):]: v for k, v in full.items() if k.startswith(prefix)})


def _geometry(clip_shapes, Lt, d, axes, seed):
    from oracle.flux_oracle import build_mask, rope_table, sequence_geometry
    g = torch.Generator().manual_seed(seed)
    clips = [torch.zeros(2, 16, *s) for s in clip_shapes]
    ids, frame_t = sequence_geometry(clips)
    enc_mask = torch.zeros(2, Lt, dtype=torch.long)
    enc_mask[0, :40] = 1
    enc_mask[1, :96] = 1
    mask = build_mask(enc_mask, frame_t)
    all_ids = torch.cat([torch.zeros(Lt, 3), ids], 0)
    if len(axes) == 1:                      # MMDiT: temporal RoPE over the whole head (mmdit:293-305)
        freqs = rope_table(all_ids[:, :1], axes)
    else:
        freqs = rope_table(all_ids, axes)
    L_img = ids.shape[0]
    x = torch.randn(2, L_img, d, generator=g).to(torch.bfloat16).float()
    c = torch.randn(2, Lt, d, generator=g).to(torch.bfloat16).float()
    temb = torch.randn(2, d, generator=g)
    return x, c, temb, enc_mask, mask, freqs


@pytest.mark.parametrize("seq", list(SEQS))
def test_flux_double_block_vs_oracle(seq):
    from pyflow_hip import synth
    from pyramid_dit import FluxTransformerBlock
    from oracle.flux_oracle import double_block
    cfg = dict(synth.MINIFLUX, num_layers=1, num_single_layers=0)
    p = "transformer_blocks.0."
    sd = _block_sd(synth.flux_param_shapes(cfg), p, 31)
    x, c, temb, enc_mask, mask, freqs = _geometry(SEQS[seq], 128, 1920, cfg["axes_dims_rope"], 32)
    c_ref, x_ref = double_block({p + k: v for k, v in sd.items()}, p, cfg, x, c, temb, mask, freqs)
    blk = FluxTransformerBlock(1920, 30, 64).load_state_dict(sd)
    c_out, x_out = blk(x.cuda(), c.cuda(), encoder_attention_mask=enc_mask, temb=temb, attention_mask=[mask],
                       hidden_length=[x.shape[1]], image_rotary_emb=[freqs[None]])
    ex, ec = rel_l2(x_out.float().cpu(), x_ref), rel_l2(c_out.float().cpu(), c_ref)
    print(f"FluxTransformerBlock d=1920 {seq}: image {ex:.3e} text {ec:.3e}")
    assert ex < 1e-2 and ec < 1e-2


@pytest.mark.parametrize("seq", list(SEQS))
def test_flux_single_block_vs_oracle(seq):
    from pyflow_hip import synth
    from pyramid_dit import FluxSingleTransformerBlock
    from oracle.flux_oracle import single_block
    cfg = dict(synth.MINIFLUX, num_layers=0, num_single_layers=1)
    p = "single_transformer_blocks.0."
    sd = _block_sd(synth.flux_param_shapes(cfg), p, 33)
    x, c, temb, enc_mask, mask, freqs = _geometry(SEQS[seq], 128, 1920, cfg["axes_dims_rope"], 34)
    h = torch.cat([c, x], dim=1)
    ref = single_block({p + k: v for k, v in sd.items()}, p, cfg, h, temb, mask, freqs)
    blk = FluxSingleTransformerBlock(1920, 30, 64).load_state_dict(sd)
    out = blk(h.cuda(), temb=temb, encoder_attention_mask=enc_mask, attention_mask=[mask], hidden_length=[x.shape[1]],
              image_rotary_emb=[freqs[None]])
    e = rel_l2(out.float().cpu(), ref)
    print(f"FluxSingleTransformerBlock d=1920 {seq}: {e:.3e}")
    assert e < 1e-2


@pytest.mark.parametrize("last", [False, True])
def test_mmdit_joint_block_vs_oracle(last):
    from pyflow_hip import synth
    from pyramid_dit import JointTransformerBlock
    from oracle.mmdit_oracle import joint_block
    cfg = dict(synth.SD3_MMDIT, num_layers=2)
    p_src = f"transformer_blocks.{1 if last else 0}."
    sd = _block_sd(synth.mmdit_param_shapes(cfg), p_src, 35)
    x, c, temb, enc_mask, mask, freqs = _geometry(SEQS["L608"], 128, 1536, [64], 36)
    p = "transformer_blocks.0."
    c_ref, x_ref = joint_block({p + k: v for k, v in sd.items()}, p, cfg, x, c, temb, mask, freqs, last)
    blk = JointTransformerBlock(1536, 24, 64, context_pre_only=last).load_state_dict(sd)
    c_out, x_out = blk(x.cuda(), c.cuda(), encoder_attention_mask=enc_mask, temb=temb, attention_mask=[mask],
                       hidden_length=[x.shape[1]], image_rotary_emb=[freqs[None]])
    ex = rel_l2(x_out.float().cpu(), x_ref)
    assert ex < 1e-2
    if last:
        assert c_out is None and c_ref is None
    else:
        assert rel_l2(c_out.float().cpu(), c_ref) < 1e-2


def test_foreign_mask_structure_is_rejected():
    from pyramid_dit import FluxSingleTransformerBlock
    from pyflow_hip import synth
    cfg = dict(synth.MINIFLUX, num_layers=0, num_single_layers=1)
    sd = _block_sd(synth.flux_param_shapes(cfg), "single_transformer_blocks.0.", 33)
    blk = FluxSingleTransformerBlock(1920, 30, 64).load_state_dict(sd)
    L, Lt = 256, 128
    m = torch.ones(2, 1, L, L, dtype=torch.bool)
    m[:, :, 10, 200] = False            # a hole in the image keys: not an interval mask
    with pytest.raises(NotImplementedError):
        blk(torch.zeros(2, L, 1920), temb=torch.zeros(2, 1920), encoder_attention_mask=torch.ones(2, Lt),
            attention_mask=m, image_rotary_emb=torch.zeros(L, 1, 32, 2, 2))
    with pytest.raises(NotImplementedError):
        blk.attn.set_processor(object())
