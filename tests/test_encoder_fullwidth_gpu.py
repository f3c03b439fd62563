"""VAE ENCODER parity at the released channel widths (128 / 256 / 512 / 512, two resnets per block, 512-channel mid
attention) against the CPU oracle -- SURVEY 8 row V-4, config C4's own encode shapes:
  * one 256 x 256 tile (what every tile of the tiled encode is),
  * one tiled 768 x 1280 frame (4 x 7 tiles at stride 192, blended; modeling_causal_vae.py:409-466) = the frame
    generate_i2v encodes in C4 (pyramid_dit_for_video_gen_pipeline.py:906-911).
Replaces CausalVaeEncoder.forward + quant_conv (modeling_enc_dec.py:55-198).  Tolerance: one forward of the
bf16 HIP path vs the fp32 oracle on the same bf16-rounded weights: rel-L2 <= 2e-2 on the posterior parameters."""
import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu


def _model():
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    ecfg = synth.VAE_ENC_DEFAULT
    assert tuple(ecfg["encoder_block_out_channels"]) == (128, 256, 512, 512) and tuple(ecfg["encoder_layers_per_block"]) == (2, 2, 2, 2)
    sd = round_sd(synth.random_state_dict(synth.vae_encoder_param_shapes(ecfg), seed=31, std=0.02, lively=True))
    cfg = dict(synth.VAE_DEFAULT)
    cfg.update(ecfg)
    ocfg = {k: ecfg[k] for k in ("encoder_block_out_channels", "encoder_layers_per_block", "encoder_spatial_down_sample",
                                 "encoder_temporal_down_sample")}
    return CausalVideoVAE(sd, cfg, "cuda"), sd, ocfg


def _image(h, w, seed):
    """smooth + noisy content in [-1, 1] (a flat random image would make every GroupNorm statistic alike)"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 6.28, h), torch.linspace(0, 9.42, w), indexing="ij")
    base = torch.stack([torch.sin(yy + 0.3 * c) * torch.cos(xx * (1 + 0.2 * c)) for c in range(3)])
    return (0.7 * base + 0.3 * torch.randn(3, h, w, generator=g)).clamp(-1, 1)[None, :, None]


def test_encoder_released_widths_one_tile_vs_oracle():
    from oracle.vae_oracle import vae_encode_moments
    vae, sd, ocfg = _model()
    x = _image(256, 256, 41)
    ref = vae_encode_moments(sd, ocfg, x)
    assert ref.shape == (1, 32, 1, 32, 32)
    got = vae.encode(x.cuda()).latent_dist.parameters.float().cpu()
    err = rel_l2(got, ref)
    print(f"VAE encoder, released widths, 256 x 256 tile: moments rel-L2 vs oracle {err:.3e}")
    assert got.shape == ref.shape and err < 2e-2
    # mean and logvar halves separately (the mean is what generate_i2v's mode() / sample() centre on)
    assert rel_l2(got[:, :16], ref[:, :16]) < 2e-2 and rel_l2(got[:, 16:], ref[:, 16:]) < 2e-2


def test_encoder_released_widths_tiled_768p_frame_vs_oracle():
    from oracle.vae_oracle import vae_encode_moments
    vae, sd, ocfg = _model()
    x = _image(768, 1280, 42)
    ref = vae_encode_moments(sd, ocfg, x, use_tiling=True, tile_sample_min_size=256)
    assert ref.shape == (1, 32, 1, 96, 160)
    vae.enable_tiling()
    got = vae.encode(x.cuda(), tile_sample_min_size=256).latent_dist.parameters.float().cpu()
    err = rel_l2(got, ref)
    print(f"VAE encoder, released widths, tiled 768 x 1280 frame (C4's encode): moments rel-L2 vs oracle {err:.3e}")
    assert got.shape == ref.shape and err < 2e-2
