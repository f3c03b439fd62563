"""GPU parity of the image-to-video path (config C4): VAE encoder (single frame, plain + tiled) and generate_i2v with
the SD3-style MMDiT, against the CPU oracle and the fixture produced by the UNMODIFIED reference's generate_i2v."""
import os

import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "i2v_tiny.pt")


def _vae_sd(g):
    from pyflow_hip import synth
    sd = round_sd(synth.random_state_dict(synth.vae_decoder_param_shapes(g["vae_cfg"]), seed=g["vae_weight_seed"], std=0.05, lively=True))
    sd.update(round_sd(synth.random_state_dict(synth.vae_encoder_param_shapes(g["vae_enc_cfg"]), seed=g["enc_weight_seed"], std=0.05, lively=True)))
    return sd


def _vae_cfg(g):
    cfg = dict(g["vae_cfg"])
    cfg.update(g["vae_enc_cfg"])
    return cfg


def _img(g):
    return (g["image"].permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5


def test_encoder_vs_reference_fixture_and_oracle():
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_encode_moments
    g = torch.load(GOLD)
    sd = _vae_sd(g)
    vae = CausalVideoVAE(sd, _vae_cfg(g), "cuda")
    x = _img(g)[None, :, None]
    post = vae.encode(x.cuda()).latent_dist
    ocfg = dict(encoder_block_out_channels=g["vae_enc_cfg"]["encoder_block_out_channels"],
                encoder_layers_per_block=g["vae_enc_cfg"]["encoder_layers_per_block"],
                encoder_spatial_down_sample=g["vae_enc_cfg"]["encoder_spatial_down_sample"],
                encoder_temporal_down_sample=g["vae_enc_cfg"]["encoder_temporal_down_sample"])
    ref = vae_encode_moments(sd, ocfg, x)
    assert post.parameters.shape == ref.shape
    # one encoder forward, bf16 HIP vs fp32 oracle / reference fixture: the forward bound of SURVEY 8c (2e-2)
    e_or, e_fx = rel_l2(post.parameters.float().cpu(), ref), rel_l2(post.parameters.float().cpu(), g["moments"].float())
    vae.enable_tiling()
    post_t = vae.encode(x.cuda(), tile_sample_min_size=32).latent_dist
    e_tl = rel_l2(post_t.parameters.float().cpu(), g["moments_tiled32"].float())
    print(f"tiny encoder: vs oracle {e_or:.3e}, vs reference fixture {e_fx:.3e}, tiled vs fixture {e_tl:.3e}")
    assert e_or < 2e-2 and e_fx < 2e-2 and e_tl < 2e-2
    # sample / mode
    eps = g["posterior_eps"]
    z = post.sample(eps=eps.cuda())
    mean, logvar = ref.chunk(2, dim=1)
    assert rel_l2(z.float().cpu(), mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * eps) < 3e-2
    assert torch.equal(post.mode(), post.mean)


def test_generate_i2v_vs_reference_fixture():
    from pyflow_hip import synth
    from pyflow_hip.pipeline import PyramidDiTForVideoGeneration
    from oracle.ref_harness import NoiseStream
    g = torch.load(GOLD)
    dsd = round_sd(synth.mmdit_state_dict(g["dit_cfg"], seed=g["dit_weight_seed"], std=0.05, lively=True))
    dsd["pos_embed.pos_embed"] = synth.mmdit_state_dict(g["dit_cfg"], seed=g["dit_weight_seed"])["pos_embed.pos_embed"]
    pipe = PyramidDiTForVideoGeneration(dit_state_dict=dsd, dit_config=g["dit_cfg"], vae_state_dict=_vae_sd(g),
                                        vae_config=_vae_cfg(g), model_name="pyramid_mmdit")
    pipe.block_noise_fn = NoiseStream(g["noise_seed"]).block_noise
    e, m, p = g["prompt_embeds"], g["prompt_mask"], g["pooled"]
    lat = pipe.generate_i2v(prompt_embeds=(e[1:2], m[1:2], p[1:2], e[0:1], m[0:1], p[0:1]), input_image=_img(g),
                            temp=g["temp"], num_inference_steps=g["steps"], guidance_scale=g["guidance"],
                            video_guidance_scale=g["video_guidance"], generator=torch.Generator().manual_seed(g["latent_seed"]),
                            output_type="latent", posterior_noise=g["posterior_eps"])
    assert lat.shape == g["latents"].shape
    err = rel_l2(lat.float().cpu(), g["latents"])
    print("i2v trajectory rel-L2 vs reference fixture:", err)
    assert err < 5e-2
    # frames come out too (decode of image latent + generated units)
    pipe.vae.enable_tiling()
    frames = pipe.generate_i2v(prompt_embeds=(e[1:2], m[1:2], p[1:2], e[0:1], m[0:1], p[0:1]), input_image=_img(g),
                               temp=g["temp"], num_inference_steps=g["steps"], guidance_scale=g["guidance"],
                               video_guidance_scale=g["video_guidance"], generator=torch.Generator().manual_seed(g["latent_seed"]),
                               output_type="uint8", posterior_noise=g["posterior_eps"])
    assert frames.shape == (1 + 8 * (g["temp"] - 1), 64, 128, 3) and frames.dtype == torch.uint8


@pytest.mark.parametrize("T", [9, 17])
def test_clip_encode_vs_oracle(T):
    """multi-frame encode in one causal pass (temporal stride-2 downsamplers, full 3-tap filters), plain and tiled"""
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_encode_moments
    g = torch.load(GOLD)
    sd = _vae_sd(g)
    vae = CausalVideoVAE(sd, _vae_cfg(g), "cuda")
    x = torch.randn(1, 3, T, 32, 48, generator=torch.Generator().manual_seed(8)).clamp(-1, 1)
    ocfg = {k: g["vae_enc_cfg"][k] for k in ("encoder_block_out_channels", "encoder_layers_per_block",
                                              "encoder_spatial_down_sample", "encoder_temporal_down_sample")}
    ref = vae_encode_moments(sd, ocfg, x)
    post = vae.encode(x.cuda()).latent_dist
    assert post.parameters.shape == ref.shape == (1, 32, 1 + (T - 1) // 8, 4, 6)
    assert rel_l2(post.parameters.float().cpu(), ref) < 3e-2
    vae.enable_tiling()
    ref_t = vae_encode_moments(sd, ocfg, x, use_tiling=True, tile_sample_min_size=32)
    post_t = vae.encode(x.cuda(), tile_sample_min_size=32).latent_dist
    assert rel_l2(post_t.parameters.float().cpu(), ref_t) < 3e-2
    # the single-frame fast path (last temporal tap only) agrees with the general path on a 1-frame clip
    one = vae.encode(x[:, :, :1].cuda(), tile_sample_min_size=32).latent_dist.parameters
    ref1 = vae_encode_moments(sd, ocfg, x[:, :, :1], use_tiling=True, tile_sample_min_size=32)
    assert rel_l2(one.float().cpu(), ref1) < 3e-2


@pytest.mark.parametrize("T,win", [(33, 8), (33, 16), (41, 16), (9, 8)])
def test_chunk_encode(T, win):
    """sliding-window chunk_encode (modeling_causal_vae.py:310-341): window + 1 frames, then `window` per call (a short
    remainder last), caches carried in the conv buffers' slots.  Same arithmetic per output element as the single pass,
    so the two HIP results agree to rounding; both are checked against the oracle (which is pinned to the reference's
    chunk_encode in tests/test_oracle_vs_reference.py)."""
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_encode_moments
    g = torch.load(GOLD)
    sd = _vae_sd(g)
    vae = CausalVideoVAE(sd, _vae_cfg(g), "cuda")
    x = torch.randn(1, 3, T, 32, 48, generator=torch.Generator().manual_seed(9)).clamp(-1, 1)
    ocfg = {k: g["vae_enc_cfg"][k] for k in ("encoder_block_out_channels", "encoder_layers_per_block",
                                              "encoder_spatial_down_sample", "encoder_temporal_down_sample")}
    ref = vae_encode_moments(sd, ocfg, x)
    one = vae.encode(x.cuda()).latent_dist.parameters
    chk = vae.encode(x.cuda(), temporal_chunk=True, window_size=win).latent_dist.parameters
    assert chk.shape == ref.shape == (1, 32, 1 + (T - 1) // 8, 4, 6)
    assert rel_l2(chk.float().cpu(), ref) < 3e-2
    assert rel_l2(chk.float().cpu(), one.float().cpu()) < 2e-3        # GroupNorm sums are atomics: order may differ
    vae.enable_tiling()
    ref_t = vae_encode_moments(sd, ocfg, x, use_tiling=True, tile_sample_min_size=32)
    chk_t = vae.encode(x.cuda(), temporal_chunk=True, window_size=win, tile_sample_min_size=32).latent_dist.parameters
    assert rel_l2(chk_t.float().cpu(), ref_t) < 3e-2
