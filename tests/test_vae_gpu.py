"""GPU parity of the HIP CausalVideoVAE decode vs the CPU fp32 oracle and the committed reference fixture."""
import os

import pytest
import torch

from util import rel_l2, round_sd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_tiny_decode.pt")


def _sd(cfg, seed=5):
    from pyflow_hip import synth
    return round_sd(synth.random_state_dict(synth.vae_decoder_param_shapes(cfg), seed=seed, std=0.05, lively=True))


def _oracle_cfg(cfg):
    return dict(decoder_block_out_channels=cfg["block_out_channels"], decoder_layers_per_block=cfg["layers_per_block"],
                decoder_spatial_up_sample=cfg["spatial_up_sample"], decoder_temporal_up_sample=cfg["temporal_up_sample"])


@pytest.mark.parametrize("coalesce", [1, 4])
@pytest.mark.parametrize("T,h,w,window", [(1, 6, 10, 1), (3, 6, 10, 1), (4, 8, 8, 2), (11, 6, 6, 1)])
def test_decode_vs_oracle(T, h, w, window, coalesce):
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_decode
    cfg = synth.TINY_VAE
    sd = _sd(cfg)
    z = torch.randn(1, 16, T, h, w, generator=torch.Generator().manual_seed(2))
    ref = vae_decode(sd, _oracle_cfg(cfg), z)
    vae = CausalVideoVAE(sd, cfg, "cuda")
    vae.chunk_coalesce = coalesce        # 1 = the reference's chunk schedule literally, 4 = the default launch shapes
    out = vae.decode(z.cuda(), temporal_chunk=True, window_size=window).sample.float().cpu()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 3e-2
    out2 = vae.decode(z.cuda(), temporal_chunk=False).sample.float().cpu()     # un-chunked == chunked (exact streaming)
    assert rel_l2(out2, ref) < 3e-2


def test_golden_fixture_plain_and_tiled():
    from pyflow_hip.vae import CausalVideoVAE
    g = torch.load(GOLD)
    cfg = g["cfg"]
    vae = CausalVideoVAE(_sd(cfg, g["weight_seed"]), cfg, "cuda")
    out = vae.decode(g["z"].cuda(), temporal_chunk=True, window_size=1).sample.float().cpu()
    assert rel_l2(out, g["out"].float()) < 3e-2
    vae.enable_tiling()
    out_t = vae.decode(g["z"].cuda(), temporal_chunk=True, window_size=1, tile_sample_min_size=32).sample.float().cpu()
    assert out_t.shape == g["out_tiled32"].shape
    assert rel_l2(out_t, g["out_tiled32"].float()) < 3e-2


def test_uint8_frames_vs_oracle():
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    from oracle.vae_oracle import vae_decode, to_uint8_frames
    cfg = synth.TINY_VAE
    sd = _sd(cfg)
    z = torch.randn(1, 16, 2, 8, 12, generator=torch.Generator().manual_seed(3))
    ref = to_uint8_frames(vae_decode(sd, _oracle_cfg(cfg), z, use_tiling=True, tile_sample_min_size=32))
    vae = CausalVideoVAE(sd, cfg, "cuda")
    vae.enable_tiling()
    u8 = vae.decode_to_uint8(z.cuda(), window_size=1, tile_sample_min_size=32).cpu()
    assert u8.shape == ref.shape and u8.dtype == torch.uint8
    diff = (u8.int() - ref.int()).abs()
    assert diff.float().mean() < 2.0 and diff.max() <= 12      # bf16 activations vs fp32 oracle on a 0..255 scale
