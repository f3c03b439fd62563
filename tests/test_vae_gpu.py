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


@pytest.mark.parametrize("co", [1, 3, 4, 5, 8])
def test_narrow_conv_every_channel_count_vs_conv3d(co):
    """conv_narrow_kernel (csrc/convnarrow.hip) multiplies filter rows 0..7: any 3x3x3, 128-input-channel CausalConv3d
    with 1..8 output channels stored at pitch 8 is exact through it -- against torch's fp32 conv3d with the causal
    padding of modeling_causal_conv.py:116-146 and against the generic implicit-GEMM path on the same descriptor."""
    import torch.nn.functional as F
    from pyflow_hip import ops
    from pyflow_hip.vae import PBuf, ConvW, conv
    T, H, W, Ci = 3, 32, 48, 128
    g = torch.Generator().manual_seed(40 + co)
    x = torch.randn(1, Ci, T, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(co, Ci, 3, 3, 3, generator=g) * 0.05).to(torch.bfloat16).float()
    b = torch.randn(co, generator=g)
    ref = F.conv3d(F.pad(x, (1, 1, 1, 1, 2, 0)), w, b)[0].permute(1, 2, 3, 0)          # [T, H, W, co]
    src = PBuf("x", T, H, W, Ci, "cuda")
    v = src.t.view(T + 2, H + 2, W + 2, src.Cp)
    v[2:, 1:-1, 1:-1, :Ci] = x[0].permute(1, 2, 3, 0).to("cuda", torch.bfloat16)
    src.cur = T
    cw = ConvW(w, b, "cuda")
    outs = {}
    for name, policy in (("narrow", 3), ("generic", -3)):
        out = torch.full((T, H, W, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.gemm_set_policy(policy)
        try:
            conv(src, None, cw, T, dst_raw=(out, H, W, 8, 0))
        finally:
            ops.gemm_set_policy(3)
        outs[name] = out.float().cpu()
        assert rel_l2(outs[name][..., :co], ref) < 5e-3, name
    assert torch.isfinite(outs["narrow"]).all()                  # every one of the 8 pitch columns was written
    assert (outs["narrow"][..., co:] == 0).all()                 # ... the padding columns with zero filters and zero bias
    assert rel_l2(outs["narrow"][..., :co], outs["generic"][..., :co]) < 2e-3
