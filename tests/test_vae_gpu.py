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


@pytest.mark.parametrize("co,with_res", [(128, False), (128, True), (256, True)])
def test_conv_epilogue_accumulates_groupnorm_statistics(co, with_res):
    """pf_conv_desc.gn_stats (round 3): the 256-row conv kernel leaves (sum, sum of squares) per output frame and channel of
    what it STORED -- exactly what a pf_gn_stats pass over the output computes (up to the order of the fp32 / double sums);
    the host then skips that pass (PBuf.gn_ready).  Plain conv and conv + shortcut add, N = 128 and 256 filters."""
    from pyflow_hip import lib as L_
    from pyflow_hip.lib import check, stream
    from pyflow_hip.vae import PBuf, ConvW, conv
    import ctypes as C
    T, H, W, Ci = 4, 128, 128, 128           # 65 536 output pixels: the 256-row kernel's domain
    g = torch.Generator().manual_seed(90 + co)
    x = torch.randn(T, H, W, Ci, generator=g)
    w = (torch.randn(co, Ci, 3, 3, 3, generator=g) * 0.05)
    b = torch.randn(co, generator=g)
    src = PBuf("x", T, H, W, Ci, "cuda")
    src.t.view(T + 2, H + 2, W + 2, src.Cp)[2:, 1:-1, 1:-1, :Ci] = x.to("cuda", torch.bfloat16)
    src.cur = T
    res = None
    if with_res:
        res = PBuf("r", T, H, W, co, "cuda")
        res.t.view(T + 2, H + 2, W + 2, res.Cp)[2:, 1:-1, 1:-1, :co] = torch.randn(T, H, W, co, generator=g).to("cuda", torch.bfloat16)
        res.cur = T
    dst = PBuf("y", T, H, W, co, "cuda")
    stats = torch.zeros(T * co * 2, dtype=torch.float64, device="cuda")
    conv(src, dst, ConvW(w, b, "cuda"), T, res=res, gn_stats=stats)
    assert dst.gn_ready is stats                                       # the kernel that ran accumulates them
    ref = torch.zeros(T * co * 2, dtype=torch.float64, device="cuda")
    check(L_.load().pf_gn_stats(C.c_void_p(dst.t.data_ptr()), C.c_void_p(ref.data_ptr()), C.c_int(T), C.c_int(co),
                                C.c_int(dst.Cp), C.c_int(H), C.c_int(W), C.c_int(dst.Hp), C.c_int(dst.Wp),
                                C.c_longlong(dst.fs), C.c_longlong(dst.off(2)), stream()))
    y = dst.t.view(T + 2, H + 2, W + 2, dst.Cp)[2:, 1:-1, 1:-1, :co].double()
    exact = torch.stack([y.sum(dim=(1, 2)), (y * y).sum(dim=(1, 2))], dim=-1).reshape(-1)      # [T][co][2]
    assert exact.abs().max() > 0
    scale = exact.view(T, co, 2)[..., 1].sqrt().max().item() * (H * W) ** 0.5
    assert (ref - exact).abs().max() <= 1e-4 * max(scale, 1.0)         # the separate pass (fp32 partial sums)
    assert (stats - exact).abs().max() <= 1e-4 * max(scale, 1.0)       # the epilogue's sums
    # a shape the epilogue cannot serve (frames of 48 x 40 = 1 920 pixels: not a multiple of 256) falls back silently
    src2 = PBuf("x2", T, 48, 40, Ci, "cuda")
    src2.cur = T
    dst2 = PBuf("y2", T, 48, 40, co, "cuda")
    conv(src2, dst2, ConvW(w, b, "cuda"), T, gn_stats=stats)
    assert dst2.gn_ready is None


def test_decode_same_with_and_without_fused_groupnorm_statistics():
    from pyflow_hip import synth
    from pyflow_hip.vae import CausalVideoVAE
    cfg = synth.TINY_VAE
    sd = _sd(cfg)
    z = torch.randn(1, 16, 3, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
    outs = []
    for fuse in (True, False):
        vae = CausalVideoVAE(sd, cfg, "cuda")
        vae.fuse_gn_stats = fuse
        outs.append(vae.decode(z, temporal_chunk=True, window_size=1).sample.float().cpu())
    assert rel_l2(outs[0], outs[1]) < 2e-3 and outs[0].abs().max() > 0
