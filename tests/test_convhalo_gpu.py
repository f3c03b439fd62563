"""The LDS-halo direct convolution (csrc/convhalo.hip) behind pf_conv3d_bf16: the decoder's full-resolution CausalConv3d
layers (3 x 3 x 3 taps, 128 filters, 128 or 256 input channels: modeling_causal_conv.py:116-146, conv1 / conv2 of
up_blocks.3's resnets, modeling_resnet.py:115-150).
  * against torch's fp32 conv3d of the same bf16 operands (the "plain PyTorch fp32 reference" of a floating-point kernel;
    tolerance of one op, SURVEY 8c: rel-L2 <= 5e-3) and against the implicit-GEMM route it replaces (policy -5),
  * the GroupNorm statistics its epilogue leaves (pf_conv_desc.gn_stats) against a float64 reduction of what it stored,
  * routing: pf_conv3d_which == -2 for these layers, the implicit GEMM for everything the kernel does not take."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from util import rel_l2

pytestmark = pytest.mark.gpu


def _desc_route(src, dst, cw, T, res=None, st=1, sh=1, sw=1, t_shift=0):
    from pyflow_hip import lib as L_
    from pyflow_hip.lib import ConvDesc, GEMM_GATE_RES
    d = ConvDesc()
    d.X, d.W, d.bias, d.Y = src.t.data_ptr(), cw.w.data_ptr(), cw.b.data_ptr(), dst.t.data_ptr()
    d.T, d.H, d.W_ = T, src.H, src.W
    d.in_sh = d.in_sw = d.in_st = 1
    d.Hp, d.Wp, d.Cin = src.Hp, src.Wp, src.Cp
    d.kt, d.kh, d.kw = cw.kt, cw.kh, cw.kw
    d.N, d.n_valid = cw.N, cw.n_valid
    d.st, d.sh, d.sw, d.out_t_shift = st, sh, sw, t_shift
    d.Cg = cw.Cg
    d.Hop, d.Wop, d.Cout_pitch = dst.Hp, dst.Wp, dst.Cp
    d.flags = GEMM_GATE_RES if res is not None else 0
    d.res = res.t.data_ptr() if res is not None else None
    d.out_scale = 1.0
    return int(L_.load().pf_conv3d_which(C.byref(d)))


@pytest.mark.parametrize("Ci,with_res,T,H,W,co", [(128, False, 3, 64, 64, 128), (128, True, 8, 256, 256, 128), (256, False, 2, 64, 96, 128),
                                                 (256, True, 1, 32, 32, 128), (128, True, 2, 16, 32, 128),
                                                 # the 256- / 512-filter resnets of up_blocks.2 / .1 (one 128-filter block per blockIdx.y)
                                                 (256, True, 4, 128, 128, 256), (512, False, 2, 128, 128, 256), (512, True, 8, 64, 64, 512)])
def test_halo_conv_vs_fp32_conv3d_and_vs_implicit_gemm(Ci, with_res, T, H, W, co):
    from pyflow_hip import ops
    from pyflow_hip.vae import PBuf, ConvW, conv
    g = torch.Generator().manual_seed(7 * Ci + T)
    x = torch.randn(T + 2, H, W, Ci, generator=g).to(torch.bfloat16)            # frames 0, 1 = the cache slots (previous chunk)
    w = (torch.randn(co, Ci, 3, 3, 3, generator=g) * 0.03).to(torch.bfloat16)
    b = torch.randn(co, generator=g)
    r = torch.randn(T, H, W, co, generator=g).to(torch.bfloat16) if with_res else None
    src = PBuf("x", T, H, W, Ci, "cuda")
    src.t.view(T + 2, H + 2, W + 2, src.Cp)[:, 1:-1, 1:-1, :Ci] = x.cuda()
    src.cur = T
    res = None
    if with_res:
        res = PBuf("r", T, H, W, co, "cuda")
        res.t.view(T + 2, H + 2, W + 2, res.Cp)[2:, 1:-1, 1:-1, :co] = r.cuda()
        res.cur = T
    cw = ConvW(w.float(), b, "cuda")
    outs = {}
    for halo in (True, False):
        ops.gemm_set_policy(5 if halo else -5)
        try:
            dst = PBuf("y", T, H, W, co, "cuda")
            rt = _desc_route(src, dst, cw, T, res)
            assert (rt == -2) == halo, rt
            stats = torch.zeros(T * co * 2, dtype=torch.float64, device="cuda")
            conv(src, dst, cw, T, res=res, gn_stats=stats)
            y = dst.t.view(T + 2, H + 2, W + 2, dst.Cp)[2:, 1:-1, 1:-1, :co].clone()
            if halo:
                assert dst.gn_ready is stats
                yd = y.double()
                exact = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=-1).reshape(-1)
                scale = exact.view(T, co, 2)[..., 1].sqrt().max().item() * (H * W) ** 0.5
                assert (stats - exact).abs().max() <= 1e-4 * max(scale, 1.0)
                # border pixels and cache slots of the output buffer stay untouched (zero)
                full = dst.t.view(T + 2, H + 2, W + 2, dst.Cp)
                assert full[:2].abs().max() == 0 and full[:, 0].abs().max() == 0 and full[:, :, 0].abs().max() == 0
                assert full[:, -1].abs().max() == 0 and full[:, :, -1].abs().max() == 0
            outs[halo] = y.float().cpu()
        finally:
            ops.gemm_set_policy(5)
    # fp32 reference: causal conv = frames [t, t+2] of the slot-extended input, zero spatial padding
    xin = x.float().permute(3, 0, 1, 2)[None].cuda()                             # [1, Ci, T+2, H, W]
    ref = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.float().cuda(), b.cuda())   # [1, co, T, H, W]
    ref = ref[0].permute(1, 2, 3, 0)
    if with_res:
        ref = ref + r.float().cuda()
    ref = ref.cpu()
    e_h, e_g = rel_l2(outs[True], ref), rel_l2(outs[False], ref)
    print(f"halo conv Ci={Ci} res={with_res} T={T} {H}x{W}: rel-L2 vs fp32 conv3d {e_h:.3e} (implicit GEMM {e_g:.3e}), "
          f"halo vs implicit GEMM {rel_l2(outs[True], outs[False]):.3e}")
    assert e_h < 5e-3 and e_g < 5e-3
    assert rel_l2(outs[True], outs[False]) < 3e-3


def test_halo_route_only_where_the_kernel_applies():
    from pyflow_hip.vae import PBuf, ConvW
    g = torch.Generator().manual_seed(1)

    def route(Ci, co, H, W, k=3):
        src = PBuf("x", 2, H, W, Ci, "cuda")
        dst = PBuf("y", 2, H, W, co, "cuda")
        cw = ConvW(torch.randn(co, Ci, k, k, k, generator=g) * 0.05, torch.zeros(co), "cuda")
        return _desc_route(src, dst, cw, 2)
    assert route(128, 128, 256, 256) == -2 and route(256, 128, 256, 256) == -2
    assert route(128, 128, 48, 40) != -2            # frames that are not whole 16 x 32 patches
    assert route(128, 192, 128, 128) != -2          # filter counts that are not whole 128-blocks
    assert route(512, 512, 32, 32) != -2            # wide layers with too few patches for half a round of the chip (T = 2)
    assert route(256, 256, 128, 128) == -2          # the 256-filter resnets at 128 x 128
    assert route(384, 128, 64, 64) != -2            # other input widths
    assert route(128, 128, 64, 64, k=1) != -2       # 1 x 1 x 1 shortcut convs


@pytest.mark.parametrize("kind,Ci,Cg,T,H,W", [("spatial", 256, 256, 4, 64, 64), ("temporal", 256, 256, 4, 64, 64),
                                             ("temporal_first", 128, 128, 4, 64, 128)])
def test_halo_conv_upsampler_output_maps_equal_the_implicit_gemm(kind, Ci, Cg, T, H, W):
    """the decoder's upsampler convs (N = 4 Cg / 2 Cg filters; pixel shuffle modeling_resnet.py:609-617, depth-to-time
    :716-729 with the first-frame drop) through the halo kernel's epilogue placement: same values in the same places as the
    implicit GEMM (whose placement the decode tests pin to the oracle), nothing written outside them"""
    from pyflow_hip import ops
    from pyflow_hip.vae import PBuf, ConvW, conv
    g = torch.Generator().manual_seed(5)
    groups = 4 if kind == "spatial" else 2
    co = groups * Cg
    x = torch.randn(T + 2, H, W, Ci, generator=g).to(torch.bfloat16)
    w = (torch.randn(co, Ci, 3, 3, 3, generator=g) * 0.03)
    b = torch.randn(co, generator=g)
    src = PBuf("x", T, H, W, Ci, "cuda")
    src.t.view(T + 2, H + 2, W + 2, src.Cp)[:, 1:-1, 1:-1, :Ci] = x.cuda()
    src.cur = T
    cw = ConvW(w, b, "cuda", groups)
    outs = {}
    for halo in (True, False):
        ops.gemm_set_policy(5 if halo else -5)
        try:
            if kind == "spatial":
                dst = PBuf("y", T, 2 * H, 2 * W, Cg, "cuda")
                assert (_desc_route(src, dst, cw, T, sh=2, sw=2) == -2) == halo
                conv(src, dst, cw, T, sh=2, sw=2)
            else:
                dst = PBuf("y", 2 * T, H, W, Cg, "cuda")
                ts = -1 if kind == "temporal_first" else 0
                assert (_desc_route(src, dst, cw, T, st=2, t_shift=ts) == -2) == halo
                conv(src, dst, cw, T, st=2, t_shift=ts)
            outs[halo] = dst.t.clone()
        finally:
            ops.gemm_set_policy(5)
    assert outs[True].abs().max() > 0
    d = (outs[True].float() - outs[False].float()).abs().max().item()
    assert rel_l2(outs[True].float().cpu(), outs[False].float().cpu()) < 3e-3 and d <= 2 ** -5 * outs[False].float().abs().max().item()
    # exactly the same set of elements was written (borders, cache slots, a dropped first frame stay zero in both); an element
    # whose sum cancels to +-1e-7 may round to an exact bf16 zero in one summation order and not in the other (measured: one
    # such element of 8.4 M): where the zero patterns differ, both values must be that small
    diff = (outs[True] != 0) != (outs[False] != 0)
    assert diff.sum().item() <= 8
    assert outs[True][diff].float().abs().max().item() < 1e-5 if diff.any() else True
    assert outs[False][diff].float().abs().max().item() < 1e-5 if diff.any() else True
    # ... and DIRECTLY against torch's fp32 conv3d of the same bf16 operands followed by the reference's rearranges
    # (filters in the reference's channel order: 'b (c p1 p2) t h w -> b c t (h p1) (w p2)' modeling_resnet.py:616,
    # 'b (c p) t h w -> b c (t p) h w' + the is_init_image first-frame drop :726-729); one op: rel-L2 <= 5e-3
    xin = x.float().permute(3, 0, 1, 2)[None].cuda()                             # [1, Ci, T + 2, H, W], frames 0, 1 = cache slots
    y = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.to(torch.bfloat16).float().cuda(), b.cuda())        # [1, co, T, H, W]
    if kind == "spatial":
        y = y.view(1, Cg, 2, 2, T, H, W).permute(0, 1, 4, 5, 2, 6, 3).reshape(1, Cg, T, 2 * H, 2 * W)
        To, Ho, Wo = T, 2 * H, 2 * W
    else:
        y = y.view(1, Cg, 2, T, H, W).permute(0, 1, 3, 2, 4, 5).reshape(1, Cg, 2 * T, H, W)
        To, Ho, Wo = 2 * T, H, W
    ref = y[0].permute(1, 2, 3, 0).cpu()                                         # [To, Ho, Wo, Cg]
    got = outs[True].view(To + 2, Ho + 2, Wo + 2, -1)[2:, 1:-1, 1:-1, :Cg].float().cpu()
    if kind == "temporal_first":            # t_shift = -1: reference frame 0 is dropped, frame f lands in slot f - 1
        ref = ref[1:]
        assert got[-1].abs().max() == 0     # the last slot of the 2T-frame buffer is not written
        got = got[:-1]
    e_t = rel_l2(got, ref)
    print(f"halo upsampler {kind} Ci={Ci} Cg={Cg} T={T} {H}x{W}: rel-L2 vs fp32 conv3d + rearrange {e_t:.3e}")
    assert e_t < 5e-3
