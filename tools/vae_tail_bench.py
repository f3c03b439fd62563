"""The decode's convolutions that do NOT run on the halo kernel (tools/vae_routes.py lists them), each ALONE on the GPU at its launch
shape: ms, TFLOP/s and the GB/s of its algorithmic traffic (input + output + filters once) -- inside a decode these launches overlap
three other lanes, so rocprofv3's durations of them are inflated; this is the standalone cost.  argv[1] = library variant."""
import os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as L
if len(sys.argv) > 1 and sys.argv[1] != "ship":
    L.use_lab_library(sys.argv[1])
import ctypes as C
from pyflow_hip.vae import PBuf, ConvW, conv
from pyflow_hip.lib import ConvDesc
g = torch.Generator().manual_seed(1)
ROUTE = {-2: "halo", -1: "narrow", 8: "gemm8p", 0: "gemm128"}
# (T, H, W, Cin, Cout, taps, (st, sh, sw), with_res)
CASES = (
    (32, 256, 256, 256, 128, 1, (1, 1, 1), False),      # conv_shortcut of up_blocks.3 (1.4 s of lane time per video)
    (32, 192, 256, 256, 128, 1, (1, 1, 1), False),
    (16, 128, 128, 512, 256, 1, (1, 1, 1), False),      # conv_shortcut of up_blocks.2
    (4, 32, 32, 512, 512, 27, (1, 1, 1), False),        # latent-resolution resnets
    (4, 32, 32, 512, 512, 27, (1, 1, 1), True),
    (4, 32, 32, 512, 2048, 27, (1, 2, 2), False),       # spatial upsampler at the latent resolution
    (4, 24, 32, 512, 512, 27, (1, 1, 1), True),
    (8, 48, 32, 512, 512, 27, (1, 1, 1), True),         # 24 x 16-latent tile: not a whole number of halo patches per round
    (4, 32, 32, 64, 512, 27, (1, 1, 1), False),         # conv_in
    (32, 256, 256, 128, 128, 27, (1, 1, 1), True),      # reference point: the halo kernel
)
lib = L.load()
for (T, H, W, Ci, Co, taps, (st, sh, sw), with_res) in CASES:
    k = 3 if taps == 27 else 1
    src = PBuf("x", T, H, W, Ci, "cuda")
    src.t.view(T + 2, H + 2, W + 2, src.Cp)[:, 1:-1, 1:-1, :Ci] = (torch.randn(T + 2, H, W, Ci, generator=g) * 0.5).to("cuda", torch.bfloat16)
    src.cur = T
    cg = Co // (st * sh * sw)
    dst = PBuf("y", T * st, H * sh, W * sw, cg, "cuda")
    res = None
    if with_res:
        res = PBuf("r", T, H, W, Co, "cuda")
        res.t.view(T + 2, H + 2, W + 2, res.Cp)[2:, 1:-1, 1:-1, :Co] = torch.randn(T, H, W, Co, generator=g).to("cuda", torch.bfloat16)
        res.cur = T
    cw = ConvW(torch.randn(Co, Ci, k, k, k, generator=g) * 0.03, torch.randn(Co, generator=g), "cuda", groups=st * sh * sw)
    for _ in range(3):
        conv(src, dst, cw, T, st=st, sh=sh, sw=sw, res=res)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv(src, dst, cw, T, st=st, sh=sh, sw=sw, res=res)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    fl = 2.0 * T * H * W * Co * taps * Ci
    by = 2.0 * T * H * W * (Ci + Co * (2 if with_res else 1)) + 2.0 * Co * taps * Ci
    # route of this launch (the descriptor conv() builds is not returned: ask with an equivalent one)
    d = ConvDesc()
    d.X = src.t.data_ptr(); d.W = cw.w.data_ptr(); d.bias = cw.b.data_ptr(); d.Y = dst.t.data_ptr()
    d.T, d.H, d.W_ = T, H, W
    d.in_sh = d.in_sw = d.in_st = 1
    d.Hp, d.Wp, d.Cin = src.Hp, src.Wp, src.Cp
    d.kt = d.kh = d.kw = k
    d.N, d.n_valid, d.st, d.sh, d.sw, d.Cg = cw.N, cw.n_valid, st, sh, sw, cw.Cg
    d.Hop, d.Wop, d.Cout_pitch = dst.Hp, dst.Wp, dst.Cp
    d.flags = L.GEMM_GATE_RES if with_res else 0
    d.res = res.t.data_ptr() if with_res else None
    d.out_scale = 1.0
    route = int(lib.pf_conv3d_which(C.byref(d)))
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'ship':8s} T={T:2d} {H:3d}x{W:<3d} {Ci:3d}->{Co:4d} taps={taps:2d} up={st}{sh}{sw}{' +res' if with_res else '     '} "
          f"{ROUTE.get(route, 'gemm256<%d>' % route):12s}: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s {by / ms / 1e6:7.0f} GB/s  "
          f"checksum {int(dst.t.view(torch.int16).long().sum()) & 0xffffffff:08x}", flush=True)
    del src, dst, res, cw
    torch.cuda.empty_cache()
