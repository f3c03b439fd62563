"""the halo convolution alone on the decoder's resnet layers (8 frames of 256 x 256: 128 -> 128 with shortcut add, 256 -> 128;
128 x 128: 256 -> 256, 512 -> 256), TFLOP/s per layer.  argv[1] = library variant (default: shipping).  One process per
variant, alternating, for a same-box A/B."""
import os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as L
if len(sys.argv) > 1 and sys.argv[1] != "ship":
    L.use_lab_library(sys.argv[1])
from pyflow_hip.vae import PBuf, ConvW, conv
g = torch.Generator().manual_seed(1)
for (T, H, W, Ci, Co, with_res) in ((8, 256, 256, 128, 128, True), (8, 256, 256, 256, 128, False), (8, 128, 128, 256, 256, True), (8, 128, 128, 512, 256, False)):
    src = PBuf("x", T, H, W, Ci, "cuda")
    src.t.view(T + 2, H + 2, W + 2, src.Cp)[:, 1:-1, 1:-1, :Ci] = torch.randn(T + 2, H, W, Ci, generator=g).to("cuda", torch.bfloat16)
    src.cur = T
    res = None
    if with_res:
        res = PBuf("r", T, H, W, Co, "cuda")
        res.t.view(T + 2, H + 2, W + 2, res.Cp)[2:, 1:-1, 1:-1, :Co] = torch.randn(T, H, W, Co, generator=g).to("cuda", torch.bfloat16)
        res.cur = T
    cw = ConvW(torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.03, torch.randn(Co, generator=g), "cuda")
    dst = PBuf("y", T, H, W, Co, "cuda")
    for _ in range(3):
        conv(src, dst, cw, T, res=res)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv(src, dst, cw, T, res=res)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    fl = 2.0 * T * H * W * Co * 27 * Ci
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'ship':10s} {T}x{H}x{W} {Ci:3d}->{Co:3d}{' +res' if with_res else '     '}: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TFLOP/s  checksum {int(dst.t.view(torch.int16).long().sum()) & 0xffffffff:08x}", flush=True)
