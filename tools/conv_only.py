"""the decoder's dominant full-resolution layer alone (8 frames of 256 x 256, 128 -> 128 channels, 3 x 3 x 3, shortcut add),
argv[1] launches through the LDS-halo direct conv and as many through the implicit GEMM it replaced (pf_gemm_set_policy(-5)) --
the target of the rocprofv3 --pmc passes that compare their L2 <-> fabric traffic (tools/gpu_pmc.sh).
Algorithmic bytes of one launch: input 10 x 258 x 258 x 128 x 2 = 170 MB, shortcut 136 MB, output 136 MB, filters 0.9 MB."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops
from pyflow_hip.vae import PBuf, ConvW, conv
T, H, W, Ci = 8, 256, 256, 128
g = torch.Generator().manual_seed(1)
src = PBuf("x", T, H, W, Ci, "cuda")
src.t.view(T + 2, H + 2, W + 2, src.Cp)[:, 1:-1, 1:-1, :Ci] = torch.randn(T + 2, H, W, Ci, generator=g).to("cuda", torch.bfloat16)
src.cur = T
res = PBuf("r", T, H, W, Ci, "cuda")
res.t.view(T + 2, H + 2, W + 2, res.Cp)[2:, 1:-1, 1:-1, :Ci] = torch.randn(T, H, W, Ci, generator=g).to("cuda", torch.bfloat16)
res.cur = T
cw = ConvW(torch.randn(Ci, Ci, 3, 3, 3, generator=g) * 0.03, torch.randn(Ci, generator=g), "cuda")
dst = PBuf("y", T, H, W, Ci, "cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for pol in (5, -5):
    ops.gemm_set_policy(pol)
    for _ in range(n):
        conv(src, dst, cw, T, res=res)
ops.gemm_set_policy(5)
torch.cuda.synchronize()
print("conv layer done")
