import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "pyramid-flow_amd"); sys.path.insert(0, ".")
from pyflow_hip import ops
from pyflow_hip.vae import PBuf, ConvW, conv
kind, Ci, Cg, T, H, W = "temporal_first", 128, 128, 4, 64, 128
g = torch.Generator().manual_seed(5)
groups = 2
co = groups * Cg
x = torch.randn(T + 2, H, W, Ci, generator=g).to(torch.bfloat16)
w = (torch.randn(co, Ci, 3, 3, 3, generator=g) * 0.03)
b = torch.randn(co, generator=g)
src = PBuf("x", T, H, W, Ci, "cuda")
src.t.view(T + 2, H + 2, W + 2, src.Cp)[:, 1:-1, 1:-1, :Ci] = x.cuda()
src.cur = T
cw = ConvW(w, b, "cuda", groups)
outs = {}
for name, pols in (("halo", [5]), ("gemm256", [-5]), ("gemm8p", [-5, 8])):
    for p_ in pols:
        ops.gemm_set_policy(p_)
    dst = PBuf("y", 2 * T, H, W, Cg, "cuda")
    conv(src, dst, cw, T, st=2, t_shift=-1)
    outs[name] = dst.t.clone().view(2 * T + 2, H + 2, W + 2, -1)
    ops.gemm_set_policy(0); ops.gemm_set_policy(5)
for a, b_ in (("halo", "gemm256"), ("halo", "gemm8p"), ("gemm256", "gemm8p")):
    A, B = outs[a], outs[b_]
    nz = (A != 0) != (B != 0)
    idx = nz.nonzero()
    print(a, b_, "mask mismatches", idx.shape[0], "max abs diff", (A.float() - B.float()).abs().max().item())
    if idx.shape[0]:
        print("  frames", idx[:, 0].unique().tolist(), "y", idx[:, 1].min().item(), idx[:, 1].max().item(), "x", idx[:, 2].min().item(), idx[:, 2].max().item(),
              "c", idx[:, 3].min().item(), idx[:, 3].max().item())
        i = idx[0].tolist()
        print("  first", i, A[tuple(i)].item(), B[tuple(i)].item())
