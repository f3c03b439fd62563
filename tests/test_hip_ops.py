"""GPU parity of each HIP entry point against a plain torch fp32 statement of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import rel_l2, bf16_round

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 256, 192), (1000, 384, 1920), (77, 128, 128)])
def test_gemm_bias(M, N, K):
    from pyflow_hip import ops
    A = _mk((M, K), 1).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 2, 0.05).to(torch.bfloat16).to(DEV)
    # asymmetric structure so a transposed C write cannot pass
    W[0, :] += 1.0
    bias = _mk((N,), 3).to(DEV)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N, bias=bias)
    ref = A.float() @ W.float().T + bias
    assert rel_l2(C.float(), ref) < 5e-3
    assert (C.float() - ref).abs().max() <= 2 ** -6 * ref.abs().max()


def test_gemm_identity_layout():
    """A = I with asymmetric W: catches row/col swaps in the MFMA C layout."""
    from pyflow_hip import ops
    M = N = K = 128
    A = torch.eye(M, dtype=torch.bfloat16, device=DEV)
    W = (torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.001).to(torch.bfloat16).to(DEV)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N)
    assert torch.equal(C.float(), W.float().T.contiguous())


def test_gemm_batched_strided_gelu_gate():
    from pyflow_hip import ops
    B, Lr, d = 2, 300, 256
    L = Lr + 16
    x = _mk((B, L, d), 4).to(torch.bfloat16).to(DEV)
    W = _mk((2 * d, d), 5, 0.06).to(torch.bfloat16).to(DEV)
    bias = _mk((2 * d,), 6, 0.1).to(DEV)
    out = torch.zeros(B, L, 2 * d, dtype=torch.bfloat16, device=DEV)
    # rows [16, L) of each batch, GELU on columns >= d
    ops.gemm(x, W, out, Lr, 2 * d, d, d, d, 2 * d, bias=bias, batch=B, strideA=L * d, strideC=L * 2 * d,
             gelu_from=d, a_off=16 * d, c_off=16 * 2 * d)
    ref = x[:, 16:].float() @ W.float().T + bias
    ref[..., d:] = F.gelu(ref[..., d:], approximate="tanh")
    assert rel_l2(out[:, 16:].float(), ref) < 5e-3
    assert out[:, :16].abs().max() == 0
    # gated residual, in place
    hid = _mk((B, L, d), 7).to(torch.bfloat16).to(DEV)
    hid0 = hid.clone()
    W2 = _mk((d, 2 * d), 8, 0.05).to(torch.bfloat16).to(DEV)
    b2 = _mk((d,), 9, 0.1).to(DEV)
    gate = _mk((B, 3 * d), 10).to(DEV)
    ops.gemm(out, W2, hid, Lr, d, 2 * d, 2 * d, 2 * d, d, bias=b2, res=hid, gate=gate, gate_off=d, ldr=d, batch=B,
             strideA=L * 2 * d, strideC=L * d, strideR=L * d, gate_stride=3 * d, flags=ops.GEMM_GATE_RES,
             a_off=16 * 2 * d, c_off=16 * d, r_off=16 * d)
    ref2 = hid0[:, 16:].float() + gate[:, None, d:2 * d] * (out[:, 16:].float() @ W2.float().T + b2)
    assert rel_l2(hid[:, 16:].float(), ref2) < 5e-3
    assert torch.equal(hid[:, :16], hid0[:, :16])


def test_gemm_out_f32():
    from pyflow_hip import ops
    M, N, K = 130, 128, 256
    A = _mk((M, K), 11).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 12, 0.05).to(torch.bfloat16).to(DEV)
    C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A, W, C, M, N, K, K, K, N, flags=ops.GEMM_OUT_F32)
    assert rel_l2(C, A.float() @ W.float().T) < 1e-5


def test_ln_modulate():
    from pyflow_hip import ops
    B, L, d = 2, 37, 1920
    x = _mk((B, L, d), 13, 2.0).to(torch.bfloat16).to(DEV)
    mod = _mk((B, 4 * d), 14, 0.5).to(DEV)
    y = torch.zeros_like(x)
    ops.ln_modulate(x, y, (mod, d), (mod, 2 * d), d, B, L - 5, L * d, L * d, d, d, 4 * d, x_off=5 * d, y_off=5 * d)
    ref = F.layer_norm(x[:, 5:].float(), (d,), eps=1e-6) * (1 + mod[:, None, 2 * d:3 * d]) + mod[:, None, d:2 * d]
    assert rel_l2(y[:, 5:].float(), ref) < 4e-3
    assert y[:, :5].abs().max() == 0


def test_gemv_and_timestep_embed():
    from pyflow_hip import ops
    B, K, N = 2, 1920, 777
    W = _mk((N, K), 15, 0.03).to(torch.bfloat16).to(DEV)
    b = _mk((N,), 16).to(DEV)
    x = _mk((B, K), 17).to(DEV)
    y = torch.zeros(B, N, device=DEV)
    ops.gemv(W, b, x, y, N, K, B, silu_in=True)
    ref = F.silu(x) @ W.float().T + b
    assert rel_l2(y, ref) < 1e-5
    ops.gemv(W, None, x, y, N, K, B, accumulate=True)
    assert rel_l2(y, ref + x @ W.float().T) < 1e-5
    out = torch.zeros(2, 256, device=DEV)
    ops.timestep_embed(out, [704.0, 13.25], 256)
    half = 128
    e = torch.tensor([704.0, 13.25])[:, None] * torch.exp(-math.log(10000) * torch.arange(half).float() / half)[None]
    ref = torch.cat([torch.cos(e), torch.sin(e)], -1)
    assert (out.cpu() - ref).abs().max() < 2e-3


def test_qk_norm_rope_and_vtranspose():
    from pyflow_hip import ops
    from pyflow_hip.plan import SequencePlan
    B, H, Lt = 2, 4, 16
    mask = torch.zeros(B, Lt, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    plan = SequencePlan([(2, 4, 8), (1, 8, 16), (1, 16, 32)], mask, [16, 24, 24], DEV)
    L, d = plan.L, H * 64
    qkv = _mk((B, L, 3 * d), 18).to(torch.bfloat16).to(DEV)
    q0 = qkv.clone()
    ws = [(1 + 0.1 * _mk((64,), 19 + i)).to(DEV) for i in range(4)]
    ops.qk_norm_rope(qkv, 3 * d, L * 3 * d, 2 * d, 0, ws[0], ws[1], ws[2], ws[3], plan.rope, B, L, Lt, H)

    def ref(x, w_img, w_txt):
        x = x.float().view(B, L, H, 64)
        w = torch.where((torch.arange(L, device=DEV) < Lt)[None, :, None, None], w_txt, w_img)
        xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
        xp = xn.view(B, L, H, 32, 2)
        c, s = plan.rope[None, :, None, :, 0], plan.rope[None, :, None, :, 1]
        o0 = c * xp[..., 0] - s * xp[..., 1]
        o1 = s * xp[..., 0] + c * xp[..., 1]
        return torch.stack([o0, o1], -1).view(B, L, d)
    assert rel_l2(qkv[..., 2 * d:].float(), ref(q0[..., 2 * d:], ws[0], ws[2])) < 4e-3
    assert rel_l2(qkv[..., :d].float(), ref(q0[..., :d], ws[1], ws[3])) < 4e-3
    assert torch.equal(qkv[..., d:2 * d], q0[..., d:2 * d])
    q2 = q0.clone()
    ops.qk_norm_rope(q2, 3 * d, L * 3 * d, 2 * d, 0, ws[0], ws[1], ws[2], ws[3], plan.rope, B, L, Lt, H, q_scale=0.18)
    assert rel_l2(q2[..., 2 * d:].float(), 0.18 * ref(q0[..., 2 * d:], ws[0], ws[2])) < 4e-3
    assert torch.equal(q2[..., :2 * d], qkv[..., :2 * d])          # k and v untouched by q_scale
    Lp = plan.Lp
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp)
    v = qkv[..., d:2 * d].view(B, L, H, 64).permute(0, 2, 3, 1)          # [B,H,64,L]
    pos = torch.arange(Lp)
    key = (pos & ~15) | (pos & 3) | ((pos & 4) << 1) | ((pos & 8) >> 1)
    exp = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device=DEV)
    valid = key < L
    exp[..., valid] = v[..., key[valid].to(DEV)]
    assert torch.equal(vT, exp)


@pytest.mark.parametrize("clips,Lt,valid", [
    ([(1, 16, 32)], 16, (5, 12)),
    ([(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)], 16, (5, 12)),
    ([(3, 8, 16), (1, 16, 32), (1, 32, 48), (1, 32, 48)], 128, (40, 96)),
])
@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_masked(clips, Lt, valid, prescaled):
    from pyflow_hip import ops
    from pyflow_hip.plan import SequencePlan
    B, H = 2, 3
    mask = torch.zeros(B, Lt, dtype=torch.long)
    for b in range(B):
        mask[b, :valid[b]] = 1
    plan = SequencePlan(clips, mask, [16, 24, 24], DEV)
    L, Lp, d = plan.L, plan.Lp, H * 64
    qkv = _mk((B, L, 3 * d), 30, 1.0).to(torch.bfloat16).to(DEV)
    # spike a few keys so the online-softmax rescale path is exercised
    qkv[0, L // 2, :64] *= 6.0
    qkv[1, L - 3, 64:128] *= 6.0
    q = qkv[..., 2 * d:].float().view(B, L, H, 64).transpose(1, 2)
    k = qkv[..., :d].float().view(B, L, H, 64).transpose(1, 2)
    v = qkv[..., d:2 * d].float().view(B, L, H, 64).transpose(1, 2)
    dm = torch.from_numpy(plan.dense_mask()).to(DEV)[:, None]
    if prescaled:
        # q already carries scale*log2(e) (as pf_qk_norm_rope(q_scale=...) leaves it): softmax base 2 of q.k
        qkv[..., 2 * d:] = (qkv[..., 2 * d:].float() * (0.125 * ops.LOG2E)).to(torch.bfloat16)
        q = qkv[..., 2 * d:].float().view(B, L, H, 64).transpose(1, 2)
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=dm, scale=math.log(2.0)).transpose(1, 2).reshape(B, L, d)
    else:
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=dm).transpose(1, 2).reshape(B, L, d)
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, d, 3 * d, L * 3 * d, B, H, L, Lp)
    ops.attention(qkv, qkv, vT, qkv, 2 * d, 0, 2 * d, 3 * d, L * 3 * d, B, H, L, Lp, Lt, plan, 0.125,
                  q_prescaled=prescaled)
    out = qkv[..., 2 * d:].float()
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
    assert (out - ref).abs().max() < 3e-2 * ref.abs().max()


def test_small_elementwise():
    from pyflow_hip import ops
    C_, H, W = 16, 8, 12
    x = _mk((C_, 1, H, W), 40).to(DEV)
    tok = torch.zeros(2, (H // 2) * (W // 2), 64, dtype=torch.bfloat16, device=DEV)
    ops.patchify(x, tok, 0, C_, 1, H, W, 64, tok.stride(0), 2)
    ref = x.permute(1, 2, 3, 0).reshape(1, H // 2, 2, W // 2, 2, C_).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64)
    assert torch.equal(tok[0].float(), bf16_round(ref))
    assert torch.equal(tok[0], tok[1])
    # cfg + euler
    n = (H // 2) * (W // 2)
    v = _mk((2, n, 128), 41).to(DEV)
    lat = _mk((C_, H, W), 42).to(DEV)
    lat0 = lat.clone()
    ops.cfg_euler_step(v, n * 128, 128, lat, C_, H, W, 5.0, True, -0.1, False)
    vv = v[..., :64].reshape(2, H // 2, W // 2, 2, 2, C_).permute(0, 5, 1, 3, 2, 4).reshape(2, C_, H, W)
    ref = lat0 + (-0.1) * (vv[0] + 5.0 * (vv[1] - vv[0]))
    assert (lat - ref).abs().max() < 1e-5
    # renoise + avgpool
    xin = _mk((C_, H // 2, W // 2), 43).to(DEV)
    noise = _mk((C_, H, W), 44).to(DEV)
    xo = torch.zeros(C_, H, W, device=DEV)
    ops.renoise_upsample(xin, noise, xo, C_, H, W, 0.6, 0.7, False)
    ref = 0.6 * F.interpolate(xin[None], size=(H, W), mode="nearest")[0] + 0.7 * noise
    assert (xo - ref).abs().max() < 1e-6
    pooled = torch.zeros(C_, H // 2, W // 2, device=DEV)
    ops.avgpool2(noise, pooled, C_, H, W, 2.0)
    ref = F.interpolate(noise[None], size=(H // 2, W // 2), mode="bilinear")[0] * 2
    assert (pooled - ref).abs().max() < 1e-5


def test_bf16_trajectory_rounding_points_bit_exact():
    """round_bf16=True branches (what generate() runs with bf16 prompt embeddings, pipeline `_round`): the fused kernels
    must reproduce the reference's op-by-op bf16 evaluation BIT FOR BIT -- CFG combine (pipeline.py:771-776), Euler step
    (scheduling_flow_matching.py:278-286: bf16(dsigma*v) then fp32 add then bf16), renoise (:729-743) and the
    bilinear /2 pyramids (:565, :1116).  The right-hand sides below are those expressions run by torch on CPU bf16
    tensors."""
    from pyflow_hip import ops
    bf = torch.bfloat16
    C_, H, W = 16, 8, 12
    n = (H // 2) * (W // 2)
    v = _mk((2, n, 128), 51).to(DEV)                                    # fp32 velocity tokens as the DiT leaves them
    lat0 = bf16_round(_mk((C_, H, W), 52))
    vv = v.cpu()[..., :64].reshape(2, H // 2, W // 2, 2, 2, C_).permute(0, 5, 1, 3, 2, 4).reshape(2, C_, H, W)
    for gs, dsig in ((5.0, -0.0526315789), (7.0, -0.111), (1.0, -0.001)):
        lat = lat0.clone().to(DEV)
        # the pipeline hands the kernel bf16(dsigma): a 0-dim f64 tensor operand is converted to bf16 before the product
        ds_k = float(torch.tensor(dsig, dtype=torch.float64).to(bf))
        ops.cfg_euler_step(v, n * 128, 128, lat, C_, H, W, gs, True, ds_k, True)
        npred = vv.to(bf)                                                # model output dtype
        vu, vt = npred[0], npred[1]
        comb = vu + gs * (vt - vu)                                       # bf16 op by op
        assert comb.dtype == bf
        step = torch.tensor(dsig, dtype=torch.float64) * comb            # 0-dim f64 x bf16 -> bf16 (Appendix B)
        assert step.dtype == bf
        ref = (lat0.float() + step).to(bf).float()
        assert torch.equal(lat.cpu(), ref), (gs, dsig)
    # no CFG: the model output is still bf16
    lat = lat0.clone().to(DEV)
    ops.cfg_euler_step(v, n * 128, 128, lat, C_, H, W, 0.0, False, float(torch.tensor(-0.05).to(bf)), True)
    ref = (lat0.float() + torch.tensor(-0.05, dtype=torch.float64) * vv[0].to(bf)).to(bf).float()
    assert torch.equal(lat.cpu(), ref)
    # renoise: alpha * nearest_up(latents) + beta * noise.to(bf16), all bf16
    xin = bf16_round(_mk((C_, H // 2, W // 2), 53))
    noise = _mk((C_, H, W), 54)
    xo = torch.zeros(C_, H, W, device=DEV)
    alpha, beta = 0.5998800255395356, 0.693028124888686              # stage-1 coefficients
    ops.renoise_upsample(xin.to(DEV), noise.to(DEV), xo, C_, H, W, alpha, beta, True)
    up = F.interpolate(xin.to(bf)[None], size=(H, W), mode="nearest")[0]
    ref = (alpha * up + beta * noise.to(bf)).float()
    assert torch.equal(xo.cpu(), ref)
    # pyramids: F.interpolate(bilinear, 1/2) [* 2] on bf16 tensors
    src = bf16_round(_mk((C_, H, W), 55))
    for mul in (1.0, 2.0):
        pooled = torch.zeros(C_, H // 2, W // 2, device=DEV)
        ops.avgpool2(src.to(DEV), pooled, C_, H, W, mul, True)
        ref = F.interpolate(src.to(bf)[None], size=(H // 2, W // 2), mode="bilinear")[0]
        if mul != 1.0:
            ref = ref * mul
        assert torch.equal(pooled.cpu(), ref.float()), mul


@pytest.mark.gpu
@pytest.mark.parametrize("M,B,N,K,flavour", [
    (128, 2, 5760, 1920, "bias"), (128, 2, 1920, 7680, "gate_res"), (128, 2, 7680, 1920, "gelu"), (77, 1, 1280, 5120, "gate_res"),
    (128, 1, 20480, 4096, "gelu_from"), (256, 1, 4096, 10240, "gate_res"), (100, 2, 1536, 1536, "f32"), (128, 2, 1920, 1920, "quick"),
])
def test_skinny_gemm_split_k(M, B, N, K, flavour):
    """split-K path of the 128x128 kernel (workspace given, < 128 tiles): every epilogue flavour against torch fp32 and
    against the same launch without the workspace"""
    import torch.nn.functional as F
    from pyflow_hip import ops
    from util import rel_l2
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(B, M, K, generator=g)).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(B, M, N, generator=g).to(torch.bfloat16).cuda()
    gate = torch.randn(B, N, generator=g).cuda()
    ws = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
    kw = dict(bias=bias, batch=B, strideA=M * K, strideC=M * N)
    ref = A.float() @ W.float().T + bias
    if flavour == "gate_res":
        kw.update(res=res, gate=gate, ldr=N, strideR=M * N, gate_stride=N, flags=ops.GEMM_GATE_RES)
        ref = res.float() + gate[:, None, :] * ref
    elif flavour == "gelu":
        kw.update(gelu_from=0)
        ref = F.gelu(ref, approximate="tanh")
    elif flavour == "gelu_from":
        kw.update(gelu_from=N // 2)
        ref[..., N // 2:] = F.gelu(ref[..., N // 2:], approximate="tanh")
    elif flavour == "quick":
        kw.update(gelu_from=0, flags=ops.GEMM_ACT_QUICK_GELU)
        ref = ref * torch.sigmoid(1.702 * ref)
    f32 = flavour == "f32"
    if f32:
        kw.update(flags=ops.GEMM_OUT_F32)
    outs = []
    for w_ in (ws, None):
        out = torch.zeros(B, M + 2, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        ops.gemm(A, W, out, M, N, K, K, K, N, **dict(kw, strideC=(M + 2) * N), workspace=w_)
        outs.append(out)
    split, plain = outs
    assert split[:, M:].abs().max() == 0
    assert rel_l2(split[:, :M].float(), ref) < 5e-3
    assert rel_l2(split[:, :M].float(), plain[:, :M].float()) < 3e-3
    # the scratch really was used (a split happened): its head holds partial sums, not garbage from torch.empty
    ops.gemm_set_policy(-2)
    out2 = torch.zeros_like(split)
    ops.gemm(A, W, out2, M, N, K, K, K, N, **dict(kw, strideC=(M + 2) * N), workspace=ws)
    ops.gemm_set_policy(2)
    assert torch.equal(out2, plain)
