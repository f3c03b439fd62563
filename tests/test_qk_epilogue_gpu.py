"""QK-RMSNorm + RoPE inside the K|V|Q projection (pf_gemm_desc.qk_*, ABI 4): the persistent GEMM's epilogue applies
RMSNorm over every 64-wide head of the K and Q column blocks (modeling_normalization.py:66-79) and the adjacent-pair RoPE
rotation (flux_block.py:34-39) to the bf16-rounded projection -- the arithmetic of pf_qk_norm_rope, shared source
(csrc/common.h), so the fused result must equal "GEMM, then the separate pass" BIT FOR BIT; that pass is what the
oracle-pinned forwards of the other test files have been checking since round 1.
  * every layout the engines use: K|V|Q (double blocks), K|V|Q|MLP + GELU (single blocks), K|V only and Q|MLP only
    (last-block forms, row offset into the RoPE table), batch 2, M tails;
  * the library's own fallback (a kernel choice without the fused epilogue) gives the same bits;
  * a whole forward with and without the fusion: identical velocity tokens."""
import pytest
import torch
import torch.nn.functional as F

from util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rope_table(rows, seed):
    ang = torch.rand(rows, 32, generator=torch.Generator().manual_seed(seed)) * 6.28
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous().to(DEV)      # [rows][32][cos, sin] fp32


CASES = {
    # name: (M, batch, d, N, q_col0, k_col0, gelu_from, row0)
    "double_kvq": (1000, 2, 256, 768, 512, 0, -1, 128),
    "single_kvqm": (777, 2, 256, 7 * 256, 512, 0, 3 * 256, 0),
    "tail_kv": (900, 2, 256, 512, -1, 0, -1, 128),
    "tail_q_mlp": (300, 2, 256, 5 * 256, 0, -1, 256, 4000),
    "miniflux_width": (2100, 2, 1920, 3 * 1920, 2 * 1920, 0, -1, 128),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("policy", [8, 0, -8])
def test_fused_qk_epilogue_equals_gemm_then_separate_pass(case, policy):
    from pyflow_hip import ops
    M, B, d, N, q0, k0, gelu_from, row0 = CASES[case]
    K = 256 if d == 256 else 1920
    H = d // 64
    A = _mk((B, M, K), 1).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 2, 0.06).to(torch.bfloat16).to(DEV)
    bias = _mk((N,), 3).to(DEV)
    wq = (1.0 + 0.3 * _mk((64,), 4)).to(DEV)
    wk = (1.0 + 0.3 * _mk((64,), 5)).to(DEV)
    rope = _rope_table(row0 + M, 6)
    qs = 0.125 * ops.LOG2E
    ops.gemm_set_policy(policy)
    try:
        if policy == 8:
            assert ops.L.load().pf_gemm_which(M, B, N, K) == 8
        fused = torch.zeros(B, M + 2, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(A, W, fused, M, N, K, K, K, N, bias=bias, batch=B, strideA=M * K, strideC=(M + 2) * N, gelu_from=gelu_from,
                 qk=dict(rope=rope, wq=wq, wk=wk, d=d, q_col0=q0, k_col0=k0, row0=row0, eps=1e-6, q_scale=qs))
        plain = torch.zeros_like(fused)
        ops.gemm(A, W, plain, M, N, K, K, K, N, bias=bias, batch=B, strideA=M * K, strideC=(M + 2) * N, gelu_from=gelu_from)
    finally:
        ops.gemm_set_policy(0)
    # the separate pass over rows [0, M) with table rows row0 ...: Lt = 0 -> every row uses (wq, wk)
    sep = plain.clone()
    ops.qk_norm_rope(sep, N, (M + 2) * N, q0, k0, wq, wk, None, None, rope[row0:], B, M, 0, H, q_scale=qs)
    assert torch.equal(fused, sep), f"{case}: fused != separate pass, rel {rel_l2(fused.float(), sep.float()):.3e}"
    assert fused[:, M:].abs().max() == 0
    # and against an fp32 restatement (RMSNorm + rotation of the bf16 projection): what the bits mean
    x = plain[:, :M].float()
    for col0, w_, sc in ((k0, wk, 1.0), (q0, wq, qs)):
        if col0 < 0:
            continue
        blk = x[:, :, col0:col0 + d].reshape(B, M, H, 64)
        n = blk * torch.rsqrt(blk.pow(2).mean(-1, keepdim=True) + 1e-6) * w_
        cs = rope[row0:row0 + M]                                   # [M, 32, 2]
        x0, x1 = n[..., 0::2], n[..., 1::2]
        c, s_ = cs[None, :, None, :, 0], cs[None, :, None, :, 1]
        ref = torch.stack([c * x0 - s_ * x1, s_ * x0 + c * x1], dim=-1).reshape(B, M, d) * sc
        got = fused[:, :M, col0:col0 + d].float()
        assert rel_l2(got, ref) < 4e-3
    # columns outside the two blocks are the plain projection (V, MLP + GELU)
    keep = torch.ones(N, dtype=torch.bool)
    for col0 in (q0, k0):
        if col0 >= 0:
            keep[col0:col0 + d] = False
    assert torch.equal(fused[:, :, keep], plain[:, :, keep])


@pytest.mark.parametrize("policy", [8, 0, -8])
@pytest.mark.parametrize("M,d,extra", [(1500, 1920, 0), (700, 256, 512)])
def test_fused_qk_epilogue_head_major_layout(policy, M, d, extra):
    """ABI 5: the sequence-parallel engine's head-major projection columns [head][k | v | q][64] (qk_head_stride = 192): K at
    column 0 and Q at column 128 of every head -- normed and rotated by the GEMM BEFORE the exchange (flux_sp.py).  Same bits
    as "plain GEMM, then pf_qk_norm_rope with head_stride = 192"; the V blocks and any columns behind the heads (the MLP
    branch) are the plain projection."""
    from pyflow_hip import ops
    B, K, H = 2, (1920 if d == 1920 else 256), d // 64
    N = 3 * d + extra
    row0 = 77
    A = _mk((B, M, K), 11).to(torch.bfloat16).to(DEV)
    W = _mk((N, K), 12, 0.06).to(torch.bfloat16).to(DEV)
    bias = _mk((N,), 13).to(DEV)
    wq = (1.0 + 0.3 * _mk((64,), 14)).to(DEV)
    wk = (1.0 + 0.3 * _mk((64,), 15)).to(DEV)
    rope = _rope_table(row0 + M, 16)
    qs = 0.125 * ops.LOG2E
    gelu_from = 3 * d if extra else -1
    ops.gemm_set_policy(policy)
    try:
        fused = torch.zeros(B, M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(A, W, fused, M, N, K, K, K, N, bias=bias, batch=B, strideA=M * K, strideC=M * N, gelu_from=gelu_from,
                 qk=dict(rope=rope, wq=wq, wk=wk, d=d, q_col0=128, k_col0=0, head_stride=192, row0=row0, eps=1e-6, q_scale=qs))
        plain = torch.zeros_like(fused)
        ops.gemm(A, W, plain, M, N, K, K, K, N, bias=bias, batch=B, strideA=M * K, strideC=M * N, gelu_from=gelu_from)
    finally:
        ops.gemm_set_policy(0)
    sep = plain.clone()
    ops.qk_norm_rope(sep, N, M * N, 128, 0, wq, wk, None, None, rope[row0:], B, M, 0, H, q_scale=qs, head_stride=192)
    assert torch.equal(fused, sep), f"head-major: fused != separate pass, rel {rel_l2(fused.float(), sep.float()):.3e}"
    hm = plain[:, :, :3 * d].view(B, M, H, 3, 64)
    fm = fused[:, :, :3 * d].view(B, M, H, 3, 64)
    assert torch.equal(fm[:, :, :, 1], hm[:, :, :, 1])                     # V blocks untouched
    assert not torch.equal(fm[:, :, :, 0], hm[:, :, :, 0]) and not torch.equal(fm[:, :, :, 2], hm[:, :, :, 2])
    if extra:
        assert torch.equal(fused[:, :, 3 * d:], plain[:, :, 3 * d:])
    # fp32 restatement of one head's K block
    blk = hm[:, :, 3, 0].float()
    n = blk * torch.rsqrt(blk.pow(2).mean(-1, keepdim=True) + 1e-6) * wk
    cs = rope[row0:row0 + M]
    x0, x1 = n[..., 0::2], n[..., 1::2]
    ref = torch.stack([cs[None, :, :, 0] * x0 - cs[None, :, :, 1] * x1, cs[None, :, :, 1] * x0 + cs[None, :, :, 0] * x1], dim=-1).reshape(B, M, 64)
    assert rel_l2(fm[:, :, 3, 0].float(), ref) < 4e-3


def test_forward_identical_with_and_without_the_fused_epilogue():
    from pyflow_hip import synth
    from pyflow_hip.flux import FluxEngine
    cfg = dict(synth.MINIFLUX, num_layers=2, num_single_layers=2)
    g = torch.Generator().manual_seed(3)
    sd = synth.random_state_dict(synth.flux_param_shapes(cfg), seed=5, std=0.02, lively=True)
    shapes = [(3, 24, 40), (1, 24, 40), (1, 48, 80), (1, 48, 80)]                    # L = 3 008: the large-tile kernels
    clips = [torch.randn(1, 16, *s_, generator=g).to(DEV) for s_ in shapes]
    enc = torch.randn(2, 128, 4096, generator=g).to(torch.bfloat16)
    mask = torch.zeros(2, 128, dtype=torch.long)
    mask[0, :40] = 1
    mask[1, :96] = 1
    pooled = torch.randn(2, 768, generator=g)
    eng = FluxEngine(sd, cfg, DEV)
    eng.group_text = False          # (the grouped double blocks need the fused epilogue: they are compared in test_gemm_grouped_gpu.py)
    plan = eng.make_plan(shapes, mask)
    eng.encode_context(enc)
    outs = {}
    for fuse in (True, False, True):
        eng.fuse_qk = fuse
        for dead in (True, False):
            eng.skip_dead_rows = dead
            v = eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True).clone()
            assert torch.isfinite(v).all() and v.abs().max() > 0
            outs.setdefault((fuse, dead), v)
            assert torch.equal(outs[(fuse, dead)], v)
    # the fused launches bring no tail-split scratch (the split's second launch has no QK epilogue), the separate-pass ones
    # do: their K|V|Q GEMMs differ in fp32 summation order where a tail is split -> two bf16 evaluations of the same 4-block
    # forward (measured 3.6e-3 apart, the rounding noise of tests/test_fullsize_gpu.py's split-vs-whole comparison), and
    # bit for bit with the split switched off
    from pyflow_hip import ops
    for dead in (True, False):
        assert rel_l2(outs[(True, dead)].cpu(), outs[(False, dead)].cpu()) < 1e-2
    ops.gemm_set_policy(-4)
    try:
        eng.skip_dead_rows = True
        eng.fuse_qk = True
        a = eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True).clone()
        eng.fuse_qk = False
        b = eng.forward_tokens(plan, clips, [500.0, 500.0], pooled, shared_clips=True).clone()
    finally:
        ops.gemm_set_policy(4)
    assert torch.equal(a, b)
