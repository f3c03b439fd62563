"""The 64-rows-per-wave attention pair (csrc/attention_w64.h): a fast pass without any running row maximum + a fix-up
pass with it that recomputes the workgroups the fast pass flagged.  Replaces F.scaled_dot_product_attention with the
[B,1,L,L] mask (flux_block.py:361-365, modeling_pyramid_flux.py:318-350) exactly like the 32-row kernel; tolerance of
one attention op: rel-L2 <= 1e-2 against an fp32 restatement with the dense mask.
  * benchmark-like scores (+-10 in base-2 units): every row against the dense reference, the pair is what runs,
  * scores far outside the window the fast pass is exact in (|s| up to ~400: exp2 overflows / every key underflows):
    the fix-up pass must deliver the right rows,
  * the last-block form (q_row_begin) and the sequence-parallel layout (head stride 192, strided output)."""
import ctypes as C

import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
D, H, LT = 1920, 30, 128
CLIPS = [(3, 24, 40), (1, 24, 40), (1, 48, 80), (1, 48, 80)]          # unit 5, stage 1: L = 3 008


def _mask():
    m = torch.zeros(2, LT, dtype=torch.long)
    m[0, :40] = 1
    m[1, :96] = 1
    return m


def _reference(qkv, plan, B, L, rows=None):
    """fp32 softmax with the dense mask rows; q is pre-scaled (scores are base-2 exponents)"""
    rows = list(range(L)) if rows is None else rows
    dm = torch.from_numpy(plan.dense_mask()[:, rows]).to(DEV)
    q = qkv[:, rows, 2 * D:].float().view(B, len(rows), H, 64).transpose(1, 2)
    k = qkv[..., :D].float().view(B, L, H, 64).transpose(1, 2)
    v = qkv[..., D:2 * D].float().view(B, L, H, 64).transpose(1, 2)
    out = torch.empty(B, len(rows), D, device=DEV)
    for b in range(B):          # per batch entry: [H, R, L] fp32 scores
        s = torch.einsum("hrd,hld->hrl", q[b], k[b]) * 0.6931471805599453
        s = s.masked_fill(~dm[b][None], float("-inf"))
        out[b] = torch.einsum("hrl,hld->hrd", torch.softmax(s, -1), v[b]).transpose(0, 1).reshape(len(rows), D)
    return out


def _run(qkv, plan, B, L, Lp, q_row_begin=0):
    from pyflow_hip import ops
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, D, 3 * D, L * 3 * D, B, H, L, Lp)
    out = torch.zeros(B, L, D, dtype=torch.bfloat16, device=DEV)
    ops.attention(qkv, qkv, vT, out, 2 * D, 0, 0, 3 * D, L * 3 * D, B, H, L, Lp, LT, plan, 0.125, q_prescaled=True,
                  ldo=D, o_bstride=L * D, q_row_begin=q_row_begin)
    return out


def _which(plan, B, L, Lp, prescaled=True):
    from pyflow_hip import lib, ops
    d = lib.AttnDesc()
    need = int(lib.load().pf_attention_workspace_bytes(C.c_int(B), C.c_int(H), C.c_int(L)))
    ws = ops._attention_workspace(torch.device(DEV, torch.cuda.current_device()), need)
    d.Q = d.K = d.Vt = ws.data_ptr()
    d.O = ws.data_ptr() + (1 << 40)               # an output range that does not touch Q (pf_attention_which only compares addresses)
    d.ldq = d.ldk = 3 * D
    d.strideQ = d.strideK = L * 3 * D
    d.ldo = D
    d.strideO = L * D
    d.B, d.H, d.L, d.Lp, d.Lt = B, H, L, Lp, LT
    d.q_prescaled = int(prescaled)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return lib.load().pf_attention_which(C.byref(d))


def test_pair_runs_for_the_benchmark_shapes_and_matches_dense_reference():
    from pyflow_hip import ops
    from pyflow_hip.plan import SequencePlan
    B = 2
    plan = SequencePlan(CLIPS, _mask(), [16, 24, 24], DEV)
    L, Lp = plan.L, plan.Lp
    assert L == 3008 and _which(plan, B, L, Lp) == 64 and _which(plan, B, L, Lp, prescaled=False) == 32
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(B, L, 3 * D, generator=g, device=DEV)
    qkv[..., 2 * D:] *= 0.125 * ops.LOG2E
    qkv = qkv.to(torch.bfloat16)
    ref = _reference(qkv, plan, B, L)
    out = _run(qkv, plan, B, L, Lp)
    err = rel_l2(out.float().cpu(), ref.cpu())
    print(f"attention pair, L = 3008, every row vs dense fp32 reference: rel-L2 {err:.3e}")
    assert err < 1e-2
    # the 32-row kernel on the same input (no scratch offered): same tolerance, and the two agree closely
    from pyflow_hip import lib
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, D, 3 * D, L * 3 * D, B, H, L, Lp)
    out32 = torch.zeros_like(out)
    d = lib.AttnDesc()
    d.Q, d.K, d.Vt, d.O = qkv.data_ptr() + 2 * 2 * D, qkv.data_ptr(), vT.data_ptr(), out32.data_ptr()
    d.ldq = d.ldk = 3 * D
    d.ldo = D
    d.strideQ = d.strideK = L * 3 * D
    d.strideO = L * D
    d.strideVt_b, d.strideVt_h = H * 64 * Lp, 64 * Lp
    d.B, d.H, d.L, d.Lp, d.Lt = B, H, L, Lp, LT
    d.a_lo, d.a_hi, d.b_hi = plan.a_lo.data_ptr(), plan.a_hi.data_ptr(), plan.b_hi.data_ptr()
    d.tile_kv_end = plan.tile_kv_end.data_ptr()
    d.scale, d.q_prescaled = 0.125, 1
    lib.check(lib.load().pf_attention_bf16(C.byref(d), lib.stream()))
    assert rel_l2(out32.float().cpu(), ref.cpu()) < 1e-2
    assert rel_l2(out.float().cpu(), out32.float().cpu()) < 5e-3
    # last-block form: rows below q_row_begin are not needed.  The remaining rows are 4 x 60 = 240 workgroups of 256 rows: since
    # round 4 the pair takes them with every query tile's key range cut in two (KV split); rows below q_row_begin untouched
    r0 = L - plan.n_cur
    tail = _run(qkv, plan, B, L, Lp, q_row_begin=r0)
    assert (tail[:, :r0] == 0).all()
    assert rel_l2(tail[:, r0:].float().cpu(), out[:, r0:].float().cpu()) < 2e-3
    assert rel_l2(tail[:, r0:].float().cpu(), out32[:, r0:].float().cpu()) < 5e-3


@pytest.mark.parametrize("case", ["overflow", "underflow", "mixed"])
def test_fix_up_pass_recomputes_rows_outside_the_fast_window(case):
    """scores far beyond what exp2 represents without a running maximum: the fast pass must flag those rows and the
    fix-up pass (running maximum, deferred rescale) must produce them; rows inside the window are untouched by it"""
    from pyflow_hip import ops
    from pyflow_hip.plan import SequencePlan
    B = 2
    plan = SequencePlan(CLIPS, _mask(), [16, 24, 24], DEV)
    L, Lp = plan.L, plan.Lp
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(B, L, 3 * D, generator=g, device=DEV)
    qkv[..., 2 * D:] *= 0.125 * ops.LOG2E
    q = qkv[..., 2 * D:]
    k = qkv[..., :D]
    if case in ("overflow", "mixed"):
        # rows 700..899 (frames 2 / 3): q shifted, 40 keys of frame 3 shifted the same way: scores of ~ +380 where visible,
        # +-100 elsewhere in those rows
        q[:, 700:900] += 2.0
        k[:, 1000:1040] += 3.0
    if case in ("underflow", "mixed"):
        # rows 2000..2199: every score ~ -250 (q anti-aligned with a component all keys share)
        q[:, 2000:2200] -= 3.0
        k += 1.3
    qkv = qkv.to(torch.bfloat16)
    ref = _reference(qkv, plan, B, L)
    assert torch.isfinite(ref).all()
    sc = torch.einsum("bld,bmd->blm", qkv[:, :, 2 * D:2 * D + 64].float(), qkv[:, :, :64].float())
    print(f"{case}: head-0 score range {sc.min().item():.0f} .. {sc.max().item():.0f} (base-2 exponents)")
    assert sc.abs().max() > 150
    out = _run(qkv, plan, B, L, Lp)
    assert torch.isfinite(out.float()).all()
    err = rel_l2(out.float().cpu(), ref.cpu())
    print(f"{case}: attention pair vs dense fp32 reference rel-L2 {err:.3e}")
    assert err < 1e-2


@pytest.mark.parametrize("case", ["benign", "overflow", "underflow", "mixed"])
def test_pair_in_place_output_over_q(case):
    """the DiT's call form (flux.py): O aliases Q -- same address, leading dimension and batch stride.  The fast pass must
    not store the rows of a wave it flags, so that the fix-up launch still reads the caller's Q there (the round-3 pair
    overwrote them and recomputed from garbage); result == the out-of-place run bit for bit."""
    from pyflow_hip import ops
    from pyflow_hip.plan import SequencePlan
    B = 2
    plan = SequencePlan(CLIPS, _mask(), [16, 24, 24], DEV)
    L, Lp = plan.L, plan.Lp
    g = torch.Generator(device=DEV).manual_seed(7)
    qkv = torch.randn(B, L, 3 * D, generator=g, device=DEV)
    qkv[..., 2 * D:] *= 0.125 * ops.LOG2E
    q = qkv[..., 2 * D:]
    k = qkv[..., :D]
    if case in ("overflow", "mixed"):
        q[:, 700:900] += 2.0
        k[:, 1000:1040] += 3.0
    if case in ("underflow", "mixed"):
        q[:, 2000:2200] -= 3.0
        k += 1.3
    qkv = qkv.to(torch.bfloat16)
    ref = _reference(qkv, plan, B, L)
    out = _run(qkv, plan, B, L, Lp)                      # separate output
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-2
    vT = torch.zeros(B, H, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, D, 3 * D, L * 3 * D, B, H, L, Lp)
    buf = qkv.clone()
    kv_before = buf[..., :2 * D].clone()
    ops.attention(buf, buf, vT, buf, 2 * D, 0, 2 * D, 3 * D, L * 3 * D, B, H, L, Lp, LT, plan, 0.125, q_prescaled=True)
    assert torch.equal(buf[..., :2 * D], kv_before)      # K | V columns untouched
    inplace = buf[..., 2 * D:]
    assert torch.isfinite(inplace.float()).all()
    assert torch.equal(inplace, out), f"{case}: in-place differs from out-of-place, rel {rel_l2(inplace.float().cpu(), out.float().cpu()):.3e}"


def test_pair_is_not_chosen_for_a_partial_overlap_of_output_and_q():
    """O inside Q's address range but not the in-place form (other leading dimension): stays with the 128-row kernel"""
    from pyflow_hip import lib, ops
    from pyflow_hip.plan import SequencePlan
    B = 2
    plan = SequencePlan(CLIPS, _mask(), [16, 24, 24], DEV)
    L, Lp = plan.L, plan.Lp
    need = int(lib.load().pf_attention_workspace_bytes(C.c_int(B), C.c_int(H), C.c_int(L)))
    ws = ops._attention_workspace(torch.device(DEV, torch.cuda.current_device()), need)
    buf = torch.empty(B, L, 3 * D, dtype=torch.bfloat16, device=DEV)
    d = lib.AttnDesc()
    d.K = d.Vt = buf.data_ptr()
    d.Q = buf.data_ptr() + 2 * 2 * D
    d.ldq = d.ldk = 3 * D
    d.strideQ = d.strideK = L * 3 * D
    d.B, d.H, d.L, d.Lp, d.Lt = B, H, L, Lp, LT
    d.q_prescaled = 1
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.O, d.ldo, d.strideO = d.Q, 3 * D, L * 3 * D
    assert lib.load().pf_attention_which(C.byref(d)) == 64          # exact in-place form
    d.O, d.ldo, d.strideO = buf.data_ptr() + 2 * 2 * D + 16 * 3 * D * 2, 3 * D, L * 3 * D   # shifted by 16 rows
    assert lib.load().pf_attention_which(C.byref(d)) == 32
    d.O, d.ldo, d.strideO = d.Q, D, L * D                          # same start, other leading dimension
    assert lib.load().pf_attention_which(C.byref(d)) == 32


# ---- KV split: launches with too few workgroups for the chip (a sequence-parallel rank's few heads) ------------------------
def _few_heads_case(Hh, clips, case, seed):
    from pyflow_hip import ops
    from pyflow_hip.plan import SequencePlan
    B, Dh = 2, Hh * 64
    plan = SequencePlan(clips, _mask(), [16, 24, 24], DEV)
    L, Lp = plan.L, plan.Lp
    g = torch.Generator(device=DEV).manual_seed(seed)
    qkv = torch.randn(B, L, 3 * Dh, generator=g, device=DEV)
    qkv[..., 2 * Dh:] *= 0.125 * ops.LOG2E
    q, k = qkv[..., 2 * Dh:], qkv[..., :Dh]
    if case in ("overflow", "mixed"):
        q[:, 700:900] += 2.0
        k[:, 1000:1040] += 3.0
    if case in ("underflow", "mixed"):
        q[:, 2000:2200] -= 3.0
        k += 1.3
    return plan, qkv.to(torch.bfloat16), B, Dh, L, Lp


def _attend(qkv, plan, B, Hh, Dh, L, Lp, q_row_begin=0, scratch=True, out=None):
    """out-of-place attention through the C ABI; scratch=False: no workspace offered -> the 128-row kernel"""
    from pyflow_hip import lib, ops
    vT = torch.zeros(B, Hh, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, Dh, 3 * Dh, L * 3 * Dh, B, Hh, L, Lp)
    out = torch.zeros(B, L, Dh, dtype=torch.bfloat16, device=DEV) if out is None else out
    d = lib.AttnDesc()
    d.Q, d.K, d.Vt, d.O = qkv.data_ptr() + 2 * 2 * Dh, qkv.data_ptr(), vT.data_ptr(), out.data_ptr()
    d.ldq = d.ldk = 3 * Dh
    d.ldo = Dh
    d.strideQ = d.strideK = L * 3 * Dh
    d.strideO = L * Dh
    d.strideVt_b, d.strideVt_h = Hh * 64 * Lp, 64 * Lp
    d.B, d.H, d.L, d.Lp, d.Lt = B, Hh, L, Lp, LT
    d.a_lo, d.a_hi, d.b_hi = plan.a_lo.data_ptr(), plan.a_hi.data_ptr(), plan.b_hi.data_ptr()
    d.tile_kv_end = plan.tile_kv_end.data_ptr()
    d.scale, d.q_prescaled, d.q_row_begin = 0.125, 1, q_row_begin
    if scratch:
        need = int(lib.load().pf_attention_workspace_bytes(C.c_int(B), C.c_int(Hh), C.c_int(L)))
        ws = ops._attention_workspace(torch.device(DEV, torch.cuda.current_device()), need)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    which = lib.load().pf_attention_which(C.byref(d))
    lib.check(lib.load().pf_attention_bf16(C.byref(d), lib.stream()))
    return out, which


def _dense_ref(qkv, plan, B, Hh, Dh, L, rows):
    dm = torch.from_numpy(plan.dense_mask()[:, rows]).to(DEV)
    q = qkv[:, rows, 2 * Dh:].float().view(B, len(rows), Hh, 64).transpose(1, 2)
    k = qkv[..., :Dh].float().view(B, L, Hh, 64).transpose(1, 2)
    v = qkv[..., Dh:2 * Dh].float().view(B, L, Hh, 64).transpose(1, 2)
    s = torch.einsum("bhrd,bhld->bhrl", q, k) * 0.6931471805599453
    s = s.masked_fill(~dm[:, None], float("-inf"))
    return torch.einsum("bhrl,bhld->bhrd", torch.softmax(s, -1), v).transpose(1, 2).reshape(B, len(rows), Dh)


@pytest.mark.parametrize("case", ["benign", "overflow", "underflow", "mixed"])
def test_kv_split_pair_few_heads_every_row(case):
    """6 heads x 2 x 12 query tiles = 144 workgroups at L = 3 008: the key range of every query tile is cut into 4 parts (SPLIT),
    the parts are added and range-checked (COMBINE), flagged waves recomputed (FIXUP).  Every row against the dense fp32
    reference (1e-2, one attention op) and against the 128-row kernel."""
    Hh = 6
    plan, qkv, B, Dh, L, Lp = _few_heads_case(Hh, CLIPS, case, 11)
    out, which = _attend(qkv, plan, B, Hh, Dh, L, Lp)
    assert which == 64 + 4
    assert torch.isfinite(out.float()).all()
    ref = _dense_ref(qkv, plan, B, Hh, Dh, L, list(range(L)))
    err = rel_l2(out.float().cpu(), ref.cpu())
    print(f"KV-split pair, 6 heads, L = 3008, {case}: rel-L2 vs dense fp32 reference {err:.3e}")
    assert err < 1e-2
    out32, which32 = _attend(qkv, plan, B, Hh, Dh, L, Lp, scratch=False)
    assert which32 == 32 and rel_l2(out.float().cpu(), out32.float().cpu()) < 5e-3
    again, _ = _attend(qkv, plan, B, Hh, Dh, L, Lp)
    assert torch.equal(out, again)                       # parts are added in part order: repeatable


def test_kv_split_pair_rank_shape_at_the_headline_length():
    """a sequence-parallel rank at P = 8: 4 heads, batch 2, L = 15 488 -> 488 workgroups, each query tile in 2 key ranges;
    sampled rows against fp32, the last-block form (q_row_begin on an odd 128-row tile) leaves the rows below untouched"""
    Hh = 4
    clips = [(28, 24, 40), (1, 48, 80), (1, 96, 160), (1, 96, 160)]
    plan, qkv, B, Dh, L, Lp = _few_heads_case(Hh, clips, "benign", 12)
    assert L == 15488
    out, which = _attend(qkv, plan, B, Hh, Dh, L, Lp)
    assert which == 64 + 2
    rows = [0, 39, 40, 127, 128, 367, 368, 6847, 6848, 7807, 7808, 11647, 11648, 15487]
    ref = _dense_ref(qkv, plan, B, Hh, Dh, L, rows)
    assert rel_l2(out[:, rows].float().cpu(), ref.cpu()) < 1e-2
    out32, _ = _attend(qkv, plan, B, Hh, Dh, L, Lp, scratch=False)
    assert rel_l2(out.float().cpu(), out32.float().cpu()) < 5e-3
    r0 = L - plan.n_cur
    tail = torch.full_like(out, 7.0)
    tail, wt = _attend(qkv, plan, B, Hh, Dh, L, Lp, q_row_begin=r0, out=tail)
    assert wt > 64                                         # 16 query tiles x 8 = 128 workgroups -> 4 parts
    assert (tail[:, :r0].float() == 7.0).all()
    assert rel_l2(tail[:, r0:].float().cpu(), out[:, r0:].float().cpu()) < 2e-3      # other part boundaries: fp32 summation order


# ---- V token-major (ABI 4: pf_attn_desc.V): the kernels transpose V on the way out of LDS (ds_read_b64_tr_b16) ------------------
@pytest.mark.parametrize("Hh,clips,expect", [(30, CLIPS, 64), (6, CLIPS, 68), (2, [(1, 24, 40), (1, 24, 40)], 32)])
def test_row_major_v_equals_the_transposed_image_path(Hh, clips, expect):
    """the same attention with V read token-major (no pf_v_transpose pass) and with the V^T image: every kernel form -- the
    pair, the KV-split pair, the 128-row kernel -- must give the SAME BITS (same values into the same MFMAs in the same
    order; only the route of V through LDS differs), plus the dense fp32 reference for the small case"""
    from pyflow_hip import lib, ops
    plan, qkv, B, Dh, L, Lp = _few_heads_case(Hh, clips, "mixed" if L_ok(clips) else "benign", 21)
    out_t, which = _attend(qkv, plan, B, Hh, Dh, L, Lp)
    assert which == expect
    out_r = torch.zeros_like(out_t)
    d = lib.AttnDesc()
    d.Q, d.K, d.O = qkv.data_ptr() + 2 * 2 * Dh, qkv.data_ptr(), out_r.data_ptr()
    d.V, d.ldv, d.strideV = qkv.data_ptr() + 2 * Dh, 3 * Dh, L * 3 * Dh
    d.ldq = d.ldk = 3 * Dh
    d.ldo = Dh
    d.strideQ = d.strideK = L * 3 * Dh
    d.strideO = L * Dh
    d.B, d.H, d.L, d.Lp, d.Lt = B, Hh, L, Lp, LT
    d.a_lo, d.a_hi, d.b_hi = plan.a_lo.data_ptr(), plan.a_hi.data_ptr(), plan.b_hi.data_ptr()
    d.tile_kv_end = plan.tile_kv_end.data_ptr()
    d.scale, d.q_prescaled = 0.125, 1
    need = int(lib.load().pf_attention_workspace_bytes(C.c_int(B), C.c_int(Hh), C.c_int(L)))
    ws = ops._attention_workspace(torch.device(DEV, torch.cuda.current_device()), need)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    assert lib.load().pf_attention_which(C.byref(d)) == expect
    lib.check(lib.load().pf_attention_bf16(C.byref(d), lib.stream()))
    assert torch.isfinite(out_r.float()).all()
    assert torch.equal(out_r, out_t), f"row-major V differs from the V^T path: rel {rel_l2(out_r.float().cpu(), out_t.float().cpu()):.3e}"
    if L <= 1024:
        ref = _dense_ref(qkv, plan, B, Hh, Dh, L, list(range(L)))
        assert rel_l2(out_r.float().cpu(), ref.cpu()) < 1e-2
    # not pre-scaled (the 128-row kernel's other instantiation)
    d.q_prescaled, d.workspace, d.workspace_bytes = 0, None, 0
    a = torch.zeros_like(out_t)
    d.O = a.data_ptr()
    lib.check(lib.load().pf_attention_bf16(C.byref(d), lib.stream()))
    vT = torch.zeros(B, Hh, 64, Lp, dtype=torch.bfloat16, device=DEV)
    ops.v_transpose(qkv, vT, Dh, 3 * Dh, L * 3 * Dh, B, Hh, L, Lp)
    d.V, d.Vt, d.strideVt_b, d.strideVt_h = None, vT.data_ptr(), Hh * 64 * Lp, 64 * Lp
    b_ = torch.zeros_like(out_t)
    d.O = b_.data_ptr()
    lib.check(lib.load().pf_attention_bf16(C.byref(d), lib.stream()))
    assert torch.equal(a, b_)


def L_ok(clips):
    return sum(c[0] * (c[1] // 2) * (c[2] // 2) for c in clips) + LT >= 3008
