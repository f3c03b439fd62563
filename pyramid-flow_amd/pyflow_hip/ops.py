"""Thin torch-tensor wrappers over the C ABI (one function per entry point of pyflow_hip.h).

Tensors are device-memory handles only; nothing here computes in torch.
"""
import ctypes as C

import torch

from . import lib as L
from .lib import (GemmDesc, AttnDesc, AttnSmallDesc, ConvDesc, check, ptr, stream, GEMM_GATE_RES,  # noqa: F401
                  GEMM_OUT_F32, GEMM_ACT_QUICK_GELU, GEMM_ACT_GELU_ERF)


class KernelProfiler:
    """HIP-event timing of individual launches on torch's current stream (the stream every pyflow kernel is
    enqueued on).  bench.py enables it for a deterministic sample of DiT forwards; disabled = zero overhead."""

    def __init__(self):
        self.enabled = False
        self.records = {}        # name -> list of (start_event, end_event, work)

    def launch(self, name, work, fn):
        if not self.enabled:
            return fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.setdefault(name, []).append((e0, e1, work))

    def summary(self):
        """name -> dict(launches, ms_total, work_total) ; call after torch.cuda.synchronize()."""
        out = {}
        for name, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            out[name] = dict(launches=len(recs), ms_total=ms, work_total=float(sum(w for _, _, w in recs)))
        return out


PROFILER = KernelProfiler()
RECORDER = None      # a cmdlist.CommandList while a launch list is being recorded (cmdlist.recording): the recordable
                     # wrappers below append to it instead of launching



def gemm(A, W, Cout, M, N, K, lda, ldw, ldc, bias=None, res=None, gate=None, ldr=0, batch=1,
         strideA=0, strideC=0, strideR=0, gate_stride=0, gelu_from=-1, flags=0,
         a_off=0, c_off=0, r_off=0, gate_off=0, w_off=0, bias_off=0, workspace=None, tail_workspace=None, qk=None, split_small=False,
         second=None):
    """C = epi(A W^T). a_off/c_off/r_off are ELEMENT offsets into A / C / res.  workspace: fp32 scratch tensor that lets
    a skinny problem split its K range and a large one split its tail tiles (pf_gemm_desc.workspace); must not be
    shared by overlapping launches.  tail_workspace: the same, but handed over only to problems that run the persistent
    256 x 256 kernel (pf_gemm_which == 8): the compute stream's scratch, which leaves the summation order of small
    problems what it is without scratch.
    qk: dict(rope, wq, wk, d, q_col0, k_col0, row0, eps, q_scale[, head_stride]) -- QK-RMSNorm + RoPE of the K / Q column blocks of C as part
    of this GEMM (pf_gemm_desc.qk_*: in the persistent kernel's epilogue, else by the library's separate pass).
    second: dict(M, W, bias, [A, C, res, gate default to the first problem's tensors], a_off, c_off, r_off, gate_off, w_off,
    bias_off, strideA, strideC, strideR, [wq, wk, row0 with qk]) -- the second problem of a GROUPED launch (pf_gemm_desc.A2 ...:
    same N, K, leading dimensions, batch, flags; the text stream of a double block beside the image stream)."""
    lib = L.load()
    esz_c = 4 if (flags & GEMM_OUT_F32) else 2
    d = GemmDesc()
    d.A = A.data_ptr() + 2 * a_off
    d.W = W.data_ptr() + 2 * w_off
    d.C = Cout.data_ptr() + esz_c * c_off
    d.bias = (bias.data_ptr() + 4 * bias_off) if bias is not None else None
    d.res = (res.data_ptr() + 2 * r_off) if res is not None else None
    d.gate = (gate.data_ptr() + 4 * gate_off) if gate is not None else None
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.ldr = M, N, K, lda, ldw, ldc, ldr
    d.strideA, d.strideC, d.strideR = strideA, strideC, strideR
    d.gate_stride, d.batch, d.gelu_from, d.flags = gate_stride, batch, gelu_from, flags
    if workspace is None and tail_workspace is not None:
        which = lib.pf_gemm_which(C.c_int(M), C.c_int(batch), C.c_int(N), C.c_int(K))
        # split_small (the sequence-parallel engine's image rows: a rank's few rows leave the 128 x 128 kernel with < 128
        # workgroups, each a long chain of K-tiles): those problems may split K through the same scratch
        if which == 8 or (split_small and which == 0):
            workspace = tail_workspace
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if qk is not None:
        d.qk_rope, d.qk_wq, d.qk_wk = qk["rope"].data_ptr(), qk["wq"].data_ptr(), qk["wk"].data_ptr()
        d.qk_d, d.qk_q_col0, d.qk_k_col0, d.qk_row0 = qk["d"], qk.get("q_col0", -1), qk.get("k_col0", -1), qk.get("row0", 0)
        d.qk_eps, d.qk_q_scale = qk.get("eps", 1e-6), qk.get("q_scale", 1.0)
        d.qk_head_stride = qk.get("head_stride", 0)
    flops = 2.0 * M * N * K * batch
    if second is not None:
        s2 = second
        A2, C2 = s2.get("A", A), s2.get("C", Cout)
        d.A2 = A2.data_ptr() + 2 * s2.get("a_off", 0)
        d.W2 = s2["W"].data_ptr() + 2 * s2.get("w_off", 0)
        d.C2 = C2.data_ptr() + esz_c * s2.get("c_off", 0)
        d.bias2 = (s2["bias"].data_ptr() + 4 * s2.get("bias_off", 0)) if s2.get("bias") is not None else None
        r2 = s2.get("res", res)
        d.res2 = (r2.data_ptr() + 2 * s2.get("r_off", 0)) if r2 is not None else None
        g2 = s2.get("gate", gate)
        d.gate2 = (g2.data_ptr() + 4 * s2.get("gate_off", 0)) if g2 is not None else None
        d.M2 = s2["M"]
        d.strideA2, d.strideC2, d.strideR2 = s2.get("strideA", strideA), s2.get("strideC", strideC), s2.get("strideR", strideR)
        if qk is not None:
            d.qk_wq2, d.qk_wk2, d.qk_row0_2 = s2["wq"].data_ptr(), s2["wk"].data_ptr(), s2.get("row0", 0)
        flops += 2.0 * s2["M"] * N * K * batch
    rec = RECORDER
    if rec is not None:
        check(lib.pf_cmdlist_gemm(rec.h, C.byref(d), C.c_int(rec.slot)))
        return
    if PROFILER.enabled:      # attribute the launch to the kernel rocprofv3 will name
        bn = lib.pf_gemm_which_desc(C.byref(d))       # pf_gemm_bf16's own routing of THIS descriptor (scratch, QK, flavour)
        if bn == 8:       # the epilogue flavour is a template argument: same names as in a rocprofv3 kernel trace
            epi = 12 if qk is not None else (1 if (flags & GEMM_GATE_RES) else (2 if (flags & GEMM_OUT_F32) else (4 if 0 <= gelu_from < N else 0)))
        name = f"gemm8p_kernel<false, {epi}>" if bn == 8 else (f"gemm256_kernel<{bn}>" if bn > 0 else "gemm_kernel(128x128)")
    else:
        name = "gemm"
    PROFILER.launch(name, flops, lambda: check(lib.pf_gemm_bf16(C.byref(d), stream())))


POLICY_GEN = 0          # bumped by every gemm_set_policy: recorded launch lists / hipGraphs hold the kernels chosen at record time


def gemm_set_policy(force):
    """0 auto | -1 128x128 kernel only | 128/192/256 force the 256xBN kernel | 8 / -8 force / forbid the persistent
    256x256 kernel gemm8p (pf_gemm_set_policy)."""
    global POLICY_GEN
    check(L.load().pf_gemm_set_policy(C.c_int(force)))
    POLICY_GEN += 1


LOG2E = 1.4426950408889634


_ATTN_WS = {}          # device -> (int32 scratch of pf_attention_bf16's two-launch form, retired buffers kept alive)


def _attention_workspace(device, nbytes):
    """per-device scratch for the flag words of the fast / fix-up attention pair.  Grow-only; a replaced buffer stays
    referenced because recorded launch lists / captured graphs may still point at it.  Attention launches of one device
    are stream-ordered on the compute stream, so one buffer serves them all."""
    ent = _ATTN_WS.get(device)
    if ent is None or ent[0].numel() * 4 < nbytes:
        t = torch.zeros(max(nbytes // 4, 1 << 16), dtype=torch.int32, device=device)
        _ATTN_WS[device] = ent = (t, (ent[1] + [ent[0]]) if ent else [])
    return ent[0]


def attention(Q, K, Vt, O, q_off, k_off, o_off, ld, bstride, B, H, Lseq, Lp, Lt, plan, scale, q_prescaled=False,
              head_stride_qk=0, ldo=None, o_bstride=None, q_row_begin=0, v_off=None):
    """v_off (elements): V is a column block of the K buffer (token-major, same leading dimension, batch and head strides) and
    the kernels transpose it on the way out of LDS -- no pf_v_transpose pass, `Vt` may be None.  v_off=None: `Vt` is the
    V^T image written by v_transpose."""
    lib = L.load()
    d = AttnDesc()
    if v_off is not None:
        d.V, d.ldv, d.strideV = K.data_ptr() + 2 * v_off, ld, bstride
    if q_prescaled:
        need = int(lib.pf_attention_workspace_bytes(C.c_int(B), C.c_int(H), C.c_int(Lseq)))
        ws = _attention_workspace(Q.device, need)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.Q = Q.data_ptr() + 2 * q_off
    d.K = K.data_ptr() + 2 * k_off
    d.Vt = Vt.data_ptr() if Vt is not None else None
    d.O = O.data_ptr() + 2 * o_off
    d.ldq = d.ldk = ld
    d.ldo = ld if ldo is None else ldo
    d.strideQ = d.strideK = bstride
    d.strideO = bstride if o_bstride is None else o_bstride
    d.head_stride_qk = head_stride_qk
    d.q_row_begin = q_row_begin
    d.strideVt_b = H * 64 * Lp
    d.strideVt_h = 64 * Lp
    d.B, d.H, d.L, d.Lp, d.Lt = B, H, Lseq, Lp, Lt
    d.a_lo, d.a_hi, d.b_hi = plan.a_lo.data_ptr(), plan.a_hi.data_ptr(), plan.b_hi.data_ptr()
    d.tile_kv_end = plan.tile_kv_end.data_ptr()
    d.scale = scale
    d.q_prescaled = int(q_prescaled)
    rec = RECORDER
    if rec is not None:
        check(lib.pf_cmdlist_attention(rec.h, C.byref(d), C.c_int(rec.slot)))
        return
    PROFILER.launch("attention", 4.0 * plan.useful_pairs(q_row_begin) * 64 * H,
                    lambda: check(lib.pf_attention_bf16(C.byref(d), stream())))


def v_transpose(V, Vt, v_off, ldv, strideV, B, H, Lseq, Lp, head_stride=0):
    lib = L.load()
    rec = RECORDER
    fn, head, tail = (lib.pf_v_transpose, (), (stream(),)) if rec is None else (lib.pf_cmdlist_v_transpose, (rec.h,), (C.c_int(rec.slot),))
    check(fn(*head, C.c_void_p(V.data_ptr() + 2 * v_off), ptr(Vt), C.c_int(ldv), C.c_longlong(strideV),
             C.c_longlong(H * 64 * Lp), C.c_longlong(64 * Lp), C.c_int(B), C.c_int(H),
             C.c_int(Lseq), C.c_int(Lp), C.c_int(head_stride), *tail))


def ln_modulate(x, y, shift, scale, D, B, rows, x_bstride, y_bstride, ldx, ldy, mod_bstride,
                x_off=0, y_off=0, eps=1e-6):
    """shift/scale: (tensor, element offset) pairs into the fp32 modulation buffer."""
    lib = L.load()
    sh = C.c_void_p(shift[0].data_ptr() + 4 * shift[1])
    sc = C.c_void_p(scale[0].data_ptr() + 4 * scale[1])
    rec = RECORDER
    fn, head, tail = (lib.pf_ln_modulate, (), (stream(),)) if rec is None else (lib.pf_cmdlist_ln_modulate, (rec.h,), (C.c_int(rec.slot),))
    check(fn(*head, C.c_void_p(x.data_ptr() + 2 * x_off), C.c_void_p(y.data_ptr() + 2 * y_off), sh, sc,
             C.c_int(D), C.c_int(B), C.c_int(rows), C.c_longlong(x_bstride),
             C.c_longlong(y_bstride), C.c_int(ldx), C.c_int(ldy), C.c_int(mod_bstride),
             C.c_float(eps), *tail))


def qk_norm_rope(qkv, ld, bstride, q_off, k_off, wq_img, wk_img, wq_txt, wk_txt, rope, B, Lseq, Lt, H, eps=1e-6,
                 q_scale=1.0, head_stride=0):
    lib = L.load()
    rec = RECORDER
    fn, head, tail = (lib.pf_qk_norm_rope, (), (stream(),)) if rec is None else (lib.pf_cmdlist_qk_norm_rope, (rec.h,), (C.c_int(rec.slot),))
    check(fn(*head, ptr(qkv), C.c_int(ld), C.c_longlong(bstride), C.c_int(q_off), C.c_int(k_off),
             ptr(wq_img), ptr(wk_img), ptr(wq_txt), ptr(wk_txt), ptr(rope), C.c_int(B),
             C.c_int(Lseq), C.c_int(Lt), C.c_int(H), C.c_float(eps), C.c_float(q_scale), C.c_int(head_stride), *tail))


def gemv(W, bias, x, y, N, K, B, ldw=None, ldx=None, ldy=None, silu_in=False, accumulate=False, y_off=0):
    lib = L.load()
    check(lib.pf_gemv_f32(ptr(W), C.c_int(ldw or K), ptr(bias), ptr(x), C.c_int(ldx or K),
                          C.c_void_p(y.data_ptr() + 4 * y_off), C.c_int(ldy or N), C.c_int(N), C.c_int(K),
                          C.c_int(B), C.c_int(int(silu_in)), C.c_int(int(accumulate)), stream()))


def timestep_embed(out, ts, dim=256):
    lib = L.load()
    arr = (C.c_float * len(ts))(*[float(t) for t in ts])
    check(lib.pf_timestep_embed(ptr(out), C.c_int(out.stride(0)), C.c_int(len(ts)), arr, C.c_int(dim), stream()))


def patchify(x, tok, tok_off, Cc, T, H, W, ld, bstride, ncopies):
    lib = L.load()
    check(lib.pf_patchify(ptr(x), C.c_int(int(x.dtype == torch.float32)), C.c_void_p(tok.data_ptr() + 2 * tok_off),
                          C.c_int(Cc), C.c_int(T), C.c_int(H), C.c_int(W), C.c_int(ld), C.c_longlong(bstride),
                          C.c_int(ncopies), stream()))


def cfg_euler_step(v, vb_stride, ld, x, Cc, H, W, guidance, use_cfg, dsigma, round_bf16):
    lib = L.load()
    check(lib.pf_cfg_euler_step(ptr(v), C.c_longlong(vb_stride), C.c_int(ld), ptr(x), C.c_int(Cc), C.c_int(H),
                                C.c_int(W), C.c_float(guidance), C.c_int(int(use_cfg)), C.c_float(dsigma),
                                C.c_int(int(round_bf16)), stream()))


def copy_rows(src, dst, rows, D, ld_src, ld_dst, src_bstride, dst_bstride, B, dst_off=0, src_off=0):
    lib = L.load()
    rec = RECORDER
    fn, head, tail = (lib.pf_copy_rows, (), (stream(),)) if rec is None else (lib.pf_cmdlist_copy_rows, (rec.h,), (C.c_int(rec.slot),))
    check(fn(*head, C.c_void_p(src.data_ptr() + 2 * src_off), C.c_void_p(dst.data_ptr() + 2 * dst_off),
             C.c_int(rows), C.c_int(D), C.c_int(ld_src), C.c_int(ld_dst), C.c_longlong(src_bstride),
             C.c_longlong(dst_bstride), C.c_int(B), *tail))


def sp_relayout(mat, chunks, rows, B, ld, mat_bstride, col0, cols, off, to_chunks, mat_off=0):
    """pack (to_chunks) / unpack the all-to-all chunks; col0 / cols / off: per-part python lists (elements)."""
    lib = L.load()
    n = len(cols)
    rec = RECORDER
    fn, head, tail = (lib.pf_sp_relayout, (), (stream(),)) if rec is None else (lib.pf_cmdlist_sp_relayout, (rec.h,), (C.c_int(rec.slot),))
    check(fn(*head, C.c_void_p(mat.data_ptr() + 2 * mat_off), ptr(chunks), C.c_int(rows), C.c_int(B), C.c_int(ld),
             C.c_longlong(mat_bstride), C.c_int(n), (C.c_int * n)(*col0), (C.c_int * n)(*cols),
             (C.c_longlong * n)(*off), C.c_int(int(to_chunks)), *tail))


def renoise_upsample(xin, noise, xout, Cc, H, W, alpha, beta, round_bf16):
    lib = L.load()
    check(lib.pf_renoise_upsample(ptr(xin), ptr(noise), ptr(xout), C.c_int(Cc), C.c_int(H), C.c_int(W),
                                  C.c_float(alpha), C.c_float(beta), C.c_int(int(round_bf16)), stream()))


def avgpool2(xin, xout, planes, H, W, mul=1.0, round_bf16=False):
    lib = L.load()
    check(lib.pf_avgpool2(ptr(xin), ptr(xout), C.c_longlong(planes), C.c_int(H), C.c_int(W), C.c_float(mul),
                          C.c_int(int(round_bf16)), stream()))


# ---------------------------------------------------------------------------------------------- prompt encoders
def embed_rows(table, ids, out, D, n, vocab, pos=None, Lseq=0, ldo=None):
    """out[r] = table[ids[r]] (+ pos[r % Lseq]); ids int32 device tensor."""
    lib = L.load()
    check(lib.pf_embed_rows(ptr(table), ptr(ids), ptr(pos), ptr(out), C.c_int(D), C.c_int(n), C.c_int(Lseq),
                            C.c_int(D if ldo is None else ldo), C.c_int(vocab), stream()))


def rmsnorm(x, y, w, D, rows, ldx=None, ldy=None, eps=1e-6):
    lib = L.load()
    check(lib.pf_rmsnorm(ptr(x), ptr(y), ptr(w), C.c_int(D), C.c_int(rows), C.c_int(D if ldx is None else ldx),
                         C.c_int(D if ldy is None else ldy), C.c_float(eps), stream()))


def glu_mul(x, y, rows, F, ldx=None, ldy=None):
    lib = L.load()
    check(lib.pf_glu_mul(ptr(x), ptr(y), C.c_int(rows), C.c_int(F), C.c_int(2 * F if ldx is None else ldx),
                         C.c_int(F if ldy is None else ldy), stream()))


def attention_small(qkv, O, q_off, k_off, v_off, ld, ldo, B, H, Lseq, scale, bias=None, key_mask=None, causal=False):
    """Q/K/V are column blocks (element offsets q_off/k_off/v_off) of one fused projection buffer [B*Lseq][ld]."""
    lib = L.load()
    d = AttnSmallDesc()
    d.Q = qkv.data_ptr() + 2 * q_off
    d.K = qkv.data_ptr() + 2 * k_off
    d.V = qkv.data_ptr() + 2 * v_off
    d.O = O.data_ptr()
    d.ldq = d.ldk = d.ldv = ld
    d.ldo = ldo
    d.strideQ = d.strideK = d.strideV = Lseq * ld
    d.strideO = Lseq * ldo
    d.B, d.H, d.L = B, H, Lseq
    d.bias = bias.data_ptr() if bias is not None else None
    d.key_mask = key_mask.data_ptr() if key_mask is not None else None
    d.causal = int(causal)
    d.scale = scale
    PROFILER.launch("attention_small", 4.0 * B * H * Lseq * Lseq * 64,
                    lambda: check(lib.pf_attention_small_bf16(C.byref(d), stream())))
