"""Sequence-parallel miniFLUX forward (one process per GPU, P ranks).

Mirrors the sequence-parallel branches of PyramidFluxTransformer.forward (modeling_pyramid_flux.py:409-426,
463-489, 522-537, 354-390) and of the attention functors (modeling_flux_block.py:266-325, 519-565) with a layout
the reference lacks:

  * every rank owns a contiguous chunk of the merged [text | image] rows for ALL 24 blocks (entry = local slice of
    replicated inputs, C7; no double->single re-partition, C12/C13);
  * around each attention two all-to-alls (C8-C11): rows x all heads  ->  all rows x my heads  ->  back.  The fused
    K|V|Q projection writes its columns head-major ([h][k|v|q][64]), so "the heads of rank p" is one contiguous
    column block; a pack copy turns it into the per-destination send chunks `[dest][row][b][cols_dest]`, and the
    received chunks `[src][row][b][my cols]` concatenate into ONE token-major matrix `[L][B][my cols]` that
    RMSNorm+RoPE, the V transpose and the attention kernel consume in place (row stride B*cols, head stride 192);
  * uneven head map (H = 30: 15|15, 8|8|7|7, 4x6|3|3) through all_to_all_single split sizes;
  * the velocity tokens of the current frame (last t*h*w rows) are summed across ranks into a replicated fp32
    buffer (C14) -- every rank then performs the identical Euler update (replaces the per-step broadcast C2).
"""
import torch

from . import ops
from .flux import FluxEngine
from .ops import GEMM_GATE_RES, GEMM_OUT_F32
from .sp import LocalComm, SPLayout


class FluxEngineSP(FluxEngine):
    HEAD_MAJOR = True

    def __init__(self, state_dict, cfg, device="cuda", comm=None):
        super().__init__(state_dict, cfg, device)
        self.comm = comm if comm is not None else LocalComm()
        self._layouts = {}

    def layout(self, plan):
        key = (plan.L, plan.Lt)
        lay = self._layouts.get(key)
        if lay is None:
            lay = SPLayout(plan.L, plan.Lt, self.w.H, self.comm.world, self.comm.rank)
            self._layouts[key] = lay
        return lay

    # ---- the two exchanges -------------------------------------------------------------------------------------
    def _exchange_qkv(self, lay, big, ld, B, send, recv, async_op=False):
        """big[B][nloc][ld] (first 3d columns head-major) -> recv[L][B][my_cols]; returns a wait handle or None"""
        nloc = lay.nloc
        s_spl, r_spl = lay.a2a1_splits(B)
        if nloc:
            parts = [p for p in range(lay.P) if lay.heads[p]]
            ops.sp_relayout(big, send, nloc, B, ld, nloc * ld, [lay.head0[p] * lay.HEAD_COLS for p in parts],
                            [lay.heads[p] * lay.HEAD_COLS for p in parts], [sum(s_spl[:p]) for p in parts], True)
        return self.comm.all_to_all(recv, send, r_spl, s_spl, async_op=async_op)

    def _exchange_out(self, lay, obuf, B, recv, dst, ld_dst, col0):
        """obuf[L][B][my_heads*64] -> dst[B][nloc][ld_dst] columns col0 .. col0 + d (all heads, head order)"""
        s_spl, r_spl = lay.a2a2_splits(B)
        self.comm.all_to_all(recv, obuf, r_spl, s_spl)
        nloc = lay.nloc
        if nloc:
            parts = [p for p in range(lay.P) if lay.heads[p]]
            ops.sp_relayout(dst, recv, nloc, B, ld_dst, nloc * ld_dst, [col0 + lay.head0[p] * 64 for p in parts],
                            [lay.heads[p] * 64 for p in parts], [sum(r_spl[:p]) for p in parts], False)

    # ---- forward ----------------------------------------------------------------------------------------------
    def forward_tokens(self, plan, clips, timesteps, pooled, ctx=None, shared_clips=False, debug=None):
        w = self.w
        d, H = w.d, w.H
        B, Lt, L, L_img, Lp = plan.B, plan.Lt, plan.L, plan.L_img, plan.Lp
        lay = self.layout(plan)
        nloc, n_txt, n_img = lay.nloc, lay.n_txt, lay.n_img
        mh, mc = lay.my_heads, lay.my_cols
        ctx = ctx if ctx is not None else self._ctx
        mod, _ = self.conditioning(timesteps, pooled)
        nm = w.n_mod
        bf = torch.bfloat16
        hidden = self._buf("sp_hidden", B * nloc * d, bf)
        xn = self._buf("sp_xn", B * nloc * d, bf)
        big = self._buf("sp_big", B * nloc * 7 * d, bf)
        send1 = self._buf("sp_send1", B * nloc * 3 * d, bf)
        recv1 = self._buf("sp_recv1", L * B * max(mc, 1), bf)
        obuf = self._buf("sp_obuf", L * B * max(mh, 1) * 64, bf)
        recv2 = self._buf("sp_recv2", B * nloc * d, bf)
        vT = self._buf("vT", B * max(mh, 1) * 64 * Lp, bf)
        # scratch of the text rows' skinny GEMMs: same K split as the single-process engine (one 128-row tile per prompt
        # either way), hence the same fp32 summation order and bit-identical text rows
        ws_txt = self._buf("splitk_txt", 8 << 20, torch.float32)
        tok = self._buf("tok", B * L_img * w.in_ch, bf)
        Ld, L3, L4, L7 = nloc * d, nloc * 3 * d, nloc * 4 * d, nloc * 7 * d
        mlp_base = B * L3
        scale = 64 ** -0.5
        qs = scale * ops.LOG2E

        # ---- embed (local rows only): text rows <- cached context, image rows <- x_embedder(patchify)
        if n_txt:
            ops.copy_rows(ctx, hidden, n_txt, d, d, d, Lt * d, Ld, B, src_off=lay.r0 * d)
        row = 0
        for cl, n in zip(clips, plan.clip_tokens):
            Cc, t, h, wd = cl.shape[1:]
            if shared_clips:
                ops.patchify(cl[0], tok, row * w.in_ch, Cc, t, h, wd, w.in_ch, L_img * w.in_ch, B)
            else:
                for b in range(B):
                    ops.patchify(cl[b], tok, (b * L_img + row) * w.in_ch, Cc, t, h, wd, w.in_ch, 0, 1)
            row += n
        if n_img and w.mmdit:     # + sincos position rows of my image tokens (mmdit embedding.py:326-352)
            ops.gemm(tok, w.x_w, hidden, n_img, d, w.in_ch, w.in_ch, w.in_ch, d, bias=w.x_b, batch=B,
                     strideA=L_img * w.in_ch, strideC=Ld, a_off=lay.img0 * w.in_ch, c_off=n_txt * d,
                     res=plan.pos, r_off=lay.img0 * d, ldr=d, strideR=0, flags=GEMM_GATE_RES)
        elif n_img:
            ops.gemm(tok, w.x_w, hidden, n_img, d, w.in_ch, w.in_ch, w.in_ch, d, bias=w.x_b, batch=B,
                     strideA=L_img * w.in_ch, strideC=Ld, a_off=lay.img0 * w.in_ch, c_off=n_txt * d)

        def ln(rows, x_off, sh, sc):
            if rows:
                ops.ln_modulate(hidden, xn, (mod, sh), (mod, sc), d, B, rows, Ld, Ld, d, d, nm, x_off=x_off, y_off=x_off)

        def attend(ld, overlap=None):
            """big (first 3d columns, head-major) -> obuf = attention output of my heads for all rows.
            `overlap`: work that does not depend on the exchange, queued while the all-to-all is in flight."""
            h = self._exchange_qkv(lay, big, ld, B, send1, recv1, async_op=overlap is not None)
            if overlap is not None:
                overlap()
            if h is not None:
                h.wait()
            if mh:
                ops.qk_norm_rope(recv1, B * mc, mc, 128, 0, *norms, plan.rope, B, L, Lt, mh, q_scale=qs,
                                 head_stride=lay.HEAD_COLS, eps=w.qk_eps)
                ops.v_transpose(recv1, vT, 64, B * mc, mc, B, mh, L, Lp, head_stride=lay.HEAD_COLS)
                ops.attention(recv1, recv1, vT, obuf, 128, 0, 0, B * mc, mc, B, mh, L, Lp, Lt, plan, scale,
                              q_prescaled=True, head_stride_qk=lay.HEAD_COLS, ldo=B * mh * 64, o_bstride=mh * 64)

        for blk in w.dbl:
            mb = blk["mod"]
            pre_only = blk["pre_only"]         # last MMDiT block: the text stream only feeds the attention
            ln(n_img, n_txt * d, mb + 0, mb + d)
            if pre_only:
                ln(n_txt, 0, mb + 7 * d, mb + 6 * d)      # AdaLayerNormContinuous: (scale, shift)
            else:
                ln(n_txt, 0, mb + 6 * d, mb + 7 * d)
            if n_img:
                ops.gemm(xn, blk["kvq_img"][0], big, n_img, 3 * d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                         strideA=Ld, strideC=L3, a_off=n_txt * d, c_off=n_txt * 3 * d)
            if n_txt:
                ops.gemm(xn, blk["kvq_txt"][0], big, n_txt, 3 * d, d, d, d, 3 * d, bias=blk["kvq_txt"][1], batch=B,
                         strideA=Ld, strideC=L3, workspace=ws_txt)
            norms = (blk["norm_q"], blk["norm_k"], blk["norm_added_q"], blk["norm_added_k"])
            attend(3 * d)
            self._exchange_out(lay, obuf, B, recv2, big, 3 * d, 0)          # attention rows -> big[..., 0:d]
            if n_img:
                ops.gemm(big, blk["o_img"][0], hidden, n_img, d, d, 3 * d, d, d, bias=blk["o_img"][1], res=hidden,
                         gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=n_txt * 3 * d, c_off=n_txt * d, r_off=n_txt * d)
            if n_txt and not pre_only:
                ops.gemm(big, blk["o_txt"][0], hidden, n_txt, d, d, 3 * d, d, d, bias=blk["o_txt"][1], res=hidden,
                         gate=mod, gate_off=mb + 8 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, workspace=ws_txt)
            ln(n_img, n_txt * d, mb + 3 * d, mb + 4 * d)
            if not pre_only:
                ln(n_txt, 0, mb + 9 * d, mb + 10 * d)
            if n_img:
                ops.gemm(xn, blk["ff1_img"][0], big, n_img, 4 * d, d, d, d, 4 * d, bias=blk["ff1_img"][1], batch=B,
                         strideA=Ld, strideC=L4, gelu_from=0, a_off=n_txt * d, c_off=mlp_base + n_txt * 4 * d)
                ops.gemm(big, blk["ff2_img"][0], hidden, n_img, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_img"][1],
                         res=hidden, gate=mod, gate_off=mb + 5 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base + n_txt * 4 * d, c_off=n_txt * d,
                         r_off=n_txt * d)
            if n_txt and not pre_only:
                ops.gemm(xn, blk["ff1_txt"][0], big, n_txt, 4 * d, d, d, d, 4 * d, bias=blk["ff1_txt"][1], batch=B,
                         strideA=Ld, strideC=L4, gelu_from=0, c_off=mlp_base, workspace=ws_txt)
                ops.gemm(big, blk["ff2_txt"][0], hidden, n_txt, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_txt"][1],
                         res=hidden, gate=mod, gate_off=mb + 11 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base, workspace=ws_txt)

        for blk in w.sgl:
            mb = blk["mod"]
            ln(nloc, 0, mb, mb + d)
            # K|V|Q first, then the MLP branch (proj_mlp + GELU, flux_block.py:921-922) while the qkv all-to-all flies
            if nloc:
                ops.gemm(xn, blk["kvqm"][0], big, nloc, 3 * d, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B, strideA=Ld,
                         strideC=L7)

            def mlp_branch(blk=blk):
                if nloc:
                    ops.gemm(xn, blk["kvqm"][0], big, nloc, 4 * d, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B,
                             strideA=Ld, strideC=L7, gelu_from=0, w_off=3 * d * d, c_off=3 * d, bias_off=3 * d)
            norms = (blk["norm_q"], blk["norm_k"], None, None)
            attend(7 * d, overlap=mlp_branch)
            self._exchange_out(lay, obuf, B, recv2, big, 7 * d, 2 * d)      # [attn | mlp] = big[..., 2d:7d)
            if nloc:
                ops.gemm(big, blk["out"][0], hidden, nloc, d, 5 * d, 7 * d, 5 * d, d, bias=blk["out"][1], res=hidden,
                         gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L7, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=2 * d)
        if debug is not None:
            debug["hidden_final_local"] = hidden[:B * nloc * d].view(B, nloc, d).clone()

        # ---- norm_out + proj_out on my part of the current frame's rows; sum the disjoint parts across ranks
        n_cur = plan.n_cur
        npad = w.proj_w.shape[0]
        vtok = self._buf("vtok", B * n_cur * npad, torch.float32)
        lo = max(lay.r0, L - n_cur)
        cnt = lay.r1 - lo
        if self.comm.world > 1:
            vtok[:B * n_cur * npad].zero_()
        if cnt > 0:
            fo = (lo - lay.r0) * d
            mf = w.mod_final
            ops.ln_modulate(hidden, xn, (mod, mf + d), (mod, mf), d, B, cnt, Ld, Ld, d, d, nm, x_off=fo, y_off=fo)
            ops.gemm(xn, w.proj_w, vtok, cnt, npad, d, d, d, npad, bias=w.proj_b, batch=B, strideA=Ld,
                     strideC=n_cur * npad, flags=GEMM_OUT_F32, a_off=fo, c_off=(lo - (L - n_cur)) * npad)
        if self.comm.world > 1:
            self.comm.all_reduce(vtok[:B * n_cur * npad])
        return vtok[:B * n_cur * npad].view(B, n_cur, npad)
