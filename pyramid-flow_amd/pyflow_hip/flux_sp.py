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

Round 3 -- production shape of the per-forward sequence:
  * BOTH exchanges are asynchronous (`trainer_misc/communicate.py:7-66` is blocking): in the single-stream blocks the MLP
    branch (proj_mlp + GELU, flux_block.py:921-922, independent of the attention) is split 2 : 1 -- two thirds of its
    columns are computed while the K|V|Q exchange (3 d columns) is in flight, the last third while the attention-output
    exchange (d columns) is; in the double-stream blocks nothing is independent of the exchanges (the text rows' K|V|Q are
    part of what is exchanged);
  * the blocks + head of a (unit, stage) are RECORDED into a launch list (pyflow_hip/cmdlist.py) when the communicator
    can be recorded -- the C-ABI communicator (`NativeComm`: pf_cmdlist_all_to_all_v / pf_cmdlist_comm_wait entries) or a
    single rank (`LocalComm`) -- and replayed by one C call per forward; torch.distributed collectives cannot be recorded
    (gloo tests, `--comm torch`): those run the same sequence eagerly;
  * the LAST block computes the attention, the MLP branch and the out-projection only for the current frame's rows
    (split_output keeps [-n_cur:], modeling_pyramid_flux.py:380): every rank saves the attention rows below them, the
    ranks that own no such row skip their MLP / out-projection launches.
Text rows stay with the ranks that own them in the merged order (rank 0 at P <= 8 for 128 text rows): their skinny GEMMs
stream the text-stream weights once and cost 0.26 ms per forward when run alone -- splitting them over
ranks would make EVERY rank stream those weights and save nothing.
"""
import ctypes as C

import torch

from . import ops
from .flux import FluxEngine
from .ops import GEMM_GATE_RES, GEMM_OUT_F32
from .sp import LocalComm, SPLayout


class _SegmentedProgram:
    """The blocks + head of one (unit, stage) for a communicator whose calls cannot be recorded (torch.distributed): the
    kernels BETWEEN two collective calls are recorded into launch lists (one C call each at replay), the collectives and the
    waits on their handles are kept as Python closures in issue order.  Replay = list, exchange, list, wait, list ... --
    ~100 short C calls + ~100 c10d calls per forward instead of ~350 Python launches (round 5; the C-ABI communicator's
    single list, which holds the exchanges too, is the other form)."""

    def __init__(self):
        from .cmdlist import CommandList
        self._new = CommandList
        self.entries = []            # ("list", CommandList) | ("call", fn, id) | ("wait", id)
        self.cur = None
        self.n_calls = 0

    def begin(self):
        assert ops.RECORDER is None, "launch lists do not nest"
        self.cur = self._new()
        ops.RECORDER = self.cur

    def _cut(self):
        """close the segment being recorded (kept if it holds anything) and open the next one"""
        if len(self.cur):
            self.entries.append(("list", self.cur))
        self.cur = self._new()
        ops.RECORDER = self.cur

    def add_call(self, fn):
        self._cut()
        self.n_calls += 1
        self.entries.append(("call", fn, self.n_calls))
        return self.n_calls

    def add_wait(self, idx):
        self._cut()
        self.entries.append(("wait", idx))

    def end(self):
        if self.cur is not None and len(self.cur):
            self.entries.append(("list", self.cur))
        self.cur = None
        ops.RECORDER = None

    def run(self, stream):
        handles = {}
        for e in self.entries:
            if e[0] == "list":
                e[1].run(stream)
            elif e[0] == "call":
                handles[e[2]] = e[1]()
            else:
                h = handles.pop(e[1], None)
                if h is not None:
                    h.wait()

    def __len__(self):
        return len(self.entries)


class _RecordingComm:
    """stands in for the engine's communicator while a _SegmentedProgram is being recorded: every exchange becomes a program
    entry (issued at replay through the real communicator), every handle wait too"""
    recordable = False

    class _Handle:
        def __init__(self, prog, idx):
            self.prog, self.idx = prog, idx

        def wait(self):
            self.prog.add_wait(self.idx)

    def __init__(self, real, prog):
        self.real, self.prog = real, prog
        self.rank, self.world = real.rank, real.world

    def all_to_all(self, recv, send, recv_splits, send_splits, async_op=False):
        r_spl, s_spl = list(recv_splits), list(send_splits)
        idx = self.prog.add_call(lambda: self.real.all_to_all(recv, send, r_spl, s_spl, async_op=async_op))
        return self._Handle(self.prog, idx) if async_op else None


class FluxEngineSP(FluxEngine):
    HEAD_MAJOR = True

    def __init__(self, state_dict, cfg, device="cuda", comm=None):
        super().__init__(state_dict, cfg, device)
        self.comm = comm if comm is not None else LocalComm()
        self._layouts = {}
        self.launch_mode = "list"          # "list": record + replay when the communicator allows it; "eager": never
        self.split_small = True            # a rank's small image GEMMs (128 x 128 kernel, < 128 workgroups) split K through scratch
        self.segment_lists = True          # torch.distributed collectives: the kernels between them replay from launch lists

    def layout(self, plan):
        key = (plan.L, plan.Lt)
        lay = self._layouts.get(key)
        if lay is None:
            lay = SPLayout(plan.L, plan.Lt, self.w.H, self.comm.world, self.comm.rank)
            self._layouts[key] = lay
        return lay

    # ---- the two exchanges (both return a wait handle or None) ----------------------------------------------------
    def _exchange_qkv(self, lay, big, ld, B, send, recv):
        """big[B][nloc][ld] (first 3d columns head-major) -> recv[L][B][my_cols]"""
        nloc = lay.nloc
        s_spl, r_spl = lay.a2a1_splits(B)
        if nloc:
            parts = [p for p in range(lay.P) if lay.heads[p]]
            ops.sp_relayout(big, send, nloc, B, ld, nloc * ld, [lay.head0[p] * lay.HEAD_COLS for p in parts],
                            [lay.heads[p] * lay.HEAD_COLS for p in parts], [sum(s_spl[:p]) for p in parts], True)
        return self.comm.all_to_all(recv, send, r_spl, s_spl, async_op=True)

    def _exchange_out_start(self, lay, obuf, B, recv):
        s_spl, r_spl = lay.a2a2_splits(B)
        return self.comm.all_to_all(recv, obuf, r_spl, s_spl, async_op=True)

    def _exchange_out_finish(self, lay, h, B, recv, dst, ld_dst, col0):
        """recv[src][row][b][src's heads x 64] -> dst[B][nloc][ld_dst] columns col0 .. col0 + d (all heads, head order)"""
        if h is not None:
            h.wait()
        _, r_spl = lay.a2a2_splits(B)
        nloc = lay.nloc
        if nloc:
            parts = [p for p in range(lay.P) if lay.heads[p]]
            ops.sp_relayout(dst, recv, nloc, B, ld_dst, nloc * ld_dst, [col0 + lay.head0[p] * 64 for p in parts],
                            [lay.heads[p] * 64 for p in parts], [sum(r_spl[:p]) for p in parts], False)

    # ---- buffers of one (plan, rank) -------------------------------------------------------------------------------
    def _sp_state(self, plan):
        w = self.w
        d = w.d
        B, L, L_img, Lp = plan.B, plan.L, plan.L_img, plan.Lp
        lay = self.layout(plan)
        nloc, mh, mc = lay.nloc, lay.my_heads, lay.my_cols
        bf = torch.bfloat16
        n_cur = plan.n_cur
        npad = w.proj_w.shape[0]
        return dict(
            lay=lay,
            hidden=self._buf("sp_hidden", B * nloc * d, bf), xn=self._buf("sp_xn", B * nloc * d, bf),
            big=self._buf("sp_big", B * nloc * 7 * d, bf), send1=self._buf("sp_send1", B * nloc * 3 * d, bf),
            recv1=self._buf("sp_recv1", L * B * max(mc, 1), bf), obuf=self._buf("sp_obuf", L * B * max(mh, 1) * 64, bf),
            recv2=self._buf("sp_recv2", B * nloc * d, bf), vT=None,
            # scratch of the text rows' skinny GEMMs: same K split as the single-process engine (one 128-row tile per
            # prompt either way), hence the same fp32 summation order and bit-identical text rows
            ws_txt=self._buf("splitk_txt", 8 << 20, torch.float32),
            ws_img=self._buf("gemm_tail", 16 << 20, torch.float32),      # 64 MiB: tail split of the large GEMMs
            tok=self._buf("tok", B * L_img * w.in_ch, bf),
            vtok=self._buf("vtok", B * n_cur * npad, torch.float32))

    def _embed_local(self, plan, clips, ctx, shared_clips, st):
        """local rows only: text rows <- cached context, image rows <- x_embedder(patchify)"""
        w = self.w
        d = w.d
        B, Lt, L_img = plan.B, plan.Lt, plan.L_img
        lay, hidden, tok = st["lay"], st["hidden"], st["tok"]
        n_txt, n_img = lay.n_txt, lay.n_img
        Ld = lay.nloc * d
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

    # ---- blocks + head over the local rows (recordable) -----------------------------------------------------------
    def _run_sp_blocks(self, plan, mod, st, debug=None):
        w = self.w
        d = w.d
        B, Lt, L, Lp = plan.B, plan.Lt, plan.L, plan.Lp
        lay = st["lay"]
        lib = ops.L.load()
        nloc, n_txt, n_img = lay.nloc, lay.n_txt, lay.n_img
        mh, mc = lay.my_heads, lay.my_cols
        hidden, xn, big, send1, recv1, obuf, recv2, vT, ws_txt, ws_img, vtok = (st[k] for k in (
            "hidden", "xn", "big", "send1", "recv1", "obuf", "recv2", "vT", "ws_txt", "ws_img", "vtok"))
        nm = w.n_mod
        Ld, L3, L4, L7 = nloc * d, nloc * 3 * d, nloc * 4 * d, nloc * 7 * d
        mlp_base = B * L3
        scale = 64 ** -0.5
        qs = scale * ops.LOG2E
        n_cur = plan.n_cur
        r_cur = L - n_cur                                   # first row of the current frame (global)
        la0 = min(max(r_cur - lay.r0, 0), nloc)             # first LOCAL row the last block has to produce
        dead_ok = self.skip_dead_rows and n_cur < L

        def ln(rows, x_off, sh, sc):
            if rows:
                ops.ln_modulate(hidden, xn, (mod, sh), (mod, sc), d, B, rows, Ld, Ld, d, d, nm, x_off=x_off, y_off=x_off)

        # QK-RMSNorm + RoPE (flux_block.py:846-858) are per row and per head: since round 5 a rank's K / Q leave its OWN
        # projection normed and rotated (pf_gemm_desc.qk_* with the head-major column layout: in the persistent kernel's
        # epilogue, by the library's separate pass for the small launches) -- BEFORE the exchange; the received matrix goes
        # straight into the attention.  `fuse_qk = False`: the round 3-4 form (one pass over the received matrix).
        fuse = self.fuse_qk

        def qk_of(wq, wk, row0):
            """descriptor part for a K|V|Q projection whose row 0 is global row `row0`"""
            if not fuse:
                return None
            return dict(rope=plan.rope, wq=wq, wk=wk, d=d, k_col0=0, q_col0=128, head_stride=lay.HEAD_COLS, row0=row0,
                        eps=w.qk_eps, q_scale=qs)

        def attend(ld, norms, overlap=None, q_row_begin=0):
            """big (first 3d columns, head-major) -> obuf = attention output of my heads for all rows.
            `overlap`: work that does not depend on the exchange, queued while the all-to-all is in flight."""
            h = self._exchange_qkv(lay, big, ld, B, send1, recv1)
            if overlap is not None:
                overlap()
            if h is not None:
                h.wait()
            if mh:
                if not fuse:
                    ops.qk_norm_rope(recv1, B * mc, mc, 128, 0, *norms, plan.rope, B, L, Lt, mh, q_scale=qs,
                                     head_stride=lay.HEAD_COLS, eps=w.qk_eps)
                # V = columns 64 .. 127 of every head's [k | v | q] block of the received matrix: read token-major
                ops.attention(recv1, recv1, None, obuf, 128, 0, 0, B * mc, mc, B, mh, L, Lp, Lt, plan, scale,
                              q_prescaled=True, head_stride_qk=lay.HEAD_COLS, ldo=B * mh * 64, o_bstride=mh * 64,
                              q_row_begin=q_row_begin, v_off=64)

        n_dbl = len(w.dbl)
        for bi, blk in enumerate(w.dbl):
            mb = blk["mod"]
            pre_only = blk["pre_only"]         # last MMDiT block: the text stream only feeds the attention
            # last block of a model without single blocks: only the current frame's rows go on
            tail = dead_ok and pre_only and not w.sgl and bi == n_dbl - 1
            i0 = max(la0, n_txt) if tail else n_txt          # first local image row computed in full after the attention
            n_act = nloc - i0
            ln(n_img, n_txt * d, mb + 0, mb + d)
            if pre_only:
                ln(n_txt, 0, mb + 7 * d, mb + 6 * d)      # AdaLayerNormContinuous: (scale, shift)
            else:
                ln(n_txt, 0, mb + 6 * d, mb + 7 * d)
            # the rank that owns the text rows (rank 0) runs every projection of the block for both streams: GROUPED launches
            # (pf_gemm_desc.A2 ...: the text rows' tiles in the image rows' persistent launch; the library runs two launches
            # where the image problem does not take the persistent kernel) -- `group_text = False`: always two launches
            # -- per projection, and only where the image problem takes the persistent kernel: everywhere else the text rows keep
            # their own launch with their own K-split scratch (the single-rank engine's summation order, bit for bit)
            grp = self.group_text and fuse and n_img > 0 and n_txt > 0

            def groups(M_, N_, K_):
                # the single-rank engine's rule (FluxEngine._groups_text): the persistent kernel as a whole-round launch (>= 192
                # tiles) -- pf_gemm_which reports 8 for mid-size problems too, which the library serves by a K split through
                # scratch and never as a grouped launch; those keep the text rows' own launch with their own scratch
                return (grp and M_ > 0 and lib.pf_gemm_which(C.c_int(M_), C.c_int(B), C.c_int(N_), C.c_int(K_)) == 8 and
                        -(-M_ // 256) * B * -(-N_ // 256) >= 192)
            g_kvq = groups(n_img, 3 * d, d)
            nq_t = blk["norm_added_q"] if blk["norm_added_q"] is not None else blk["norm_q"]
            nk_t = blk["norm_added_k"] if blk["norm_added_k"] is not None else blk["norm_k"]
            if n_img:
                ops.gemm(xn, blk["kvq_img"][0], big, n_img, 3 * d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                         strideA=Ld, strideC=L3, a_off=n_txt * d, c_off=n_txt * 3 * d, tail_workspace=ws_img, split_small=self.split_small,
                         qk=qk_of(blk["norm_q"], blk["norm_k"], lay.r0 + n_txt),
                         second=dict(M=n_txt, W=blk["kvq_txt"][0], bias=blk["kvq_txt"][1], wq=nq_t, wk=nk_t, row0=lay.r0) if g_kvq else None)
            if n_txt and not g_kvq:         # text rows: the added-projection gains (norm_added_q / k; the image gains where the model has none)
                ops.gemm(xn, blk["kvq_txt"][0], big, n_txt, 3 * d, d, d, d, 3 * d, bias=blk["kvq_txt"][1], batch=B,
                         strideA=Ld, strideC=L3, workspace=ws_txt, qk=qk_of(nq_t, nk_t, lay.r0))
            norms = (blk["norm_q"], blk["norm_k"], blk["norm_added_q"], blk["norm_added_k"])
            attend(3 * d, norms, q_row_begin=r_cur if tail else 0)
            h2 = self._exchange_out_start(lay, obuf, B, recv2)
            self._exchange_out_finish(lay, h2, B, recv2, big, 3 * d, 0)          # attention rows -> big[..., 0:d]
            post = grp and n_act > 0 and not pre_only          # the post-attention GEMMs of both streams
            g_o = post and groups(n_act, d, d)
            g_ff = post and groups(n_act, 4 * d, d) and groups(n_act, d, 4 * d)          # (the MLP's two GEMMs: both or neither)
            if n_act > 0:
                ops.gemm(big, blk["o_img"][0], hidden, n_act, d, d, 3 * d, d, d, bias=blk["o_img"][1], res=hidden,
                         gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=i0 * 3 * d, c_off=i0 * d, r_off=i0 * d, tail_workspace=ws_img, split_small=self.split_small,
                         second=dict(M=n_txt, W=blk["o_txt"][0], bias=blk["o_txt"][1], gate_off=mb + 8 * d) if g_o else None)
            if n_txt and not pre_only and not g_o:
                ops.gemm(big, blk["o_txt"][0], hidden, n_txt, d, d, 3 * d, d, d, bias=blk["o_txt"][1], res=hidden,
                         gate=mod, gate_off=mb + 8 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, workspace=ws_txt)
            ln(n_act if n_act > 0 else 0, i0 * d, mb + 3 * d, mb + 4 * d)
            if not pre_only:
                ln(n_txt, 0, mb + 9 * d, mb + 10 * d)
            if n_act > 0:
                ops.gemm(xn, blk["ff1_img"][0], big, n_act, 4 * d, d, d, d, 4 * d, bias=blk["ff1_img"][1], batch=B,
                         strideA=Ld, strideC=L4, gelu_from=0, a_off=i0 * d, c_off=mlp_base + i0 * 4 * d, tail_workspace=ws_img, split_small=self.split_small,
                         second=dict(M=n_txt, W=blk["ff1_txt"][0], bias=blk["ff1_txt"][1], c_off=mlp_base) if g_ff else None)
                ops.gemm(big, blk["ff2_img"][0], hidden, n_act, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_img"][1],
                         res=hidden, gate=mod, gate_off=mb + 5 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base + i0 * 4 * d, c_off=i0 * d,
                         r_off=i0 * d, tail_workspace=ws_img, split_small=self.split_small,
                         second=dict(M=n_txt, W=blk["ff2_txt"][0], bias=blk["ff2_txt"][1], a_off=mlp_base, gate_off=mb + 11 * d) if g_ff else None)
            if n_txt and not pre_only and not g_ff:
                ops.gemm(xn, blk["ff1_txt"][0], big, n_txt, 4 * d, d, d, d, 4 * d, bias=blk["ff1_txt"][1], batch=B,
                         strideA=Ld, strideC=L4, gelu_from=0, c_off=mlp_base, workspace=ws_txt)
                ops.gemm(big, blk["ff2_txt"][0], hidden, n_txt, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_txt"][1],
                         res=hidden, gate=mod, gate_off=mb + 11 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base, workspace=ws_txt)

        # MLP branch of a single block in two column groups (2 : 1): the first under the K|V|Q exchange, the second under
        # the attention-output exchange.  256-column multiples so that both launches take the same tile kernels.
        n1 = (4 * d * 2 // 3) // 256 * 256
        n_sgl = len(w.sgl)
        for bi, blk in enumerate(w.sgl):
            mb = blk["mod"]
            last = dead_ok and bi == n_sgl - 1
            i0 = la0 if last else 0                        # first local row whose MLP branch / out-projection is needed
            n_act = nloc - i0
            ln(nloc, 0, mb, mb + d)
            # K|V|Q first, then the MLP branch (proj_mlp + GELU, flux_block.py:921-922) while the exchanges fly
            if nloc:
                ops.gemm(xn, blk["kvqm"][0], big, nloc, 3 * d, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B, strideA=Ld,
                         strideC=L7, tail_workspace=ws_img, split_small=self.split_small,
                         qk=qk_of(blk["norm_q"], blk["norm_k"], lay.r0))

            def mlp_cols(c0, nc, blk=blk, i0=i0, n_act=n_act):
                if n_act > 0 and nc > 0:
                    ops.gemm(xn, blk["kvqm"][0], big, n_act, nc, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B,
                             strideA=Ld, strideC=L7, gelu_from=0, a_off=i0 * d, w_off=(3 * d + c0) * d,
                             c_off=i0 * 7 * d + 3 * d + c0, bias_off=3 * d + c0, tail_workspace=ws_img, split_small=self.split_small)
            norms = (blk["norm_q"], blk["norm_k"], None, None)
            attend(7 * d, norms, overlap=lambda: mlp_cols(0, n1), q_row_begin=r_cur if last else 0)
            h2 = self._exchange_out_start(lay, obuf, B, recv2)
            mlp_cols(n1, 4 * d - n1)
            self._exchange_out_finish(lay, h2, B, recv2, big, 7 * d, 2 * d)      # [attn | mlp] = big[..., 2d:7d)
            if n_act > 0:
                ops.gemm(big, blk["out"][0], hidden, n_act, d, 5 * d, 7 * d, 5 * d, d, bias=blk["out"][1], res=hidden,
                         gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L7, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=i0 * 7 * d + 2 * d, c_off=i0 * d, r_off=i0 * d, tail_workspace=ws_img, split_small=self.split_small)
        if debug is not None:
            debug["hidden_final_local"] = hidden[:B * nloc * d].view(B, nloc, d).clone()

        # ---- norm_out + proj_out on my part of the current frame's rows (summed over the ranks by the caller)
        npad = w.proj_w.shape[0]
        lo = max(lay.r0, r_cur)
        cnt = lay.r1 - lo
        if cnt > 0:
            fo = (lo - lay.r0) * d
            mf = w.mod_final
            ops.ln_modulate(hidden, xn, (mod, mf + d), (mod, mf), d, B, cnt, Ld, Ld, d, d, nm, x_off=fo, y_off=fo)
            ops.gemm(xn, w.proj_w, vtok, cnt, npad, d, d, d, npad, bias=w.proj_b, batch=B, strideA=Ld,
                     strideC=n_cur * npad, flags=GEMM_OUT_F32, a_off=fo, c_off=(lo - r_cur) * npad)

    def _run_sp_list(self, plan, mod, st):
        """the blocks + head of this (unit, stage) through a launch list recorded once: the kernels, the relayouts and --
        with the C-ABI communicator -- the exchanges and their waits are re-issued by ONE C call per forward"""
        from .cmdlist import CommandList, recording
        n_mod = plan.B * self.w.n_mod
        for _ in range(2):
            ms = self._buf("mod_fixed", n_mod, torch.float32)
            key = (id(self), self._ws_gen, self.skip_dead_rows, ops.POLICY_GEN, id(self.comm), self.split_small, self.fuse_qk, self.group_text)
            ent = getattr(plan, "_sp_list", None)
            if ent is not None and ent[0] == key:
                break
            if getattr(self.comm, "recordable", False):
                cl = CommandList()
                with recording(cl):
                    self._run_sp_blocks(plan, ms, st)
            else:            # torch.distributed: launch-list segments between the collectives (_SegmentedProgram)
                cl = _SegmentedProgram()
                real = self.comm
                self.comm = _RecordingComm(real, cl)
                cl.begin()
                try:
                    self._run_sp_blocks(plan, ms, st)
                finally:
                    cl.end()
                    self.comm = real
            if key[1] != self._ws_gen:           # a buffer was (re)allocated while recording: record again
                st = self._sp_state(plan)
                continue
            plan._sp_list = ent = (key, cl)
            if getattr(self, "_listed_plans", None) is None:
                import weakref
                self._listed_plans = weakref.WeakSet()
            self._listed_plans.add(plan)
            break
        else:
            raise RuntimeError("launch list: the workspace did not settle")
        self._ws["mod_fixed"][:n_mod].copy_(mod.reshape(-1)[:n_mod])
        ent[1].run(torch.cuda.current_stream())

    def reserve(self, B, L, L_img, n_cur):
        """(a rank's buffers hold its row chunk only and are sized per plan: nothing to pre-size)"""
        return

    def _buf(self, name, numel, dtype):
        gen = self._ws_gen
        t = super()._buf(name, numel, dtype)
        if gen != self._ws_gen:                  # stale pointers in every recorded sequence-parallel list as well
            for p_ in list(getattr(self, "_listed_plans", None) or ()):
                p_.__dict__.pop("_sp_list", None)
        return t

    # ---- forward ----------------------------------------------------------------------------------------------
    def forward_tokens(self, plan, clips, timesteps, pooled, ctx=None, shared_clips=False, debug=None):
        w = self.w
        B = plan.B
        ctx = ctx if ctx is not None else self._ctx
        mod, _ = self.conditioning(timesteps, pooled)
        st = self._sp_state(plan)
        self._embed_local(plan, clips, ctx, shared_clips, st)
        n_cur = plan.n_cur
        npad = w.proj_w.shape[0]
        vtok = st["vtok"]
        if self.comm.world > 1:
            vtok[:B * n_cur * npad].zero_()
        # (the C-ABI communicator and a single rank: ONE list incl. the exchanges; torch.distributed: list segments between
        #  the collectives -- FluxEngineSP.segment_lists = False keeps those eager, the round 3-4 behaviour)
        recordable = (self.launch_mode != "eager" and debug is None and not ops.PROFILER.enabled
                      and (getattr(self.comm, "recordable", False) or self.segment_lists))
        if recordable:
            self._run_sp_list(plan, mod, st)
        else:
            self._run_sp_blocks(plan, mod, st, debug)
        if self.comm.world > 1:
            self.comm.all_reduce(vtok[:B * n_cur * npad])
        return vtok[:B * n_cur * npad].view(B, n_cur, npad)
