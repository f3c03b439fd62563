"""CausalVideoVAE decode on MI355X -- drop-in for video_vae/modeling_causal_vae.py:39-519 (decode side).

Data layout (DESIGN.md "VAE data layout"): every activation is a channels-last bf16 buffer
``[2 + Tmax][H+2][W+2][Cp]``: a one-pixel zero border (the conv's spatial padding, never written) and two
leading temporal slots that hold the previous chunk's last two frames -- the reference's
``cache_front_feat`` (modeling_causal_conv.py:128-143) -- or zeros for the first chunk (== the causal
zero padding).  A CausalConv3d is then a pure implicit GEMM (pf_conv3d_bf16): rows = output pixels,
K = 27*Cp, no bounds checks.  Temporal chunking streams the latent frame by frame exactly like
``chunk_decode`` (:347-374); tiling (:468-519) loops the same program over latent windows and blends.
Pixel-shuffle / depth-to-time rearranges (modeling_resnet.py:609-617, 716-729) are folded into the conv's
store addressing (filter rows permuted at load time).
"""
import ctypes as C
import math

import torch

from . import lib as L
from . import ops
from .lib import ConvDesc, check, stream, GEMM_GATE_RES

from .refapi import DeviceModuleAPI, load_diffusers_dir  # noqa: E402

def _ru(x, m):
    return (x + m - 1) // m * m


class PBuf:
    """padded channels-last activation with 2 temporal cache slots."""

    def __init__(self, name, Tmax, H, W, Cc, device, pool=None):
        self.name, self.T, self.H, self.W, self.C = name, Tmax, H, W, Cc
        self.Cp, self.Hp, self.Wp = _ru(Cc, 64), H + 2, W + 2
        self.fs = self.Hp * self.Wp * self.Cp
        n = (Tmax + 2) * self.fs
        if pool is None:
            self.t = torch.zeros(n, dtype=torch.bfloat16, device=device)
        else:
            # storage shared by the tile programs of one decode lane (they run one at a time): the buffer of this name,
            # grown to the largest geometry seen; the view is zeroed because the padding ring must read as zeros
            st = pool.get(name)
            if st is None or st.numel() < n:
                st = torch.empty(n, dtype=torch.bfloat16, device=device)
                pool[name] = st
            self.t = st[:n]
            self.t.zero_()
        self.cur = 0          # frames valid in slots [2, 2+cur)
        self.pending = None   # list shared with the owning tile program: deferred cache-slot updates
        self.halo = None      # context-parallel mode: comm whose previous rank supplies the two cache slots

    def off(self, slot):
        return slot * self.fs + (self.Wp + 1) * self.Cp

    def exchange_halo(self):
        """temporal context parallelism (cp_pass_from_previous_rank, context_parallel_ops.py:76-114): my last two frames
        go to rank+1, slots[0:2] <- the last two frames of rank-1 (rank 0 keeps the causal zeros)."""
        n, fs = self.cur, self.fs
        assert n >= 2, "context parallelism needs >= 2 frames per rank at every temporal level"
        self.halo.shift(self.t[n * fs:(n + 2) * fs], self.t[0:2 * fs])

    def exchange_halo_start(self):
        """the same exchange without waiting for it (the reference blocks on req_recv.wait() before the conv,
        context_parallel_ops.py:110): returns a handle whose wait() orders the CURRENT stream behind the received frames,
        or None when the transport completed inline (gloo tests, one rank).  Output frames >= 2 of the conv that follows
        do not read the two cache slots and are launched before the wait."""
        n, fs = self.cur, self.fs
        assert n >= 2, "context parallelism needs >= 2 frames per rank at every temporal level"
        start = getattr(self.halo, "shift_start", None)
        if start is None:
            self.halo.shift(self.t[n * fs:(n + 2) * fs], self.t[0:2 * fs])
            return None
        return start(self.t[n * fs:(n + 2) * fs], self.t[0:2 * fs])

    def shift_cache(self):
        """slots[0:2] <- last two of slots[0:2+cur]  (cache_front_feat update, causal_conv.py:132,143).  Inside a tile
        program the update is only RECORDED (`pending`): all buffers of a chunk are shifted by one pf_shift_caches launch
        at the end of the chunk -- nothing reads the cache slots again before the next chunk."""
        if self.halo is not None or getattr(self, "transient", False):
            return            # context-parallel / one-pass mode: no chunk follows (the slots are filled by exchange_halo())
        n = self.cur
        if self.pending is not None:
            if n >= 1:
                self.pending.append((self, n))
            return
        fs = self.fs
        if n >= 2:
            self.t[0:2 * fs].copy_(self.t[n * fs:(n + 2) * fs])
        elif n == 1:
            self.t[0:fs].copy_(self.t[fs:2 * fs])
            self.t[fs:2 * fs].copy_(self.t[2 * fs:3 * fs])

    def reset(self):
        self.t[:2 * self.fs].zero_()
        self.cur = 0


class ConvW:
    """filters repacked to [Np][taps][Cp_in] bf16 (+fp32 bias), rows permuted for the upsampling stores."""

    def __init__(self, w, b, device, groups=1):
        co, ci, kt, kh, kw = w.shape
        self.kt, self.kh, self.kw = kt, kh, kw
        cp = _ru(ci, 64)
        cg = co // groups
        # output column n = g*cg + c  <-  original filter c*groups + g
        idx = torch.arange(co).view(cg, groups).t().reshape(-1)
        w = w[idx]
        b = b[idx]
        n_valid = co
        Np = _ru(co, 128)
        wp = torch.zeros(Np, kt * kh * kw, cp, dtype=torch.float32)
        wp[:co, :, :ci] = w.permute(0, 2, 3, 4, 1).reshape(co, kt * kh * kw, ci)
        bp = torch.zeros(Np, dtype=torch.float32)
        bp[:co] = b
        self.w = wp.reshape(Np, -1).to(device=device, dtype=torch.bfloat16).contiguous()
        self.b = bp.to(device)
        self.N, self.n_valid, self.Cg, self.groups, self.cin_p = Np, n_valid, cg, groups, cp


def conv(src, dst, cw, Tc, st=1, sh=1, sw=1, res=None, t_shift=0, dst_raw=None, down=1, tdown=1, later_chunk=False,
         gn_stats=None):
    """dst[frames] = conv(src frames [0, Tc+2)) (+ res).  dst: PBuf (interior, slots from 2) or raw tuple.
    down = 2: spatially strided conv (CausalDownsample2x, modeling_resnet.py:291-336): output grid = src grid / 2.
    tdown = 2: CausalTemporalDownsample2x (:458-502).  First chunk / whole clip: windows start at the first cache slot
    (two zero frames in front).  later_chunk: only ONE frame of context (modeling_causal_conv.py:139-140), i.e. the
    windows start at the second cache slot and an even chunk of Tc frames gives Tc / 2 outputs.
    Context-parallel mode (src.halo): the two cache slots come from the previous rank; output frames >= 2 do not read them
    and are launched while that exchange is in flight, frames 0-1 after it (CausalConv3d.context_parallel_forward,
    modeling_causal_conv.py:95-114, without its blocking wait).
    gn_stats: zeroed double [>= Tc][dst.C][2] region for the GroupNorm that reads dst: where the conv kernel can, its
    epilogue accumulates the statistics (pf_conv_desc.gn_stats) and dst.gn_ready = that region, so the norm skips its
    statistics pass."""
    lib = L.load()
    slot_shift = 0
    T_in = Tc
    if tdown > 1:          # causal temporal stride: frames [0, 0, x0 .. x(Tc-1)] -> floor((Tc - 1) / 2) + 1 outputs
        if later_chunk:
            assert Tc % tdown == 0, "later chunks of a strided temporal conv must hold an even number of frames"
            Tc, slot_shift = Tc // tdown, 1
        else:
            Tc = (Tc - 1) // tdown + 1
    # first temporal slot the taps touch (kt = 3: the two cache slots + the frame; kt = 1: the frame itself) and, for
    # spatial taps, the padded origin (row -1, col -1) instead of the first interior pixel
    slot0 = 2 - (cw.kt - 1) + slot_shift
    assert cw.cin_p == src.Cp

    def launch(f0, nf):
        """output frames [f0, f0 + nf) of the launch (input frames f0 * tdown ...)"""
        d = ConvDesc()
        d.X = src.t.data_ptr()
        d.W = cw.w.data_ptr()
        d.bias = cw.b.data_ptr()
        d.T, d.H, d.W_ = nf, src.H // down, src.W // down
        d.in_sh = d.in_sw = down
        d.in_st = tdown
        d.Hp, d.Wp, d.Cin = src.Hp, src.Wp, src.Cp
        d.kt, d.kh, d.kw = cw.kt, cw.kh, cw.kw
        d.in_base_off = (slot0 + f0 * tdown) * src.fs + ((src.Wp + 1) * src.Cp if cw.kh == 1 else 0)
        d.N, d.n_valid = cw.N, cw.n_valid
        d.st, d.sh, d.sw, d.Cg = st, sh, sw, cw.Cg
        # the first output frame of the WHOLE conv may be dropped (t_shift = -1: is_init_image); a sub-range that starts
        # later carries the shift in its base offset instead (its own first frame must not be dropped)
        ts_k, ts_off = (t_shift, 0) if f0 == 0 else (0, t_shift)
        if dst_raw is not None:
            t, Ht, Wt, Cp, frame0 = dst_raw
            d.Y = t.data_ptr()
            d.Hop, d.Wop, d.Cout_pitch = Ht, Wt, Cp
            d.out_base_off = (frame0 + f0 * st + ts_off) * Ht * Wt * Cp
            d.n_valid = Cp
            d.Cg = Cp
        else:
            d.Y = dst.t.data_ptr()
            d.Hop, d.Wop, d.Cout_pitch = dst.Hp, dst.Wp, dst.Cp
            d.out_base_off = dst.off(2) + (f0 * st + ts_off) * dst.fs
            assert dst.H == src.H * sh // down and dst.W == src.W * sw // down
        d.flags = GEMM_GATE_RES if res is not None else 0
        d.res = res.t.data_ptr() if res is not None else None
        if res is not None:
            assert (res.Hp, res.Wp, res.Cp) == (dst.Hp, dst.Wp, dst.Cp)
        d.out_scale = 1.0
        d.out_t_shift = ts_k
        if gn_stats is not None:
            d.gn_stats = gn_stats.data_ptr() + f0 * dst.C * 2 * 8
            d.gn_C = dst.C
            fused.append(bool(lib.pf_conv3d_fuses_gn_stats(C.byref(d))))
        name = "conv3d"
        if ops.PROFILER.enabled:          # attribute the launch to the kernel rocprofv3 will name (pf_conv3d_which)
            route = int(lib.pf_conv3d_which(C.byref(d)))
            name = "conv3d:" + {-2: "conv_halo128_kernel", -1: "conv_narrow_kernel", 8: "gemm8p_kernel<true, 0>",
                                0: "gemm_kernel<true>"}.get(route, f"gemm256_kernel<{route}, true>")
        ops.PROFILER.launch(name, 2.0 * nf * (src.H // down) * (src.W // down) * cw.n_valid * cw.kt * cw.kh * cw.kw * src.C,
                            lambda: check(lib.pf_conv3d_bf16(C.byref(d), stream())))

    fused = []
    if cw.kt == 3 and src.halo is not None:
        if tdown == 1 and Tc > 2:
            h = src.exchange_halo_start()
            launch(2, Tc - 2)                  # these frames read slots >= 2 only
            if h is not None:
                h.wait()
            launch(0, 2)
        else:
            src.exchange_halo()
            launch(0, Tc)
    else:
        launch(0, Tc)
    if dst is not None:
        dst.cur = Tc * st + t_shift
        # statistics are complete only if EVERY launch of this conv accumulated them
        dst.gn_ready = gn_stats if (gn_stats is not None and fused and all(fused)) else None


class _TileProgram:
    """all buffers + the layer sequence for one latent tile geometry (th x tw)."""

    def __init__(self, vae, th, tw, t_first, t_later, encoder=False, pool=None):
        self.vae, self.th, self.tw = vae, th, tw
        self.bufs = {}
        self.pool = pool            # dict name -> storage shared with the other programs of the same lane (or None)
        self.dev = vae.dev
        self.cw = vae.convs         # conv weight set (the clip encoder swaps in the full 3-tap encoder filters)
        cfg = vae.cfg
        # frames per chunk at each temporal level (first chunk / later chunks)
        self.tmax = [max(t_first, t_later)]
        tf_, tl = t_first, t_later
        for up in ([] if encoder else cfg["temporal_up_sample"]):
            if up:
                tf_, tl = 2 * tf_ - 1, 2 * tl
                self.tmax.append(max(tf_, tl))
        if encoder:                 # frames per temporal level of the encoder: T -> (T - 1) // 2 + 1 per temporal downsample
            for dn in vae.enc_cfg["temporal_down_sample"]:
                if dn:
                    tf_ = (tf_ - 1) // 2 + 1
                    self.tmax.append(tf_)
        # GroupNorm statistics: ONE arena with a region per norm layer, zeroed by one fill per chunk (the stats kernel
        # accumulates atomically) instead of one fill per layer
        self.stats = torch.zeros(100 * max(self.tmax) * 512 * 2, dtype=torch.float64, device=self.dev)
        self._stats_regions, self._stats_used = {}, 0
        self.fuse_gn = getattr(vae, "fuse_gn_stats", True)     # conv epilogues accumulate the next norm's statistics
        self.pending = []           # (buffer, frames) cache-slot updates of the running chunk, flushed in one launch
        n = th * tw
        ca = vae.attn_pitch
        npad = _ru(n, 128)
        self.n_tok, self.npad = n, npad
        tm = self.tmax[0]
        z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=self.dev)  # noqa: E731
        self.a_x, self.a_q, self.a_k, self.a_o = z(tm, npad, ca), z(tm, npad, ca), z(tm, npad, ca), z(tm, npad, ca)
        self.a_vt = z(ca, npad)
        self.q_block = n if n <= 4096 else 2048
        self.a_s = z(_ru(self.q_block, 128), npad)

    def buf(self, name, level_t, H, W, Cc):
        b = self.bufs.get(name)
        if b is None:
            b = PBuf(name, self.tmax[level_t], H, W, Cc, self.dev, self.pool)
            b.halo = getattr(self, "halo", None)
            b.transient = getattr(self, "transient", False)
            b.pending = self.pending
            self.bufs[name] = b
        return b

    def release(self, b):
        """TRANSIENT mode (the one-pass un-tiled decode of a context-parallel rank: no chunk follows, so no cache slot has
        to survive): a buffer's storage goes back to the allocator after its last reader has been queued -- the live set
        is then a few layers (x, n1, h at the widest level) instead of all ~70 activations of the decoder, which is what
        lets a rank hold 16 latent frames (128 frames of 768 x 1280) in HBM.  Everything runs on one stream, so the
        caching allocator may hand the block to the next layer at once.  No-op for the chunked tile programs."""
        if getattr(self, "transient", False) and b is not None and self.bufs.get(b.name) is b:
            del self.bufs[b.name]
            b.t = None

    def reset(self):
        if self.pool is not None and self.pool.get("__owner__") is not self:
            # another geometry used this lane's storage since: its data sits where this program's padding rings are
            for b in self.bufs.values():
                b.t.zero_()
                b.cur = 0
            self.pool["__owner__"] = self
            return
        # a new clip / tile starts: the two cache slots of EVERY buffer read as the causal zero padding again (causal_conv.py:
        # 128-131) -- one pf_shift_caches launch with n = 0 per 64 buffers (ABI 7; through round 5: one torch fill per buffer,
        # ~2 500 launches per video)
        items = [b for b in self.bufs.values() if b.t is not None]
        lib = L.load()
        for i0 in range(0, len(items), 64):
            part = items[i0:i0 + 64]
            k = len(part)
            check(lib.pf_shift_caches(C.c_int(k), (C.c_void_p * k)(*[b.t.data_ptr() for b in part]),
                                      (C.c_longlong * k)(*[b.fs for b in part]), (C.c_int * k)(*([0] * k)), stream()))
        for b in items:
            b.cur = 0

    def begin_chunk(self):
        self.stats[:max(self._stats_used, 1)].zero_() if self._stats_used else self.stats.zero_()
        del self.pending[:]
        for b in self.bufs.values():
            b.gn_ready = None       # statistics accumulated by a conv epilogue belong to the chunk that produced them

    def end_chunk(self):
        """flush the recorded cache-slot updates: one launch for every buffer of the program"""
        items = self.pending
        lib = L.load()
        for i0 in range(0, len(items), 64):
            part = items[i0:i0 + 64]
            k = len(part)
            check(lib.pf_shift_caches(C.c_int(k), (C.c_void_p * k)(*[b.t.data_ptr() for b, _ in part]),
                                      (C.c_longlong * k)(*[b.fs for b, _ in part]), (C.c_int * k)(*[n for _, n in part]),
                                      stream()))
        del self.pending[:]

    # ---- layer helpers -------------------------------------------------------------------------
    def gn(self, src, dst, name, silu=True, dst_raw=None):
        v = self.vae
        g, bt = v.norms[name]
        Tc = src.cur
        reg = self._stats_regions.get(name)
        if reg is None:
            size = max(self.tmax) * src.C * 2
            assert self._stats_used + size <= self.stats.numel(), "GroupNorm statistics arena too small"
            reg = self._stats_regions[name] = (self._stats_used, size)
            self._stats_used += size
        st = self.stats[reg[0]:reg[0] + Tc * src.C * 2]
        lib = L.load()
        nbytes = 2.0 * Tc * src.H * src.W * src.C                       # one pass over the activation (bf16)
        ready = getattr(src, "gn_ready", None)
        src.gn_ready = None
        if ready is not None:        # the conv that wrote src accumulated (sum, sum of squares) in its epilogue
            st = ready[:Tc * src.C * 2]
        else:
            ops.PROFILER.launch("gn_stats", nbytes, lambda: check(lib.pf_gn_stats(
                C.c_void_p(src.t.data_ptr()), C.c_void_p(st.data_ptr()), C.c_int(Tc), C.c_int(src.C),
                C.c_int(src.Cp), C.c_int(src.H), C.c_int(src.W), C.c_int(src.Hp), C.c_int(src.Wp),
                C.c_longlong(src.fs), C.c_longlong(src.off(2)), stream())))
        if dst_raw is None:
            args = (dst.t.data_ptr(), dst.Cp, dst.Hp, dst.Wp, dst.fs, dst.off(2))
            dst.cur = Tc
        else:
            t, cp = dst_raw
            args = (t.data_ptr(), cp, src.H, src.W, t.stride(0), 0)
        ops.PROFILER.launch("gn_apply", 2.0 * nbytes, lambda: check(lib.pf_gn_apply(
            C.c_void_p(src.t.data_ptr()), C.c_void_p(args[0]), C.c_void_p(st.data_ptr()),
            C.c_void_p(g.data_ptr()), C.c_void_p(bt.data_ptr()), C.c_int(Tc), C.c_int(src.C),
            C.c_int(v.groups), C.c_int(src.H), C.c_int(src.W), C.c_int(src.Cp), C.c_int(src.Hp),
            C.c_int(src.Wp), C.c_longlong(src.fs), C.c_longlong(src.off(2)), C.c_int(args[1]),
            C.c_int(args[2]), C.c_int(args[3]), C.c_longlong(args[4]), C.c_longlong(args[5]),
            C.c_float(1e-6), C.c_int(int(silu)), stream())))

    def stats_region(self, key, channels):
        """a zeroed-per-chunk region of the statistics arena for [tmax][channels][2] doubles"""
        reg = self._stats_regions.get(key)
        if reg is None:
            size = max(self.tmax) * channels * 2
            assert self._stats_used + size <= self.stats.numel(), "GroupNorm statistics arena too small"
            reg = self._stats_regions[key] = (self._stats_used, size)
            self._stats_used += size
        return self.stats[reg[0]:reg[0] + reg[1]]

    def resnet(self, x, p, lvl, out_name, cout):
        """CausalResnetBlock3D.forward (modeling_resnet.py:115-150).  The statistics of norm2's input and of the block's
        output (the next layer's norm input) are accumulated by the epilogues of conv1 / conv2 where the kernel can."""
        v = self.vae
        Tc = x.cur
        n1 = self.buf(p + "n1", lvl, x.H, x.W, x.C)
        self.gn(x, n1, p + "norm1")
        h = self.buf(p + "h", lvl, x.H, x.W, cout)
        conv(n1, h, self.cw[p + "conv1"], Tc, gn_stats=self.stats_region(p + "conv1.out", cout) if self.fuse_gn else None)
        n1.shift_cache()
        self.release(n1)
        res = x
        if (p + "conv_shortcut") in self.cw:
            res = self.buf(p + "sc", lvl, x.H, x.W, cout)
            conv(x, res, self.cw[p + "conv_shortcut"], Tc)
            self.release(x)                     # the wide input is dead once the shortcut has been taken from it
        n2 = self.buf(p + "n2", lvl, x.H, x.W, cout)
        self.gn(h, n2, p + "norm2")
        self.release(h)
        out = self.buf(out_name, lvl, x.H, x.W, cout)
        conv(n2, out, self.cw[p + "conv2"], Tc, res=res,
             gn_stats=self.stats_region(p + "conv2.out", cout) if self.fuse_gn else None)
        n2.shift_cache()
        self.release(n2)
        self.release(res)                       # = x without a shortcut conv: the block's input is not read again
        return out

    def mid_attention(self, x, out_name, side="decoder", lvl=0):
        """per-frame 1-head attention (modeling_block.py:456-460 + diffusers Attention, deprecated-attn-block form)."""
        v = self.vae
        Tc, n, npad, ca = x.cur, self.n_tok, self.npad, v.attn_pitch
        self.gn(x, None, side + ".mid_block.attentions.0.group_norm", silu=False, dst_raw=(self.a_x, ca))
        attn = v.attn if side == "decoder" else v.enc_attn
        wq, bq = attn["to_q"]
        wk, bk = attn["to_k"]
        wv, bv = attn["to_v"]
        wo, bo = attn["to_out.0"]
        ops.gemm(self.a_x, wq, self.a_q, npad, ca, ca, ca, ca, ca, bias=bq, batch=Tc, strideA=npad * ca, strideC=npad * ca)
        ops.gemm(self.a_x, wk, self.a_k, npad, ca, ca, ca, ca, ca, bias=bk, batch=Tc, strideA=npad * ca, strideC=npad * ca)
        out = self.buf(out_name, lvl, x.H, x.W, x.C)
        qb = self.q_block                       # query rows per score block (= n for tile-sized frames)
        for f in range(Tc):
            fo = f * npad * ca
            # V^T = Wv . X^T  (bias folded into the PV epilogue: rows of P sum to 1)
            ops.gemm(wv, self.a_x, self.a_vt, ca, npad, ca, ca, ca, npad, c_off=0, a_off=0, w_off=fo)
            # scores of a BLOCK of query rows at a time: S[qb, npad] instead of [n, npad] (an un-tiled 768p frame has
            # 15 360 tokens: 472 MB of scores per frame if materialised whole; 63 MB with 2 048-row blocks)
            for q0 in range(0, n, qb):
                nq = min(qb, n - q0)
                ops.gemm(self.a_q, self.a_k, self.a_s, nq, npad, ca, ca, ca, npad, a_off=fo + q0 * ca, w_off=fo)
                check(L.load().pf_softmax_rows(C.c_void_p(self.a_s.data_ptr()), C.c_int(npad), C.c_int(n), C.c_int(npad),
                                               C.c_int(nq), C.c_float(v.attn_scale), stream()))
                ops.gemm(self.a_s, self.a_vt, self.a_o, nq, ca, npad, npad, npad, ca, bias=bv, c_off=fo + q0 * ca)
        # to_out + residual, written into the padded image (1x1x1 conv form, A un-padded)
        lib = L.load()
        for f in range(Tc):
            d = ConvDesc()
            d.X = self.a_o.data_ptr() + 2 * f * npad * ca
            d.W, d.bias = wo.data_ptr(), bo.data_ptr()
            d.T, d.H, d.W_ = 1, x.H, x.W
            d.Hp, d.Wp, d.Cin = x.H, x.W, ca
            d.kt = d.kh = d.kw = 1
            d.in_base_off = 0
            d.N, d.n_valid = wo.shape[0], x.C
            d.st = d.sh = d.sw = 1
            d.Cg = wo.shape[0]
            d.Y = out.t.data_ptr()
            d.res = x.t.data_ptr()
            d.Hop, d.Wop, d.Cout_pitch = out.Hp, out.Wp, out.Cp
            d.out_base_off = out.off(2 + f)
            d.flags, d.out_scale, d.out_t_shift = GEMM_GATE_RES, 1.0, 0
            check(lib.pf_conv3d_bf16(C.byref(d), stream()))
        out.cur = Tc
        out.gn_ready = None         # written outside conv(): no statistics came with it
        return out

    # ---- one chunk through post_quant_conv + decoder (modeling_enc_dec.py:302-366) -------------------
    def run_chunk(self, z, t0, nt, h0, w0, first, out_tile, out_frame0, affine):
        v = self.vae
        cfg = v.cfg
        th, tw = self.th, self.tw
        lat = cfg["latent_channels"]
        zb = self.buf("z", 0, th, tw, lat)
        lib = L.load()
        self.begin_chunk()
        Zc, ZT, ZH, ZW = z.shape
        check(lib.pf_latent_to_nhwc(C.c_void_p(z.data_ptr()), C.c_void_p(zb.t.data_ptr()), C.c_int(Zc), C.c_int(ZT),
                                    C.c_int(ZH), C.c_int(ZW), C.c_int(t0), C.c_int(nt), C.c_int(h0), C.c_int(w0),
                                    C.c_int(th), C.c_int(tw), C.c_int(zb.Cp), C.c_int(zb.Hp), C.c_int(zb.Wp),
                                    C.c_longlong(zb.fs), C.c_longlong(zb.off(2)), C.c_float(affine[0]), C.c_float(affine[1]),
                                    C.c_float(affine[2]), C.c_float(affine[3]), stream()))
        zb.cur = nt
        pq = self.buf("pq", 0, th, tw, lat)
        conv(zb, pq, v.convs["post_quant_conv"], nt)
        self.release(zb)
        top = cfg["block_out_channels"][-1]
        x = self.buf("conv_in", 0, th, tw, top)
        conv(pq, x, v.convs["decoder.conv_in"], nt)
        pq.shift_cache()
        self.release(pq)
        x = self.resnet(x, "decoder.mid_block.resnets.0.", 0, "mid.r0", top)
        xa = self.mid_attention(x, "mid.attn")
        self.release(x)
        x = self.resnet(xa, "decoder.mid_block.resnets.1.", 0, "mid.r1", top)
        rev = list(reversed(cfg["block_out_channels"]))
        lvl = 0
        for i, co in enumerate(rev):
            p = f"decoder.up_blocks.{i}."
            for j in range(cfg["layers_per_block"][i]):
                x = self.resnet(x, p + f"resnets.{j}.", lvl, f"up{i}.r{j}", co)
            if cfg["spatial_up_sample"][i]:
                y = self.buf(f"up{i}.sp", lvl, x.H * 2, x.W * 2, co)
                conv(x, y, v.convs[p + "upsamplers.0.conv"], x.cur, sh=2, sw=2)
                x.shift_cache()
                self.release(x)
                x = y
            if cfg["temporal_up_sample"][i]:
                y = self.buf(f"up{i}.tp", lvl + 1, x.H, x.W, co)
                conv(x, y, v.convs[p + "temporal_upsamplers.0.conv"], x.cur, st=2, t_shift=-1 if first else 0)
                x.shift_cache()
                self.release(x)
                x = y
                lvl += 1
        n = self.buf("norm_out", lvl, x.H, x.W, x.C)
        self.gn(x, n, "decoder.conv_norm_out")
        self.release(x)
        conv(n, None, v.convs["decoder.conv_out"], n.cur, dst_raw=(out_tile, x.H, x.W, 8, out_frame0))
        nf = n.cur
        n.shift_cache()
        self.release(n)
        self.end_chunk()
        return nf


    # ---- a frame or a clip through encoder + quant_conv (modeling_enc_dec.py:154-198), un-chunked ---------------------
    def run_encoder(self, img, h0, w0, out_tile, t0=0, nt=None, chunked=False, first=True, out_frame0=0):
        """img [3,T,H,W] fp32 in [-1,1]; frames [t0, t0+nt) of the window (h0, w0, 8*th, 8*tw) -> moments tile
        [T'][th][tw][64] bf16 from frame `out_frame0` on.
        T = 1 (image-to-video): a single frame sees two zero frames in front of every causal conv
        (modeling_causal_conv.py:116-146), so each 3x3x3 filter reduces to its last temporal tap and the filters are
        packed as kt = 1.  T > 1: the full filters over [0, 0, x0 ..] with the temporal stride-2 downsamplers.
        chunked (chunk_encode, modeling_causal_vae.py:310-341): the two previous input frames of every 3-tap conv stay
        in the cache slots between calls (cache_front_feat, causal_conv.py:128-143); `first` = the chunk that starts
        the clip."""
        v = self.vae
        ecfg = v.enc_cfg
        T = img.shape[1] if nt is None else nt
        whole = img.shape[1]
        later = chunked and not first
        s_ = 2 ** sum(ecfg["spatial_down_sample"])
        ph, pw = self.th * s_, self.tw * s_
        xin = self.buf("e.img", 0, ph, pw, 3)
        lib = L.load()
        self.begin_chunk()
        Zc, ZT, ZH, ZW = img.shape
        check(lib.pf_latent_to_nhwc(C.c_void_p(img.data_ptr()), C.c_void_p(xin.t.data_ptr()), C.c_int(Zc), C.c_int(ZT),
                                    C.c_int(ZH), C.c_int(ZW), C.c_int(t0), C.c_int(T), C.c_int(h0), C.c_int(w0),
                                    C.c_int(ph), C.c_int(pw), C.c_int(xin.Cp), C.c_int(xin.Hp), C.c_int(xin.Wp),
                                    C.c_longlong(xin.fs), C.c_longlong(xin.off(2)), C.c_float(1.0), C.c_float(0.0),
                                    C.c_float(1.0), C.c_float(0.0), stream()))
        xin.cur = T
        self.chunked = chunked
        boc = ecfg["block_out_channels"]
        lvl = 0
        x = self.buf("e.conv_in", 0, ph, pw, boc[0])
        conv(xin, x, self.cw["encoder.conv_in"], T)
        if chunked:
            xin.shift_cache()
        for i, co in enumerate(boc):
            p = f"encoder.down_blocks.{i}."
            for j in range(ecfg["layers_per_block"][i]):
                x = self.resnet(x, p + f"resnets.{j}.", lvl, f"e.d{i}.r{j}", co)
            if ecfg["spatial_down_sample"][i]:
                y = self.buf(f"e.d{i}.sp", lvl, x.H // 2, x.W // 2, co)
                conv(x, y, self.cw[p + "downsamplers.0.conv"], x.cur, down=2)
                if chunked:
                    x.shift_cache()
                x = y
            if ecfg["temporal_down_sample"][i]:
                y = self.buf(f"e.d{i}.tp", lvl + 1, x.H, x.W, co)
                conv(x, y, self.cw[p + "temporal_downsamplers.0.conv"], x.cur, tdown=2 if whole > 1 else 1, later_chunk=later)
                if chunked:
                    x.shift_cache()
                x = y
                lvl += 1
        x = self.resnet(x, "encoder.mid_block.resnets.0.", lvl, "e.mid.r0", boc[-1])
        x = self.mid_attention(x, "e.mid.attn", side="encoder", lvl=lvl)
        x = self.resnet(x, "encoder.mid_block.resnets.1.", lvl, "e.mid.r1", boc[-1])
        n = self.buf("e.norm_out", lvl, x.H, x.W, x.C)
        self.gn(x, n, "encoder.conv_norm_out")
        m = self.buf("e.moments", lvl, x.H, x.W, 2 * ecfg["latent_channels"])
        conv(n, m, self.cw["encoder.conv_out"], n.cur)
        if chunked:
            n.shift_cache()
        conv(m, None, self.cw["quant_conv"], m.cur, dst_raw=(out_tile, x.H, x.W, out_tile.shape[-1], out_frame0))
        if chunked:
            self.end_chunk()
        else:
            del self.pending[:]          # a whole clip in one pass: the cache slots are not read again
        return m.cur


class DiagonalGaussianDistribution:
    """modeling_enc_dec.py:369-421 (the members the sampling path uses)."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, eps=None):
        if eps is None:
            eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device if generator is None
                              else generator.device, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * eps.to(self.mean.device, self.mean.dtype)

    def mode(self):
        return self.mean


class EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class CausalVideoVAE(DeviceModuleAPI):
    """decode(z, is_init_image=True, temporal_chunk=False, return_dict=True, window_size=2, tile_sample_min_size=256)
    -> DecoderOutput(sample [B,3,T,H,W]);  enable_tiling()/disable_tiling()  (modeling_causal_vae.py:183-196, 376-395)."""

    def __init__(self, state_dict, cfg=None, device="cuda"):
        from . import synth
        self.dev = torch.device(device)
        cfg_in = dict(cfg) if cfg else None
        self.cfg_in = cfg_in
        cfg = dict(cfg or synth.VAE_DEFAULT)
        if "decoder_block_out_channels" in cfg:      # reference-style config (causal_vae.py:73-116)
            cfg = dict(latent_channels=cfg.get("decoder_in_channels", 4),
                       block_out_channels=tuple(cfg["decoder_block_out_channels"]),
                       layers_per_block=tuple(cfg.get("decoder_layers_per_block", (3, 3, 3, 3))),
                       spatial_up_sample=tuple(cfg.get("decoder_spatial_up_sample", (True, True, True, False))),
                       temporal_up_sample=tuple(cfg.get("decoder_temporal_up_sample", (True, True, True, False))),
                       out_channels=cfg.get("decoder_out_channels", 3),
                       norm_num_groups=cfg.get("decoder_norm_num_groups", 32))
        self.cfg = cfg
        self.groups = cfg["norm_num_groups"]
        self.use_tiling = False
        self.downsample_scale = 8
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()
              if k.startswith(("decoder.", "post_quant_conv.", "encoder.", "quant_conv."))}
        self.has_encoder = any(k.startswith("encoder.") for k in sd)
        if self.has_encoder:
            ref_cfg = dict(cfg_in or {})
            boc = tuple(ref_cfg.get("encoder_block_out_channels", (128, 256, 512, 512)))
            self.enc_cfg = dict(latent_channels=sd["quant_conv.conv.weight"].shape[0] // 2, block_out_channels=boc,
                                layers_per_block=tuple(ref_cfg.get("encoder_layers_per_block", (2,) * len(boc))),
                                spatial_down_sample=tuple(ref_cfg.get("encoder_spatial_down_sample", (True, True, True, False))),
                                temporal_down_sample=tuple(ref_cfg.get("encoder_temporal_down_sample", (True, True, True, False))))
        self.convs, self.norms, self.attn, self.enc_attn = {}, {}, {}, {}
        self._enc_full, self.convs_clip = {}, None
        for k in sd:
            if k.endswith(".conv.weight"):
                name = k[:-len(".conv.weight")]
                groups = 4 if ".upsamplers." in name else (2 if ".temporal_upsamplers." in name else 1)
                wt = sd[k]
                if name.startswith(("encoder.", "quant_conv")) and wt.shape[2] == 3:
                    self._enc_full[name] = (wt, sd[name + ".conv.bias"])        # packed on the first clip encode
                    wt = wt[:, :, 2:3]          # single-frame encode: only the last temporal tap meets data
                self.convs[name] = ConvW(wt, sd[name + ".conv.bias"], self.dev, groups)
            elif k.endswith(".weight") and sd[k].ndim == 1:
                name = k[:-len(".weight")]
                self.norms[name] = (sd[k].to(self.dev), sd[name + ".bias"].to(self.dev))
        top = cfg["block_out_channels"][-1]
        self.attn_pitch = ca = _ru(top, 128)
        self.attn_scale = top ** -0.5
        for side, store in (("decoder", self.attn), ("encoder", self.enc_attn)):
            a = side + ".mid_block.attentions.0."
            if (a + "to_q.weight") not in sd:
                continue
            assert sd[a + "to_q.weight"].shape[0] == top, "encoder / decoder mid blocks share the attention width"
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                w = torch.zeros(ca, ca)
                w[:top, :top] = sd[a + n + ".weight"]
                b = torch.zeros(ca)
                b[:top] = sd[a + n + ".bias"]
                store[n] = (w.to(self.dev, torch.bfloat16).contiguous(), b.to(self.dev))
        self._programs = {}
        self._lane_pools = {}
        self.n_streams = 4          # concurrent tile decodes (HIP streams); 1 = strictly sequential
        # Temporal chunks of chunk_decode are run `chunk_coalesce` windows at a time.  The chunk cache makes every
        # chunking of the clip compute the same values (causal_conv.py:128-143; per-frame GroupNorm), so this changes only
        # the launch shapes: with the reference's window_size = 1 the latent-resolution layers are 1 024-row GEMMs (32
        # workgroups on 256 CUs); four windows at once quadruple their rows.  Activation memory grows by the same factor
        # (a few GB per tile program at 768p); 1 = the reference's schedule literally.
        self.chunk_coalesce = 4
        self._streams = []

    @classmethod
    def from_pretrained(cls, pretrained_model_path, torch_dtype=None, device="cuda", interpolate=False, **kwargs):
        """pipeline.py:156: `CausalVideoVAE.from_pretrained(os.path.join(model_path, 'causal_video_vae'),
        torch_dtype=, interpolate=False)` -- config.json + safetensors directory"""
        sd, cfg = load_diffusers_dir(pretrained_model_path)
        return cls(sd, cfg, device)

    @property
    def config(self):
        import types
        return types.SimpleNamespace(**dict(self.cfg_in or self.cfg))

    def enable_tiling(self, use_tiling=True):
        self.use_tiling = use_tiling

    def disable_tiling(self):
        self.enable_tiling(False)

    # ---- chunk schedule of chunk_decode (:347-374)
    @staticmethod
    def chunk_sizes(num_frames, window_size, temporal_chunk):
        if not temporal_chunk:
            return [num_frames]
        init = min(window_size + 1, num_frames)
        sizes = [init]
        fid = init
        for _ in range((num_frames - init) // window_size):
            sizes.append(window_size)
            fid += window_size
        if fid < num_frames:
            sizes.append(num_frames - fid)
        return sizes

    def _program(self, th, tw, sizes, lane=0):
        key = (th, tw, sizes[0], max(sizes[1:] or [sizes[0]]), lane)
        p = self._programs.get(key)
        if p is None:
            # the (up to 4) tile geometries of a frame size and every chunk schedule alias one storage pool per lane:
            # activation memory = n_streams x the largest program instead of the sum over geometries
            p = _TileProgram(self, th, tw, key[2], key[3], pool=self._lane_pools.setdefault(lane, {}))
            self._programs[key] = p
        return p

    def _tile_out(self, T, th, tw):
        f = 2 ** sum(self.cfg["temporal_up_sample"])
        s = 2 ** sum(self.cfg["spatial_up_sample"])
        return torch.empty(1 + f * (T - 1), th * s, tw * s, 8, dtype=torch.bfloat16, device=self.dev)

    def _decode_tile(self, z, h0, w0, th, tw, sizes, affine, out=None, lane=0):
        """-> bf16 [T_out, 8 th, 8 tw, 8] (channels 0..2 valid)."""
        T = z.shape[1]
        n_t = sum(self.cfg["temporal_up_sample"])
        f = 2 ** n_t
        T_out = 1 + f * (T - 1)
        if out is None:
            out = self._tile_out(T, th, tw)
        prog = self._program(th, tw, sizes, lane)
        prog.reset()
        t0, fo = 0, 0
        for ci, nt in enumerate(sizes):
            fo += prog.run_chunk(z, t0, nt, h0, w0, ci == 0, out, fo, affine)
            t0 += nt
        assert fo == T_out, (fo, T_out)
        return out

    def _blend(self, a, b, blend, vertical, a_w=None, cp=8):
        """blend_v / blend_h (:397-407): b updated in place from the bottom rows / right columns of a."""
        lib = L.load()
        Tt, Hb, Wb, _ = b.shape
        check(lib.pf_blend_tiles(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_int(Tt),
                                 C.c_int(a.shape[1]), C.c_int(a.shape[2] if a_w is None else a_w), C.c_int(Hb), C.c_int(Wb),
                                 C.c_int(cp), C.c_int(blend), C.c_int(int(vertical)), stream()))

    @torch.no_grad()
    def decode_tiles(self, z, temporal_chunk, window_size, tile_sample_min_size, affine=(1.0, 0.0, 1.0, 0.0), comm=None):
        """z [1,C,T,h,w] fp32 -> (tiles grid, geometry) following decode/tiled_decode (:376-395, 468-519).

        comm (SPComm with world > 1): tile-parallel decode -- the tile COLUMNS are split into contiguous blocks over
        the ranks (the reference decodes on rank 0 only, pipeline.py:1223-1224).  Tiles are independent until the
        blend; blend_v is column-local, blend_h needs the right 64-px strip of the left neighbour AFTER its vertical
        blend, which crosses a rank boundary once per tile row: rank r receives that strip from r-1 and forwards its own
        (a 4-step pipeline of point-to-point messages).  Same arithmetic in the same per-element order as the
        sequential loop, so the result is bit-identical.  Returns this rank's column block (col0, rows-of-tiles)."""
        assert z.shape[0] == 1, "batch size 1"
        z = z[0].to(self.dev, torch.float32).contiguous()
        Cc, T, H, W = z.shape
        tl = int(tile_sample_min_size / self.downsample_scale)
        sizes = tuple(self.chunk_sizes(T, window_size * max(1, self.chunk_coalesce), temporal_chunk))
        if not (self.use_tiling and (W > tl or H > tl)):
            if comm is not None and comm.world > 1 and comm.rank != 0:
                return None, None                  # nothing to split: rank 0 decodes alone (reference behaviour)
            return [[self._decode_tile(z, 0, 0, H, W, sizes, affine)]], None
        overlap = int(tl * 0.75)
        blend = int(tile_sample_min_size * 0.25)
        limit = tile_sample_min_size - blend
        i_list = list(range(0, H, overlap))
        j_list = list(range(0, W, overlap))
        world = comm.world if comm is not None else 1
        rank = comm.rank if comm is not None else 0
        from .sp import even_split, starts_of
        ncols = even_split(len(j_list), world)
        c0 = starts_of(ncols)[rank]
        my_js = j_list[c0:c0 + ncols[rank]]
        self.tile_cols = (c0, ncols[rank], len(j_list))
        # the tiles are independent until the blend: decode them round-robin on a few HIP streams (each with its own
        # buffer set), so that the small latent-resolution layers of one tile -- 32 workgroups on 256 CUs -- overlap
        # with the large layers of another
        rows = [[self._tile_out(T, min(tl, H - i), min(tl, W - j)) for j in my_js] for i in i_list]
        ns = max(1, min(self.n_streams, len(i_list) * len(my_js)))
        if ns == 1:
            for a, i in enumerate(i_list):
                for b, j in enumerate(my_js):
                    self._decode_tile(z, i, j, min(tl, H - i), min(tl, W - j), sizes, affine, out=rows[a][b])
        else:
            if len(self._streams) < ns:
                self._streams += [torch.cuda.Stream(device=self.dev) for _ in range(ns - len(self._streams))]
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)
            k = 0
            for a, i in enumerate(i_list):
                for b, j in enumerate(my_js):
                    st = self._streams[k % ns]
                    if k < ns:
                        st.wait_event(ready)
                    with torch.cuda.stream(st):
                        self._decode_tile(z, i, j, min(tl, H - i), min(tl, W - j), sizes, affine, out=rows[a][b], lane=k % ns)
                    k += 1
            for st in self._streams[:ns]:
                cur.wait_stream(st)
        has_left = world > 1 and c0 > 0 and ncols[rank] > 0
        # the next rank that owns columns (ranks beyond the column count own none)
        nxt = rank + 1 if (world > 1 and rank + 1 < world and ncols[rank + 1] > 0 and ncols[rank] > 0) else None
        for i, row in enumerate(rows):
            for j, tile in enumerate(row):
                if i > 0:
                    self._blend(rows[i - 1][j], tile, blend, True)
                if nxt is not None and j == len(row) - 1:
                    comm.send(tile[:, :, tile.shape[2] - blend:, :], nxt)      # right strip, after the vertical blend
                if j > 0:
                    self._blend(row[j - 1], tile, blend, False)
                elif has_left:
                    strip = torch.empty(tile.shape[0], tile.shape[1], blend, 8, dtype=torch.bfloat16, device=self.dev)
                    comm.recv(strip, rank - 1)
                    self._blend(strip, tile, blend, False)
        return rows, limit

    @torch.no_grad()
    def decode_to_uint8(self, z, window_size=1, tile_sample_min_size=256, temporal_chunk=True, affine=(1.0, 0.0, 1.0, 0.0),
                        comm=None):
        """fused decode + (x*127.5+127.5).clamp.byte -> uint8 [T,H,W,3] on device (pipeline.py:1234-1240).
        With comm (world > 1) every rank decodes its block of tile columns and rank 0 assembles the frames
        (returns None elsewhere, like the reference's non-zero ranks)."""
        rows, limit = self.decode_tiles(z, temporal_chunk, window_size, tile_sample_min_size, affine, comm=comm)
        lib = L.load()
        if rows is None:
            return None
        if limit is None:
            t = rows[0][0]
            T_out = t.shape[0]
            H, W = t.shape[1], t.shape[2]
            out = torch.empty(T_out, H, W, 3, dtype=torch.uint8, device=self.dev)
            check(lib.pf_to_uint8(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int(T_out), C.c_int(H), C.c_int(W),
                                  C.c_int(8), C.c_int(H), C.c_int(W), C.c_int(H), C.c_int(W), C.c_int(0), C.c_int(0), stream()))
            return out
        s_up = self.downsample_scale
        Hs, Ws = z.shape[3] * s_up, z.shape[4] * s_up
        T_out = 1 + (2 ** sum(self.cfg["temporal_up_sample"])) * (z.shape[2] - 1)
        overlap_px = int(tile_sample_min_size * 0.75)
        c0, nc, ncols_total = self.tile_cols
        # pixel extent of this rank's column block in the final frame (each tile contributes `limit` columns, the
        # last one what is left)
        x_begin = c0 * overlap_px
        x_end = min((c0 + nc) * overlap_px, Ws) if (c0 + nc) < ncols_total else Ws
        if nc == 0:
            x_begin = x_end = 0
        Wloc = max(x_end - x_begin, 0)
        out = torch.empty(T_out, Hs, max(Wloc, 1), 3, dtype=torch.uint8, device=self.dev)
        if nc:
            hs = [min(r[0].shape[1], limit) for r in rows]
            ws = [min(t.shape[2], limit) for t in rows[0]]
            assert sum(hs) == Hs and sum(ws) == Wloc, (hs, ws, Hs, Wloc)
            y0 = 0
            for i, row in enumerate(rows):
                x0 = 0
                for j, t in enumerate(row):
                    check(lib.pf_to_uint8(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int(T_out), C.c_int(t.shape[1]),
                                          C.c_int(t.shape[2]), C.c_int(8), C.c_int(hs[i]), C.c_int(ws[j]), C.c_int(Hs), C.c_int(Wloc),
                                          C.c_int(y0), C.c_int(x0), stream()))
                    x0 += ws[j]
                y0 += hs[i]
        if comm is None or comm.world == 1:
            return out
        # ---- assemble on rank 0 (column blocks are strided in the final frames: receive, then place)
        from .sp import even_split, starts_of
        ncs = even_split(ncols_total, comm.world)
        c0s = starts_of(ncs)
        if comm.rank != 0:
            if nc:
                comm.send(out, 0)
            return None
        full = torch.empty(T_out, Hs, Ws, 3, dtype=torch.uint8, device=self.dev)
        full[:, :, x_begin:x_end] = out[:, :, :Wloc]
        for r in range(1, comm.world):
            if not ncs[r]:
                continue
            xb = c0s[r] * overlap_px
            xe = min((c0s[r] + ncs[r]) * overlap_px, Ws) if (c0s[r] + ncs[r]) < ncols_total else Ws
            blk = torch.empty(T_out, Hs, xe - xb, 3, dtype=torch.uint8, device=self.dev)
            comm.recv(blk, r)
            full[:, :, xb:xe] = blk
        return full

    @torch.no_grad()
    def decode_context_parallel(self, z, comm, affine=(1.0, 0.0, 1.0, 0.0), to_uint8=True):
        """Temporal context-parallel decode (config C5; the reference wires this up for training only:
        modeling_causal_vae.py:540-567, context_parallel_ops.py:14-114, CausalConv3d.context_parallel_forward
        modeling_causal_conv.py:95-114).  The T latent frames are split into contiguous, possibly uneven ranges over the
        ranks (the reference's _conv_split needs (T-1) % P == 0); every rank decodes its range UN-tiled in one pass, layer
        by layer in lockstep: before each 3-tap temporal conv the last two frames of the layer input travel to the next
        rank (one point-to-point message per conv and boundary) and fill that rank's two cache slots; rank 0 keeps the
        causal zeros and drops the first frame after each temporal upsample, the others keep it (:561-565).
        Equals the single-process un-tiled decode.  Returns uint8 frames [T_out,H,W,3] (or the bf16 image) on rank 0,
        None elsewhere."""
        from .sp import even_split, starts_of
        assert z.shape[0] == 1
        zc = z[0].to(self.dev, torch.float32).contiguous()
        Cc, T, H, W = zc.shape
        P, r = comm.world, comm.rank
        counts = even_split(T, P)
        assert min(counts) >= 2, "context parallelism needs >= 2 latent frames per rank"
        f0 = starts_of(counts)[r]
        nt = counts[r]
        key = ("cp", H, W, nt, r == 0)
        prog = self._programs.get(key)
        if prog is None:
            prog = _TileProgram(self, H, W, nt, nt)
            prog.halo = comm if P > 1 else None
            prog.transient = True              # one pass per rank: activations are handed back as soon as they are dead
            self._programs[key] = prog
        prog.reset()
        n_t = sum(self.cfg["temporal_up_sample"])
        f = 2 ** n_t
        s_up = 2 ** sum(self.cfg["spatial_up_sample"])
        t_out = f * nt - (f - 1 if r == 0 else 0)
        img = torch.empty(t_out, H * s_up, W * s_up, 8, dtype=torch.bfloat16, device=self.dev)
        got = prog.run_chunk(zc, f0, nt, 0, 0, r == 0, img, 0, affine)
        assert got == t_out, (got, t_out)
        lib = L.load()
        if to_uint8:
            loc = torch.empty(t_out, H * s_up, W * s_up, 3, dtype=torch.uint8, device=self.dev)
            check(lib.pf_to_uint8(C.c_void_p(img.data_ptr()), C.c_void_p(loc.data_ptr()), C.c_int(t_out), C.c_int(H * s_up),
                                  C.c_int(W * s_up), C.c_int(8), C.c_int(H * s_up), C.c_int(W * s_up), C.c_int(H * s_up),
                                  C.c_int(W * s_up), C.c_int(0), C.c_int(0), stream()))
        else:
            loc = img[..., :3].contiguous()
        if P == 1:
            return loc
        # frames are contiguous in time: rank 0 concatenates (conv_gather, context_parallel_ops.py:41-73)
        if r != 0:
            comm.send(loc, 0)
            return None
        parts = [loc]
        for p_ in range(1, P):
            tp = f * counts[p_]
            part = torch.empty((tp,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=self.dev)
            comm.recv(part, p_)
            parts.append(part)
        return torch.cat(parts, dim=0)

    @torch.no_grad()
    def encode(self, x, return_dict=True, is_init_image=True, temporal_chunk=False, window_size=16, tile_sample_min_size=256):
        """modeling_causal_vae.py:274-308 / tiled_encode :409-466: x [1,3,T,H,W] in [-1,1] -> latent_dist over
        [1,C,T',H/8,W/8] (T' = 1 + (T-1)/8).  T = 1 is what generate_i2v encodes (pyramid_dit_for_video_gen_pipeline.py:
        906-911); clips run in one causal pass (temporal_chunk=False, the default) or, with temporal_chunk=True, through
        the sliding window of chunk_encode (:310-341): window_size + 1 frames first, then window_size frames per call,
        the two previous input frames of every temporal conv kept in its cache slots -- bounded activation memory for
        long clips, same result as the single pass."""
        if not self.has_encoder:
            raise RuntimeError("this CausalVideoVAE was built without encoder weights")
        assert x.shape[0] == 1 and x.shape[1] == 3
        T = x.shape[2]
        chunks = [(0, T)]
        if temporal_chunk and T > window_size + 1:
            assert (T - 1) % self.downsample_scale == 0                                   # :314
            td = 2 ** sum(self.enc_cfg["temporal_down_sample"])
            assert window_size % td == 0, "window_size must be a multiple of the temporal downsample factor"
            chunks, fid = [(0, window_size + 1)], window_size + 1
            while fid < T:
                chunks.append((fid, min(window_size, T - fid)))
                fid += window_size
        chunked = len(chunks) > 1
        img = x[0].to(self.dev, torch.float32).contiguous()
        _, _, H, W = img.shape
        s_ = self.downsample_scale
        lat = self.enc_cfg["latent_channels"]
        ts = tile_sample_min_size
        lib = L.load()
        Tl = T
        for dn in self.enc_cfg["temporal_down_sample"]:
            if dn:
                Tl = (Tl - 1) // 2 + 1
        if T > 1 and self.convs_clip is None:      # full 3-tap encoder filters, packed once
            self.convs_clip = dict(self.convs)
            for name, (wt, bs) in self._enc_full.items():
                self.convs_clip[name] = ConvW(wt, bs, self.dev, 1)
        tiled = self.use_tiling and (W > ts or H > ts)
        if not tiled:
            tiles, i_list, j_list = None, [0], [0]
        else:
            overlap = int(ts * 0.75)
            i_list, j_list = list(range(0, H, overlap)), list(range(0, W, overlap))
        rows = []
        for i in i_list:
            row = []
            for j in j_list:
                ph, pw = (min(ts, H - i), min(ts, W - j)) if tiled else (H, W)
                assert ph % s_ == 0 and pw % s_ == 0
                Tprog = chunks[0][1]
                key = ("enc", ph // s_, pw // s_, Tprog, T > 1)
                prog = self._programs.get(key)
                if prog is None:
                    prog = _TileProgram(self, ph // s_, pw // s_, Tprog, Tprog, encoder=True)
                    if T > 1:
                        prog.cw = self.convs_clip
                    self._programs[key] = prog
                prog.reset()
                t = torch.empty(Tl, ph // s_, pw // s_, 64, dtype=torch.bfloat16, device=self.dev)
                got = 0
                for ci, (t0, nt) in enumerate(chunks):
                    got += prog.run_encoder(img, i, j, t, t0=t0, nt=nt, chunked=chunked, first=(ci == 0), out_frame0=got)
                assert got == Tl, (got, Tl)
                row.append(t)
            rows.append(row)
        h, w = H // s_, W // s_
        moments = torch.empty(2 * lat, Tl, h, w, dtype=torch.float32, device=self.dev)
        if not tiled:
            t = rows[0][0]
            check(lib.pf_nhwc_to_planar_f32(C.c_void_p(t.data_ptr()), C.c_void_p(moments.data_ptr()), C.c_int(Tl), C.c_int(h),
                                            C.c_int(w), C.c_int(64), C.c_int(2 * lat), C.c_int(h), C.c_int(w), C.c_int(h),
                                            C.c_int(w), C.c_int(0), C.c_int(0), stream()))
        else:
            tl = ts // s_
            blend = int(tl * 0.25)
            limit = tl - blend
            y0 = 0
            for i, row in enumerate(rows):
                x0 = 0
                for j, t in enumerate(row):
                    if i > 0:
                        self._blend(rows[i - 1][j], t, blend, True, cp=64)
                    if j > 0:
                        self._blend(row[j - 1], t, blend, False, cp=64)
                    ch_, cw_ = min(t.shape[1], limit), min(t.shape[2], limit)
                    check(lib.pf_nhwc_to_planar_f32(C.c_void_p(t.data_ptr()), C.c_void_p(moments.data_ptr()), C.c_int(Tl),
                                                    C.c_int(t.shape[1]), C.c_int(t.shape[2]), C.c_int(64), C.c_int(2 * lat),
                                                    C.c_int(ch_), C.c_int(cw_), C.c_int(h), C.c_int(w), C.c_int(y0), C.c_int(x0),
                                                    stream()))
                    x0 += cw_
                y0 += min(row[0].shape[1], limit)
        post = DiagonalGaussianDistribution(moments[None])
        if not return_dict:
            return (post,)
        return EncoderOutput(post)

    @torch.no_grad()
    def decode(self, z, is_init_image=True, temporal_chunk=False, return_dict=True, window_size=2, tile_sample_min_size=256):
        assert is_init_image, "inference decode always starts at the first frame"
        rows, limit = self.decode_tiles(z, temporal_chunk, window_size, tile_sample_min_size)
        if limit is None:
            img = rows[0][0][..., :3]
        else:
            img = torch.cat([torch.cat([t[:, :limit, :limit, :3] for t in row], dim=2) for row in rows], dim=1)
        sample = img.permute(3, 0, 1, 2)[None].contiguous()          # [1,3,T,H,W] (layout plumbing only)
        if not return_dict:
            return (sample,)
        return DecoderOutput(sample)
