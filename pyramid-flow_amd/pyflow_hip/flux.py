"""miniFLUX DiT forward on MI355X: weight packing + kernel sequencing.

Mirrors ``PyramidFluxTransformer.forward`` (pyramid_dit/flux_modules/modeling_pyramid_flux.py:392-542)
for one pyramid stage per call (the only form inference uses, pipeline.py:760).  The token layout is
ONE buffer ``hidden[B][Lt + L_img][d]`` (text rows first) for all double and single blocks -- the
reference's `cat([text, image])` at :480 is the identity here, so there is no re-partition step.
Fused projections: per double block K|V|Q share one GEMM per stream; per single block K|V|Q|MLP
(N = 7d) with the GELU fused on the MLP columns; attention writes O over the Q columns so that the
out-projection reads `[O | mlp]` (= the reference's `cat([attn, mlp])`, flux_block.py:936) in place.
"""
import contextlib
import ctypes as C
import time

import torch

from . import ops
from .ops import GEMM_GATE_RES, GEMM_OUT_F32
from .plan import SequencePlan
from .refapi import DeviceModuleAPI, load_diffusers_dir


def _bf16(t, device):
    return t.detach().to(device=device, dtype=torch.bfloat16).contiguous()


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _pad_k(w, mult=64):
    k = w.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w
    out = torch.zeros(w.shape[0], kp, dtype=w.dtype)
    out[:, :k] = w
    return out


def _pad_n(w, b, mult=128):
    n = w.shape[0]
    np_ = (n + mult - 1) // mult * mult
    if np_ == n:
        return w, b
    wo = torch.zeros(np_, w.shape[1], dtype=w.dtype)
    wo[:n] = w
    bo = torch.zeros(np_, dtype=b.dtype)
    bo[:n] = b
    return wo, bo


class FluxWeights:
    """Packs a reference-keyed state dict (SURVEY 8b key layout) into fused bf16 matrices."""

    def __init__(self, sd, cfg, device, head_major=False, blocks_only=False):
        """head_major: order the fused K|V|Q output columns per head ([h][k|v|q][64]) instead of per type -- the
        layout of the sequence-parallel exchange buffers (a rank's heads are then one contiguous column block).
        blocks_only: `sd` holds transformer blocks only (the block-level operators of pyflow_hip/blocks.py): no
        embedders, conditioning MLPs or output head are packed."""
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.cfg = cfg
        self.head_major = head_major
        H, hd = cfg["num_attention_heads"], cfg["attention_head_dim"]
        assert hd == 64, "attention kernel is specialised for head_dim 64"
        d = H * hd
        assert d % 128 == 0, "model width must be a multiple of 128"
        self.d, self.H = d, H
        # variant: miniFLUX ("pyramid_flux": x_embedder Linear, 3-axis RoPE, double + single blocks) or the SD3-style
        # MMDiT ("pyramid_mmdit", mmdit_modules/modeling_pyramid_mmdit.py:60-149: PatchEmbed3D conv + sincos table,
        # temporal RoPE over the whole head, joint blocks only, last block context_pre_only, QK-norm eps 1e-5)
        self.mmdit = "pos_embed.proj.weight" in sd or bool(cfg.get("_mmdit_blocks"))
        nd, ns = cfg["num_layers"], (0 if self.mmdit else cfg["num_single_layers"])
        self.qk_eps = 1e-5 if self.mmdit else 1e-6
        self.rope_axes = [hd] if self.mmdit else list(cfg["axes_dims_rope"])
        dev = device
        if self.mmdit and blocks_only:
            self.pos_table = None
            added_q, added_k = "norm_add_q", "norm_add_k"
        elif self.mmdit:
            # Conv2d(k=2, s=2) on a frame == Linear over the (c, p1, p2) patch; tokens arrive as (p1, p2, c)
            wc = sd["pos_embed.proj.weight"]
            sd["x_embedder.weight"] = wc.permute(0, 2, 3, 1).reshape(wc.shape[0], -1).contiguous()
            sd["x_embedder.bias"] = sd["pos_embed.proj.bias"]
            self.pos_table = sd["pos_embed.pos_embed"][0].clone()           # [max*max, d] fp32 (host)
            self.pos_max = int(round(self.pos_table.shape[0] ** 0.5))
            added_q, added_k = "norm_add_q", "norm_add_k"
        else:
            self.pos_table = None
            added_q, added_k = "norm_added_q", "norm_added_k"

        def W(name):
            return sd[name + ".weight"]

        def Bv(name):
            return sd[name + ".bias"]

        if head_major:
            t_, h_, e_ = torch.meshgrid(torch.arange(3), torch.arange(H), torch.arange(hd), indexing="ij")
            perm = torch.empty(3 * d, dtype=torch.long)
            perm[(h_ * 3 * hd + t_ * hd + e_).reshape(-1)] = (t_ * d + h_ * hd + e_).reshape(-1)
        else:
            perm = None

        def kvq(wk, wv, wq, extra=None):
            """fused [K; V; Q (; extra)] weight + bias, columns optionally head-major"""
            w_ = torch.cat([W(wk), W(wv), W(wq)])
            b_ = torch.cat([Bv(wk), Bv(wv), Bv(wq)])
            if perm is not None:
                w_, b_ = w_[perm], b_[perm]
            if extra is not None:
                w_ = torch.cat([w_, W(extra)])
                b_ = torch.cat([b_, Bv(extra)])
            return _bf16(w_, dev), _f32(b_, dev)

        if not blocks_only:
            self._pack_embedders(W, Bv, dev)
        self._pack_blocks(sd, W, Bv, kvq, nd, ns, d, dev, added_q, added_k, blocks_only)

    def _pack_embedders(self, W, Bv, dev):
        # conditioning MLPs (gemv, bf16 weights / fp32 bias)
        self.t1 = (_bf16(W("time_text_embed.timestep_embedder.linear_1"), dev), _f32(Bv("time_text_embed.timestep_embedder.linear_1"), dev))
        self.t2 = (_bf16(W("time_text_embed.timestep_embedder.linear_2"), dev), _f32(Bv("time_text_embed.timestep_embedder.linear_2"), dev))
        self.p1 = (_bf16(_pad_k(W("time_text_embed.text_embedder.linear_1"), 8), dev), _f32(Bv("time_text_embed.text_embedder.linear_1"), dev))
        self.p2 = (_bf16(W("time_text_embed.text_embedder.linear_2"), dev), _f32(Bv("time_text_embed.text_embedder.linear_2"), dev))
        self.pooled_k = self.p1[0].shape[1]
        # embedders
        self.ctx_w = _bf16(_pad_k(W("context_embedder")), dev)
        self.ctx_b = _f32(Bv("context_embedder"), dev)
        self.ctx_k = self.ctx_w.shape[1]
        self.x_w = _bf16(_pad_k(W("x_embedder")), dev)
        self.x_b = _f32(Bv("x_embedder"), dev)
        self.in_ch = W("x_embedder").shape[1]
        assert self.x_w.shape[1] == self.in_ch, "in_channels must be a multiple of 64"

    def _pack_blocks(self, sd, W, Bv, kvq, nd, ns, d, dev, added_q, added_k, blocks_only):
        # all AdaLN linears as one matrix (one gemv per forward)
        mods, mod_b = [], []
        self.dbl, self.sgl = [], []
        off = 0
        for i in range(nd):
            p = f"transformer_blocks.{i}."
            mods += [W(p + "norm1.linear"), W(p + "norm1_context.linear")]
            mod_b += [Bv(p + "norm1.linear"), Bv(p + "norm1_context.linear")]
            # context_pre_only (last MMDiT block, mmdit_block.py:585-599): the text stream only feeds the attention:
            # AdaLayerNormContinuous (2d: scale, shift) instead of AdaLayerNormZero, no to_add_out, no ff_context
            pre_only = (p + "attn.to_add_out.weight") not in sd
            blk = dict(mod=off, pre_only=pre_only)
            off += (8 if pre_only else 12) * d
            blk["kvq_img"] = kvq(p + "attn.to_k", p + "attn.to_v", p + "attn.to_q")
            blk["kvq_txt"] = kvq(p + "attn.add_k_proj", p + "attn.add_v_proj", p + "attn.add_q_proj")
            blk["o_img"] = (_bf16(W(p + "attn.to_out.0"), dev), _f32(Bv(p + "attn.to_out.0"), dev))
            blk["ff1_img"] = (_bf16(W(p + "ff.net.0.proj"), dev), _f32(Bv(p + "ff.net.0.proj"), dev))
            blk["ff2_img"] = (_bf16(W(p + "ff.net.2"), dev), _f32(Bv(p + "ff.net.2"), dev))
            if not pre_only:
                blk["o_txt"] = (_bf16(W(p + "attn.to_add_out"), dev), _f32(Bv(p + "attn.to_add_out"), dev))
                blk["ff1_txt"] = (_bf16(W(p + "ff_context.net.0.proj"), dev), _f32(Bv(p + "ff_context.net.0.proj"), dev))
                blk["ff2_txt"] = (_bf16(W(p + "ff_context.net.2"), dev), _f32(Bv(p + "ff_context.net.2"), dev))
            for nm, key in (("norm_q", "norm_q"), ("norm_k", "norm_k"), ("norm_added_q", added_q), ("norm_added_k", added_k)):
                blk[nm] = _f32(sd[p + f"attn.{key}.weight"], dev)
            self.dbl.append(blk)
        for j in range(ns):
            p = f"single_transformer_blocks.{j}."
            mods.append(W(p + "norm.linear"))
            mod_b.append(Bv(p + "norm.linear"))
            blk = dict(mod=off)
            off += 3 * d
            blk["kvqm"] = kvq(p + "attn.to_k", p + "attn.to_v", p + "attn.to_q", extra=p + "proj_mlp")
            blk["out"] = (_bf16(W(p + "proj_out"), dev), _f32(Bv(p + "proj_out"), dev))
            blk["norm_q"] = _f32(sd[p + "attn.norm_q.weight"], dev)
            blk["norm_k"] = _f32(sd[p + "attn.norm_k.weight"], dev)
            self.sgl.append(blk)
        if not blocks_only:
            mods.append(W("norm_out.linear"))
            mod_b.append(Bv("norm_out.linear"))
            self.mod_final = off
            off += 2 * d
            pw, pb = _pad_n(W("proj_out"), Bv("proj_out"))
            self.proj_w, self.proj_b = _bf16(pw, dev), _f32(pb, dev)
            self.out_cols = W("proj_out").shape[0]
        self.n_mod = off
        self.mod_w = _bf16(torch.cat(mods), dev)
        self.mod_b = _f32(torch.cat(mod_b), dev)
        self.n_params = sum(v.numel() for v in sd.values())


class FluxEngine(DeviceModuleAPI):
    """`PyramidFluxTransformer` / `PyramidDiffusionMMDiT` of the reference (modeling_pyramid_flux.py:60-542,
    modeling_pyramid_mmdit.py:60-497), inference half: same constructor source (`from_pretrained` on a diffusers
    directory), same `forward(sample, encoder_hidden_states, encoder_attention_mask, pooled_projections,
    timestep_ratio)` call and `.config`; the variant is read off the state dict."""
    HEAD_MAJOR = False

    def __init__(self, state_dict, cfg, device="cuda"):
        import types
        self.dev = torch.device(device)
        self.w = FluxWeights(state_dict, cfg, self.dev, head_major=self.HEAD_MAJOR)
        self.cfg = cfg
        self.config = types.SimpleNamespace(**dict(cfg))          # `.config.in_channels` (pipeline.py:1098)
        self._ws = {}
        self._ctx = None
        self._mod_cache = None
        self.overlap_text = True        # text stream of the double blocks on a side HIP stream
        self.skip_dead_rows = True      # last block: Q / MLP / attention / proj_out only for the current frame's rows
        self.fuse_qk = True             # QK-RMSNorm + RoPE inside the K|V|Q projections (pf_gemm_desc.qk_*); False = separate pass
        self.group_text = True          # double blocks: the text stream's GEMMs ride in the image stream's persistent launches
                                        # (pf_gemm_desc.A2 ..., round 5) wherever those run the persistent kernel; False = side stream
        self.v_rowmajor = True          # attention reads V token-major (hardware transpose read); False = pf_v_transpose + V^T image
        self._side = None
        # how the ~300 launches of the blocks + head of one forward reach the device (cmdlist.py):
        #   "eager": one ctypes call per launch;  "list": recorded once per plan, re-issued from C by one call;
        #   "graph": the same list captured into a hipGraph after its first replay (one hipGraphLaunch per forward)
        self.launch_mode = "graph"
        # host-side bookkeeping of the launch lists: how many were recorded / captured and what that cost (bench.py reports it)
        self.list_stats = dict(records=0, record_s=0.0, instantiates=0, instantiate_s=0.0, replays=0)
        self._ws_gen = 0                # bumped when a workspace buffer is re-allocated (recorded pointers go stale)
        self._listed_plans = None       # weak set of the plans that hold a launch list of this engine

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    # ---- workspace (grow-only) ----
    def _buf(self, name, numel, dtype):
        t = self._ws.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.zeros(numel, dtype=dtype, device=self.dev) if name == "vT" \
                else torch.empty(numel, dtype=dtype, device=self.dev)
            self._ws[name] = t
            self._ws_gen += 1
            # every recorded list / captured hipGraph of this engine now points at freed storage: drop them NOW, not when
            # their plan happens to run again (a stale graph exec would otherwise stay attached to a cached plan)
            for p_ in list(getattr(self, "_listed_plans", None) or ()):
                p_.__dict__.pop("_launch_list", None)
        return t

    def reserve(self, B, L, L_img, n_cur):
        """grow the workspace to what a sequence of L rows (L_img image rows, n_cur of the current frame) needs, NOW: every
        later forward of the job then finds its buffers in place and no recorded launch list goes stale mid-video"""
        w, d = self.w, self.w.d
        self._buf("hidden", B * L * d, torch.bfloat16)
        self._buf("xn", B * L * d, torch.bfloat16)
        self._buf("big", B * L * 7 * d, torch.bfloat16)
        if not self.v_rowmajor:
            self._buf("vT", B * w.H * 64 * ((L + 63) // 64 * 64 + 64), torch.bfloat16)
        self._buf("tok", B * L_img * w.in_ch, torch.bfloat16)
        self._buf("vtok", B * n_cur * w.proj_w.shape[0], torch.float32)
        self._buf("mod_fixed", B * w.n_mod, torch.float32)
        if w.dbl:
            self._buf("splitk_txt", 8 << 20, torch.float32)
        self._buf("gemm_tail", 16 << 20, torch.float32)

    def make_plan(self, clip_shapes, enc_mask, cfg_pair=False):
        """cfg_pair: the two rows of `enc_mask` are the [negative | positive] guidance pair of ONE sample (pipeline.py:747);
        only the guidance-parallel engine (flux_cfg.py) treats that differently from a genuine batch of 2"""
        plan = SequencePlan(clip_shapes, enc_mask, self.w.rope_axes, self.dev)
        if self.w.mmdit:
            plan.pos = self._pos_rows(clip_shapes)
        return plan

    def _pos_rows(self, clip_shapes):
        """additive 2-D sincos position rows [L_img, d] bf16 of PatchEmbed3D (mmdit_modules/modeling_embedding.py:
        269-308, 326-352 with interp_condition_pos): the table window of the CURRENT clip's token grid, bilinearly
        resized to each lower-resolution history clip's grid, repeated over the clip's frames.  Host arithmetic once
        per (unit, stage)."""
        import torch.nn.functional as F
        w = self.w
        ms = w.pos_max
        oh, ow = clip_shapes[-1][1] // 2, clip_shapes[-1][2] // 2
        top, left = (ms - oh) // 2, (ms - ow) // 2
        win = w.pos_table.reshape(1, ms, ms, -1)[:, top:top + oh, left:left + ow, :]
        rows = []
        for (t, h, wd) in clip_shapes:
            h2, w2 = h // 2, wd // 2
            pe = win
            if (h2, w2) != (oh, ow):
                pe = F.interpolate(win.permute(0, 3, 1, 2), size=(h2, w2), mode="bilinear").permute(0, 2, 3, 1)
            rows.append(pe.reshape(1, h2 * w2, -1).expand(t, -1, -1).reshape(t * h2 * w2, -1))
        return torch.cat(rows, 0).to(self.dev, torch.bfloat16).contiguous()

    def encode_context(self, enc, cfg_pair=False):
        """context_embedder (flux:401): enc [B, Lt, C] -> cached bf16 [B, Lt, d].  cfg_pair: see make_plan."""
        w = self.w
        B, Lt, Cc = enc.shape
        x = torch.zeros(B * Lt, w.ctx_k, dtype=torch.bfloat16, device=self.dev)
        x[:, :Cc] = enc.reshape(B * Lt, Cc).to(self.dev, torch.bfloat16)
        out = torch.empty(B, Lt, w.d, dtype=torch.bfloat16, device=self.dev)
        ops.gemm(x, w.ctx_w, out, B * Lt, w.d, w.ctx_k, w.ctx_k, w.ctx_k, w.d, bias=w.ctx_b)
        self._ctx = out
        self._ctx_keep = x
        self._mod_cache = {}          # conditioning depends on (timestep, pooled): new prompt -> new cache
        return out

    def conditioning(self, timesteps, pooled):
        """time_text_embed (modeling_embedding.py:185-200) + every block's AdaLN linear -> mod [B, n_mod] fp32."""
        w = self.w
        B = len(timesteps)
        d = w.d
        # the modulation vectors are a function of (timestep, pooled prompt) only: every later unit repeats the
        # (stage, step) timesteps of unit 1, so 870 of the 960 forwards of a video hit this cache (1.07 GB of AdaLN
        # weights not re-read per forward).  Keyed on the timestep values and the CONTENT of the pooled tensor: the
        # bytes are fetched once per distinct tensor object / version (the tensor is kept referenced meanwhile, so its
        # address cannot be recycled for another prompt), not per forward.
        last = getattr(self, "_pooled_seen", None)
        if last is None or last[0] is not pooled or last[1] != pooled._version:
            last = (pooled, pooled._version, pooled.detach().to("cpu", torch.float32).numpy().tobytes())
            self._pooled_seen = last
        key = (tuple(float(t) for t in timesteps), last[2])
        cache = getattr(self, "_mod_cache", None)
        if cache is not None and key in cache:
            return cache[key], None
        tproj = self._buf("tproj", B * 256, torch.float32).view(B, 256)
        ops.timestep_embed(tproj, timesteps, 256)
        h1 = self._buf("h1", B * d, torch.float32)
        temb = self._buf("temb", B * d, torch.float32)
        ops.gemv(w.t1[0], w.t1[1], tproj, h1, d, 256, B)
        ops.gemv(w.t2[0], w.t2[1], h1, temb, d, d, B, silu_in=True)
        pk = w.pooled_k
        pp = torch.zeros(B, pk, dtype=torch.float32, device=self.dev)
        pp[:, :pooled.shape[1]] = pooled.to(self.dev, torch.float32)
        ops.gemv(w.p1[0], w.p1[1], pp, h1, d, pk, B)
        ops.gemv(w.p2[0], w.p2[1], h1, temb, d, d, B, silu_in=True, accumulate=True)
        mod = self._buf("mod", B * w.n_mod, torch.float32)
        ops.gemv(w.mod_w, w.mod_b, temb, mod, w.n_mod, d, B, silu_in=True)
        if cache is not None:
            if len(cache) >= 128:             # one video uses <= 90 distinct (stage, step) timesteps: bound the cache
                cache.pop(next(iter(cache)))
            mod = mod[:B * w.n_mod].clone()
            cache[key] = mod
        return mod, temb

    def forward_tokens(self, plan, clips, timesteps, pooled, ctx=None, shared_clips=False, debug=None):
        """clips: list of device tensors [B,C,t,h,w] (or [1,C,t,h,w] with shared_clips=True: the CFG
        duplicate of pipeline.py:747).  Returns v tokens fp32 [B, n_cur, 128] (first 4C columns valid)."""
        ctx = ctx if ctx is not None else self._ctx
        mod, _ = self.conditioning(timesteps, pooled)
        self._embed_tokens(plan, clips, ctx, shared_clips, debug)
        if self.launch_mode == "eager" or debug is not None or ops.PROFILER.enabled:
            self._run_blocks(plan, mod, debug=debug)
            return self._head(plan, mod)
        return self._run_launch_list(plan, mod)

    def _run_launch_list(self, plan, mod):
        """blocks + head through the plan's launch list: every step of a (unit, stage) runs the same kernels on the same
        buffers -- only the CONTENT of the modulation vector (copied into a fixed buffer first) and of `hidden` differs."""
        from .cmdlist import CommandList, recording
        w = self.w
        n_mod = plan.B * w.n_mod
        for _ in range(2):
            ms = self._buf("mod_fixed", n_mod, torch.float32)
            # what a recorded sequence depends on besides the plan: the workspace pointers (generation), the stream
            # structure, the row restriction of the last block, and the GEMM dispatch policy / split-K state in force
            # when the descriptors were recorded (pf_gemm_set_policy picks kernels at record time)
            key = (id(self), self._ws_gen, self.overlap_text, self.skip_dead_rows, self.launch_mode != "list",
                   ops.POLICY_GEN, self.fuse_qk, self.v_rowmajor, self.group_text)
            ent = getattr(plan, "_launch_list", None)
            if ent is not None and ent[0] == key:
                break
            cl = CommandList()
            t_rec = time.perf_counter()
            with recording(cl):
                self._run_blocks(plan, ms)
                out = self._head(plan, ms)
            self.list_stats["records"] += 1
            self.list_stats["record_s"] += time.perf_counter() - t_rec
            if key[1] != self._ws_gen:        # a workspace buffer was (re)allocated while recording: pointers of the
                continue                      # earlier entries may be stale -> record again, now without allocations
            plan._launch_list = ent = (key, cl, out)
            if getattr(self, "_listed_plans", None) is None:
                import weakref
                self._listed_plans = weakref.WeakSet()
            self._listed_plans.add(plan)
            break
        else:
            raise RuntimeError("launch list: the workspace did not settle")
        _, cl, out = ent
        ms = self._ws["mod_fixed"]
        ms[:n_mod].copy_(mod.reshape(-1)[:n_mod])
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_text else main
        if self.launch_mode == "graph" and cl.runs >= 1 and not cl.is_graph:
            # after one plain replay (every lazy one-time set-up has happened eagerly).  Captured on two private
            # streams -- torch's default stream is the legacy stream, which cannot be captured; the graph is then
            # launched on the caller's stream like any kernel.
            if getattr(self, "_cap_streams", None) is None:
                self._cap_streams = (torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev))
            try:
                t_inst = time.perf_counter()
                cl.instantiate(*self._cap_streams)
                self.list_stats["instantiates"] += 1
                self.list_stats["instantiate_s"] += time.perf_counter() - t_inst
            except RuntimeError as e:         # no graph support for this sequence on this stack: keep replaying the list
                import warnings
                warnings.warn(f"launch list: hipGraph capture failed ({e}); falling back to list replay")
                self.launch_mode = "list"
        self.list_stats["replays"] += 1
        try:
            cl.run(main, side)
        except RuntimeError:
            if not cl.is_graph:
                raise
            import warnings                   # a graph launch failed: drop to list replay for the rest of the run
            warnings.warn("launch list: hipGraphLaunch failed; falling back to list replay")
            self.launch_mode = "list"
            del plan._launch_list
            return self._run_launch_list(plan, mod)
        return out

    def _groups_text(self, plan, tail, n_act, L_img, B, d):
        """does a double block with this row geometry take the GROUPED form (text rows' tiles appended to the image rows'
        persistent launches)?  True when every image GEMM of the block runs the persistent kernel as whole-round launches
        (>= 192 tiles: what runs it without K-split scratch and with the QK epilogue).  Asked of the library once per
        (plan, geometry, dispatch policy), not per block and forward."""
        memo = plan.__dict__.setdefault("_group_memo", {})
        key = (id(self), ops.POLICY_GEN, bool(tail), n_act)
        if key not in memo:
            lib = ops.L.load()
            memo[key] = all(
                lib.pf_gemm_which(C.c_int(M_), C.c_int(B), C.c_int(N_), C.c_int(K_)) == 8 and
                -(-M_ // 256) * B * -(-N_ // 256) >= 192
                for M_, N_, K_ in ((L_img, (2 if tail else 3) * d, d), (n_act, d, d), (n_act, 4 * d, d), (n_act, d, 4 * d)))
        return memo[key]

    def _geometry(self, plan):
        w = self.w
        d, H = w.d, w.H
        B, Lt, L, L_img, Lp = plan.B, plan.Lt, plan.L, plan.L_img, plan.Lp
        hidden = self._buf("hidden", B * L * d, torch.bfloat16)
        xn = self._buf("xn", B * L * d, torch.bfloat16)
        big = self._buf("big", B * L * 7 * d, torch.bfloat16)
        vT = None if self.v_rowmajor else self._buf("vT", B * H * 64 * Lp, torch.bfloat16)      # V^T image of the older V path only
        return w, d, H, B, Lt, L, L_img, Lp, hidden, xn, big, vT

    def _embed_tokens(self, plan, clips, ctx, shared_clips=False, debug=None):
        """text rows <- cached context, image rows <- x_embedder(patchify) (flux:284-290, 401)"""
        w, d, H, B, Lt, L, L_img, Lp, hidden, xn, big, vT = self._geometry(plan)
        tok = self._buf("tok", B * L_img * w.in_ch, torch.bfloat16)
        Ld = L * d
        # ---- embed: text rows <- cached context, image rows <- x_embedder(patchify) ----
        ops.copy_rows(ctx, hidden, Lt, d, d, d, Lt * d, Ld, B)
        row = 0
        for cl, n in zip(clips, plan.clip_tokens):
            Cc, t, h, wd = cl.shape[1:]
            if shared_clips:
                ops.patchify(cl[0], tok, row * w.in_ch, Cc, t, h, wd, w.in_ch, L_img * w.in_ch, B)
            else:
                for b in range(B):
                    ops.patchify(cl[b], tok, (b * L_img + row) * w.in_ch, Cc, t, h, wd, w.in_ch, 0, 1)
            row += n
        if w.mmdit:       # + sincos position rows (same rows for every batch entry: strideR = 0)
            ops.gemm(tok, w.x_w, hidden, L_img, d, w.in_ch, w.in_ch, w.in_ch, d, bias=w.x_b, batch=B,
                     strideA=L_img * w.in_ch, strideC=Ld, c_off=Lt * d, res=plan.pos, ldr=d, strideR=0,
                     flags=GEMM_GATE_RES)
        else:
            ops.gemm(tok, w.x_w, hidden, L_img, d, w.in_ch, w.in_ch, w.in_ch, d, bias=w.x_b, batch=B,
                     strideA=L_img * w.in_ch, strideC=Ld, c_off=Lt * d)
        if debug is not None:
            debug["hidden0"] = hidden[:B * L * d].view(B, L, d).clone()


    def _run_blocks(self, plan, mod, dbl=None, sgl=None, last_block_tail=True, debug=None):
        """the double-stream and single-stream blocks over the resident `hidden` buffer (flux:447-520).  `dbl` / `sgl`:
        the packed blocks to run (default: all); last_block_tail=False computes every row of the last block too (the
        block-level operators return all rows)."""
        w, d, H, B, Lt, L, L_img, Lp, hidden, xn, big, vT = self._geometry(plan)
        dbl = w.dbl if dbl is None else dbl
        sgl = w.sgl if sgl is None else sgl
        skip_dead = self.skip_dead_rows and last_block_tail
        nm = w.n_mod
        Ld, L3, L4, L7 = L * d, L * 3 * d, L * 4 * d, L * 7 * d
        mlp_base = B * L3          # mlp region of `big` for the double blocks
        scale = 64 ** -0.5
        qs = scale * ops.LOG2E     # folded into q by qk_norm_rope; attention then works in base-2 exponents

        def ln(rows, x_off, sh, sc):
            ops.ln_modulate(hidden, xn, (mod, sh), (mod, sc), d, B, rows, Ld, Ld, d, d, nm, x_off=x_off, y_off=x_off)

        # The text stream of a double block (rows [0, Lt): 6 small GEMMs, 2 workgroup waves on 256 CUs) is independent
        # of the image stream between the two joins around the attention: it runs on a side HIP stream and fills CUs
        # the image GEMMs' tails leave idle.  Rows / buffer regions of the two streams are disjoint.
        # scratch of the text stream's GEMMs (128 rows per prompt: their K range is split, pf_gemm_desc.workspace); the
        # image stream's skinny launches (small early units) run on the other stream and get their own
        ws_txt = self._buf("splitk_txt", 8 << 20, torch.float32) if dbl else None
        # scratch of the compute stream's GEMMs: 64 MiB = one 256-KiB slot per workgroup of the persistent kernel, which
        # splits the tiles that do not fill its last round along K (pf_gemm_desc.workspace)
        ws_img = self._buf("gemm_tail", 16 << 20, torch.float32)
        lib = ops.L.load()
        main = torch.cuda.current_stream()
        side = self._side_stream() if (self.overlap_text and dbl) else None
        rec = ops.RECORDER                # recording a launch list: stream switches / joins become list entries

        def join(frm, to):
            """stream slot `to` (0 = main, 1 = side) waits for the work queued so far on slot `frm`"""
            if side is None:
                return
            if rec is not None:
                rec.join(frm, to)
            else:
                (side if to else main).wait_stream(main if to else side)

        def on_side():
            if side is None:
                return contextlib.nullcontext()
            return rec.on_slot(1) if rec is not None else torch.cuda.stream(side)

        join(0, 1)

        n_cur = plan.n_cur
        prev_grouped = None
        for blk in dbl:
            mb = blk["mod"]
            pre_only = blk["pre_only"]
            # last MMDiT block (context_pre_only, no single blocks follow): only the current frame's rows reach the
            # output -> K, V for every row, but Q / attention rows / to_out / MLP only for the last n_cur image rows
            tail = pre_only and skip_dead and not sgl and n_cur < L_img
            r0 = L - n_cur if tail else Lt                      # first image row that is computed in full
            n_act = L - r0
            # QK-RMSNorm + RoPE (flux_block.py:846-858): the image rows' K / Q blocks leave the projection's epilogue normed
            # and rotated (pf_gemm_desc.qk_*); the 128 text rows get the separate pass on their own stream
            fuse = self.fuse_qk
            qk_img = dict(rope=plan.rope, wq=blk["norm_q"], wk=blk["norm_k"], d=d, eps=w.qk_eps, q_scale=qs) if fuse else None
            # GROUPED form (round 5): every projection of the block runs ONCE for both streams -- the text rows' tiles are
            # appended to the image rows' persistent launch (same N, K, flavour; other weights, rows, gates, gains).  Taken
            # when all of the block's image GEMMs run the persistent kernel (pf_gemm_which == 8: from ~3 000 image rows on);
            # shorter sequences keep the two-stream form below.  No side stream, no joins, no K-split scratch.
            # (whole-round launches only: >= 192 tiles, what runs the persistent kernel without K-split scratch and with the QK epilogue)
            grouped = self.group_text and fuse and self._groups_text(plan, tail, n_act, L_img, B, d)
            # the two forms order the text rows differently (grouped: main stream only; two-stream: side stream between
            # joins), and the choice is per block (an MMDiT's last block in tail form has fewer tiles than the others): where
            # the form changes, the stream that takes over waits for the other one's work on `hidden` / `xn` / `big`
            if prev_grouped is not None and prev_grouped != grouped:
                join(0, 1) if prev_grouped else join(1, 0)
            prev_grouped = grouped
            if grouped:
                nq_t = blk["norm_added_q"] if blk["norm_added_q"] is not None else blk["norm_q"]
                nk_t = blk["norm_added_k"] if blk["norm_added_k"] is not None else blk["norm_k"]
                if pre_only:
                    ln(Lt, 0, mb + 7 * d, mb + 6 * d)
                else:
                    ln(Lt, 0, mb + 6 * d, mb + 7 * d)
                ln(L_img, Lt * d, mb + 0, mb + d)
                txt_kvq = dict(M=Lt, W=blk["kvq_txt"][0], bias=blk["kvq_txt"][1], wq=nq_t, wk=nk_t, row0=0)
                if tail:          # K | V of every image row (+ the text rows' K | V), then Q of the current frame's rows only
                    ops.gemm(xn, blk["kvq_img"][0], big, L_img, 2 * d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                             strideA=Ld, strideC=L3, a_off=Lt * d, c_off=Lt * 3 * d, qk=dict(qk_img, k_col0=0, row0=Lt),
                             second=txt_kvq)
                    ops.gemm(xn, blk["kvq_img"][0], big, n_act, d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                             strideA=Ld, strideC=L3, a_off=r0 * d, c_off=r0 * 3 * d + 2 * d, w_off=2 * d * d, bias_off=2 * d,
                             qk=dict(qk_img, q_col0=0, row0=r0))
                else:
                    ops.gemm(xn, blk["kvq_img"][0], big, L_img, 3 * d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                             strideA=Ld, strideC=L3, a_off=Lt * d, c_off=Lt * 3 * d,
                             qk=dict(qk_img, k_col0=0, q_col0=2 * d, row0=Lt), second=txt_kvq)
                if not self.v_rowmajor:
                    ops.v_transpose(big, vT, d, 3 * d, L3, B, H, L, Lp)
                ops.attention(big, big, vT, big, 2 * d, 0, 2 * d, 3 * d, L3, B, H, L, Lp, Lt, plan, scale, q_prescaled=True,
                              q_row_begin=r0 if tail else 0, v_off=d if self.v_rowmajor else None)
                txt = not pre_only
                ops.gemm(big, blk["o_img"][0], hidden, n_act, d, d, 3 * d, d, d, bias=blk["o_img"][1], res=hidden,
                         gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld, gate_stride=nm,
                         flags=GEMM_GATE_RES, a_off=r0 * 3 * d + 2 * d, c_off=r0 * d, r_off=r0 * d, tail_workspace=ws_img,
                         second=dict(M=Lt, W=blk["o_txt"][0], bias=blk["o_txt"][1], a_off=2 * d, gate_off=mb + 8 * d) if txt else None)
                ln(n_act, r0 * d, mb + 3 * d, mb + 4 * d)
                if txt:
                    ln(Lt, 0, mb + 9 * d, mb + 10 * d)
                ops.gemm(xn, blk["ff1_img"][0], big, n_act, 4 * d, d, d, d, 4 * d, bias=blk["ff1_img"][1], batch=B,
                         strideA=Ld, strideC=L4, gelu_from=0, a_off=r0 * d, c_off=mlp_base + r0 * 4 * d, tail_workspace=ws_img,
                         second=dict(M=Lt, W=blk["ff1_txt"][0], bias=blk["ff1_txt"][1], c_off=mlp_base) if txt else None)
                ops.gemm(big, blk["ff2_img"][0], hidden, n_act, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_img"][1],
                         res=hidden, gate=mod, gate_off=mb + 5 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                         gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base + r0 * 4 * d, c_off=r0 * d, r_off=r0 * d,
                         tail_workspace=ws_img,
                         second=dict(M=Lt, W=blk["ff2_txt"][0], bias=blk["ff2_txt"][1], a_off=mlp_base, gate_off=mb + 11 * d) if txt else None)
                if debug is not None and ("hidden_d0" not in debug or "blocks" in debug):
                    snap = hidden[:B * L * d].view(B, L, d).clone()
                    debug.setdefault("hidden_d0", snap)
                    if "blocks" in debug:
                        debug["blocks"].append(snap)
                continue
            with on_side():
                if pre_only:          # AdaLayerNormContinuous: (scale, shift) = chunks 0, 1 of the 2d modulation
                    ln(Lt, 0, mb + 7 * d, mb + 6 * d)
                else:
                    ln(Lt, 0, mb + 6 * d, mb + 7 * d)
                ops.gemm(xn, blk["kvq_txt"][0], big, Lt, (2 if tail else 3) * d, d, d, d, 3 * d, bias=blk["kvq_txt"][1],
                         batch=B, strideA=Ld, strideC=L3, workspace=ws_txt)
                if fuse:              # rows [0, Lt) with the text stream's gains (norm_added_q / k), K only in the tail form
                    ops.qk_norm_rope(big, 3 * d, L3, -1 if tail else 2 * d, 0, blk["norm_added_q"] if blk["norm_added_q"] is not None else blk["norm_q"],
                                     blk["norm_added_k"] if blk["norm_added_k"] is not None else blk["norm_k"], None, None,
                                     plan.rope, B, Lt, 0, H, q_scale=qs, eps=w.qk_eps)
            ln(L_img, Lt * d, mb + 0, mb + d)
            if tail:
                ops.gemm(xn, blk["kvq_img"][0], big, L_img, 2 * d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                         strideA=Ld, strideC=L3, a_off=Lt * d, c_off=Lt * 3 * d, tail_workspace=None if fuse else ws_img,
                         qk=dict(qk_img, k_col0=0, row0=Lt) if fuse else None)
                ops.gemm(xn, blk["kvq_img"][0], big, n_act, d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                         strideA=Ld, strideC=L3, a_off=r0 * d, c_off=r0 * 3 * d + 2 * d, w_off=2 * d * d, bias_off=2 * d,
                         tail_workspace=None if fuse else ws_img, qk=dict(qk_img, q_col0=0, row0=r0) if fuse else None)
            else:
                ops.gemm(xn, blk["kvq_img"][0], big, L_img, 3 * d, d, d, d, 3 * d, bias=blk["kvq_img"][1], batch=B,
                         strideA=Ld, strideC=L3, a_off=Lt * d, c_off=Lt * 3 * d, tail_workspace=None if fuse else ws_img,
                         qk=dict(qk_img, k_col0=0, q_col0=2 * d, row0=Lt) if fuse else None)
            join(1, 0)
            if not fuse:
                ops.qk_norm_rope(big, 3 * d, L3, 2 * d, 0, blk["norm_q"], blk["norm_k"], blk["norm_added_q"],
                                 blk["norm_added_k"], plan.rope, B, L, Lt, H, q_scale=qs, eps=w.qk_eps)
            if not self.v_rowmajor:
                ops.v_transpose(big, vT, d, 3 * d, L3, B, H, L, Lp)
            ops.attention(big, big, vT, big, 2 * d, 0, 2 * d, 3 * d, L3, B, H, L, Lp, Lt, plan, scale, q_prescaled=True,
                          q_row_begin=r0 if tail else 0, v_off=d if self.v_rowmajor else None)
            join(0, 1)
            if not pre_only:
                with on_side():
                    ops.gemm(big, blk["o_txt"][0], hidden, Lt, d, d, 3 * d, d, d, bias=blk["o_txt"][1], res=hidden,
                             gate=mod, gate_off=mb + 8 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld, gate_stride=nm,
                             flags=GEMM_GATE_RES, a_off=2 * d, workspace=ws_txt)
                    ln(Lt, 0, mb + 9 * d, mb + 10 * d)
                    ops.gemm(xn, blk["ff1_txt"][0], big, Lt, 4 * d, d, d, d, 4 * d, bias=blk["ff1_txt"][1], batch=B,
                             strideA=Ld, strideC=L4, gelu_from=0, c_off=mlp_base, workspace=ws_txt)
                    ops.gemm(big, blk["ff2_txt"][0], hidden, Lt, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_txt"][1],
                             res=hidden, gate=mod, gate_off=mb + 11 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                             gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base, workspace=ws_txt)
            ops.gemm(big, blk["o_img"][0], hidden, n_act, d, d, 3 * d, d, d, bias=blk["o_img"][1], res=hidden,
                     gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L3, strideC=Ld, strideR=Ld, gate_stride=nm,
                     flags=GEMM_GATE_RES, a_off=r0 * 3 * d + 2 * d, c_off=r0 * d, r_off=r0 * d, tail_workspace=ws_img)
            ln(n_act, r0 * d, mb + 3 * d, mb + 4 * d)
            ops.gemm(xn, blk["ff1_img"][0], big, n_act, 4 * d, d, d, d, 4 * d, bias=blk["ff1_img"][1], batch=B,
                     strideA=Ld, strideC=L4, gelu_from=0, a_off=r0 * d, c_off=mlp_base + r0 * 4 * d, tail_workspace=ws_img)
            ops.gemm(big, blk["ff2_img"][0], hidden, n_act, d, 4 * d, 4 * d, 4 * d, d, bias=blk["ff2_img"][1],
                     res=hidden, gate=mod, gate_off=mb + 5 * d, ldr=d, batch=B, strideA=L4, strideC=Ld, strideR=Ld,
                     gate_stride=nm, flags=GEMM_GATE_RES, a_off=mlp_base + r0 * 4 * d, c_off=r0 * d, r_off=r0 * d, tail_workspace=ws_img)
            if debug is not None and ("hidden_d0" not in debug or "blocks" in debug):
                join(1, 0)
                snap = hidden[:B * L * d].view(B, L, d).clone()
                debug.setdefault("hidden_d0", snap)
                if "blocks" in debug:             # caller asked for the state after EVERY block (parity tests)
                    debug["blocks"].append(snap)
        join(1, 0)

        n_cur = plan.n_cur
        for bi, blk in enumerate(sgl):
            mb = blk["mod"]
            ln(L, 0, mb, mb + d)
            if skip_dead and bi == len(sgl) - 1 and n_cur < L:
                # LAST block: only the current frame's rows reach the output (split_output keeps [-n_cur:],
                # modeling_pyramid_flux.py:380).  K and V are still needed for every row, but Q, the MLP branch,
                # the attention rows and proj_out only for the last n_cur rows -- identical values, less work.
                r0 = L - n_cur
                fuse = self.fuse_qk
                qk_s = dict(rope=plan.rope, wq=blk["norm_q"], wk=blk["norm_k"], d=d, eps=w.qk_eps, q_scale=qs) if fuse else None
                ops.gemm(xn, blk["kvqm"][0], big, L, 2 * d, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B, strideA=Ld,
                         strideC=L7, tail_workspace=None if fuse else ws_img, qk=dict(qk_s, k_col0=0) if fuse else None)
                ops.gemm(xn, blk["kvqm"][0], big, n_cur, 5 * d, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B, strideA=Ld,
                         strideC=L7, gelu_from=d, a_off=r0 * d, c_off=r0 * 7 * d + 2 * d, w_off=2 * d * d, bias_off=2 * d,
                         tail_workspace=None if fuse else ws_img, qk=dict(qk_s, q_col0=0, row0=r0) if fuse else None)
                if not fuse:
                    ops.qk_norm_rope(big, 7 * d, L7, 2 * d, 0, blk["norm_q"], blk["norm_k"], None, None, plan.rope, B, L, Lt, H,
                                     q_scale=qs, eps=w.qk_eps)
                if not self.v_rowmajor:
                    ops.v_transpose(big, vT, d, 7 * d, L7, B, H, L, Lp)
                ops.attention(big, big, vT, big, 2 * d, 0, 2 * d, 7 * d, L7, B, H, L, Lp, Lt, plan, scale, q_prescaled=True,
                              q_row_begin=r0, v_off=d if self.v_rowmajor else None)
                ops.gemm(big, blk["out"][0], hidden, n_cur, d, 5 * d, 7 * d, 5 * d, d, bias=blk["out"][1], res=hidden,
                         gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L7, strideC=Ld, strideR=Ld, gate_stride=nm,
                         flags=GEMM_GATE_RES, a_off=r0 * 7 * d + 2 * d, c_off=r0 * d, r_off=r0 * d, tail_workspace=ws_img)
                continue
            fuse = self.fuse_qk
            ops.gemm(xn, blk["kvqm"][0], big, L, 7 * d, d, d, d, 7 * d, bias=blk["kvqm"][1], batch=B, strideA=Ld,
                     strideC=L7, gelu_from=3 * d, tail_workspace=None if fuse else ws_img,
                     qk=dict(rope=plan.rope, wq=blk["norm_q"], wk=blk["norm_k"], d=d, eps=w.qk_eps, q_scale=qs, k_col0=0,
                             q_col0=2 * d) if fuse else None)
            if not fuse:
                ops.qk_norm_rope(big, 7 * d, L7, 2 * d, 0, blk["norm_q"], blk["norm_k"], None, None, plan.rope, B, L, Lt, H, q_scale=qs)
            if not self.v_rowmajor:
                ops.v_transpose(big, vT, d, 7 * d, L7, B, H, L, Lp)
            ops.attention(big, big, vT, big, 2 * d, 0, 2 * d, 7 * d, L7, B, H, L, Lp, Lt, plan, scale, q_prescaled=True,
                          v_off=d if self.v_rowmajor else None)
            ops.gemm(big, blk["out"][0], hidden, L, d, 5 * d, 7 * d, 5 * d, d, bias=blk["out"][1], res=hidden,
                     gate=mod, gate_off=mb + 2 * d, ldr=d, batch=B, strideA=L7, strideC=Ld, strideR=Ld, gate_stride=nm,
                     flags=GEMM_GATE_RES, a_off=2 * d, tail_workspace=ws_img)
            if debug is not None and "blocks" in debug:
                debug["blocks"].append(hidden[:B * L * d].view(B, L, d).clone())
        if debug is not None:
            debug["hidden_final"] = hidden[:B * L * d].view(B, L, d).clone()


    def _head(self, plan, mod):
        """norm_out + proj_out on the current frame's tokens only (split_output keeps [-n_cur:], flux:380)"""
        w, d, H, B, Lt, L, L_img, Lp, hidden, xn, big, vT = self._geometry(plan)
        nm = w.n_mod
        Ld = L * d
        n_cur = plan.n_cur
        fo = (L - n_cur) * d
        mf = w.mod_final
        ops.ln_modulate(hidden, xn, (mod, mf + d), (mod, mf), d, B, n_cur, Ld, Ld, d, d, nm, x_off=fo, y_off=fo)
        npad = w.proj_w.shape[0]
        vtok = self._buf("vtok", B * n_cur * npad, torch.float32)
        ops.gemm(xn, w.proj_w, vtok, n_cur, npad, d, d, d, npad, bias=w.proj_b, batch=B, strideA=Ld,
                 strideC=n_cur * npad, flags=GEMM_OUT_F32, a_off=fo)
        return vtok[:B * n_cur * npad].view(B, n_cur, npad)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, torch_dtype=None, device="cuda", **config_overrides):
        """pipeline.py:73-87: `PyramidFluxTransformer.from_pretrained(os.path.join(model_path, variant), torch_dtype=,
        use_gradient_checkpointing=, use_flash_attn=, use_temporal_causal=, interp_condition_pos=, axes_dims_rope=, ...)`
        -- keyword arguments override entries of config.json, exactly like diffusers' ModelMixin."""
        sd, cfg = load_diffusers_dir(pretrained_model_path)
        cfg.update({k: v for k, v in config_overrides.items() if v is not None})
        if cfg.get("use_flash_attn"):
            raise NotImplementedError("use_flash_attn=True drops the temporal mask in the reference (SURVEY 8c); the "
                                      "masked attention kernel is the only attention path here")
        return cls(sd, cfg, device)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, sample, encoder_hidden_states=None, encoder_attention_mask=None, pooled_projections=None,
                timestep_ratio=None):
        """modeling_pyramid_flux.py:392-399 / modeling_pyramid_mmdit.py:420-427.
        Reference form: `sample` = List[n_stage] of List[clip [B,C,t,h,w]] (oldest .. current) -> List[n_stage] of
        the current clip's prediction [B,C,t,h,w] in the dtype of the clips.  Engine form (tests, smoke): `sample` =
        flat list of clips of ONE stage -> one fp32 tensor."""
        if len(sample) and isinstance(sample[0], (list, tuple)):
            outs = []
            for stage_clips in sample:
                o = self._forward_stage(stage_clips, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                                        timestep_ratio)
                outs.append(o.to(stage_clips[-1].dtype))
            return outs
        return self._forward_stage(sample, encoder_hidden_states, encoder_attention_mask, pooled_projections, timestep_ratio)

    def _forward_stage(self, clips, enc, enc_mask, pooled, timesteps):
        """one pyramid stage: returns [B, C, t, h, w] fp32 of the LAST clip (flux:392-542)."""
        clips = [c.to(self.dev, torch.float32).contiguous() for c in clips]
        shapes = [tuple(c.shape[2:]) for c in clips]
        plan = self.make_plan(shapes, enc_mask)
        ctx = self.encode_context(enc)
        vt = self.forward_tokens(plan, clips, [float(t) for t in timesteps], pooled, ctx)
        B = plan.B
        t, h, w_ = plan.cur
        Cc = self.w.out_cols // 4
        x = vt[:, :, :self.w.out_cols].reshape(B, t, h // 2, w_ // 2, 2, 2, Cc)
        x = x.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, t, h, w_, Cc).permute(0, 4, 1, 2, 3)
        return x.contiguous()


# the reference's two transformer classes are one engine here (the variant is read off the state dict)
PyramidFluxTransformer = FluxEngine       # pyramid_dit/flux_modules/modeling_pyramid_flux.py:60
PyramidDiffusionMMDiT = FluxEngine        # pyramid_dit/mmdit_modules/modeling_pyramid_mmdit.py:60
