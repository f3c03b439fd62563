"""PyramidDiTForVideoGeneration -- drop-in for pyramid_dit/pyramid_dit_for_video_gen_pipeline.py:114-1279
(inference half: generate / generate_one_unit / decode_latent), driving the HIP engines.

Host loop = the reference's (autoregressive units x pyramid stages x Euler steps, pipeline.py:1126-1203,
706-788) with these MI355X-side changes, none of which alters the arithmetic:
  * the per-step DiT call, CFG combine and Euler update never leave the device: the latent state is an fp32
    device tensor updated in place by pf_cfg_euler_step; timesteps/sigmas are host scalars passed by value;
  * RoPE tables and the block-causal mask plan are built once per (unit, stage), not per forward;
  * text embedding of the context (`context_embedder`) is computed once per video;
  * block-correlated renoise noise is drawn vectorised (one randn(N,4) @ L^T) instead of N Python-loop draws
    -- same distribution, different stream; `block_noise_fn` lets callers inject the reference's stream;
  * no gc.collect()/empty_cache() per unit, no per-step broadcast (ranks are bit-deterministic).
"""
import json
import math
import time
import os

import numpy as np
import torch

from . import ops
from . import sp as sp_mod
from .flux import FluxEngine
from .scheduler import PyramidFlowMatchEulerDiscreteScheduler

DEFAULT_NEGATIVE = ("cartoon style, worst quality, low quality, blurry, absolute black, absolute white, low res, "
                    "extra limbs, extra digits, misplaced objects, mutated anatomy, monochrome, horror")


def block_noise_cholesky(gamma):
    """L with L L^T = (1+g) I - g 11^T (diagonal 1, off-diagonal -g: pipeline.py:699).  The matrix is singular at
    g = 1/3 (the four values of a 2x2 block sum to
    zero), so the factor is written in closed form instead of calling a (CPU-dependent) Cholesky on it."""
    L = torch.zeros(4, 4, dtype=torch.float64)
    a, b = 1.0, -gamma                  # diagonal / off-diagonal of the covariance
    for j in range(4):
        s = sum(L[j, k] ** 2 for k in range(j))
        L[j, j] = math.sqrt(max(a - s, 0.0))
        for i in range(j + 1, 4):
            t = sum(L[i, k] * L[j, k] for k in range(j))
            L[i, j] = (b - t) / L[j, j] if L[j, j] > 1e-12 else 0.0
    return L.float()


class PyramidDiTForVideoGeneration:
    def __init__(self, model_path=None, model_dtype="bf16", model_name="pyramid_mmdit", use_gradient_checkpointing=False,
                 return_log=True, model_variant="diffusion_transformer_768p", timestep_shift=1.0,
                 stage_range=[0, 1 / 3, 2 / 3, 1], sample_ratios=[1, 1, 1], scheduler_gamma=1 / 3,
                 use_mixed_training=False, use_flash_attn=False, load_text_encoder=True, load_vae=True,
                 max_temporal_length=31, frame_per_unit=1, use_temporal_causal=True, corrupt_ratio=1 / 3,
                 interp_condition_pos=True, stages=[1, 2, 4], video_sync_group=8, gradient_checkpointing_ratio=0.6,
                 dit_state_dict=None, dit_config=None, vae_state_dict=None, vae_config=None, text_encoder=None,
                 device="cuda", **kwargs):
        if model_name not in ("pyramid_flux", "pyramid_mmdit"):
            raise NotImplementedError("Unsupported DiT architecture, please set the model_name to `pyramid_flux` or "
                                      "`pyramid_mmdit`")                                     # pipeline.py:88-89
        assert use_temporal_causal and interp_condition_pos and not use_flash_attn
        assert frame_per_unit == 1, "fixed unit implementation (pipeline.py:181-186)"
        self.stages = list(stages)
        self.sample_ratios = sample_ratios
        self.model_name = model_name
        self.model_dtype = model_dtype
        self._device = torch.device(device)
        if dit_state_dict is None:
            dit_state_dict, dit_config = _load_diffusers_dir(os.path.join(model_path, model_variant))
        # like the reference (flux_block.py:734-743), the sequence-parallel form is chosen at construction time:
        # init_sequence_parallel_group() must have run before the model is built
        self.sp = sp_mod.get_sequence_parallel_comm() if sp_mod.is_sequence_parallel_initialized() else None
        if self.sp is not None and self.sp.world == 2 and sp_mod.is_guidance_parallel():
            from .flux_cfg import FluxEngineCFG          # two ranks: one branch of the guidance pair each, no all-to-all
            self.dit = FluxEngineCFG(dit_state_dict, dit_config, device, comm=self.sp)
        elif self.sp is not None and self.sp.world > 1:
            from .flux_sp import FluxEngineSP
            self.dit = FluxEngineSP(dit_state_dict, dit_config, device, comm=self.sp)
        else:
            self.sp = None
            self.dit = FluxEngine(dit_state_dict, dit_config, device)
        self.text_encoder = text_encoder
        self.load_text_encoder = load_text_encoder
        if text_encoder is None and load_text_encoder and model_path is not None \
                and os.path.isdir(os.path.join(model_path, "text_encoder")):                   # pipeline.py:99-103, 120-123
            from .text_encoder import FluxTextEncoderWithMask, SD3TextEncoderWithMask
            cls = FluxTextEncoderWithMask if model_name == "pyramid_flux" else SD3TextEncoderWithMask
            self.text_encoder = cls(model_path, device=device)
        self.vae = None
        self.load_vae = load_vae
        if load_vae:
            from .vae import CausalVideoVAE
            if vae_state_dict is None and model_path is not None:
                vae_state_dict, vae_config = _load_diffusers_dir(os.path.join(model_path, "causal_video_vae"))
            if vae_state_dict is not None:
                self.vae = CausalVideoVAE(vae_state_dict, vae_config, device)
        if model_name == "pyramid_flux":
            self.vae_shift_factor, self.vae_scale_factor = -0.04, 1 / 1.8726        # :165-167
        else:
            self.vae_shift_factor, self.vae_scale_factor = 0.1490, 1 / 1.8415       # :168-170
        self.vae_video_shift_factor, self.vae_video_scale_factor = -0.2343, 1 / 3.0986
        self.downsample = 8
        self.frame_per_unit = frame_per_unit
        self.max_temporal_length = max_temporal_length
        self.scheduler = PyramidFlowMatchEulerDiscreteScheduler(shift=timestep_shift, stages=len(self.stages),
                                                                stage_range=stage_range, gamma=scheduler_gamma)
        self.sequential_offload_enabled = False
        self.block_noise_fn = None          # (bs, ch, t, h, w) -> CPU fp32 tensor; None = vectorised global-RNG draw
        self._noise_slots = {}              # shape -> two [pinned staging tensor, event] slots (_to_device_async)
        self._chol = {}                     # gamma -> closed-form Cholesky factor (numpy)
        self._mask_keys = {}                # id(mask) -> (mask, bytes) for the current generate() call (_mask_key)
        self._plans = {}
        self.plan_cache_size = 256
        self.timers = {}

    # ---- reference properties (:1261-1279)
    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return torch.bfloat16 if self.model_dtype == "bf16" else torch.float32

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def video_guidance_scale(self):
        return self._video_guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 0

    # ---- raw .pth checkpoints (:213-241): same key handling as the reference, then the weights are re-packed
    @staticmethod
    def remap_dit_checkpoint(checkpoint):
        """:213-224: drop `vae*` / `text_encoder*` entries, strip a leading `dit.`"""
        out = {}
        for key, val in checkpoint.items():
            if key.startswith("vae") or key.startswith("text_encoder"):
                continue
            out[key.split(".", 1)[1] if key.startswith("dit") else key] = val
        return out

    def load_checkpoint(self, checkpoint_path, model_key="model", **kwargs):
        checkpoint = torch.load(checkpoint_path, map_location="cpu")
        sd = self.remap_dit_checkpoint(checkpoint)
        cfg = self.dit.cfg
        self.dit = type(self.dit)(sd, cfg, self._device, **({"comm": self.sp} if self.sp is not None else {}))
        self._plans = {}
        print(f"Load checkpoint from {checkpoint_path}: {len(sd)} tensors re-packed")

    def load_vae_checkpoint(self, vae_checkpoint_path, model_key="model"):
        from .vae import CausalVideoVAE
        checkpoint = torch.load(vae_checkpoint_path, map_location="cpu")[model_key]
        sd = {k.split(".", 1)[1]: v for k, v in checkpoint.items() if k.startswith("vae.")}          # :235-239
        tiling = self.vae.use_tiling if self.vae is not None else False
        self.vae = CausalVideoVAE(sd, None if self.vae is None else self.vae.cfg_in, self._device)
        self.vae.enable_tiling(tiling)
        print(f"Load the VAE from {vae_checkpoint_path}: {len(sd)} tensors re-packed")

    def enable_sequential_cpu_offload(self):
        # 288 GB of HBM: offloading is a no-op by design (pipeline.py:201-211 exists to fit 8-12 GB cards)
        self.sequential_offload_enabled = False

    # ---- noise (:676-703)
    def prepare_latents(self, batch_size, num_channels_latents, temp, height, width, dtype, device, generator):
        shape = (batch_size, num_channels_latents, int(temp), int(height) // self.downsample, int(width) // self.downsample)
        gdev = generator.device if generator is not None else "cpu"
        return torch.randn(shape, generator=generator, device=gdev, dtype=dtype)

    def _block_noise_into(self, out):
        """default draw, written into `out` (a contiguous CPU fp32 tensor [bs, ch, t, h, w], possibly pinned): ONE
        torch.randn(N, 4) from the global generator (N = 2 x 2 blocks), block k's four values = L . eps_k laid out
        `(b c t h w) (p q) -> b c t (h p) (w q)` (:697-703).  The 4 x 4 product and the rearrangement are written as eight
        numpy multiply-adds on strided views: a 4-column matmul and a permuting copy of 60 k floats each start an OpenMP
        region in torch, which costs 3-5 ms per stage boundary on a 64-core host -- more than a stage-0 forward of config C2
        lasts on the device (tools/host_stage_boundary.py)."""
        bs, ch, temp, height, width = out.shape
        g = float(self.scheduler.config.gamma)
        L = self._chol.get(g)
        if L is None:
            L = self._chol[g] = block_noise_cholesky(g).numpy().copy()
        n = bs * ch * temp * (height // 2) * (width // 2)
        eps = torch.randn(n, 4).numpy().reshape(bs, ch, temp, height // 2, width // 2, 4)
        o = out.numpy().reshape(bs, ch, temp, height // 2, 2, width // 2, 2)
        for i in range(4):                          # output position (p, q) = (i // 2, i % 2) of every block; L is lower triangular
            dst = o[:, :, :, :, i // 2, :, i % 2]
            np.multiply(eps[..., 0], L[i, 0], out=dst)
            for k in range(1, i + 1):
                dst += eps[..., k] * L[i, k]
        return out

    def sample_block_noise(self, bs, ch, temp, height, width):
        if self.block_noise_fn is not None:
            return self.block_noise_fn(bs, ch, temp, height, width)
        if not hasattr(self, "_chol"):
            self._chol = {}
        return self._block_noise_into(torch.empty(bs, ch, temp, height, width, dtype=torch.float32))

    def _to_device_async(self, z, shape=None):
        """a host fp32 tensor (the stage boundary's block noise) -> device WITHOUT blocking the host.  `.to(device)` from pageable
        memory is a synchronous copy: it returns when the stream has drained, i.e. the host -- which runs a whole stage ahead of
        the device (one graph launch per step) -- stopped at every stage boundary, and the device then idled for the copy plus the
        host's way to the next forward: 2.2 + 0.7 ms x 62 boundaries of a 241-frame video (round 6: tools/gap_analysis.py on a
        kernel trace of bench.py).  Here the values go through a pinned staging slot (two per shape, guarded by an event: a slot
        is rewritten only after the copy that read it has executed) and an asynchronous copy on the current stream, ordered
        behind the stage that is still running and in front of the re-noising kernel that consumes it.  Same draw, same values."""
        if z is not None and z.is_cuda:
            return z.to(self._device, torch.float32).contiguous()
        shape = tuple(shape) if z is None else tuple(z.shape)
        ring = self._noise_slots.setdefault(shape, [])
        if len(ring) < 2:
            slot = [torch.empty(shape, dtype=torch.float32).pin_memory(), None]
            ring.append(slot)
        else:
            slot = ring.pop(0)
            ring.append(slot)
            if slot[1] is not None:
                slot[1].synchronize()
        if z is None:
            self._block_noise_into(slot[0])          # the default draw goes straight into the staging slot
        else:
            slot[0].copy_(z)
        out = torch.empty(shape, dtype=torch.float32, device=self._device)
        out.copy_(slot[0], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return out

    def _mask_key(self, mask):
        """bytes of a prompt mask, read back ONCE per mask tensor and generate() call (a device-resident mask -- the text encoders'
        output -- costs a blocking device-to-host copy per (unit, stage) otherwise)"""
        hit = self._mask_keys.get(id(mask))
        if hit is not None and hit[0] is mask:
            return hit[1]
        key = mask.cpu().numpy().tobytes()
        self._mask_keys[id(mask)] = (mask, key)
        return key

    # ---- plan cache
    def _plan(self, shapes, mask):
        pair = bool(self.do_classifier_free_guidance)        # the rows of `mask` are one sample's [negative | positive] pair
        key = (tuple(shapes), self._mask_key(mask), pair)
        p = self._plans.pop(key, None)
        if p is None:
            # LRU over a WHOLE schedule (the headline job has 93 (unit, stage) sequences, C2 / C4 48): the plan carries its
            # recorded launch list / captured hipGraph (flux.py: _run_launch_list), so every video after the first one with
            # the same geometry and prompt mask replays its graphs instead of re-recording and re-instantiating ~350 launches
            # per plan (round 5 cleared this cache at 9 entries: every video re-recorded everything).  A plan is a few MB
            # (RoPE table + mask vectors + the graph): 256 of them are noise next to 288 GB.
            while len(self._plans) >= self.plan_cache_size:
                self._plans.pop(next(iter(self._plans)))
            p = self.dit.make_plan(shapes, mask, cfg_pair=pair)
        self._plans[key] = p          # (re)inserted last = most recently used
        return p

    def _reserve_for(self, num_units, h0, w0, Lt):
        """size the DiT's workspace for the LONGEST sequence of this job before its first forward (the last unit's last stage:
        the history rule of _history, counted in tokens), so that no buffer is re-allocated half-way through the first video
        -- a re-allocation invalidates every launch list / graph recorded so far (flux.py: _buf)."""
        if not hasattr(self.dit, "reserve"):
            return
        n_st = len(self.stages)
        tok = [(h0 * 2 ** i // 2) * (w0 * 2 ** i // 2) for i in range(n_st)]       # tokens per latent frame, stage 0 .. n_st-1
        u, i_s = num_units - 1, n_st - 1
        frames = [i_s]                              # the current frame's predecessor at this stage ...
        cur, ptx = i_s, 1
        while ptx < u:
            cur = max(cur - 1, 0)
            if cur == 0:
                break
            ptx += 1
            frames.append(cur)
        hist = sum(tok[c] for c in frames) if u >= 1 else 0
        if u >= 1 and cur == 0 and ptx < u:
            hist += (u - ptx) * tok[0]
        L_img = hist + tok[i_s]
        self.dit.reserve(2 if self.do_classifier_free_guidance else 1, Lt + L_img, L_img, tok[i_s])

    def _pyramid(self, x, n_down):
        """get_pyramid_latent (:555-570): x [C,T,H,W] fp32 device -> list low..high."""
        out = [x]
        for _ in range(n_down):
            C, T, H, W = x.shape
            y = torch.empty(C, T, H // 2, W // 2, dtype=torch.float32, device=x.device)
            ops.avgpool2(x, y, C * T, H, W, 1.0, self._round)
            out.append(y)
            x = y
        return list(reversed(out))

    def _history(self, clean, unit_index):
        """:1159-1182 -> per stage list of clips [1,C,t,h,w] oldest -> newest (CFG duplicate is implicit)."""
        res = []
        for i_s in range(len(self.stages)):
            stage_input = [clean[i_s][:, -1:]]
            cur_stage, ptx = i_s, 1
            while ptx < unit_index:
                cur_stage = max(cur_stage - 1, 0)
                if cur_stage == 0:
                    break
                ptx += 1
                stage_input.append(clean[cur_stage][:, -ptx:clean[cur_stage].shape[1] - (ptx - 1)])
            if cur_stage == 0 and ptx < unit_index:
                stage_input.append(clean[0][:, :-ptx])
            res.append([c.contiguous()[None] for c in reversed(stage_input)])
        return res

    @torch.no_grad()
    def generate_one_unit(self, x, past, prompt_mask, pooled, num_inference_steps, is_first_frame):
        """:706-788.  x: fp32 device latent [C,1,h0,w0] at stage-0 resolution.  Returns list of per-stage latents."""
        return self._one_unit_batch([x], [past], [(None, prompt_mask, pooled)], num_inference_steps, is_first_frame)[0]

    def _one_unit_batch(self, xs, pasts, ctxs, num_inference_steps, is_first_frame):
        """generate_one_unit for a batch of samples (:706-788 with latents.shape[0] > 1).  The samples share nothing but
        the random stream: per stage ONE block-noise draw of batch shape (as the reference's, :735), then each sample
        runs its own step loop (its own prompt context and launch plan).  ctxs[b] = (embeds | None, mask, pooled);
        embeds None = the context already encoded by the caller.  Returns outs[b][stage]."""
        nb = len(xs)
        outs = [[] for _ in range(nb)]
        C = xs[0].shape[0]
        B = 2 if self.do_classifier_free_guidance else 1
        xs = list(xs)
        for i_s in range(len(self.stages)):
            if i_s > 0:
                h, w = xs[0].shape[-2] * 2, xs[0].shape[-1] * 2
                self.scheduler.set_timesteps(num_inference_steps[i_s], i_s, device=None)
                ori_sigma = 1 - self.scheduler.ori_start_sigmas[i_s]
                gamma = self.scheduler.config.gamma
                alpha = 1 / (math.sqrt(1 + (1 / gamma)) * (1 - ori_sigma) + ori_sigma)
                beta = alpha * (1 - ori_sigma) / math.sqrt(gamma)
                if self.block_noise_fn is None:
                    noise = self._to_device_async(None, (nb, C, 1, h, w))          # drawn into the pinned staging slot
                else:
                    noise = self._to_device_async(self.sample_block_noise(nb, C, 1, h, w))
                if self.sp is not None:          # rank-local RNG streams may differ: rank 0's draw is the one used
                    self.sp.broadcast(noise, 0)
                for b in range(nb):
                    xn = torch.empty(C, 1, h, w, dtype=torch.float32, device=self._device)
                    ops.renoise_upsample(xs[b], noise[b:b + 1], xn, C, h, w, alpha, beta, self._round)
                    xs[b] = xn
            for b in range(nb):
                x = xs[b]
                emb, prompt_mask, pooled = ctxs[b]
                if emb is not None:
                    self.dit.encode_context(emb, cfg_pair=bool(self.do_classifier_free_guidance))
                self.scheduler.set_timesteps(num_inference_steps[i_s], i_s, device=None)
                clips = pasts[b][i_s] + [x[None]]
                shapes = [tuple(c.shape[2:]) for c in clips]
                plan = self._plan(shapes, prompt_mask)
                gs = self._guidance_scale if is_first_frame else self._video_guidance_scale
                h, w = x.shape[-2], x.shape[-1]
                for t in self.scheduler._timesteps_host:
                    tv = float(t)
                    if self._round:
                        tv = float(torch.tensor(tv, dtype=torch.float64).to(torch.bfloat16))      # pipeline.py:750
                    vtok = self.dit.forward_tokens(plan, clips, [tv] * B, pooled, shared_clips=True)
                    ds = self.scheduler.dsigma()
                    if self._round:
                        # scheduling_flow_matching.py:283: `(sigma_next - sigma) * model_output` multiplies a 0-dim
                        # float64 TENSOR with a bf16 tensor -- the 0-dim operand is converted to the common dtype (bf16)
                        # before the product, so the reference's bf16 path steps with bf16(dsigma)
                        ds = float(torch.tensor(ds, dtype=torch.float64).to(torch.bfloat16))
                    ops.cfg_euler_step(vtok, vtok.stride(0), vtok.stride(1), x, C, h, w, gs, B == 2, ds, self._round)
                outs[b].append(x)
        return outs

    @torch.no_grad()
    def generate(self, prompt=None, height=None, width=None, temp=1, num_inference_steps=28,
                 video_num_inference_steps=28, guidance_scale=7.0, video_guidance_scale=7.0, min_guidance_scale=2.0,
                 use_linear_guidance=False, alpha=0.5, negative_prompt=DEFAULT_NEGATIVE, num_images_per_prompt=1,
                 generator=None, output_type="pil", save_memory=True, cpu_offloading=False, inference_multigpu=False,
                 callback=None, prompt_embeds=None):
        """:1006-1219.  `prompt_embeds=(embeds[B,Lt,C], mask[B,Lt], pooled[B,Cp], neg_embeds, neg_mask, neg_pooled)`
        bypasses the text encoders (synthetic-prompt benchmark).  A list of B prompts (x `num_images_per_prompt`) is a
        batch of B*n independent samples drawn from ONE latent / block-noise stream of batch shape (:1049-1053, :1100);
        the negative prompt (one string, or one per prompt) is repeated to the batch -- the reference needs the caller to
        pass B negative prompts, its [negative | positive] concatenation (:1082-1085) does not broadcast."""
        assert (temp - 1) % self.frame_per_unit == 0, "The frames should be divided by frame_per unit"
        assert height % 64 == 0 and width % 64 == 0, "height/width must be multiples of 64 (8 VAE x 4 pyramid x 2 patch)"
        n_st = len(self.stages)
        self._mask_keys = {}          # (a caller may refill a mask tensor in place between calls)
        if isinstance(num_inference_steps, int):
            num_inference_steps = [num_inference_steps] * n_st
        if isinstance(video_num_inference_steps, int):
            video_num_inference_steps = [video_num_inference_steps] * n_st
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise RuntimeError("no text encoder loaded: pass prompt_embeds=(...) (synthetic prompts) or a text_encoder")
            if isinstance(prompt, str):
                prompt = prompt + ", hyper quality, Ultra HD, 8K"
            else:
                assert isinstance(prompt, list) and len(prompt) >= 1
                prompt = [p + ", hyper quality, Ultra HD, 8K" for p in prompt]
            pe, pm, pp = self.text_encoder(prompt, self._device)
            ne, nm, npool = self.text_encoder(negative_prompt or "", self._device)
        else:
            pe, pm, pp, ne, nm, npool = prompt_embeds
        nb = pe.shape[0]
        assert ne.shape[0] in (1, nb), "negative_prompt: one string, or one per prompt"
        n_img = int(num_images_per_prompt)
        self._guidance_scale = guidance_scale
        self._video_guidance_scale = video_guidance_scale
        if use_linear_guidance:
            guidance_scale_list = [max(guidance_scale - alpha * t_, min_guidance_scale) for t_ in range(temp)]
        self._round = (self.model_dtype == "bf16") and pe.dtype == torch.bfloat16
        ctxs = []                # per sample: ([negative | positive] embeds, mask, pooled) -- :1082-1085 for its row pair
        for b in range(nb):
            nbi = b if ne.shape[0] == nb else 0
            if self.do_classifier_free_guidance:
                ctx = (torch.cat([ne[nbi:nbi + 1], pe[b:b + 1]], dim=0), torch.cat([nm[nbi:nbi + 1], pm[b:b + 1]], dim=0),
                       torch.cat([npool[nbi:nbi + 1], pp[b:b + 1]], dim=0))
            else:
                ctx = (pe[b:b + 1], pm[b:b + 1], pp[b:b + 1])
            ctxs += [ctx] * n_img                # repeat_interleave order of the text encoders' num_images_per_prompt
        nb *= n_img
        if nb == 1:                              # one sample: its context is encoded once for the whole run
            self.dit.encode_context(ctxs[0][0], cfg_pair=bool(self.do_classifier_free_guidance))
            ctxs = [(None, ctxs[0][1], ctxs[0][2])]
        C = self.dit.w.out_cols // 4
        latents = self.prepare_latents(nb, C, temp, height, width, pe.dtype, self._device, generator)
        x = latents.to(self._device, torch.float32).contiguous()                    # [B,C,T,H,W]
        if self.sp is not None:              # pipeline.py:1089-1095 / 752-756: one broadcast instead of one per step
            self.sp.broadcast(x, 0)
        for _ in range(n_st - 1):                                                     # :1112-1116
            Bc, Cc, T, H, W = x.shape
            y = torch.empty(Bc, Cc, T, H // 2, W // 2, dtype=torch.float32, device=self._device)
            ops.avgpool2(x, y, Bc * Cc * T, H, W, 2.0, self._round)
            x = y
        num_units = 1 + (temp - 1) // self.frame_per_unit
        generated = [[] for _ in range(nb)]
        self._reserve_for(num_units, x.shape[-2], x.shape[-1], ctxs[0][1].shape[1])
        phases = getattr(self, "phase_times", None)      # bench.py: a dict that accumulates wall seconds per phase
        if phases is not None:
            torch.cuda.synchronize()
            t_ph = time.perf_counter()
        for unit_index in range(num_units):
            if callback:
                callback(unit_index, num_units)
            if use_linear_guidance:
                self._guidance_scale = guidance_scale_list[unit_index]
                self._video_guidance_scale = guidance_scale_list[unit_index]
            xs = [x[b, :, unit_index:unit_index + 1].contiguous() for b in range(nb)]
            if unit_index == 0:
                pasts = [[[] for _ in range(n_st)] for _ in range(nb)]
                outs = self._one_unit_batch(xs, pasts, ctxs, num_inference_steps, True)
            else:
                pasts = [self._history(self._pyramid(torch.cat(generated[b], dim=1), n_st - 1), unit_index)
                         for b in range(nb)]
                outs = self._one_unit_batch(xs, pasts, ctxs, video_num_inference_steps, False)
            for b in range(nb):
                generated[b].append(outs[b][-1])
        gen = torch.stack([torch.cat(g_, dim=1) for g_ in generated])                # [B,C,T,h,w] fp32
        if phases is not None:
            torch.cuda.synchronize()
            phases["sampling_s"] = phases.get("sampling_s", 0.0) + time.perf_counter() - t_ph
            t_ph = time.perf_counter()
        if output_type == "latent":
            return gen.to(pe.dtype) if self._round else gen
        out = self.decode_latent(gen, save_memory=save_memory, inference_multigpu=inference_multigpu,
                                 output_type=output_type)
        if phases is not None:
            torch.cuda.synchronize()
            phases["decode_s"] = phases.get("decode_s", 0.0) + time.perf_counter() - t_ph
        return out

    @torch.no_grad()
    def decode_latent(self, latents, save_memory=True, inference_multigpu=False, output_type="pil"):
        """:1221-1243.  Returns list[PIL] (or uint8 [T,H,W,3] tensor for output_type='uint8')."""
        if self.sp is None and inference_multigpu and torch.distributed.is_initialized() \
                and torch.distributed.get_rank() != 0:
            return None
        if self.vae is None:
            raise RuntimeError("VAE not loaded")
        if latents.shape[0] > 1:            # "B C T H W -> (B T) H W C" (:1239): the samples decode one after another
            parts = [self.decode_latent(latents[b:b + 1], save_memory, inference_multigpu, "uint8")
                     for b in range(latents.shape[0])]
            return self._finish_frames(None if parts[0] is None else torch.cat(parts, dim=0), output_type)
        z = latents.to(self._device, torch.float32)
        # un-normalisation (:1226-1230) is folded into the latent -> channels-last load of the decoder
        aff = (1.0 / self.vae_scale_factor, self.vae_shift_factor,
               1.0 / self.vae_video_scale_factor, self.vae_video_shift_factor)
        # with a sequence-parallel group the decode is tile-parallel over the same ranks (the reference leaves every
        # rank but 0 idle here, pipeline.py:1223-1224); frames are assembled on rank 0, other ranks return None
        comm = self.sp if (self.sp is not None and self.vae.use_tiling) else None
        # a context-parallel group (utils.initialize_context_parallel, the reference's switch: every CausalConv3d diverts
        # to the halo-exchange path whenever that group exists, modeling_causal_conv.py:119-120) selects the temporal
        # context-parallel decode: un-tiled, frame ranges over the ranks, one halo exchange per causal conv
        from . import cp as cp_mod
        if cp_mod.is_context_parallel_initialized() and cp_mod.get_context_parallel_world_size() > 1:
            return self._finish_frames(self.vae.decode_context_parallel(z, cp_mod.get_context_parallel_comm(), affine=aff),
                                       output_type)
        if self.sp is not None and comm is None and self.sp.rank != 0:
            return None
        if save_memory:
            u8 = self.vae.decode_to_uint8(z, window_size=1, tile_sample_min_size=256, affine=aff, comm=comm)
        else:
            u8 = self.vae.decode_to_uint8(z, window_size=2, tile_sample_min_size=512, affine=aff, comm=comm)
        return self._finish_frames(u8, output_type)

    @staticmethod
    def _finish_frames(u8, output_type):
        if u8 is None or output_type == "uint8":
            return u8
        arr = u8.cpu().numpy()
        from PIL import Image
        return [Image.fromarray(a) for a in arr]

    @torch.no_grad()
    def generate_i2v(self, prompt="", input_image=None, temp=1, num_inference_steps=28, guidance_scale=7.0,
                     video_guidance_scale=4.0, min_guidance_scale=2.0, use_linear_guidance=False, alpha=0.5,
                     negative_prompt=DEFAULT_NEGATIVE, num_images_per_prompt=1, generator=None, output_type="pil",
                     save_memory=True, cpu_offloading=False, inference_multigpu=False, callback=None,
                     prompt_embeds=None, posterior_noise=None):
        """:791-1003.  input_image: PIL.Image (as the reference) or a float tensor [3,H,W] already in [-1,1].
        `posterior_noise` ([1,C,1,h,w]) replaces the global-RNG draw of latent_dist.sample() (:911) for
        reproducible comparisons; `prompt_embeds` as in generate()."""
        if self.vae is None or not self.vae.has_encoder:
            raise RuntimeError("generate_i2v needs a VAE with encoder weights")
        if isinstance(input_image, torch.Tensor):
            img = input_image.float()
        else:       # transforms.ToTensor + Normalize(0.5, 0.5) of :906-909
            arr = torch.from_numpy(np.asarray(input_image.convert("RGB"), dtype=np.uint8).copy())
            img = (arr.permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
        height, width = img.shape[-2], img.shape[-1]
        assert temp % self.frame_per_unit == 0, "The frames should be divided by frame_per unit"
        assert height % 64 == 0 and width % 64 == 0, "height/width must be multiples of 64 (8 VAE x 4 pyramid x 2 patch)"
        n_st = len(self.stages)
        self._mask_keys = {}          # (a caller may refill a mask tensor in place between calls)
        if isinstance(num_inference_steps, int):
            num_inference_steps = [num_inference_steps] * n_st
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise RuntimeError("no text encoder loaded: pass prompt_embeds=(...) (synthetic prompts) or a text_encoder")
            prompt = prompt + ", hyper quality, Ultra HD, 8K" if isinstance(prompt, str) else \
                [p_ + ", hyper quality, Ultra HD, 8K" for p_ in prompt]
            pe, pm, pp = self.text_encoder(prompt, self._device)
            ne, nm, npool = self.text_encoder(negative_prompt or "", self._device)
        else:
            pe, pm, pp, ne, nm, npool = prompt_embeds
        self._guidance_scale = guidance_scale
        self._video_guidance_scale = video_guidance_scale
        if use_linear_guidance:
            guidance_scale_list = [max(guidance_scale - alpha * t_, min_guidance_scale) for t_ in range(temp + 1)]
        if self.do_classifier_free_guidance:
            pe = torch.cat([ne, pe], dim=0)
            pp = torch.cat([npool, pp], dim=0)
            pm = torch.cat([nm, pm], dim=0)
        self._round = (self.model_dtype == "bf16") and pe.dtype == torch.bfloat16
        self.dit.encode_context(pe, cfg_pair=bool(self.do_classifier_free_guidance))
        C = self.dit.w.out_cols // 4
        latents = self.prepare_latents(1, C, temp, height, width, pe.dtype, self._device, generator)
        x = latents[0].to(self._device, torch.float32).contiguous()
        if self.sp is not None:
            self.sp.broadcast(x, 0)
        for _ in range(n_st - 1):
            Cc, T, H, W = x.shape
            y = torch.empty(Cc, T, H // 2, W // 2, dtype=torch.float32, device=self._device)
            ops.avgpool2(x, y, Cc * T, H, W, 2.0, self._round)
            x = y
        # image latent (:906-911): encode, sample the posterior, normalise with the IMAGE statistics
        post = self.vae.encode(img[None, :, None].to(self._device)).latent_dist
        z = post.sample(eps=posterior_noise) if posterior_noise is not None else post.sample()
        z = ((z - self.vae_shift_factor) * self.vae_scale_factor)[0].float().contiguous()        # [C,1,h,w]
        if self._round:
            z = z.to(torch.bfloat16).float()
        if self.sp is not None:
            self.sp.broadcast(z, 0)                                                           # :913-917
        num_units = temp // self.frame_per_unit
        generated = [z]
        for unit_index in range(1, num_units):
            if callback:
                callback(unit_index, num_units)
            if use_linear_guidance:
                self._guidance_scale = guidance_scale_list[unit_index]
                self._video_guidance_scale = guidance_scale_list[unit_index]
            clean = self._pyramid(torch.cat(generated, dim=1), n_st - 1)
            past = self._history(clean, unit_index)
            outs = self.generate_one_unit(x[:, unit_index - 1:unit_index].contiguous(), past, pm, pp,
                                          num_inference_steps, False)
            generated.append(outs[-1])
        gen = torch.cat(generated, dim=1)[None]
        if output_type == "latent":
            return gen.to(pe.dtype) if self._round else gen
        return self.decode_latent(gen, save_memory=save_memory, inference_multigpu=inference_multigpu,
                                  output_type=output_type)


from .refapi import load_diffusers_dir as _load_diffusers_dir  # noqa: E402  (config.json + *.safetensors, pipeline.py:73,156)
