"""Block-level operators with the reference's class names, constructor arguments and forward signatures:

  FluxTransformerBlock        pyramid_dit/flux_modules/modeling_flux_block.py:945-1044  (double stream)
  FluxSingleTransformerBlock  pyramid_dit/flux_modules/modeling_flux_block.py:877-942   (single stream)
  JointTransformerBlock       pyramid_dit/mmdit_modules/modeling_mmdit_block.py:567-671 (SD3-style, context_pre_only)

In the reference these are nn.Modules that the transformer stacks (and that FSDP wraps).  Here a block is a thin
operator over the SAME kernel sequence the whole-model engine runs (`FluxEngine._run_blocks` on a one-block weight
pack): AdaLN modulation by one gemv of silu(temb), fused K|V|Q (|MLP) projection GEMM, QK-RMSNorm + RoPE, masked
flash attention, gated out-projection / MLP GEMMs.  They exist so that callers (and tests) can address the model at the
granularity the reference exposes; the pipeline itself uses the whole-model engine.

Weights: `load_state_dict()` takes the block's own keys exactly as the reference's module would (`norm1.linear.weight`,
`attn.to_q.weight`, ...).  Inputs follow the reference's single-stage inference call: `attention_mask` is the
[B, 1, L, L] bool mask (or a one-element list of it) over the [text | image] sequence, `image_rotary_emb` the
[.., L, 1, hd/2, 2, 2] table of `EmbedND` / `get_1d_rotary_pos_embed` (or a one-element list), `hidden_length` is
accepted and checked.  The mask must have the structure the model produces (per row: one run of text keys, one prefix
of image keys) -- it is converted to the interval form the attention kernel consumes.
"""
import types

import numpy as np
import torch

from . import ops
from .flux import FluxEngine, FluxWeights
from .refapi import DeviceModuleAPI


class _MaskPlan:
    """the attention-plan interface of pyflow_hip.plan.SequencePlan, derived from explicit mask / rotary tensors"""
    QTILE = 128

    def __init__(self, mask, rope, Lt, n_cur, device):
        m = np.asarray(mask.detach().cpu().numpy(), dtype=bool)
        if m.ndim == 4:
            m = m[:, 0]
        B, L, L2 = m.shape
        assert L == L2, "attention_mask must be [B, 1, L, L]"
        self.B, self.L, self.Lt = B, L, Lt
        self.L_img = L - Lt
        self.Lp = (L + 63) // 64 * 64
        self.n_cur = n_cur
        txt, img = m[:, :, :Lt], m[:, :, Lt:]
        tn = txt.sum(-1)
        a_lo = np.where(tn > 0, txt.argmax(-1), 0).astype(np.int32)
        a_hi = (a_lo + tn).astype(np.int32)
        b_hi = (Lt + img.sum(-1)).astype(np.int32)
        j = np.arange(L)[None, None, :]
        rebuilt = np.where(j < Lt, (j >= a_lo[..., None]) & (j < a_hi[..., None]), j < b_hi[..., None])
        if not np.array_equal(rebuilt, m):
            raise NotImplementedError("attention_mask is not of the block-causal form the model builds "
                                      "(modeling_pyramid_flux.py:318-350): one run of text keys + one prefix of image keys per row")
        nqt = (L + self.QTILE - 1) // self.QTILE
        tile_end = np.zeros((B, nqt), np.int32)
        for qt in range(nqt):
            sl = slice(qt * self.QTILE, min((qt + 1) * self.QTILE, L))
            tile_end[:, qt] = np.maximum(b_hi[:, sl].max(axis=1), Lt)
        self.host = dict(a_lo=a_lo, a_hi=a_hi, b_hi=b_hi, tile_kv_end=tile_end)
        self.a_lo = torch.from_numpy(a_lo).to(device)
        self.a_hi = torch.from_numpy(a_hi).to(device)
        self.b_hi = torch.from_numpy(b_hi).to(device)
        self.tile_kv_end = torch.from_numpy(tile_end).to(device)
        # rotary table [.., L, 1, hd/2, 2, 2] = [[cos, -sin], [sin, cos]]  ->  [L, hd/2, (cos, sin)] fp32
        r = rope
        while r.ndim > 5:
            r = r[0]
        r = r.reshape(L, -1, 2, 2).float()
        self.rope = torch.stack([r[:, :, 0, 0], r[:, :, 1, 0]], dim=-1).contiguous().to(device)

    def useful_pairs(self, q_row_begin=0):
        h = self.host
        r = q_row_begin
        return int((h["a_hi"][:, r:] - h["a_lo"][:, r:]).astype(np.int64).sum()
                   + (h["b_hi"][:, r:] - self.Lt).astype(np.int64).sum())


def _one(x):
    return x[0] if isinstance(x, (list, tuple)) else x


class _BlockOp(DeviceModuleAPI):
    _PREFIX = ""
    _MMDIT = False

    def __init__(self, dim, num_attention_heads, attention_head_dim, device="cuda"):
        assert dim == num_attention_heads * attention_head_dim and attention_head_dim == 64
        self.dim, self.heads, self.head_dim = dim, num_attention_heads, attention_head_dim
        self.dev = torch.device(device)
        self._eng = None
        # the reference block owns an `attn` sub-module with a processor plug point (modeling_flux_block.py:703-704);
        # the fused HIP sequence is the only processor of this build
        self.attn = types.SimpleNamespace(processor="pyflow_hip fused attention", set_processor=self._set_processor)

    def _set_processor(self, processor):
        raise NotImplementedError("the attention of this block is the fused MI355X kernel sequence; foreign attention "
                                  "processors (torch callables over q/k/v modules) cannot be plugged into it")

    def _cfg(self):
        raise NotImplementedError

    def load_state_dict(self, state_dict, strict=True):
        sd = {self._PREFIX + k: v for k, v in state_dict.items()}
        cfg = self._cfg()
        eng = FluxEngine.__new__(FluxEngine)
        eng.dev = self.dev
        eng.w = FluxWeights(sd, cfg, self.dev, blocks_only=True)
        eng.cfg, eng._ws, eng._ctx, eng._mod_cache = cfg, {}, None, None
        eng.overlap_text, eng.skip_dead_rows, eng._side = False, False, None
        eng.fuse_qk = eng.v_rowmajor = eng.group_text = True
        eng.launch_mode, eng._ws_gen = "eager", 0
        eng._listed_plans = None
        self._eng = eng
        return self

    def _mod(self, temb):
        w = self._eng.w
        B = temb.shape[0]
        t = temb.to(self.dev, torch.float32).contiguous()
        mod = torch.empty(B * w.n_mod, dtype=torch.float32, device=self.dev)
        ops.gemv(w.mod_w, w.mod_b, t, mod, w.n_mod, w.d, B, silu_in=True)
        return mod

    def _fill(self, plan, enc, hid):
        eng = self._eng
        w, d, H, B, Lt, L, L_img, Lp, hidden, xn, big, vT = eng._geometry(plan)
        h = hidden[:B * L * d].view(B, L, d)
        if enc is not None:
            h[:, :Lt].copy_(enc.to(self.dev, torch.bfloat16))
            h[:, Lt:].copy_(hid.to(self.dev, torch.bfloat16))
        else:
            h.copy_(hid.to(self.dev, torch.bfloat16))
        return h


class FluxTransformerBlock(_BlockOp):
    """forward(hidden_states [B,L_img,d], encoder_hidden_states [B,Lt,d], encoder_attention_mask, temb [B,d],
    attention_mask, hidden_length, image_rotary_emb) -> (encoder_hidden_states, hidden_states)   (:992-1044)"""
    _PREFIX = "transformer_blocks.0."

    def __init__(self, dim, num_attention_heads, attention_head_dim, qk_norm="rms_norm", eps=1e-6, use_flash_attn=False,
                 device="cuda"):
        assert qk_norm == "rms_norm" and not use_flash_attn
        super().__init__(dim, num_attention_heads, attention_head_dim, device)

    def _cfg(self):
        return dict(num_layers=1, num_single_layers=0, num_attention_heads=self.heads, attention_head_dim=self.head_dim,
                    axes_dims_rope=[16, 24, 24], _mmdit_blocks=self._MMDIT)

    def forward(self, hidden_states, encoder_hidden_states, encoder_attention_mask=None, temb=None, attention_mask=None,
                hidden_length=None, image_rotary_emb=None):
        B, L_img, d = hidden_states.shape
        Lt = encoder_hidden_states.shape[1]
        if hidden_length is not None:
            assert sum(hidden_length) == L_img, "single-stage call: hidden_length must add up to the image tokens"
        plan = _MaskPlan(_one(attention_mask), _one(image_rotary_emb), Lt, L_img, self.dev)
        h = self._fill(plan, encoder_hidden_states, hidden_states)
        self._eng._run_blocks(plan, self._mod(temb), last_block_tail=False)
        pre_only = self._eng.w.dbl[0]["pre_only"]
        # copies, never views of the engine's `hidden` workspace: the next call of this (or any) block overwrites it
        enc_out = None if pre_only else h[:, :Lt].to(encoder_hidden_states.dtype, copy=True)
        return enc_out, h[:, Lt:].to(hidden_states.dtype, copy=True)

    __call__ = forward


class JointTransformerBlock(FluxTransformerBlock):
    """SD3-style joint block (modeling_mmdit_block.py:567-671): QK-norm eps 1e-5, key names `attn.norm_add_q/k`,
    `context_pre_only=True` = the last block (AdaLayerNormContinuous on the text stream, no text output)."""
    _MMDIT = True

    def __init__(self, dim, num_attention_heads, attention_head_dim, qk_norm=None, context_pre_only=False,
                 use_flash_attn=False, device="cuda"):
        assert not use_flash_attn
        _BlockOp.__init__(self, dim, num_attention_heads, attention_head_dim, device)
        self.context_pre_only = context_pre_only


class FluxSingleTransformerBlock(_BlockOp):
    """forward(hidden_states [B,L,d] (text rows first), temb, encoder_attention_mask, attention_mask, hidden_length,
    image_rotary_emb) -> hidden_states   (:914-942)"""
    _PREFIX = "single_transformer_blocks.0."

    def __init__(self, dim, num_attention_heads, attention_head_dim, mlp_ratio=4.0, use_flash_attn=False, device="cuda"):
        assert mlp_ratio == 4.0 and not use_flash_attn
        super().__init__(dim, num_attention_heads, attention_head_dim, device)

    def _cfg(self):
        return dict(num_layers=0, num_single_layers=1, num_attention_heads=self.heads, attention_head_dim=self.head_dim,
                    axes_dims_rope=[16, 24, 24])

    def forward(self, hidden_states, temb=None, encoder_attention_mask=None, attention_mask=None, hidden_length=None,
                image_rotary_emb=None):
        B, L, d = hidden_states.shape
        Lt = encoder_attention_mask.shape[1] if encoder_attention_mask is not None else 0
        plan = _MaskPlan(_one(attention_mask), _one(image_rotary_emb), Lt, L - Lt, self.dev)
        h = self._fill(plan, None, hidden_states)
        self._eng._run_blocks(plan, self._mod(temb), last_block_tail=False)
        return h.to(hidden_states.dtype, copy=True)          # a copy: `h` is the engine's workspace

    __call__ = forward
