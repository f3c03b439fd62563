"""Prompt encoders on the MI355X path: T5 encoder + CLIP text towers behind the reference's wrapper classes.

Mirrors ``FluxTextEncoderWithMask`` (pyramid_dit/flux_modules/modeling_text_encoder.py:15-134: CLIP-L ``pooler_output``
:98, T5 with ``max_sequence_length=128`` :39) and ``SD3TextEncoderWithMask``
(pyramid_dit/mmdit_modules/modeling_text_encoder.py:15-139: two ``CLIPTextModelWithProjection`` pooled outputs
concatenated :127, same T5 call).  The encoder arithmetic the reference gets from transformers (pinned ==4.39.3):

* ``T5EncoderModel`` (models/t5/modeling_t5.py): pre-norm blocks of T5LayerNorm (RMS, no bias) -> q/k/v/o without
  bias and WITHOUT 1/sqrt(d) scaling, additive bucketed relative-position bias shared by all layers (learned in
  block 0) plus the key-padding mask, softmax in fp32 -> T5DenseGatedActDense ``wo(gelu_new(wi_0 x) * wi_1 x)`` ->
  final T5LayerNorm.
* ``CLIPTextModel`` (models/clip/modeling_clip.py): token + learned position embedding, pre-LayerNorm blocks with
  biased q/k/v/out projections, causal mask, scale head_dim**-0.5, MLP fc1 -> quick_gelu | gelu -> fc2, final
  LayerNorm, pooled = the EOS token's row (argmax of the ids for the legacy ``eos_token_id == 2`` configs), optional
  bias-free ``text_projection``.

Everything on the device goes through the C ABI (pf_gemm_bf16, pf_attention_small_bf16, pf_rmsnorm, pf_ln_modulate,
pf_glu_mul, pf_embed_rows, pf_copy_rows); torch is used for buffers only.  Tokenisation is host string work and stays
with the transformers tokenizers, exactly as in the reference (injectable for tests: any callable with the
tokenizer call signature returning ``input_ids`` / ``attention_mask``).
"""
import json
import math
import os

import torch

from . import ops
from .lib import GEMM_GATE_RES, GEMM_ACT_QUICK_GELU, GEMM_ACT_GELU_ERF
from .refapi import DeviceModuleAPI


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def t5_relative_position_buckets(L, num_buckets=32, max_distance=128):
    """int64 [L][L] bucket of (key j - query i), bidirectional (T5Attention._relative_position_bucket,
    modeling_t5.py).  Half of the buckets per sign; the first half of those exact, the rest log-spaced up to
    max_distance.  The log is evaluated in fp32 with the same operation order as transformers so that the truncation
    lands on the same side at the exact powers."""
    half = num_buckets // 2
    max_exact = half // 2
    dist = torch.arange(L, dtype=torch.long)
    large = max_exact + (torch.log(dist.float() / max_exact) / math.log(max_distance / max_exact)
                         * (half - max_exact)).to(torch.long)          # dist = 0 -> -inf -> huge negative, unused
    large = torch.clamp(large, max=half - 1)
    of_dist = torch.where(dist < max_exact, dist, large)               # [L]
    rel = torch.arange(L)[None, :] - torch.arange(L)[:, None]          # j - i
    return of_dist[rel.abs()] + (rel > 0).long() * half


def _bf16(t, device):
    return t.detach().to(device=device, dtype=torch.bfloat16).contiguous()


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


_SPLITK_WS = {}


def _splitk_scratch(device):
    """fp32 scratch that lets the encoders' skinny GEMMs (M = 77..256 rows) split their K range (pf_gemm_desc.workspace);
    one per device, shared by the encoders: their launches are all on the current stream, in order"""
    key = str(device)
    t = _SPLITK_WS.get(key)
    if t is None:
        t = _SPLITK_WS[key] = torch.empty(8 << 20, dtype=torch.float32, device=device)
    return t


class T5EncoderHIP:
    """``T5EncoderModel(input_ids, attention_mask)[0]`` -> last_hidden_state [B, L, d_model] bf16."""

    def __init__(self, state_dict, config, device="cuda"):
        sd = state_dict
        g = lambda k, d=None: _cfg_get(config, k, d)
        self.d_model, self.d_ff = g("d_model"), g("d_ff")
        self.H, self.n_layers = g("num_heads"), g("num_layers")
        self.eps = g("layer_norm_epsilon", 1e-6)
        self.vocab = g("vocab_size")
        self.num_buckets = g("relative_attention_num_buckets", 32)
        self.max_distance = g("relative_attention_max_distance", 128)
        if g("d_kv") != 64:
            raise ValueError("T5EncoderHIP: d_kv must be 64")
        ffp = g("feed_forward_proj", "gated-gelu")
        if ffp != "gated-gelu":
            raise ValueError(f"T5EncoderHIP: only the gated-gelu FFN of T5 v1.1 is implemented (got {ffp})")
        self.inner = self.H * 64
        for n, what in ((self.d_model, "d_model"), (self.d_ff, "d_ff")):
            if n % 128:
                raise ValueError(f"T5EncoderHIP: {what} must be a multiple of 128")
        self.device = device
        emb = sd["shared.weight"] if "shared.weight" in sd else sd["encoder.embed_tokens.weight"]
        self.embed = _bf16(emb, device)
        self.rel_bias = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach() \
            .to(torch.bfloat16).float().cpu()                                                  # [buckets][H]
        self.layers = []
        for i in range(self.n_layers):
            a = f"encoder.block.{i}.layer.0."
            f = f"encoder.block.{i}.layer.1."
            self.layers.append(dict(
                ln1=_f32(sd[a + "layer_norm.weight"].to(torch.bfloat16), device),
                wqkv=_bf16(torch.cat([sd[a + "SelfAttention.q.weight"], sd[a + "SelfAttention.k.weight"],
                                      sd[a + "SelfAttention.v.weight"]], 0), device),
                wo=_bf16(sd[a + "SelfAttention.o.weight"], device),
                ln2=_f32(sd[f + "layer_norm.weight"].to(torch.bfloat16), device),
                # [linear half | gelu half]: the GEMM epilogue applies GELU-tanh (== gelu_new) to columns >= d_ff
                wi=_bf16(torch.cat([sd[f + "DenseReluDense.wi_1.weight"], sd[f + "DenseReluDense.wi_0.weight"]], 0), device),
                wff=_bf16(sd[f + "DenseReluDense.wo.weight"], device),
            ))
        self.ln_f = _f32(sd["encoder.final_layer_norm.weight"].to(torch.bfloat16), device)
        self._bias_cache = {}

    @property
    def dtype(self):
        return torch.bfloat16

    def position_bias(self, L):
        """fp32 [H][L][L] device table (T5Attention.compute_bias); cached per L."""
        if L not in self._bias_cache:
            buckets = t5_relative_position_buckets(L, self.num_buckets, self.max_distance)
            self._bias_cache[L] = self.rel_bias[buckets].permute(2, 0, 1).contiguous().to(self.device)
        return self._bias_cache[L]

    def forward(self, input_ids, attention_mask=None):
        B, L = input_ids.shape
        if L > 256:
            raise ValueError("T5EncoderHIP: at most 256 tokens")
        rows, d, inner, dff = B * L, self.d_model, self.inner, self.d_ff
        dev = self.device
        ids = input_ids.to(device=dev, dtype=torch.int32).contiguous()
        km = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int32).contiguous()
        bias = self.position_bias(L)
        h = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        n = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        qkv = torch.empty(rows, 3 * inner, dtype=torch.bfloat16, device=dev)
        att = torch.empty(rows, inner, dtype=torch.bfloat16, device=dev)
        ff = torch.empty(rows, 2 * dff, dtype=torch.bfloat16, device=dev)
        gl = torch.empty(rows, dff, dtype=torch.bfloat16, device=dev)
        ws = _splitk_scratch(dev)
        ops.embed_rows(self.embed, ids, h, d, rows, self.vocab)
        for w in self.layers:
            ops.rmsnorm(h, n, w["ln1"], d, rows, eps=self.eps)
            ops.gemm(n, w["wqkv"], qkv, rows, 3 * inner, d, d, d, 3 * inner, workspace=ws)
            ops.attention_small(qkv, att, 0, inner, 2 * inner, 3 * inner, inner, B, self.H, L, 1.0, bias=bias, key_mask=km)
            ops.gemm(att, w["wo"], h, rows, d, inner, inner, inner, d, res=h, ldr=d, flags=GEMM_GATE_RES, workspace=ws)
            ops.rmsnorm(h, n, w["ln2"], d, rows, eps=self.eps)
            ops.gemm(n, w["wi"], ff, rows, 2 * dff, d, d, d, 2 * dff, gelu_from=dff, workspace=ws)
            ops.glu_mul(ff, gl, rows, dff)
            ops.gemm(gl, w["wff"], h, rows, d, dff, dff, dff, d, res=h, ldr=d, flags=GEMM_GATE_RES, workspace=ws)
        ops.rmsnorm(h, n, self.ln_f, d, rows, eps=self.eps)
        return n.view(B, L, d)

    __call__ = forward


class CLIPTextHIP:
    """``CLIPTextModel`` / ``CLIPTextModelWithProjection``: returns ``(last_hidden_state, pooled)`` where pooled is
    ``pooler_output`` (no projection weight in the state dict) or ``text_embeds`` (with ``text_projection.weight``)."""

    def __init__(self, state_dict, config, device="cuda"):
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()}
        g = lambda k, d=None: _cfg_get(config, k, d)
        self.d, self.dff = g("hidden_size"), g("intermediate_size")
        self.H, self.n_layers = g("num_attention_heads"), g("num_hidden_layers")
        self.eps = g("layer_norm_eps", 1e-5)
        self.vocab, self.max_pos = g("vocab_size"), g("max_position_embeddings", 77)
        self.eos_token_id = g("eos_token_id", 2)
        act = g("hidden_act", "quick_gelu")
        if act == "quick_gelu":
            self.act_flag = GEMM_ACT_QUICK_GELU
        elif act == "gelu":
            self.act_flag = GEMM_ACT_GELU_ERF
        elif act in ("gelu_new", "gelu_pytorch_tanh"):
            self.act_flag = 0
        else:
            raise ValueError(f"CLIPTextHIP: unsupported hidden_act {act}")
        if self.d != self.H * 64:
            raise ValueError("CLIPTextHIP: head_dim must be 64")
        if self.d % 128 or self.dff % 128:
            raise ValueError("CLIPTextHIP: hidden / intermediate size must be multiples of 128")
        self.device = device
        self.tok = _bf16(sd["embeddings.token_embedding.weight"], device)
        self.pos = _bf16(sd["embeddings.position_embedding.weight"], device)

        def ln(p):
            # pf_ln_modulate computes LN(x) * (1 + scale) + shift
            gamma = sd[p + ".weight"].to(torch.bfloat16).float()
            return (_f32(sd[p + ".bias"].to(torch.bfloat16), device), _f32(gamma - 1.0, device))

        self.layers = []
        for i in range(self.n_layers):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            self.layers.append(dict(
                ln1=ln(p + "layer_norm1"), ln2=ln(p + "layer_norm2"),
                wqkv=_bf16(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0), device),
                bqkv=_f32(torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0)
                          .to(torch.bfloat16), device),
                wo=_bf16(sd[a + "out_proj.weight"], device), bo=_f32(sd[a + "out_proj.bias"].to(torch.bfloat16), device),
                w1=_bf16(sd[p + "mlp.fc1.weight"], device), b1=_f32(sd[p + "mlp.fc1.bias"].to(torch.bfloat16), device),
                w2=_bf16(sd[p + "mlp.fc2.weight"], device), b2=_f32(sd[p + "mlp.fc2.bias"].to(torch.bfloat16), device),
            ))
        self.ln_f = ln("final_layer_norm")
        self.proj = None
        if "text_projection.weight" in state_dict:
            w = state_dict["text_projection.weight"]
            self.proj_dim = w.shape[0]
            n_pad = -(-self.proj_dim // 128) * 128                                    # GEMM N granularity
            wp = torch.zeros(n_pad, self.d, dtype=w.dtype)
            wp[:self.proj_dim] = w
            self.proj = _bf16(wp, device)

    @property
    def dtype(self):
        return torch.bfloat16

    def _ln(self, x, y, p, rows):
        ops.ln_modulate(x, y, (p[0], 0), (p[1], 0), self.d, 1, rows, 0, 0, self.d, self.d, 0, eps=self.eps)

    def eos_positions(self, input_ids):
        """CLIPTextTransformer.forward: row of the EOS token per prompt (host ints)."""
        ids = input_ids.cpu().long()
        if self.eos_token_id == 2:
            return ids.argmax(-1).tolist()
        return (ids == self.eos_token_id).int().argmax(-1).tolist()

    def forward(self, input_ids):
        B, L = input_ids.shape
        if L > self.max_pos or L > 256:
            raise ValueError("CLIPTextHIP: too many tokens")
        rows, d, dff, dev = B * L, self.d, self.dff, self.device
        ids = input_ids.to(device=dev, dtype=torch.int32).contiguous()
        h = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        n = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        qkv = torch.empty(rows, 3 * d, dtype=torch.bfloat16, device=dev)
        att = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        ff = torch.empty(rows, dff, dtype=torch.bfloat16, device=dev)
        ws = _splitk_scratch(dev)
        ops.embed_rows(self.tok, ids, h, d, rows, self.vocab, pos=self.pos, Lseq=L)
        for w in self.layers:
            self._ln(h, n, w["ln1"], rows)
            ops.gemm(n, w["wqkv"], qkv, rows, 3 * d, d, d, d, 3 * d, bias=w["bqkv"], workspace=ws)
            ops.attention_small(qkv, att, 0, d, 2 * d, 3 * d, d, B, self.H, L, 0.125, causal=True)
            ops.gemm(att, w["wo"], h, rows, d, d, d, d, d, bias=w["bo"], res=h, ldr=d, flags=GEMM_GATE_RES, workspace=ws)
            self._ln(h, n, w["ln2"], rows)
            ops.gemm(n, w["w1"], ff, rows, dff, d, d, d, dff, bias=w["b1"], gelu_from=0, flags=self.act_flag, workspace=ws)
            ops.gemm(ff, w["w2"], h, rows, d, dff, dff, dff, d, bias=w["b2"], res=h, ldr=d, flags=GEMM_GATE_RES, workspace=ws)
        self._ln(h, n, self.ln_f, rows)
        pooled = torch.empty(B, d, dtype=torch.bfloat16, device=dev)
        for b, e in enumerate(self.eos_positions(input_ids)):
            ops.copy_rows(n, pooled, 1, d, d, d, 0, 0, 1, dst_off=b * d, src_off=(b * L + e) * d)
        if self.proj is not None:
            n_pad = self.proj.shape[0]
            out = torch.empty(B, n_pad, dtype=torch.bfloat16, device=dev)
            ops.gemm(pooled, self.proj, out, B, n_pad, d, d, d, n_pad)
            pooled = out[:, :self.proj_dim]
        return n.view(B, L, d), pooled

    __call__ = forward


# ------------------------------------------------------------------------------------------------ loading helpers
def _load_dir(path):
    """(state_dict, config dict) of a transformers model directory: config.json + *.safetensors shards."""
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    sd = {}
    shards = sorted(p for p in os.listdir(path) if p.endswith(".safetensors"))
    if not shards:
        raise FileNotFoundError(f"no .safetensors weights under {path}")
    for s in shards:
        sd.update(load_file(os.path.join(path, s)))
    return sd, cfg


def _tokenizers(model_path, names):
    from transformers import CLIPTokenizer, T5TokenizerFast
    out = []
    for n, kind in names:
        cls = CLIPTokenizer if kind == "clip" else T5TokenizerFast
        out.append(cls.from_pretrained(os.path.join(model_path, n)))
    return out


class _TextEncoderBase(DeviceModuleAPI):
    T5_MAX_LEN = 128              # modeling_text_encoder.py:39 (flux) / :43 (mmdit)

    def _t5(self, prompt, device, num_images_per_prompt=1):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        ti = self.t5_tokenizer(prompt, padding="max_length", max_length=self.T5_MAX_LEN, truncation=True,
                               add_special_tokens=True, return_tensors="pt")
        mask = ti.attention_mask.to(device)
        emb = self.t5(ti.input_ids, attention_mask=mask)
        B, L, _ = emb.shape
        emb = emb.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, L, -1)
        mask = mask.view(B, -1).repeat(num_images_per_prompt, 1)
        return emb, mask

    def _clip(self, tokenizer, model, prompt, num_images_per_prompt=1):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        ti = tokenizer(prompt, padding="max_length", max_length=self.tokenizer_max_length, truncation=True,
                       return_tensors="pt")
        _, pooled = model(ti.input_ids)
        B = len(prompt)
        return pooled.repeat(1, num_images_per_prompt).view(B * num_images_per_prompt, -1)

    def forward(self, input_prompts, device):
        return self.encode_prompt(input_prompts, 1, device=device)

    __call__ = forward


class FluxTextEncoderWithMask(_TextEncoderBase):
    """flux_modules/modeling_text_encoder.py:15-134.  ``model_path`` holds tokenizer/, tokenizer_2/, text_encoder/
    (CLIP-L) and text_encoder_2/ (T5); alternatively pass built engines and tokenizers."""

    def __init__(self, model_path=None, torch_dtype=torch.bfloat16, device="cuda", clip=None, t5=None,
                 tokenizer=None, tokenizer_2=None):
        if model_path is not None:
            tokenizer, tokenizer_2 = _tokenizers(model_path, [("tokenizer", "clip"), ("tokenizer_2", "t5")])
            clip = CLIPTextHIP(*_load_dir(os.path.join(model_path, "text_encoder")), device=device)
            t5 = T5EncoderHIP(*_load_dir(os.path.join(model_path, "text_encoder_2")), device=device)
        self.dev = torch.device(getattr(t5, "dev", device))
        self.tokenizer, self.text_encoder = tokenizer, clip
        self.t5_tokenizer, self.t5 = tokenizer_2, t5
        self.tokenizer_2, self.text_encoder_2 = tokenizer_2, t5
        self.tokenizer_max_length = getattr(tokenizer, "model_max_length", 77)

    def encode_prompt(self, prompt, num_images_per_prompt=1, device=None):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        pooled = self._clip(self.tokenizer, self.text_encoder, prompt, num_images_per_prompt)
        emb, mask = self._t5(prompt, device, num_images_per_prompt)
        return emb, mask, pooled


class SD3TextEncoderWithMask(_TextEncoderBase):
    """mmdit_modules/modeling_text_encoder.py:15-139: CLIP-L and CLIP-G projected pooled outputs concatenated, T5 in
    text_encoder_3."""

    def __init__(self, model_path=None, torch_dtype=torch.bfloat16, device="cuda", clip=None, clip_2=None, t5=None,
                 tokenizer=None, tokenizer_2=None, tokenizer_3=None):
        if model_path is not None:
            tokenizer, tokenizer_2, tokenizer_3 = _tokenizers(
                model_path, [("tokenizer", "clip"), ("tokenizer_2", "clip"), ("tokenizer_3", "t5")])
            clip = CLIPTextHIP(*_load_dir(os.path.join(model_path, "text_encoder")), device=device)
            clip_2 = CLIPTextHIP(*_load_dir(os.path.join(model_path, "text_encoder_2")), device=device)
            t5 = T5EncoderHIP(*_load_dir(os.path.join(model_path, "text_encoder_3")), device=device)
        self.dev = torch.device(getattr(t5, "dev", device))
        self.tokenizer, self.text_encoder = tokenizer, clip
        self.tokenizer_2, self.text_encoder_2 = tokenizer_2, clip_2
        self.tokenizer_3, self.text_encoder_3 = tokenizer_3, t5
        self.t5_tokenizer, self.t5 = tokenizer_3, t5
        self.tokenizer_max_length = getattr(tokenizer, "model_max_length", 77)

    def encode_prompt(self, prompt, num_images_per_prompt=1, clip_skip=None, device=None):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        p1 = self._clip(self.tokenizer, self.text_encoder, prompt, num_images_per_prompt)
        p2 = self._clip(self.tokenizer_2, self.text_encoder_2, prompt, num_images_per_prompt)
        pooled = torch.cat([p1, p2], dim=-1)
        emb, mask = self._t5(prompt, device, num_images_per_prompt)
        return emb, mask, pooled
