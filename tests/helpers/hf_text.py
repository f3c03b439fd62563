"""Builders for the prompt-encoder parity tests: tiny transformers T5 / CLIP text models (the third-party code the
reference's text-encoder wrappers call) with seeded, lively, bf16-representable weights -- the oracle for
pyflow_hip.text_encoder.  transformers is part of the image on both boxes; nothing here reads /root/reference."""
import math

import torch


def lively_(model, seed, qk_gain):
    """Overwrite every parameter with seeded values that make attention / norms / biases non-trivial."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            new[k] = v
            continue
        if "relative_attention_bias" in k:
            t = torch.randn(v.shape, generator=g) * 1.5
        elif v.ndim == 2 and ("embed" in k or "shared" in k):
            t = torch.randn(v.shape, generator=g)
        elif v.ndim == 2:
            gain = qk_gain if any(s in k for s in (".q.", ".k.", "q_proj", "k_proj")) else 1.0
            t = torch.randn(v.shape, generator=g) * gain / math.sqrt(v.shape[1])
        elif k.endswith("norm.weight") or "layer_norm" in k and k.endswith("weight") or "layer_norm1.weight" in k \
                or "layer_norm2.weight" in k:
            t = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        else:
            t = 0.1 * torch.randn(v.shape, generator=g)
        new[k] = t.to(torch.bfloat16).float()
    model.load_state_dict(new)
    return model.eval()


def tiny_t5(seed=0, d_model=256, heads=4, d_ff=512, layers=2, vocab=512):
    from transformers import T5Config, T5EncoderModel
    cfg = T5Config(vocab_size=vocab, d_model=d_model, d_kv=64, d_ff=d_ff, num_layers=layers, num_heads=heads,
                   relative_attention_num_buckets=32, relative_attention_max_distance=128,
                   feed_forward_proj="gated-gelu", dropout_rate=0.0)
    return lively_(T5EncoderModel(cfg), seed, 0.6), cfg


def tiny_clip(seed=0, hidden=128, heads=2, inter=256, layers=2, vocab=512, act="quick_gelu", projection=0):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=77, hidden_act=act, eos_token_id=2,
                         bos_token_id=0, pad_token_id=1, projection_dim=projection or hidden)
    cls = CLIPTextModelWithProjection if projection else CLIPTextModel
    return lively_(cls(cfg), seed, 2.0), cfg


class StubTokenizer:
    """Deterministic stand-in for CLIPTokenizer / T5TokenizerFast (no vocab files in the image): hashes characters to
    ids, appends EOS (the largest id, as in CLIP), pads to max_length; returns input_ids / attention_mask tensors."""

    def __init__(self, vocab, model_max_length=77, pad_id=0):
        self.vocab, self.model_max_length, self.pad_id = vocab, model_max_length, pad_id

    def __call__(self, prompts, padding="max_length", max_length=None, truncation=True, add_special_tokens=True,
                 return_tensors="pt"):
        L = max_length or self.model_max_length
        ids = torch.full((len(prompts), L), self.pad_id, dtype=torch.long)
        mask = torch.zeros(len(prompts), L, dtype=torch.long)
        for b, p in enumerate(prompts):
            toks = [3 + (ord(c) * 7 + i) % (self.vocab - 4) for i, c in enumerate(p)][:L - 1] + [self.vocab - 1]
            ids[b, :len(toks)] = torch.tensor(toks)
            mask[b, :len(toks)] = 1

        class Out:
            pass
        o = Out()
        o.input_ids, o.attention_mask = ids, mask
        return o
