"""Prompt-encoder timing at production sizes (T5-v1.1-XXL encoder 24 x (d 4096, 64 heads, ff 10240), CLIP-L 12 x 768,
CLIP-G 32 x 1280) with random weights generated on the device; one prompt of 128 / 77 tokens, as generate() issues.
usage: python tools/text_bench.py [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops                                                                    # noqa: E402
from pyflow_hip.text_encoder import T5EncoderHIP, CLIPTextHIP                                 # noqa: E402

T5_XXL = dict(vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
              relative_attention_num_buckets=32, relative_attention_max_distance=128, feed_forward_proj="gated-gelu",
              layer_norm_epsilon=1e-6)
CLIP_L = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
              max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2)
CLIP_G = dict(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
              max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5, eos_token_id=2, projection_dim=1280)


def rnd(shape, std, g):
    return (torch.randn(shape, device="cuda", dtype=torch.float32, generator=g) * std).bfloat16()


def t5_sd(c, g):
    d, inner, ff = c["d_model"], c["num_heads"] * 64, c["d_ff"]
    sd = {"shared.weight": rnd((c["vocab_size"], d), 1.0, g),
          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": rnd((32, c["num_heads"]), 1.0, g),
          "encoder.final_layer_norm.weight": torch.ones(d, device="cuda")}
    for i in range(c["num_layers"]):
        a, f = f"encoder.block.{i}.layer.0.", f"encoder.block.{i}.layer.1."
        for n in "qkv":
            sd[a + f"SelfAttention.{n}.weight"] = rnd((inner, d), d ** -0.5 * (0.35 if n in "qk" else 1.0), g)
        sd[a + "SelfAttention.o.weight"] = rnd((d, inner), inner ** -0.5, g)
        sd[a + "layer_norm.weight"] = torch.ones(d, device="cuda")
        sd[f + "layer_norm.weight"] = torch.ones(d, device="cuda")
        sd[f + "DenseReluDense.wi_0.weight"] = rnd((ff, d), d ** -0.5, g)
        sd[f + "DenseReluDense.wi_1.weight"] = rnd((ff, d), d ** -0.5, g)
        sd[f + "DenseReluDense.wo.weight"] = rnd((d, ff), ff ** -0.5, g)
    return sd


def clip_sd(c, g):
    d, ff = c["hidden_size"], c["intermediate_size"]
    sd = {"embeddings.token_embedding.weight": rnd((c["vocab_size"], d), 0.02, g),
          "embeddings.position_embedding.weight": rnd((77, d), 0.02, g),
          "final_layer_norm.weight": torch.ones(d, device="cuda"), "final_layer_norm.bias": torch.zeros(d, device="cuda")}
    if "projection_dim" in c:
        sd["text_projection.weight"] = rnd((c["projection_dim"], d), d ** -0.5, g)
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{n}.weight"] = rnd((d, d), d ** -0.5, g)
            sd[p + f"self_attn.{n}.bias"] = rnd((d,), 0.02, g)
        for n in ("layer_norm1", "layer_norm2"):
            sd[p + n + ".weight"] = torch.ones(d, device="cuda")
            sd[p + n + ".bias"] = torch.zeros(d, device="cuda")
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rnd((ff, d), d ** -0.5, g), rnd((ff,), 0.02, g)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rnd((d, ff), ff ** -0.5, g), rnd((d,), 0.02, g)
    return sd


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    for name, cfg, mk, cls, L in (("t5_xxl", T5_XXL, t5_sd, T5EncoderHIP, 128), ("clip_l", CLIP_L, clip_sd, CLIPTextHIP, 77),
                                  ("clip_g", CLIP_G, clip_sd, CLIPTextHIP, 77)):
        sd = mk(cfg, g)
        nbytes = sum(v.numel() for k, v in sd.items() if v.ndim == 2 and "embed" not in k and "shared" not in k) * 2
        eng = cls(sd, cfg, "cuda")
        del sd
        ids = torch.randint(3, 30000, (1, L))
        ids[0, 20] = cfg["vocab_size"] - 1
        mask = torch.zeros(1, L, dtype=torch.long)
        mask[0, :21] = 1
        fn = (lambda: eng(ids, attention_mask=mask)) if cls is T5EncoderHIP else (lambda: eng(ids))
        ms = timeit(fn, iters)
        r = fn()
        r = r if torch.is_tensor(r) else r[1]
        assert torch.isfinite(r.float()).all()
        ops.PROFILER.enabled = True
        ops.PROFILER.records = {}
        fn()
        torch.cuda.synchronize()
        prof = {k: round(v["ms_total"], 3) for k, v in ops.PROFILER.summary().items()}
        ops.PROFILER.enabled = False
        out[name] = dict(ms=round(ms, 3), weight_GB=round(nbytes / 1e9, 3), weight_stream_GBps=round(nbytes / ms / 1e6, 1),
                         kernels_ms=prof)
        print(name, out[name], flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
