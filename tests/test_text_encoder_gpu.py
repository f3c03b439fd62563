"""GPU parity of the prompt encoders: the HIP path in bf16 vs transformers' T5EncoderModel /
CLIPTextModel(WithProjection) in CPU fp32 on the same bf16-rounded weights and token ids.  Tolerance: rel-L2 <= 2e-2
per encoder output (same bar as one DiT forward); single kernels <= 1e-2."""
import os
import sys

import pytest
import torch

from util import rel_l2, bf16_round

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
pytestmark = pytest.mark.gpu


def test_rmsnorm_glu_embed():
    from pyflow_hip import ops
    g = torch.Generator().manual_seed(0)
    for D, rows in [(256, 7), (1536, 130), (4096, 33)]:
        x = bf16_round(torch.randn(rows, D, generator=g) * 3)
        w = 1 + 0.3 * torch.randn(D, generator=g)
        ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
        y = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
        ops.rmsnorm(x.cuda().bfloat16(), y, w.cuda(), D, rows)
        assert rel_l2(y.float().cpu(), ref) < 4e-3
    x = bf16_round(torch.randn(37, 2 * 640, generator=g))
    y = torch.empty(37, 640, dtype=torch.bfloat16, device="cuda")
    ops.glu_mul(x.cuda().bfloat16(), y, 37, 640)
    assert torch.equal(y.cpu(), (x[:, :640] * x[:, 640:]).bfloat16())
    table = bf16_round(torch.randn(100, 128, generator=g))
    pos = bf16_round(torch.randn(10, 128, generator=g))
    ids = torch.randint(0, 100, (30,), generator=g)
    out = torch.empty(30, 128, dtype=torch.bfloat16, device="cuda")
    ops.embed_rows(table.cuda().bfloat16(), ids.int().cuda(), out, 128, 30, 100)
    assert torch.equal(out.cpu().float(), table[ids])
    ops.embed_rows(table.cuda().bfloat16(), ids.int().cuda(), out, 128, 30, 100, pos=pos.cuda().bfloat16(), Lseq=10)
    assert torch.equal(out.cpu(), (table[ids] + pos[torch.arange(30) % 10]).bfloat16())


@pytest.mark.parametrize("L,H,B,mode", [(128, 4, 2, "t5"), (77, 2, 3, "clip"), (200, 3, 1, "t5"), (5, 1, 1, "clip"),
                                         (160, 2, 2, "clip")])
def test_attention_small(L, H, B, mode):
    from pyflow_hip import ops
    g = torch.Generator().manual_seed(L + H)
    d = H * 64
    qkv = bf16_round(torch.randn(B, L, 3 * d, generator=g))
    bias = mask = None
    scale = 1.0 if mode == "t5" else 0.125
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mode == "t5":
        qkv[..., :d] *= 0.3
        q = qkv[..., :d].view(B, L, H, 64).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2))
        bias = torch.randn(H, L, L, generator=g) * 2
        mask = torch.ones(B, L, dtype=torch.int32)
        mask[0, L // 2:] = 0
        if B > 1:
            mask[1, 3] = 0                                   # a hole, not only right padding
        s = s + bias[None] + torch.where(mask[:, None, None, :] > 0, 0.0, float("-inf"))
    else:
        s = s + torch.full((L, L), float("-inf")).triu(1)
    ref = torch.matmul(s.softmax(-1), v).transpose(1, 2).reshape(B, L, d)
    O = torch.empty(B * L, d, dtype=torch.bfloat16, device="cuda")
    ops.attention_small(qkv.cuda().bfloat16().view(B * L, 3 * d), O, 0, d, 2 * d, 3 * d, d, B, H, L, scale,
                        bias=None if bias is None else bias.cuda(), key_mask=None if mask is None else mask.cuda(),
                        causal=(mode == "clip"))
    assert rel_l2(O.float().cpu().view(B, L, d), ref) < 6e-3


@pytest.mark.parametrize("flag_name,fn", [("GEMM_ACT_QUICK_GELU", lambda x: x * torch.sigmoid(1.702 * x)),
                                           ("GEMM_ACT_GELU_ERF", lambda x: torch.nn.functional.gelu(x))])
def test_gemm_clip_activations(flag_name, fn):
    from pyflow_hip import ops, lib
    g = torch.Generator().manual_seed(1)
    M, N, K = 77, 256, 128
    A = bf16_round(torch.randn(M, K, generator=g))
    W = bf16_round(torch.randn(N, K, generator=g) * 0.2)
    b = torch.randn(N, generator=g)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(A.cuda().bfloat16(), W.cuda().bfloat16(), out, M, N, K, K, K, N, bias=b.cuda(), gelu_from=0,
             flags=getattr(lib, flag_name))
    assert rel_l2(out.float().cpu(), fn(A @ W.t() + b)) < 4e-3


def _ids(B, L, vocab, seed, valid):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab - 1, (B, L), generator=g)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(valid):
        ids[b, n - 1] = vocab - 1          # EOS = largest id
        ids[b, n:] = 0
        mask[b, :n] = 1
    return ids, mask


@pytest.mark.parametrize("L,valid", [(128, [9, 128]), (32, [32, 1])])
def test_t5_encoder_vs_transformers(L, valid):
    from hf_text import tiny_t5
    from pyflow_hip.text_encoder import T5EncoderHIP
    model, cfg = tiny_t5(seed=3)
    ids, mask = _ids(2, L, cfg.vocab_size, 5, valid)
    with torch.no_grad():
        ref = model(ids, attention_mask=mask)[0]
    eng = T5EncoderHIP(model.state_dict(), cfg, "cuda")
    out = eng(ids, attention_mask=mask).float().cpu()
    assert out.shape == ref.shape
    for b, n in enumerate(valid):          # the rows the DiT can see (padded keys are masked there) ...
        assert rel_l2(out[b, :n], ref[b, :n]) < 2e-2
    assert rel_l2(out, ref) < 2e-2         # ... and the padded rows, computed the same way as transformers does


def test_t5_encoder_deeper_wider():
    """closer to T5-XXL proportions: 64-dim heads with inner != d_model, 6 layers"""
    from hf_text import tiny_t5
    from pyflow_hip.text_encoder import T5EncoderHIP
    model, cfg = tiny_t5(seed=11, d_model=384, heads=8, d_ff=1280, layers=6)
    ids, mask = _ids(1, 128, cfg.vocab_size, 6, [40])
    import copy
    with torch.no_grad():
        ref = model(ids, attention_mask=mask)[0]
        # the reference executes the encoder in bf16 (torch_dtype): its own distance to fp32 is the noise floor at depth
        floor = rel_l2(copy.deepcopy(model).bfloat16()(ids, attention_mask=mask)[0].float(), ref)
    out = T5EncoderHIP(model.state_dict(), cfg, "cuda")(ids, attention_mask=mask).float().cpu()
    err = rel_l2(out, ref)
    print("6-layer T5: HIP vs fp32", err, " transformers bf16 vs fp32", floor)
    assert err < max(2e-2, floor)


@pytest.mark.parametrize("act,projection", [("quick_gelu", 0), ("gelu", 256), ("quick_gelu", 64)])
def test_clip_text_vs_transformers(act, projection):
    from hf_text import tiny_clip
    from pyflow_hip.text_encoder import CLIPTextHIP
    model, cfg = tiny_clip(seed=4, act=act, projection=projection, layers=3)
    ids, _ = _ids(2, 77, cfg.vocab_size, 8, [12, 77])
    with torch.no_grad():
        o = model(ids)
    eng = CLIPTextHIP(model.state_dict(), cfg, "cuda")
    assert eng.eos_positions(ids) == [11, 76]
    last, pooled = eng(ids)
    assert rel_l2(last.float().cpu(), o.last_hidden_state) < 2e-2
    ref_pooled = o.text_embeds if projection else o.pooler_output
    assert pooled.shape == ref_pooled.shape
    assert rel_l2(pooled.float().cpu(), ref_pooled) < 2e-2


def test_flux_and_sd3_wrappers(tmp_path):
    """The reference's wrapper contract: (prompt_embeds [B,128,C] bf16, mask [B,128] int64, pooled [B,Cp]) -- and the
    loaders: engines built from save_pretrained() directories (config.json + safetensors)."""
    from hf_text import tiny_t5, tiny_clip, StubTokenizer
    from pyflow_hip.text_encoder import (FluxTextEncoderWithMask, SD3TextEncoderWithMask, T5EncoderHIP, CLIPTextHIP,
                                         _load_dir)
    t5, t5cfg = tiny_t5(seed=1)
    cl, clcfg = tiny_clip(seed=2)
    cg, cgcfg = tiny_clip(seed=3, act="gelu", projection=128)
    clp, clpcfg = tiny_clip(seed=2, projection=128)
    for m, n in ((t5, "t5"), (cl, "clip_l"), (cg, "clip_g")):
        m.save_pretrained(tmp_path / n, safe_serialization=True)
    tok_c, tok_t = StubTokenizer(clcfg.vocab_size, 77), StubTokenizer(t5cfg.vocab_size, 128)
    prompts = ["a red panda eating bamboo, hyper quality"]
    enc = FluxTextEncoderWithMask(clip=CLIPTextHIP(*_load_dir(str(tmp_path / "clip_l")), device="cuda"),
                                  t5=T5EncoderHIP(*_load_dir(str(tmp_path / "t5")), device="cuda"),
                                  tokenizer=tok_c, tokenizer_2=tok_t)
    emb, mask, pooled = enc(prompts, "cuda")
    ti = tok_t(prompts, max_length=128)
    ci = tok_c(prompts, max_length=77)
    with torch.no_grad():
        ref_emb = t5(ti.input_ids, attention_mask=ti.attention_mask)[0]
        ref_pool = cl(ci.input_ids).pooler_output
    assert emb.shape == (1, 128, t5cfg.d_model) and emb.dtype == torch.bfloat16
    assert mask.dtype == torch.int64 and mask.is_cuda and torch.equal(mask.cpu(), ti.attention_mask)
    assert pooled.shape == (1, clcfg.hidden_size)
    assert rel_l2(emb.float().cpu(), ref_emb) < 2e-2 and rel_l2(pooled.float().cpu(), ref_pool) < 2e-2

    sd3 = SD3TextEncoderWithMask(clip=CLIPTextHIP(clp.state_dict(), clpcfg, "cuda"),
                                 clip_2=CLIPTextHIP(*_load_dir(str(tmp_path / "clip_g")), device="cuda"), t5=enc.t5,
                                 tokenizer=tok_c, tokenizer_2=tok_c, tokenizer_3=tok_t)
    emb3, mask3, pooled3 = sd3(prompts, "cuda")
    with torch.no_grad():
        ref3 = torch.cat([clp(ci.input_ids).text_embeds, cg(ci.input_ids).text_embeds], -1)
    assert torch.equal(emb3, emb) and torch.equal(mask3, mask)
    assert pooled3.shape == ref3.shape and rel_l2(pooled3.float().cpu(), ref3) < 2e-2


def test_pipeline_with_on_device_text_encoder():
    """generate() driven by a text prompt end to end on the device path (tiny DiT + tiny encoders), against the same
    run fed with transformers' CPU fp32 embeddings of the same token ids."""
    from hf_text import tiny_t5, tiny_clip, StubTokenizer
    from pyflow_hip.text_encoder import FluxTextEncoderWithMask, T5EncoderHIP, CLIPTextHIP
    from pyflow_hip.pipeline import PyramidDiTForVideoGeneration
    from pyflow_hip import synth
    cfg = dict(synth.TINY_FLUX, joint_attention_dim=256, pooled_projection_dim=128)
    t5, t5cfg = tiny_t5(seed=1)
    cl, clcfg = tiny_clip(seed=2)
    tok_c, tok_t = StubTokenizer(clcfg.vocab_size, 77), StubTokenizer(t5cfg.vocab_size, 128)
    enc = FluxTextEncoderWithMask(clip=CLIPTextHIP(cl.state_dict(), clcfg, "cuda"),
                                  t5=T5EncoderHIP(t5.state_dict(), t5cfg, "cuda"), tokenizer=tok_c, tokenizer_2=tok_t)
    sd = synth.random_state_dict(synth.flux_param_shapes(cfg), seed=3, std=0.05, lively=True)
    pipe = PyramidDiTForVideoGeneration(model_name="pyramid_flux", dit_state_dict=sd, dit_config=cfg, load_vae=False,
                                        text_encoder=enc)
    kw = dict(height=64, width=64, temp=2, num_inference_steps=[2, 2, 2], video_num_inference_steps=[1, 1, 1],
              guidance_scale=7.0, video_guidance_scale=5.0, output_type="latent")
    neg = "blurry, low quality"
    torch.manual_seed(11)                      # the stage-transition block noise is drawn from the global RNG
    out = pipe.generate(prompt="a corgi surfing", negative_prompt=neg, generator=torch.Generator().manual_seed(5), **kw)
    assert torch.isfinite(out.float()).all()

    def hf(prompt):
        ti, ci = tok_t([prompt], max_length=128), tok_c([prompt], max_length=77)
        with torch.no_grad():
            return (t5(ti.input_ids, attention_mask=ti.attention_mask)[0].bfloat16(), ti.attention_mask,
                    cl(ci.input_ids).pooler_output.bfloat16())
    embeds = tuple(t.cuda() for t in hf("a corgi surfing, hyper quality, Ultra HD, 8K") + hf(neg))
    torch.manual_seed(11)
    ref = pipe.generate(prompt_embeds=embeds, generator=torch.Generator().manual_seed(5), **kw)
    err = rel_l2(out.float().cpu(), ref.float().cpu())
    print("latents, on-device prompt encoders vs transformers embeddings:", err)
    assert err < 5e-2
