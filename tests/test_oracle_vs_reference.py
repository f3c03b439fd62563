"""Pins the oracle restatement against the UNMODIFIED reference imported from /root/reference (dev container only;
skipped on the GPU box), and checks the synthetic key/shape tables against the reference modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def ref():
    from oracle import shims
    return shims.load_reference()


def test_state_dict_layout_matches_reference(ref):
    from pyflow_hip import synth
    m = ref.PyramidFluxTransformer(**synth.TINY_FLUX)
    shapes = synth.flux_param_shapes(synth.TINY_FLUX)
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    v = ref.CausalVideoVAE(encoder_out_channels=16, decoder_in_channels=16,
                           encoder_block_out_channels=(32, 32, 64, 64), decoder_block_out_channels=(32, 32, 64, 64),
                           encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(2, 2, 2, 2))
    vs = synth.vae_decoder_param_shapes(synth.TINY_VAE)
    vsd = {k: t for k, t in v.state_dict().items() if k.startswith(("decoder.", "post_quant_conv."))}
    assert set(vsd) == set(vs)
    assert all(tuple(vsd[k].shape) == tuple(s) for k, s in vs.items())


def test_full_size_param_counts():
    from pyflow_hip import synth
    n = sum(int(np.prod(s)) for s in synth.flux_param_shapes(synth.MINIFLUX).values())
    assert n == 1972059200                     # SURVEY 8a probe of the instantiated reference
    nv = sum(int(np.prod(s)) for s in synth.vae_decoder_param_shapes(synth.VAE_DEFAULT).values())
    assert nv == 225768707 + 16 * 16 + 16      # decoder + post_quant_conv


def test_flux_oracle_bit_exact(ref):
    from oracle import ref_harness as rh
    from oracle.flux_oracle import flux_forward
    m = rh.build_ref_dit()
    g = torch.Generator().manual_seed(5)
    clips = [torch.randn(2, 16, 2, 4, 8, generator=g), torch.randn(2, 16, 1, 8, 16, generator=g),
             torch.randn(2, 16, 1, 16, 32, generator=g), torch.randn(2, 16, 1, 16, 32, generator=g)]
    enc = torch.randn(2, 16, 32, generator=g)
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, 16, generator=g)
    t = torch.tensor([704.262, 704.262])
    with torch.no_grad():
        r = m(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled,
              timestep_ratio=t)[0]
    o = flux_forward(m.state_dict(), rh.TINY_DIT, clips, enc, mask, pooled, t)
    assert (r - o).abs().max().item() < 1e-5
    # the grouped-head evaluation the oracle switches to above L = 8 192 (bounded memory at the headline length,
    # tests/test_fullsize_gpu.py) is the same function: forced here on the tiny sequence, pinned to the reference too
    from oracle import flux_oracle
    keep = flux_oracle.HEAD_CHUNK_ABOVE_L, flux_oracle.HEAD_CHUNK
    flux_oracle.HEAD_CHUNK_ABOVE_L, flux_oracle.HEAD_CHUNK = 0, 1
    try:
        o2 = flux_forward(m.state_dict(), rh.TINY_DIT, clips, enc, mask, pooled, t)
    finally:
        flux_oracle.HEAD_CHUNK_ABOVE_L, flux_oracle.HEAD_CHUNK = keep
    assert (r - o2).abs().max().item() < 1e-5


def test_vae_oracle(ref):
    from oracle import ref_harness as rh
    from oracle.vae_oracle import vae_decode
    v = rh.build_ref_vae()
    z = torch.randn(1, 16, 3, 6, 10, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a = v.decode(z, temporal_chunk=False).sample
        b = v.decode(z, temporal_chunk=True, window_size=1).sample
        o = vae_decode(v.state_dict(), dict(v.config), z)
        assert (a - o).abs().max() < 1e-5 and (b - o).abs().max() < 5e-5
        v.enable_tiling()
        bt = v.decode(z, temporal_chunk=True, window_size=1, tile_sample_min_size=32).sample
        ot = vae_decode(v.state_dict(), dict(v.config), z, use_tiling=True, tile_sample_min_size=32)
        assert (bt - ot).abs().max() < 5e-5


def test_generate_oracle_vs_reference_generate(ref):
    from oracle import ref_harness as rh
    from oracle.pipeline_oracle import generate_latents
    dit, vae = rh.build_ref_dit(), rh.build_ref_vae()
    pipe = rh.build_ref_pipeline(dit, vae)
    rh.patch_block_noise(pipe, rh.NoiseStream(1))
    with torch.no_grad():
        lat_ref = pipe.generate(prompt="a cat", height=64, width=128, temp=3, num_inference_steps=[3, 3, 3],
                                video_num_inference_steps=[2, 2, 2], guidance_scale=7.0, video_guidance_scale=5.0,
                                generator=torch.Generator().manual_seed(0), output_type="latent")
    te = pipe.text_encoder
    pe, pm, pp = te("a cat, hyper quality, Ultra HD, 8K", None)
    neg = ("cartoon style, worst quality, low quality, blurry, absolute black, absolute white, low res, extra limbs, "
           "extra digits, misplaced objects, mutated anatomy, monochrome, horror")
    ne, nm, npool = te(neg, None)
    init = torch.randn((1, 16, 3, 8, 16), generator=torch.Generator().manual_seed(0))
    lat = generate_latents(dit.state_dict(), rh.TINY_DIT, torch.cat([ne, pe]), torch.cat([nm, pm]),
                           torch.cat([npool, pp]), init, rh.NoiseStream(1).block_noise, [3, 3, 3], [2, 2, 2], 7.0, 5.0)
    assert (lat - lat_ref).abs().max() < 1e-4


def test_block_noise_constants_match_torch():
    from oracle.pipeline_oracle import block_noise_cholesky
    g = 1 / 3
    cov = torch.eye(4) * (1 + g) - torch.ones(4, 4) * g
    L = torch.distributions.MultivariateNormal(torch.zeros(4), cov, validate_args=False).scale_tril
    assert torch.equal(L, block_noise_cholesky())
    # each reference draw is randn(4) @ L^T from the global generator
    torch.manual_seed(3)
    d = torch.distributions.MultivariateNormal(torch.zeros(4), cov, validate_args=False).sample()
    torch.manual_seed(3)
    assert torch.equal(d, L @ torch.randn(4)) or torch.allclose(d, torch.randn(4) @ L.T)


TINY_MMDIT = dict(sample_size=32, patch_size=2, in_channels=16, num_layers=3, attention_head_dim=64, num_attention_heads=4,
                  caption_projection_dim=256, pooled_projection_dim=16, pos_embed_max_size=48, joint_attention_dim=32,
                  qk_norm="rms_norm", pos_embed_type="sincos", temp_pos_embed_type="rope", use_flash_attn=False,
                  use_temporal_causal=True, use_t5_mask=True, add_temp_pos_embed=True, interp_condition_pos=True)


def test_mmdit_oracle_vs_reference(ref):
    """SD3-style variant as the pipeline builds it (pipeline.py:80-87): sincos abs-pos (cropped / bilinear for the
    lower-resolution history clips) + temporal RoPE + temporal-causal mask + context_pre_only last block."""
    from oracle import ref_harness as rh
    from oracle.mmdit_oracle import mmdit_forward, sincos_2d_table
    from pyflow_hip import synth
    m = rh.seed_weights(ref.PyramidDiffusionMMDiT(**TINY_MMDIT).eval(), 77)
    sd = m.state_dict()
    # the persistent buffer is the published sincos table
    tab = sincos_2d_table(256, 48, 32 // 2)
    assert (sd["pos_embed.pos_embed"][0] - tab).abs().max() < 1e-6
    # key / shape table of the synthetic-weight generator
    shapes = synth.mmdit_param_shapes(synth.tiny_mmdit_cfg())
    assert set(shapes) == set(sd), (set(shapes) ^ set(sd))
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    g = torch.Generator().manual_seed(5)
    clips = [torch.randn(2, 16, 2, 4, 8, generator=g), torch.randn(2, 16, 1, 8, 16, generator=g),
             torch.randn(2, 16, 1, 16, 32, generator=g), torch.randn(2, 16, 1, 16, 32, generator=g)]
    enc = torch.randn(2, 16, 32, generator=g)
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, 16, generator=g)
    t = torch.tensor([704.262, 704.262])
    with torch.no_grad():
        r = m(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled,
              timestep_ratio=t)[0]
    o = mmdit_forward(sd, TINY_MMDIT, clips, enc, mask, pooled, t)
    assert (r - o).abs().max().item() < 2e-5


def test_vae_encode_oracle(ref):
    """encoder (strided causal convs, single frame and a short clip) + tiled_encode vs the reference."""
    from oracle import ref_harness as rh
    from oracle.vae_oracle import vae_encode_moments, posterior_sample
    v = rh.build_ref_vae()
    cfg = dict(v.config)
    g = torch.Generator().manual_seed(4)
    img = torch.randn(1, 3, 1, 64, 96, generator=g).clamp(-1, 1)
    clip = torch.randn(1, 3, 9, 32, 32, generator=g).clamp(-1, 1)
    with torch.no_grad():
        for x in (img, clip):
            post = v.encode(x).latent_dist
            o = vae_encode_moments(v.state_dict(), cfg, x)
            assert (post.parameters - o).abs().max() < 2e-5
            assert (post.mode() - posterior_sample(o)).abs().max() < 2e-5
        v.enable_tiling()
        post = v.encode(img, tile_sample_min_size=32).latent_dist
        o = vae_encode_moments(v.state_dict(), cfg, img, use_tiling=True, tile_sample_min_size=32)
        assert post.parameters.shape == o.shape
        assert (post.parameters - o).abs().max() < 2e-5


def test_vae_chunk_encode_equals_single_pass(ref):
    """chunk_encode (causal_vae.py:310-341, the sliding window with cache_front_feat) of the reference equals the
    oracle's single causal pass: the restatement the HIP chunked encode is checked against; plain and tiled."""
    from oracle import ref_harness as rh
    from oracle.vae_oracle import vae_encode_moments
    v = rh.build_ref_vae().eval()
    cfg = dict(v.config)
    x = torch.randn(1, 3, 33, 32, 48, generator=torch.Generator().manual_seed(6)).clamp(-1, 1)
    with torch.no_grad():
        for win in (8, 16):                  # 33 frames: 9 + 3 x 8 / 17 + 16
            post = v.encode(x, temporal_chunk=True, window_size=win).latent_dist
            o = vae_encode_moments(v.state_dict(), cfg, x)
            assert post.parameters.shape == o.shape == (1, 32, 5, 4, 6)
            assert (post.parameters - o).abs().max() < 5e-5
        v.enable_tiling()
        post = v.encode(x, temporal_chunk=True, window_size=8, tile_sample_min_size=32).latent_dist
        o = vae_encode_moments(v.state_dict(), cfg, x, use_tiling=True, tile_sample_min_size=32)
        assert (post.parameters - o).abs().max() < 5e-5


def test_vae_encoder_key_table(ref):
    from pyflow_hip import synth
    v = ref.CausalVideoVAE(encoder_out_channels=16, decoder_in_channels=16,
                           encoder_block_out_channels=(32, 32, 64, 64), decoder_block_out_channels=(32, 32, 64, 64),
                           encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(2, 2, 2, 2))
    es = synth.vae_encoder_param_shapes(synth.TINY_VAE_ENC)
    esd = {k: t for k, t in v.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}
    assert set(esd) == set(es), set(esd) ^ set(es)
    assert all(tuple(esd[k].shape) == tuple(s) for k, s in es.items())


def test_generate_i2v_oracle_vs_reference(ref):
    """SD3-style MMDiT + image conditioning: the reference's own generate_i2v (PIL input, torchvision transform stubs,
    global-RNG posterior draw) vs the oracle (encode -> sample -> normalise -> units 1.. with the video schedule)."""
    import numpy as np
    from PIL import Image
    from oracle import ref_harness as rh
    from oracle.mmdit_oracle import mmdit_forward
    from oracle.pipeline_oracle import generate_latents
    from oracle.vae_oracle import vae_encode_moments, posterior_sample
    dit = rh.seed_weights(ref.PyramidDiffusionMMDiT(**TINY_MMDIT).eval(), 77)
    vae = rh.build_ref_vae()
    pipe = rh.build_ref_pipeline(dit, vae)
    pipe.model_name = "pyramid_mmdit"
    pipe.vae_shift_factor, pipe.vae_scale_factor = 0.1490, 1 / 1.8415
    rh.patch_block_noise(pipe, rh.NoiseStream(1))
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 256, (64, 128, 3), dtype=np.uint8))
    torch.manual_seed(123)
    with torch.no_grad():
        lat_ref = pipe.generate_i2v(prompt="a cat", input_image=img, temp=3, num_inference_steps=[2, 2, 2],
                                    guidance_scale=7.0, video_guidance_scale=4.0,
                                    generator=torch.Generator().manual_seed(0), output_type="latent")
    te = pipe.text_encoder
    pe, pm, pp = te("a cat, hyper quality, Ultra HD, 8K", None)
    neg = ("cartoon style, worst quality, low quality, blurry, absolute black, absolute white, low res, extra limbs, "
           "extra digits, misplaced objects, mutated anatomy, monochrome, horror")
    ne, nm, npool = te(neg, None)
    x = (torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
    moments = vae_encode_moments(vae.state_dict(), dict(vae.config), x[None, :, None])
    torch.manual_seed(123)
    eps = torch.randn(1, 16, 1, 8, 16)
    z = (posterior_sample(moments, eps) - 0.1490) * (1 / 1.8415)
    init = torch.randn((1, 16, 3, 8, 16), generator=torch.Generator().manual_seed(0))
    lat = generate_latents(dit.state_dict(), TINY_MMDIT, torch.cat([ne, pe]), torch.cat([nm, pm]), torch.cat([npool, pp]),
                           init, rh.NoiseStream(1).block_noise, None, [2, 2, 2], 7.0, 4.0,
                           forward_fn=mmdit_forward, image_latent=z)
    assert lat.shape == lat_ref.shape == (1, 16, 3, 8, 16)
    assert (lat - lat_ref).abs().max() < 1e-4


def test_text_encoder_wrappers_vs_reference(ref):
    """The composition the prompt-encoder GPU tests compare against (transformers' T5 encoder on the padded ids with the
    tokenizer's mask; CLIP pooler_output / text_embeds; concatenation order) IS what the reference's wrapper classes
    return: FluxTextEncoderWithMask / SD3TextEncoderWithMask instantiated without from_pretrained (tiny seeded
    transformers models + the stub tokenizer), called through their own forward()."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
    from hf_text import tiny_t5, tiny_clip, StubTokenizer
    from pyramid_dit.flux_modules.modeling_text_encoder import FluxTextEncoderWithMask as RefFlux
    from pyramid_dit.mmdit_modules.modeling_text_encoder import SD3TextEncoderWithMask as RefSD3
    t5, t5cfg = tiny_t5(seed=1)
    cl, clcfg = tiny_clip(seed=2)
    clp, _ = tiny_clip(seed=2, projection=128)
    cg, _ = tiny_clip(seed=3, act="gelu", projection=128)

    class Tok(StubTokenizer):                       # the reference passes a few extra tokenizer kwargs
        def __call__(self, prompts, **kw):
            kw = {k: v for k, v in kw.items() if k in ("padding", "max_length", "truncation", "add_special_tokens", "return_tensors")}
            return super().__call__(prompts, **kw)
    tok_c, tok_t = Tok(clcfg.vocab_size, 77), Tok(t5cfg.vocab_size, 128)
    prompts = ["a red panda eating bamboo, hyper quality"]
    ti, ci = tok_t(prompts, max_length=128), tok_c(prompts, max_length=77)
    with torch.no_grad():
        exp_emb = t5(ti.input_ids, attention_mask=ti.attention_mask)[0]

        rf = RefFlux.__new__(RefFlux)
        torch.nn.Module.__init__(rf)
        rf.tokenizer, rf.tokenizer_max_length, rf.text_encoder = tok_c, 77, cl
        rf.tokenizer_2, rf.text_encoder_2 = tok_t, t5
        emb, mask, pooled = rf(prompts, torch.device("cpu"))
        assert torch.equal(emb, exp_emb) and torch.equal(mask, ti.attention_mask)
        assert torch.equal(pooled, cl(ci.input_ids).pooler_output)

        rs = RefSD3.__new__(RefSD3)
        torch.nn.Module.__init__(rs)
        rs.tokenizer, rs.tokenizer_max_length, rs.text_encoder = tok_c, 77, clp
        rs.tokenizer_2, rs.text_encoder_2 = tok_c, cg
        rs.tokenizer_3, rs.text_encoder_3 = tok_t, t5
        emb3, mask3, pooled3 = rs(prompts, torch.device("cpu"))
        assert torch.equal(emb3, exp_emb) and torch.equal(mask3, ti.attention_mask)
        assert torch.equal(pooled3, torch.cat([clp(ci.input_ids).text_embeds, cg(ci.input_ids).text_embeds], -1))


def test_flux_oracle_at_released_width(ref):
    """the oracle against the unmodified reference at the RELEASED miniFLUX width (d = 1920, 30 heads -- a head count
    that is not a power of two), one double + one single block, history clips + padded text: the configuration the GPU
    full-width tests (tests/test_fullwidth_oracle_gpu.py) compare the HIP kernels with"""
    from oracle import ref_harness as rh
    from oracle.flux_oracle import flux_forward
    from pyflow_hip import synth
    cfg = dict(synth.MINIFLUX, num_layers=1, num_single_layers=1)
    m = rh.build_ref_dit(cfg, seed=77)
    g = torch.Generator().manual_seed(9)
    clips = [torch.randn(2, 16, 2, 6, 10, generator=g), torch.randn(2, 16, 1, 12, 20, generator=g),
             torch.randn(2, 16, 1, 12, 20, generator=g)]
    enc = torch.randn(2, 24, 4096, generator=g)
    mask = torch.zeros(2, 24, dtype=torch.long)
    mask[0, :7] = 1
    mask[1, :19] = 1
    pooled = torch.randn(2, 768, generator=g)
    t = torch.tensor([431.5, 431.5])
    with torch.no_grad():
        r = m(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled,
              timestep_ratio=t)[0]
    o = flux_forward(m.state_dict(), cfg, clips, enc, mask, pooled, t)
    assert o.shape == r.shape
    assert (r - o).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())


def test_vae_oracle_at_released_width(ref):
    """the decoder oracle against the reference at the released channel widths (512 / 512 / 256 / 128, three resnets per
    block, 512-channel mid attention) on a small latent"""
    from oracle import ref_harness as rh
    from oracle.vae_oracle import vae_decode
    cfg = dict(encoder_out_channels=16, decoder_in_channels=16,
               encoder_block_out_channels=(32, 32, 64, 64), decoder_block_out_channels=(128, 256, 512, 512),
               encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(3, 3, 3, 3))
    v = rh.build_ref_vae(cfg, seed=55)
    z = torch.randn(1, 16, 2, 6, 8, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        r = v.decode(z, temporal_chunk=False).sample
    ocfg = dict(decoder_block_out_channels=(128, 256, 512, 512), decoder_layers_per_block=(3, 3, 3, 3),
                decoder_spatial_up_sample=(True, True, True, False), decoder_temporal_up_sample=(True, True, True, False))
    o = vae_decode({k: t for k, t in v.state_dict().items() if k.startswith(("decoder.", "post_quant_conv."))}, ocfg, z)
    assert o.shape == r.shape
    assert (r - o).abs().max().item() <= 2e-4 * max(1.0, r.abs().max().item())


def test_mmdit_oracle_at_released_width(ref):
    """the SD3-style variant at its released width (d = 1536, 24 heads), one joint block + the context_pre_only last block"""
    from oracle import ref_harness as rh
    from oracle.mmdit_oracle import mmdit_forward
    from pyflow_hip import synth
    cfg = dict(TINY_MMDIT, **dict(synth.SD3_MMDIT, num_layers=2))
    m = rh.seed_weights(ref.PyramidDiffusionMMDiT(**cfg).eval(), 78)
    g = torch.Generator().manual_seed(6)
    clips = [torch.randn(2, 16, 2, 6, 10, generator=g), torch.randn(2, 16, 1, 12, 20, generator=g),
             torch.randn(2, 16, 1, 12, 20, generator=g)]
    enc = torch.randn(2, 24, 4096, generator=g)
    mask = torch.zeros(2, 24, dtype=torch.long)
    mask[0, :7] = 1
    mask[1, :19] = 1
    pooled = torch.randn(2, 2048, generator=g)
    t = torch.tensor([431.5, 431.5])
    with torch.no_grad():
        r = m(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled,
              timestep_ratio=t)[0]
    o = mmdit_forward(m.state_dict(), cfg, clips, enc, mask, pooled, t)
    assert o.shape == r.shape
    assert (r - o).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())


def test_bf16_trajectory_fixture_is_the_references(ref, tmp_path, monkeypatch):
    """tests/golden/generate_tiny_latents_bf16.pt (what the GPU's `_round = True` trajectory is compared with) is
    reproduced bit for bit by re-running the unmodified reference with a bf16 DiT under CPU bf16 autocast."""
    import os
    from oracle import gen_golden as gg
    committed = torch.load(os.path.join(gg.OUT, "generate_tiny_latents_bf16.pt"))
    monkeypatch.setattr(gg, "OUT", str(tmp_path))
    gg.generate_bf16_fixture()
    fresh = torch.load(os.path.join(str(tmp_path), "generate_tiny_latents_bf16.pt"))
    assert fresh["latents"].dtype == torch.bfloat16
    assert torch.equal(fresh["latents"], committed["latents"])
    assert torch.equal(fresh["prompt_embeds"], committed["prompt_embeds"])


def test_batch_fixture_is_the_references_and_rows_are_independent(ref, tmp_path, monkeypatch):
    """tests/golden/generate_tiny_latents_batch2.pt (two prompts through the reference's generate()) is reproduced bit
    for bit; its two samples differ (two prompts, two rows of one random stream)."""
    import os
    from oracle import gen_golden as gg
    committed = torch.load(os.path.join(gg.OUT, "generate_tiny_latents_batch2.pt"))
    monkeypatch.setattr(gg, "OUT", str(tmp_path))
    gg.generate_batch_fixture()
    fresh = torch.load(os.path.join(str(tmp_path), "generate_tiny_latents_batch2.pt"))
    assert torch.equal(fresh["latents"], committed["latents"])
    assert fresh["latents"].shape[0] == 2 and not torch.equal(fresh["latents"][0], fresh["latents"][1])


def test_eight_unit_fixture_is_the_references_and_the_oracle_follows_it(ref, tmp_path, monkeypatch):
    """tests/golden/generate_tiny_latents_8units.pt (round 6: eight autoregressive units, fp32 and the production bf16 form)
    is reproduced bit for bit by re-running the unmodified reference, and the oracle's own host loop
    (oracle/pipeline_oracle.py) follows the fp32 trajectory through all eight units."""
    import os
    from oracle import gen_golden as gg
    from oracle.pipeline_oracle import generate_latents
    from oracle.ref_harness import NoiseStream
    committed = torch.load(os.path.join(gg.OUT, "generate_tiny_latents_8units.pt"))
    monkeypatch.setattr(gg, "OUT", str(tmp_path))
    gg.generate_long_fixture()
    fresh = torch.load(os.path.join(str(tmp_path), "generate_tiny_latents_8units.pt"))
    for k in ("fp32", "bf16"):
        assert torch.equal(fresh[k]["latents"], committed[k]["latents"]) and fresh[k]["latents"].shape[2] == 8
    g = committed
    init = torch.randn((1, 16, g["temp"], g["height"] // 8, g["width"] // 8), generator=torch.Generator().manual_seed(g["latent_seed"]))
    f = g["fp32"]
    o = generate_latents(gg.synth_dit_sd(), g["dit_cfg"], f["prompt_embeds"], f["prompt_mask"], f["pooled"], init,
                         NoiseStream(g["noise_seed"]).block_noise, g["steps"], g["video_steps"], g["guidance"], g["video_guidance"])
    per_unit = [((o[:, :, u] - f["latents"][:, :, u]).norm() / f["latents"][:, :, u].norm()).item() for u in range(8)]
    assert max(per_unit) < 1e-3, per_unit


def test_reference_cfg_pair_equals_two_single_row_forwards(ref):
    """The premise of guidance parallelism (pyflow_hip/flux_cfg.py: one classifier-free-guidance branch per rank): in the
    UNMODIFIED reference the two rows of the `torch.cat([latents] * 2)` forward (pyramid_dit_for_video_gen_pipeline.py:747-776)
    do not interact -- every op of flux:392-542 is per sample, the padded-text mask included -- so row r of the batch-of-2
    forward equals the batch-of-1 forward of prompt row r (fp32, CPU; up to the summation order of batched matmuls)."""
    from oracle import ref_harness as rh
    m = rh.build_ref_dit()
    g = torch.Generator().manual_seed(11)
    one = [torch.randn(1, 16, 2, 4, 8, generator=g), torch.randn(1, 16, 1, 8, 16, generator=g),
           torch.randn(1, 16, 1, 16, 32, generator=g), torch.randn(1, 16, 1, 16, 32, generator=g)]
    clips = [c.repeat(2, 1, 1, 1, 1) for c in one]                      # the CFG duplicate of one latent
    enc = torch.randn(2, 16, 32, generator=g)
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, 16, generator=g)
    t = torch.tensor([512.0, 512.0])
    with torch.no_grad():
        both = m(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled,
                 timestep_ratio=t)[0]
        for r in range(2):
            alone = m(sample=[[c.clone() for c in one]], encoder_hidden_states=enc[r:r + 1], encoder_attention_mask=mask[r:r + 1],
                      pooled_projections=pooled[r:r + 1], timestep_ratio=t[r:r + 1])[0]
            assert alone.shape == both[r:r + 1].shape
            assert (alone - both[r:r + 1]).abs().max().item() < 2e-5 * max(1.0, both.abs().max().item())
    assert (both[0] - both[1]).abs().max().item() > 1e-3                # the two prompts do give two different velocities
