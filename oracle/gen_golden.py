"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.pt by running the UNMODIFIED reference
(/root/reference, CPU fp32) on seeded tiny models.  Run in the dev container:
    python -m oracle.gen_golden
Fixtures are small (bf16-rounded weights stored as bf16) and committed; the GPU box has no reference.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))

from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NEG = ("cartoon style, worst quality, low quality, blurry, absolute black, absolute white, low res, extra limbs, "
       "extra digits, misplaced objects, mutated anatomy, monochrome, horror")


from pyflow_hip import synth  # noqa: E402  (host-side shape tables only; no HIP involved)

DIT_SEED, VAE_SEED = 3, 5
VAE_CFG_REF = dict(encoder_out_channels=16, decoder_in_channels=16,
                   encoder_block_out_channels=(32, 32, 64, 64), decoder_block_out_channels=(32, 32, 64, 64),
                   encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(2, 2, 2, 2))


def synth_dit_sd():
    sd = synth.random_state_dict(synth.flux_param_shapes(synth.TINY_FLUX), seed=DIT_SEED, std=0.05, lively=True)
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def synth_vae_sd():
    sd = synth.random_state_dict(synth.vae_decoder_param_shapes(synth.TINY_VAE), seed=VAE_SEED, std=0.05, lively=True)
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def build_dit():
    """reference module carrying the seeded synthetic weights (weights are re-derivable from the seed,
    so fixtures only store inputs and outputs)."""
    ref = rh.shims.load_reference()
    m = ref.PyramidFluxTransformer(**synth.TINY_FLUX).eval()
    m.load_state_dict(synth_dit_sd(), strict=True)
    return m


def build_vae():
    ref = rh.shims.load_reference()
    v = ref.CausalVideoVAE(**VAE_CFG_REF).eval()
    missing, unexpected = v.load_state_dict(synth_vae_sd(), strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing)
    return v


def flux_forward_fixture():
    dit = build_dit()
    g = torch.Generator().manual_seed(11)
    B = 2
    shapes = [(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)]
    clips = [torch.randn(B, 16, *s, generator=g).to(torch.bfloat16).float() for s in shapes]
    enc = torch.randn(B, 16, 32, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(B, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(B, 16, generator=g)
    t = torch.tensor([704.0, 704.0])
    with torch.no_grad():
        out = dit(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask,
                  pooled_projections=pooled, timestep_ratio=t)[0]
    torch.save(dict(cfg=synth.TINY_FLUX, weight_seed=DIT_SEED,
                    clips=clips, enc=enc, mask=mask, pooled=pooled, timestep=t, out=out),
               os.path.join(OUT, "flux_tiny_forward.pt"))


MMDIT_SEED = 9
MMDIT_REF_KW = dict(qk_norm="rms_norm", pos_embed_type="sincos", temp_pos_embed_type="rope", use_flash_attn=False,
                    use_temporal_causal=True, use_t5_mask=True, add_temp_pos_embed=True, interp_condition_pos=True)


def mmdit_forward_fixture():
    """SD3-style variant as the pipeline configures it (pyramid_dit_for_video_gen_pipeline.py:80-87)."""
    ref = rh.shims.load_reference()
    cfg = synth.tiny_mmdit_cfg()
    m = ref.PyramidDiffusionMMDiT(**cfg, **MMDIT_REF_KW).eval()
    sd = {k: v.to(torch.bfloat16).float() for k, v in synth.mmdit_state_dict(cfg, seed=MMDIT_SEED, std=0.05, lively=True).items()}
    sd["pos_embed.pos_embed"] = synth.mmdit_state_dict(cfg, seed=MMDIT_SEED)["pos_embed.pos_embed"]      # table stays fp32
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(13)
    B = 2
    shapes = [(2, 4, 8), (1, 8, 16), (1, 16, 32), (1, 16, 32)]
    clips = [torch.randn(B, 16, *s, generator=g).to(torch.bfloat16).float() for s in shapes]
    enc = torch.randn(B, 16, 32, generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(B, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(B, 16, generator=g)
    t = torch.tensor([704.0, 704.0])
    with torch.no_grad():
        out = m(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask,
                pooled_projections=pooled, timestep_ratio=t)[0]
    torch.save(dict(cfg=cfg, weight_seed=MMDIT_SEED, clips=clips, enc=enc, mask=mask, pooled=pooled, timestep=t, out=out),
               os.path.join(OUT, "mmdit_tiny_forward.pt"))


ENC_SEED = 6


def synth_vae_full_sd():
    """decoder + encoder synthetic weights (bf16-rounded)"""
    sd = dict(synth_vae_sd())
    enc = synth.random_state_dict(synth.vae_encoder_param_shapes(synth.TINY_VAE_ENC), seed=ENC_SEED, std=0.05, lively=True)
    sd.update({k: v.to(torch.bfloat16).float() for k, v in enc.items()})
    return sd


def build_vae_full():
    ref = rh.shims.load_reference()
    v = ref.CausalVideoVAE(**VAE_CFG_REF).eval()
    v.load_state_dict(synth_vae_full_sd(), strict=True)
    return v


def build_mmdit():
    ref = rh.shims.load_reference()
    cfg = synth.tiny_mmdit_cfg()
    m = ref.PyramidDiffusionMMDiT(**cfg, **MMDIT_REF_KW).eval()
    sd = {k: v.to(torch.bfloat16).float() for k, v in synth.mmdit_state_dict(cfg, seed=MMDIT_SEED, std=0.05, lively=True).items()}
    sd["pos_embed.pos_embed"] = synth.mmdit_state_dict(cfg, seed=MMDIT_SEED)["pos_embed.pos_embed"]
    m.load_state_dict(sd, strict=True)
    return m, cfg


def i2v_fixture():
    """VAE encode (plain + tiled) of a seeded image and the reference's generate_i2v (MMDiT variant) on it."""
    import numpy as np
    from PIL import Image
    vae = build_vae_full()
    dit, dcfg = build_mmdit()
    rng = np.random.RandomState(3)
    # smooth-ish image: low-res noise upsampled, so the tiled blend regions carry structure
    base = torch.from_numpy(rng.rand(1, 3, 8, 16).astype("float32"))
    arr = (torch.nn.functional.interpolate(base, size=(64, 128), mode="bilinear")[0].permute(1, 2, 0) * 255).round().byte().numpy()
    img = Image.fromarray(arr)
    x = (torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
    with torch.no_grad():
        mom = vae.encode(x[None, :, None]).latent_dist.parameters
        vae.enable_tiling()
        mom_t = vae.encode(x[None, :, None], tile_sample_min_size=32).latent_dist.parameters
        vae.disable_tiling()
    pipe = rh.build_ref_pipeline(dit, vae)
    pipe.model_name = "pyramid_mmdit"
    pipe.vae_shift_factor, pipe.vae_scale_factor = 0.1490, 1 / 1.8415
    rh.patch_block_noise(pipe, rh.NoiseStream(1))
    torch.manual_seed(321)
    with torch.no_grad():
        lat = pipe.generate_i2v(prompt="a cat", input_image=img, temp=3, num_inference_steps=[2, 2, 2], guidance_scale=7.0,
                                video_guidance_scale=4.0, generator=torch.Generator().manual_seed(0), output_type="latent")
    torch.manual_seed(321)
    eps = torch.randn(1, 16, 1, 8, 16)             # the global-RNG draw of latent_dist.sample() inside generate_i2v
    te = pipe.text_encoder
    pe, pm, pp = te("a cat, hyper quality, Ultra HD, 8K", None)
    ne, nm, npool = te(NEG, None)
    torch.save(dict(dit_cfg=dcfg, dit_weight_seed=MMDIT_SEED, vae_cfg=synth.TINY_VAE, vae_enc_cfg=synth.TINY_VAE_ENC,
                    vae_weight_seed=VAE_SEED, enc_weight_seed=ENC_SEED, image=torch.from_numpy(arr.copy()),
                    moments=mom.to(torch.bfloat16), moments_tiled32=mom_t.to(torch.bfloat16), posterior_eps=eps,
                    prompt_embeds=torch.cat([ne, pe]), prompt_mask=torch.cat([nm, pm]), pooled=torch.cat([npool, pp]),
                    temp=3, steps=[2, 2, 2], guidance=7.0, video_guidance=4.0, latent_seed=0, noise_seed=1, latents=lat),
               os.path.join(OUT, "i2v_tiny.pt"))


def vae_fixture():
    vae = build_vae()
    g = torch.Generator().manual_seed(12)
    z = torch.randn(1, 16, 3, 6, 10, generator=g)
    with torch.no_grad():
        out = vae.decode(z, temporal_chunk=True, window_size=1).sample
        vae.enable_tiling()
        out_t = vae.decode(z, temporal_chunk=True, window_size=1, tile_sample_min_size=32).sample
    torch.save(dict(cfg=synth.TINY_VAE, weight_seed=VAE_SEED, z=z, out=out.to(torch.bfloat16),
                    out_tiled32=out_t.to(torch.bfloat16)), os.path.join(OUT, "vae_tiny_decode.pt"))


def generate_fixture():
    dit = build_dit()
    vae = build_vae()
    pipe = rh.build_ref_pipeline(dit, vae)
    rh.patch_block_noise(pipe, rh.NoiseStream(1))
    H, W, temp = 64, 128, 3
    with torch.no_grad():
        lat = pipe.generate(prompt="a cat", height=H, width=W, temp=temp, num_inference_steps=[3, 3, 3],
                            video_num_inference_steps=[2, 2, 2], guidance_scale=7.0, video_guidance_scale=5.0,
                            generator=torch.Generator().manual_seed(0), output_type="latent")
    te = pipe.text_encoder
    pe, pm, pp = te("a cat, hyper quality, Ultra HD, 8K", None)
    ne, nm, npool = te(NEG, None)
    torch.save(dict(dit_cfg=synth.TINY_FLUX, vae_cfg=synth.TINY_VAE, dit_weight_seed=DIT_SEED, vae_weight_seed=VAE_SEED,
                    prompt_embeds=torch.cat([ne, pe]), prompt_mask=torch.cat([nm, pm]), pooled=torch.cat([npool, pp]),
                    height=H, width=W, temp=temp, steps=[3, 3, 3], video_steps=[2, 2, 2], guidance=7.0,
                    video_guidance=5.0, latent_seed=0, noise_seed=1, latents=lat),
               os.path.join(OUT, "generate_tiny_latents.pt"))


class _Bf16TextEncoder:
    """the harness's stub prompt encoder with bf16 outputs, as the reference's encoders return under `model_dtype='bf16'`"""

    def __init__(self):
        self.inner = rh.StubTextEncoder()

    def to(self, *_a, **_k):
        return self

    def __call__(self, prompt, device):
        e, m, p = self.inner(prompt, device)
        return e.to(torch.bfloat16), m, p.to(torch.bfloat16)


def generate_bf16_fixture():
    """The trajectory production runs (`_round = True`): the UNMODIFIED reference's generate() with the DiT in bf16, bf16
    prompt embeddings (so the latents are bf16: pyramid_dit_for_video_gen_pipeline.py:1100) under CPU bf16 autocast --
    the reference's own inference setup (inference_multigpu.py:60-64: `torch.cuda.amp.autocast(dtype=bf16)`) moved to
    the CPU.  Every rounding point of the host loop (CFG combine, Euler step with the 0-dim float64 sigma difference,
    renoise, both pyramids, the bf16-rounded timestep) is therefore the reference's own."""
    dit = build_dit().to(torch.bfloat16)
    vae = build_vae()
    pipe = rh.build_ref_pipeline(dit, vae, text_encoder=_Bf16TextEncoder())
    rh.patch_block_noise(pipe, rh.NoiseStream(1))
    H, W, temp = 64, 128, 4
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        lat = pipe.generate(prompt="a cat", height=H, width=W, temp=temp, num_inference_steps=[3, 3, 3],
                            video_num_inference_steps=[2, 2, 2], guidance_scale=7.0, video_guidance_scale=5.0,
                            generator=torch.Generator().manual_seed(0), output_type="latent")
    assert lat.dtype == torch.bfloat16
    te = pipe.text_encoder
    pe, pm, pp = te("a cat, hyper quality, Ultra HD, 8K", None)
    ne, nm, npool = te(NEG, None)
    torch.save(dict(dit_cfg=synth.TINY_FLUX, dit_weight_seed=DIT_SEED,
                    prompt_embeds=torch.cat([ne, pe]), prompt_mask=torch.cat([nm, pm]), pooled=torch.cat([npool, pp]),
                    height=H, width=W, temp=temp, steps=[3, 3, 3], video_steps=[2, 2, 2], guidance=7.0,
                    video_guidance=5.0, latent_seed=0, noise_seed=1, latents=lat),
               os.path.join(OUT, "generate_tiny_latents_bf16.pt"))


def generate_long_fixture():
    """EIGHT units (temp = 8 -> 57 frames) through the UNMODIFIED reference: the autoregressive history (the previous
    units' latents re-noised down the pyramid, pipeline.py:1112-1186) is seven units deep, so an error of one unit is fed
    back seven times -- the longest trajectory compared before round 6 had three.  Both the fp32 run and the production
    form (bf16 DiT + bf16 embeddings under CPU bf16 autocast, see generate_bf16_fixture) are stored."""
    H, W, temp = 64, 128, 8
    out = {}
    for name in ("fp32", "bf16"):
        dit = build_dit()
        if name == "bf16":
            dit = dit.to(torch.bfloat16)
        pipe = rh.build_ref_pipeline(dit, build_vae(), **({"text_encoder": _Bf16TextEncoder()} if name == "bf16" else {}))
        rh.patch_block_noise(pipe, rh.NoiseStream(1))
        import contextlib
        with torch.no_grad(), (torch.autocast("cpu", dtype=torch.bfloat16) if name == "bf16" else contextlib.nullcontext()):
            lat = pipe.generate(prompt="a cat", height=H, width=W, temp=temp, num_inference_steps=[2, 2, 2],
                                video_num_inference_steps=[2, 2, 2], guidance_scale=7.0, video_guidance_scale=5.0,
                                generator=torch.Generator().manual_seed(0), output_type="latent")
        te = pipe.text_encoder
        pe, pm, pp = te("a cat, hyper quality, Ultra HD, 8K", None)
        ne, nm, npool = te(NEG, None)
        out[name] = dict(prompt_embeds=torch.cat([ne, pe]), prompt_mask=torch.cat([nm, pm]), pooled=torch.cat([npool, pp]),
                         latents=lat)
    assert out["bf16"]["latents"].dtype == torch.bfloat16 and out["fp32"]["latents"].shape[2] == temp
    torch.save(dict(dit_cfg=synth.TINY_FLUX, dit_weight_seed=DIT_SEED, height=H, width=W, temp=temp, steps=[2, 2, 2],
                    video_steps=[2, 2, 2], guidance=7.0, video_guidance=5.0, latent_seed=0, noise_seed=1, **out),
               os.path.join(OUT, "generate_tiny_latents_8units.pt"))


class _BatchTextEncoder:
    """the harness's stub prompt encoder over a LIST of prompts: one row per prompt (what the reference's encoders return)"""

    def __init__(self):
        self.inner = rh.StubTextEncoder()

    def to(self, *_a, **_k):
        return self

    def __call__(self, prompt, device):
        rows = [self.inner(p_, device) for p_ in ([prompt] if isinstance(prompt, str) else prompt)]
        return tuple(torch.cat([r[i] for r in rows]) for i in range(3))


def generate_batch_fixture():
    """A batch of two prompts through the UNMODIFIED reference's generate() (pyramid_dit_for_video_gen_pipeline.py:1049-1053:
    batch_size = len(prompt); one negative prompt per prompt, which is what its [negative | positive] concatenation
    needs): the latents and every block-noise draw have batch shape, so the two samples share one random stream."""
    dit = build_dit()
    vae = build_vae()
    pipe = rh.build_ref_pipeline(dit, vae, text_encoder=_BatchTextEncoder())
    rh.patch_block_noise(pipe, rh.NoiseStream(1))
    H, W, temp = 64, 128, 2
    prompts = ["a cat", "a red fox in the snow"]
    with torch.no_grad():
        lat = pipe.generate(prompt=prompts, negative_prompt=[NEG, NEG], height=H, width=W, temp=temp,
                            num_inference_steps=[2, 2, 2], video_num_inference_steps=[2, 2, 2], guidance_scale=7.0,
                            video_guidance_scale=5.0, generator=torch.Generator().manual_seed(0), output_type="latent")
    assert lat.shape[0] == 2
    te = pipe.text_encoder
    pe, pm, pp = te([p_ + ", hyper quality, Ultra HD, 8K" for p_ in prompts], None)
    ne, nm, npool = te(NEG, None)
    torch.save(dict(dit_cfg=synth.TINY_FLUX, dit_weight_seed=DIT_SEED, pos=(pe, pm, pp), neg=(ne, nm, npool),
                    height=H, width=W, temp=temp, steps=[2, 2, 2], video_steps=[2, 2, 2], guidance=7.0,
                    video_guidance=5.0, latent_seed=0, noise_seed=1, latents=lat),
               os.path.join(OUT, "generate_tiny_latents_batch2.pt"))


def scheduler_fixture():
    ref = rh.shims.load_reference()
    out = {}
    for name, kw in (("default", {}), ("one_stage", dict(stages=1, stage_range=[0, 1]))):
        s = ref.PyramidFlowMatchEulerDiscreteScheduler(**kw)
        rec = dict(start=dict(s.start_sigmas), end=dict(s.end_sigmas), ori=dict(s.ori_start_sigmas),
                   ratios=dict(s.timestep_ratios), tables={})
        for st in range(s.config.stages):
            for n in (10, 20):
                s.set_timesteps(n, st)
                rec["tables"][(st, n)] = (s.timesteps.clone(), s.sigmas.clone())
        out[name] = rec
    torch.save(out, os.path.join(OUT, "scheduler_tables.pt"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    flux_forward_fixture()
    mmdit_forward_fixture()
    i2v_fixture()
    vae_fixture()
    generate_fixture()
    generate_bf16_fixture()
    generate_batch_fixture()
    generate_long_fixture()
    scheduler_fixture()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
