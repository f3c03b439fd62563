"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU fp32 restatement of
``PyramidDiTForVideoGeneration.generate`` (pyramid_dit/
pyramid_dit_for_video_gen_pipeline.py:1006-1243) with its helpers
``generate_one_unit`` (:706-788), ``get_pyramid_latent`` (:555-570),
``sample_block_noise`` (:697-703, restated as the Cholesky form the reference's
MultivariateNormal evaluates; noise is INJECTED so both sides see identical
draws) and ``decode_latent`` (:1221-1243).  miniFLUX only.  Pinned against the
imported reference's own generate() in tests/test_oracle_vs_reference.py.
Never imported by the product path.
"""
import math

import torch
import torch.nn.functional as F

from .flux_oracle import flux_forward
from .scheduler_oracle import SchedulerOracle
from .vae_oracle import vae_decode, to_uint8_frames


# scale_tril that torch's MultivariateNormal(zeros(4), (1+g)I - g 11^T) evaluates for the default g = 1/3 in
# fp32 (the covariance is singular, so the last pivot is pure round-off and the PositiveDefinite validation of
# the distribution is CPU-dependent: pinned here as constants; tests/test_oracle_vs_reference.py checks them
# against torch in the dev container).
_L_DEFAULT = [[1.0, 0.0, 0.0, 0.0],
              [-0.3333333432674408, 0.9428090453147888, 0.0, 0.0],
              [-0.3333333432674408, -0.471404492855072, 0.8164966106414795, 0.0],
              [-0.3333333432674408, -0.471404492855072, -0.8164964914321899, 0.00042286395910196006]]


def block_noise_cholesky(gamma=1 / 3):
    if abs(gamma - 1 / 3) < 1e-12:
        return torch.tensor(_L_DEFAULT, dtype=torch.float32)
    cov = torch.eye(4) * (1 + gamma) - torch.ones(4, 4) * gamma
    return torch.linalg.cholesky(cov)


def block_noise_from_normal(eps, bs, ch, t, h, w, gamma=1 / 3):
    """eps [bs*ch*t*(h/2)*(w/2), 4] standard normal -> [bs,ch,t,h,w] (pipeline.py:697-703)."""
    L = block_noise_cholesky(gamma)
    n = eps @ L.T
    n = n.reshape(bs, ch, t, h // 2, w // 2, 2, 2).permute(0, 1, 2, 3, 5, 4, 6)
    return n.reshape(bs, ch, t, h, w)


def pyramid_latent(x, stage_num):
    # :555-570 -> list low-res ... full-res
    out = [x]
    b, c, t, h, w = x.shape
    for _ in range(stage_num):
        h //= 2
        w //= 2
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, x.shape[-2], x.shape[-1])
        y = F.interpolate(y, size=(h, w), mode="bilinear")
        x = y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
        out.append(x)
    return list(reversed(out))


def history_clips(clean, unit_index, n_stages, cfg_dup):
    """:1155-1182. clean = pyramid list (low->high) of all generated latents.
    Returns per-stage list of clips oldest->newest."""
    res = []
    for i_s in range(n_stages):
        dup = (lambda t: torch.cat([t] * 2)) if cfg_dup else (lambda t: t)
        stage_input = [dup(clean[i_s][:, :, -1:])]
        cur_stage, ptx = i_s, 1
        while ptx < unit_index:
            cur_stage = max(cur_stage - 1, 0)
            if cur_stage == 0:
                break
            ptx += 1
            stage_input.append(dup(clean[cur_stage][:, :, -ptx:-(ptx - 1)]))
        if cur_stage == 0 and ptx < unit_index:
            stage_input.append(dup(clean[0][:, :, :-ptx]))
        res.append(list(reversed(stage_input)))
    return res


def generate_latents(dit_sd, dit_cfg, prompt_embeds, prompt_mask, pooled, init_latents,
                     block_noise_fn, steps_first, steps_video, guidance_scale, video_guidance_scale,
                     stages=(1, 2, 4), sched_kwargs=None, record=None, forward_fn=None, image_latent=None):
    """generate() up to ``output_type='latent'``.  prompt_* are already [neg, pos] concatenated.
    init_latents [1,16,temp,H/8,W/8] = randn_tensor output.  block_noise_fn(bs,ch,t,h,w) -> tensor.
    forward_fn: flux_forward (default) or mmdit_oracle.mmdit_forward.
    image_latent [1,16,1,h,w] (already normalised): generate_i2v (:791-1003) -- the image is unit 0, units
    1..temp-1 are sampled from noise slices 0..temp-2 with the video schedule / video guidance."""
    forward_fn = forward_fn or flux_forward
    sched = SchedulerOracle(**(sched_kwargs or {"stages": len(stages)}))
    n_st = len(stages)
    cfg_on = guidance_scale > 0
    lat = init_latents.float()
    b, c, temp, h, w = lat.shape
    y = lat.permute(0, 2, 1, 3, 4).reshape(b * temp, c, h, w)
    for _ in range(n_st - 1):                                   # :1112-1116
        h //= 2
        w //= 2
        y = F.interpolate(y, size=(h, w), mode="bilinear") * 2
    lat = y.reshape(b, temp, c, h, w).permute(0, 2, 1, 3, 4)
    generated = [] if image_latent is None else [image_latent.float()]
    for u in range(len(generated), temp):
        if u == 0:
            past = [[] for _ in range(n_st)]
            x = lat[:, :, :1]
            steps, gs = steps_first, guidance_scale
        else:
            clean = pyramid_latent(torch.cat(generated, dim=2), n_st - 1)
            past = history_clips(clean, u, n_st, cfg_on)
            x = lat[:, :, u:u + 1] if image_latent is None else lat[:, :, u - 1:u]       # :1185 vs :969
            steps, gs = steps_video, video_guidance_scale
        hh, ww = h, w
        for i_s in range(n_st):                                  # :725-786
            sched.set_timesteps(steps[i_s], i_s)
            if i_s > 0:
                hh *= 2
                ww *= 2
                xb = x.permute(0, 2, 1, 3, 4).reshape(-1, c, x.shape[-2], x.shape[-1])
                xb = F.interpolate(xb, size=(hh, ww), mode="nearest")
                x = xb.reshape(b, 1, c, hh, ww).permute(0, 2, 1, 3, 4)
                alpha, beta = sched.renoise_coeffs(i_s)
                x = alpha * x + beta * block_noise_fn(b, c, 1, hh, ww).to(x.dtype)
            for t in sched.timesteps:
                inp = torch.cat([x] * 2) if cfg_on else x
                timestep = t.expand(inp.shape[0]).to(inp.dtype)
                v = forward_fn(dit_sd, dit_cfg, past[i_s] + [inp], prompt_embeds, prompt_mask,
                               pooled, timestep)
                if cfg_on:
                    vu, vt = v.chunk(2)
                    v = vu + gs * (vt - vu)
                x = sched.step(v, x)
                if record is not None:
                    record.append(x.clone())
        generated.append(x)
    return torch.cat(generated, dim=2)


def decode_latents(vae_sd, vae_cfg, latents, model_name="pyramid_flux", use_tiling=False,
                   tile_sample_min_size=256):
    """decode_latent :1221-1243 -> uint8 [T,H,W,3]."""
    shift, scale = (-0.04, 1 / 1.8726) if model_name == "pyramid_flux" else (0.1490, 1 / 1.8415)
    vshift, vscale = -0.2343, 1 / 3.0986
    z = latents.float().clone()
    z[:, :, :1] = z[:, :, :1] / scale + shift
    if z.shape[2] > 1:
        z[:, :, 1:] = z[:, :, 1:] / vscale + vshift
    img = vae_decode(vae_sd, vae_cfg, z, use_tiling=use_tiling, tile_sample_min_size=tile_sample_min_size)
    return to_uint8_frames(img), img
