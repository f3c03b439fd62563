"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU fp32 restatement of the SD3-style
``PyramidDiffusionMMDiT.forward`` (single pyramid stage per call, SDPA path, sincos absolute
position + temporal RoPE, temporal-causal mask) as the pipeline instantiates it
(pyramid_dit_for_video_gen_pipeline.py:80-87: use_t5_mask, add_temp_pos_embed, temp_pos_embed_type='rope',
use_temporal_causal, interp_condition_pos).  Functional over a state dict with the reference's key names.
Reference file:line cited per function (``mm:`` = pyramid_dit/mmdit_modules/modeling_pyramid_mmdit.py, ``blk:`` =
modeling_mmdit_block.py, ``emb:`` = mmdit_modules/modeling_embedding.py, ``nrm:`` = mmdit_modules/modeling_normalization.py).
Pinned against the imported reference by tests/test_oracle_vs_reference.py.  Never imported by the product path.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .flux_oracle import (_lin, apply_rope, attention, build_mask, layer_norm, rope_table, time_text_embed,
                          unpatchify)


def sincos_1d(embed_dim, pos):
    # emb:56-74
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d_table(embed_dim, grid_size, base_size):
    """emb:23-54 get_2d_sincos_pos_embed(embed_dim, grid_size, base_size=..., interpolation_scale=1) -> [g*g, D] fp32"""
    gh = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size)
    gw = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    emb_h = sincos_1d(embed_dim // 2, grid[0])
    emb_w = sincos_1d(embed_dim // 2, grid[1])
    return torch.from_numpy(np.concatenate([emb_h, emb_w], axis=1)).float()


def cropped_pos_embed(table, max_size, h, w, ori_h, ori_w):
    """emb:269-308 with interp_condition_pos=True; h, w, ori_h, ori_w in TOKENS. table [max*max, D] -> [h*w, D]"""
    top, left = (max_size - ori_h) // 2, (max_size - ori_w) // 2
    pe = table.reshape(1, max_size, max_size, -1)[:, top:top + ori_h, left:left + ori_w, :]
    if ori_h != h or ori_w != w:
        pe = F.interpolate(pe.permute(0, 3, 1, 2), size=(h, w), mode="bilinear").permute(0, 2, 3, 1)
    return pe.reshape(-1, pe.shape[-1])


def rms_norm_mm(x, w, eps):
    # nrm:38-67: fp32 variance, x*rsqrt, (to weight dtype), *w, back to input dtype
    var = x.float().pow(2).mean(-1, keepdim=True)
    return (x * torch.rsqrt(var + eps) * w).to(x.dtype)


def patch_embed(sd, clips, cfg):
    """emb:310-390 PatchEmbed3D.forward for ONE stage's clip list: Conv2d k2 s2 per frame + cropped sincos pos."""
    w, b = sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"]
    table = sd["pos_embed.pos_embed"][0]
    ms = cfg["pos_embed_max_size"]
    oh, ow = clips[-1].shape[-2] // 2, clips[-1].shape[-1] // 2
    outs = []
    for cl in clips:
        bsz, c, t, h, ww = cl.shape
        x = F.conv2d(cl.permute(0, 2, 1, 3, 4).reshape(bsz * t, c, h, ww), w, b, stride=2)       # (b t) d h/2 w/2
        x = x.flatten(2).transpose(1, 2)                                                      # (b t) n d
        x = x + cropped_pos_embed(table, ms, h // 2, ww // 2, oh, ow)[None]
        outs.append(x.reshape(bsz, t * x.shape[1], -1))
    return torch.cat(outs, dim=1)


def joint_block(sd, p, cfg, x, c, temb, mask, freqs, last):
    # blk:624-671 + JointAttention blk:489-562 + functor blk:277-323
    H = cfg["num_attention_heads"]
    hd = cfg["attention_head_dim"]
    B = x.shape[0]
    e = _lin(sd, p + "norm1.linear", F.silu(temb))
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [t[:, None] for t in e.chunk(6, dim=1)]   # nrm:108
    xn = layer_norm(x) * (1 + scale_msa) + shift_msa
    ec = _lin(sd, p + "norm1_context.linear", F.silu(temb))
    if last:                                                     # AdaLayerNormContinuous nrm:112-118: (scale, shift)
        c_scale, c_shift = [t[:, None] for t in ec.chunk(2, dim=1)]
        cn = layer_norm(c) * (1 + c_scale) + c_shift
    else:
        c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = [t[:, None] for t in ec.chunk(6, dim=1)]
        cn = layer_norm(c) * (1 + c_scale_msa) + c_shift_msa

    def heads(t):
        return t.view(B, -1, H, hd)
    eps = 1e-5                                                   # JointAttention default eps (blk:409)
    q = rms_norm_mm(heads(_lin(sd, p + "attn.to_q", xn)), sd[p + "attn.norm_q.weight"], eps)
    k = rms_norm_mm(heads(_lin(sd, p + "attn.to_k", xn)), sd[p + "attn.norm_k.weight"], eps)
    v = heads(_lin(sd, p + "attn.to_v", xn))
    cq = rms_norm_mm(heads(_lin(sd, p + "attn.add_q_proj", cn)), sd[p + "attn.norm_add_q.weight"], eps)
    ck = rms_norm_mm(heads(_lin(sd, p + "attn.add_k_proj", cn)), sd[p + "attn.norm_add_k.weight"], eps)
    cv = heads(_lin(sd, p + "attn.add_v_proj", cn))
    Q = apply_rope(torch.cat([cq, q], 1), freqs)
    K = apply_rope(torch.cat([ck, k], 1), freqs)
    V = torch.cat([cv, v], 1)
    o = attention(Q, K, V, mask)
    Lt = c.shape[1]
    co, xo = o[:, :Lt], o[:, Lt:]
    x = x + gate_msa * _lin(sd, p + "attn.to_out.0", xo)
    xn2 = layer_norm(x) * (1 + scale_mlp) + shift_mlp
    x = x + gate_mlp * _lin(sd, p + "ff.net.2", F.gelu(_lin(sd, p + "ff.net.0.proj", xn2), approximate="tanh"))
    if last:
        return None, x
    c = c + c_gate_msa * _lin(sd, p + "attn.to_add_out", co)
    cn2 = layer_norm(c) * (1 + c_scale_mlp) + c_shift_mlp
    c = c + c_gate_mlp * _lin(sd, p + "ff_context.net.2", F.gelu(_lin(sd, p + "ff_context.net.0.proj", cn2), approximate="tanh"))
    return c, x


def mmdit_forward(sd, cfg, clips, enc, enc_mask, pooled, timestep, return_intermediates=False):
    """mm:420-497 for ``sample=[clips]`` (one stage). Returns [B,16,t,h,w] of the LAST clip."""
    sd = {k: v.float() for k, v in sd.items()}
    clips = [c.float() for c in clips]
    enc, pooled = enc.float(), pooled.float()
    temb = time_text_embed(sd, timestep, pooled)
    c = _lin(sd, "context_embedder", enc)
    Lt = c.shape[1]
    # temporal RoPE ids (mm:233-262): one axis of dim head_dim, text ids 0
    frame_t, start = [], 0
    for cl in clips:
        _, _, t, h, w = cl.shape
        frame_t.append(torch.arange(start, start + t)[:, None].repeat(1, (h // 2) * (w // 2)).reshape(-1))
        start += t
    frame_t = torch.cat(frame_t).float()
    ids = torch.cat([torch.zeros(Lt), frame_t])[:, None]
    freqs = rope_table(ids, [cfg["attention_head_dim"]])
    x = patch_embed(sd, clips, cfg)
    mask = build_mask(enc_mask, frame_t)
    inter = {"temb": temb, "x0": x, "c0": c}
    n = cfg["num_layers"]
    for i in range(n):
        c, x = joint_block(sd, f"transformer_blocks.{i}.", cfg, x, c, temb, mask, freqs, last=(i == n - 1))
        if i == 0:
            inter["x_after_block0"] = x
    inter["x_final"] = x
    e = _lin(sd, "norm_out.linear", F.silu(temb))
    scale, shift = e.chunk(2, dim=1)
    x = layer_norm(x) * (1 + scale[:, None]) + shift[:, None]
    x = _lin(sd, "proj_out", x)
    _, _, t, hh, ww = clips[-1].shape
    nt = t * (hh // 2) * (ww // 2)
    out = unpatchify(x[:, -nt:], t, hh // 2, ww // 2)
    if return_intermediates:
        return out, inter
    return out
