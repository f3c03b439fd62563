"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU fp32 restatement of the miniFLUX
``PyramidFluxTransformer.forward`` (single pyramid stage per call, no sequence
parallelism, SDPA path ``use_flash_attn=False``).  Functional over a state
dict with the reference's key names.  Reference file:line cited per function
(``flux:`` = pyramid_dit/flux_modules/modeling_pyramid_flux.py, ``blk:`` =
modeling_flux_block.py, ``nrm:`` = modeling_normalization.py, ``emb:`` =
modeling_embedding.py).  Pinned against the imported reference by
``tests/test_oracle_vs_reference.py``.  Never imported by the product path.
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def timestep_embedding(t, dim=256):
    # emb:11-62 with flip_sin_to_cos=True, downscale_freq_shift=0 -> [cos | sin]
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    e = t[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)


def time_text_embed(sd, timestep, pooled):
    # emb:185-200
    tp = timestep_embedding(timestep).to(pooled.dtype)
    te = _lin(sd, "time_text_embed.timestep_embedder.linear_2",
              F.silu(_lin(sd, "time_text_embed.timestep_embedder.linear_1", tp)))
    pe = _lin(sd, "time_text_embed.text_embedder.linear_2",
              F.silu(_lin(sd, "time_text_embed.text_embedder.linear_1", pooled)))
    return te + pe


def image_ids(temp, h, w, train_h, train_w, start_t):
    # flux:186-211
    ids = torch.zeros(temp, h, w, 3)
    ids[..., 0] += torch.arange(start_t, start_t + temp)[:, None, None]
    hp = F.interpolate(torch.arange(train_h)[None, None].float(), h, mode="linear")[0, 0] \
        if h != train_h else torch.arange(train_h).float()
    wp = F.interpolate(torch.arange(train_w)[None, None].float(), w, mode="linear")[0, 0] \
        if w != train_w else torch.arange(train_w).float()
    ids[..., 1] += hp[None, :, None]
    ids[..., 2] += wp[None, None, :]
    return ids.reshape(-1, 3)


def rope_table(ids, axes_dim, theta=10000):
    # flux:28-57 ; ids [L,3] -> [L, sum(axes)/2, 2, 2] fp32
    outs = []
    for i, dim in enumerate(axes_dim):
        scale = torch.arange(0, dim, 2, dtype=torch.float64) / dim
        omega = 1.0 / (theta ** scale)
        out = ids[:, i].double()[:, None] * omega[None]
        c, s = torch.cos(out), torch.sin(out)
        outs.append(torch.stack([c, -s, s, c], dim=-1).view(-1, dim // 2, 2, 2))
    return torch.cat(outs, dim=1).float()


def apply_rope(x, freqs):
    # blk:34-39 ; x [B,L,H,hd], freqs [L,hd/2,2,2]
    x_ = x.float().reshape(*x.shape[:-1], -1, 1, 2)
    f = freqs[None, :, None]
    out = f[..., 0] * x_[..., 0] + f[..., 1] * x_[..., 1]
    return out.reshape(*x.shape).type_as(x)


def rms_norm(x, w, eps):
    # nrm:66-79
    var = x.float().pow(2).mean(-1, keepdim=True)
    return (x * torch.rsqrt(var + eps)).to(x.dtype) * w


def layer_norm(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def build_mask(enc_mask, frame_t_img):
    """flux:318-350. enc_mask [B,Lt] int; frame_t_img [L_img] temporal id per image token.
    Returns bool [B,1,L,L]."""
    B, Lt = enc_mask.shape
    tok = torch.arange(1, B + 1)[:, None].repeat(1, Lt)
    tok[enc_mask == 0] = 0
    img = torch.arange(1, B + 1)[:, None].repeat(1, frame_t_img.numel())
    ids = torch.cat([tok, img], dim=1)
    order = torch.cat([torch.zeros(Lt), frame_t_img.float()])[None].repeat(B, 1)
    m = ids[:, None, :, None] == ids[:, None, None, :]
    m = m & (order[:, None, :, None] >= order[:, None, None, :])
    return m


HEAD_CHUNK_ABOVE_L = 8192      # longer sequences run the (per-head independent) SDPA a few heads at a time: bounded memory
HEAD_CHUNK = 5


def attention(q, k, v, mask):
    # blk:361-365 ; q,k,v [B,L,H,hd].  Heads are independent in SDPA, so evaluating them in groups is the same arithmetic
    # per head (tests/test_oracle_vs_reference.py pins the grouped form to the one-call form); it only bounds the memory
    # of a fallback that materialises [B, H, L, L] scores at the headline sequence length (L = 15 488: 57 GB in one call)
    qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    if q.shape[1] <= HEAD_CHUNK_ABOVE_L:
        o = F.scaled_dot_product_attention(qt, kt, vt, attn_mask=mask)
    else:
        o = torch.cat([F.scaled_dot_product_attention(qt[:, h:h + HEAD_CHUNK], kt[:, h:h + HEAD_CHUNK], vt[:, h:h + HEAD_CHUNK],
                                                      attn_mask=mask) for h in range(0, qt.shape[1], HEAD_CHUNK)], dim=1)
    return o.transpose(1, 2).flatten(2, 3)


def double_block(sd, p, cfg, x, c, temb, mask, freqs):
    # blk:992-1044 + processor blk:805-874 + functor blk:328-376
    H, hd = cfg["num_attention_heads"], cfg["attention_head_dim"]
    B = x.shape[0]
    e = _lin(sd, p + "norm1.linear", F.silu(temb))
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [t[:, None] for t in e.chunk(6, dim=1)]
    ec = _lin(sd, p + "norm1_context.linear", F.silu(temb))
    c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = [t[:, None] for t in ec.chunk(6, dim=1)]
    xn = layer_norm(x) * (1 + scale_msa) + shift_msa
    cn = layer_norm(c) * (1 + c_scale_msa) + c_shift_msa

    def heads(t):
        return t.view(B, -1, H, hd)
    q = rms_norm(heads(_lin(sd, p + "attn.to_q", xn)), sd[p + "attn.norm_q.weight"], 1e-6)
    k = rms_norm(heads(_lin(sd, p + "attn.to_k", xn)), sd[p + "attn.norm_k.weight"], 1e-6)
    v = heads(_lin(sd, p + "attn.to_v", xn))
    cq = rms_norm(heads(_lin(sd, p + "attn.add_q_proj", cn)), sd[p + "attn.norm_added_q.weight"], 1e-6)
    ck = rms_norm(heads(_lin(sd, p + "attn.add_k_proj", cn)), sd[p + "attn.norm_added_k.weight"], 1e-6)
    cv = heads(_lin(sd, p + "attn.add_v_proj", cn))
    Q = apply_rope(torch.cat([cq, q], 1), freqs)
    K = apply_rope(torch.cat([ck, k], 1), freqs)
    V = torch.cat([cv, v], 1)
    o = attention(Q, K, V, mask)
    Lt = c.shape[1]
    co, xo = o[:, :Lt], o[:, Lt:]
    xo = _lin(sd, p + "attn.to_out.0", xo)
    co = _lin(sd, p + "attn.to_add_out", co)
    x = x + gate_msa * xo
    xn2 = layer_norm(x) * (1 + scale_mlp) + shift_mlp
    ff = _lin(sd, p + "ff.net.2", F.gelu(_lin(sd, p + "ff.net.0.proj", xn2), approximate="tanh"))
    x = x + gate_mlp * ff
    c = c + c_gate_msa * co
    cn2 = layer_norm(c) * (1 + c_scale_mlp) + c_shift_mlp
    cff = _lin(sd, p + "ff_context.net.2", F.gelu(_lin(sd, p + "ff_context.net.0.proj", cn2), approximate="tanh"))
    c = c + c_gate_mlp * cff
    return c, x


def single_block(sd, p, cfg, x, temb, mask, freqs):
    # blk:914-942 + processor blk:745-785 + functor blk:568-606
    H, hd = cfg["num_attention_heads"], cfg["attention_head_dim"]
    B = x.shape[0]
    e = _lin(sd, p + "norm.linear", F.silu(temb))
    shift, scale, gate = [t[:, None] for t in e.chunk(3, dim=1)]
    xn = layer_norm(x) * (1 + scale) + shift
    mlp = F.gelu(_lin(sd, p + "proj_mlp", xn), approximate="tanh")
    q = rms_norm(_lin(sd, p + "attn.to_q", xn).view(B, -1, H, hd), sd[p + "attn.norm_q.weight"], 1e-6)
    k = rms_norm(_lin(sd, p + "attn.to_k", xn).view(B, -1, H, hd), sd[p + "attn.norm_k.weight"], 1e-6)
    v = _lin(sd, p + "attn.to_v", xn).view(B, -1, H, hd)
    o = attention(apply_rope(q, freqs), apply_rope(k, freqs), v, mask)
    return x + gate * _lin(sd, p + "proj_out", torch.cat([o, mlp], dim=2))


def patchify(clip):
    # flux:284-288 : b c t h w -> b (t h w) (p1 p2 c)
    b, c, t, h, w = clip.shape
    x = clip.permute(0, 2, 3, 4, 1).reshape(b, t, h // 2, 2, w // 2, 2, c)
    return x.permute(0, 1, 2, 4, 3, 5, 6).reshape(b, t * (h // 2) * (w // 2), 4 * c)


def unpatchify(tok, t, h, w):
    # flux:383-388 ; tok [B, t*h*w, 4*c] -> [B,c,t,2h,2w]
    b = tok.shape[0]
    c = tok.shape[-1] // 4
    x = tok.reshape(b, t, h, w, 2, 2, c).permute(0, 1, 2, 4, 3, 5, 6).reshape(b, t, 2 * h, 2 * w, c)
    return x.permute(0, 4, 1, 2, 3)


def sequence_geometry(clips):
    """(ids [L_img,3], frame_t [L_img]) for a clip list oldest->newest (flux:213-237)."""
    th, tw = clips[-1].shape[-2] // 2, clips[-1].shape[-1] // 2
    ids, start = [], 0
    for cl in clips:
        _, _, t, h, w = cl.shape
        ids.append(image_ids(t, h // 2, w // 2, th, tw, start))
        start += t
    ids = torch.cat(ids, 0)
    return ids, ids[:, 0].clone()


def flux_forward(sd, cfg, clips, enc, enc_mask, pooled, timestep, return_intermediates=False):
    """flux:392-542 for ``sample=[clips]`` (one stage). Returns [B,16,t,h,w] of the LAST clip."""
    sd = {k: v.float() for k, v in sd.items()}
    clips = [c.float() for c in clips]
    enc, pooled = enc.float(), pooled.float()
    temb = time_text_embed(sd, timestep, pooled)
    c = _lin(sd, "context_embedder", enc)
    Lt = c.shape[1]
    ids, frame_t = sequence_geometry(clips)
    all_ids = torch.cat([torch.zeros(Lt, 3), ids], 0)
    freqs = rope_table(all_ids, cfg["axes_dims_rope"])
    x = _lin(sd, "x_embedder", torch.cat([patchify(cl) for cl in clips], dim=1))
    mask = build_mask(enc_mask, frame_t)
    inter = {"temb": temb, "x0": x, "c0": c}
    for i in range(cfg["num_layers"]):
        c, x = double_block(sd, f"transformer_blocks.{i}.", cfg, x, c, temb, mask, freqs)
        if i == 0:
            inter["x_after_double0"], inter["c_after_double0"] = x, c
        inter.setdefault("blocks", []).append(torch.cat([c, x], dim=1))       # [text | image] rows after every block
    h = torch.cat([c, x], dim=1)
    for i in range(cfg["num_single_layers"]):
        h = single_block(sd, f"single_transformer_blocks.{i}.", cfg, h, temb, mask, freqs)
        inter.setdefault("blocks", []).append(h)
    x = h[:, Lt:]
    inter["x_final"] = x
    e = _lin(sd, "norm_out.linear", F.silu(temb))
    scale, shift = e.chunk(2, dim=1)                      # nrm:119 (scale, shift)
    x = layer_norm(x) * (1 + scale[:, None]) + shift[:, None]
    x = _lin(sd, "proj_out", x)
    _, _, t, hh, ww = clips[-1].shape
    n = t * (hh // 2) * (ww // 2)
    out = unpatchify(x[:, -n:], t, hh // 2, ww // 2)
    if return_intermediates:
        return out, inter
    return out
