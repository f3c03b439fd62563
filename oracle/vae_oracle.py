"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU fp32 restatement of
``CausalVideoVAE.decode`` (video_vae/modeling_causal_vae.py:347-519) and the
decoder stack under it (modeling_enc_dec.py:302-366, modeling_block.py:449-464,
744-759, modeling_resnet.py:115-150, 609-617, 716-729, modeling_causal_conv.py:
36-146).  Temporal chunking is restated as its mathematical equivalent -- one
un-chunked causal pass (the reference's chunk cache carries exactly the two
previous padded frames, causal_conv.py:128-143; verified equal to 5e-6 in
tests/test_oracle_vs_reference.py) -- tiling is restated literally, including
the in-place blend order.  The mid-block attention arithmetic lives in
``diffusers.models.attention_processor.Attention`` (pin diffusers>=0.30.1,
absent from /root/reference): restated from its published algorithm.
Never imported by the product path.
"""
import torch
import torch.nn.functional as F


def causal_conv3d(sd, name, x):
    # causal_conv.py:116-146 (constant pad: 2 zero frames in front, 1 px each side)
    w = sd[name + ".conv.weight"]
    b = sd.get(name + ".conv.bias")
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, w, b)


def group_norm_per_frame(sd, name, x, groups=32, eps=1e-6):
    # causal_conv.py:36-43
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.group_norm(y, groups, sd[name + ".weight"], sd[name + ".bias"], eps)
    return y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def resnet(sd, p, x):
    # modeling_resnet.py:115-150
    h = F.silu(group_norm_per_frame(sd, p + "norm1", x))
    h = causal_conv3d(sd, p + "conv1", h)
    h = F.silu(group_norm_per_frame(sd, p + "norm2", h))
    h = causal_conv3d(sd, p + "conv2", h)
    if (p + "conv_shortcut.conv.weight") in sd:
        x = causal_conv3d(sd, p + "conv_shortcut", x)
    return x + h


def mid_attention(sd, p, x):
    # modeling_block.py:456-460 + diffusers Attention (deprecated attn block form)
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    res = y
    g = F.group_norm(y, 32, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], 1e-6)
    tok = g.reshape(b * t, c, h * w).transpose(1, 2)
    q = F.linear(tok, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(tok, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(tok, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    s = torch.matmul(q, k.transpose(1, 2)) * (c ** -0.5)
    o = torch.matmul(s.float().softmax(-1).to(v.dtype), v)
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    o = o.transpose(1, 2).reshape(b * t, c, h, w) + res
    return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def decoder_forward(sd, cfg, z, is_init_image=True):
    """post_quant_conv + CausalVaeDecoder.forward on a full (un-chunked) clip."""
    x = causal_conv3d(sd, "post_quant_conv", z)
    x = causal_conv3d(sd, "decoder.conv_in", x)
    x = resnet(sd, "decoder.mid_block.resnets.0.", x)
    x = mid_attention(sd, "decoder.mid_block.attentions.0.", x)
    x = resnet(sd, "decoder.mid_block.resnets.1.", x)
    nb = len(cfg["decoder_block_out_channels"])
    for i in range(nb):
        p = f"decoder.up_blocks.{i}."
        for j in range(cfg["decoder_layers_per_block"][i]):
            x = resnet(sd, p + f"resnets.{j}.", x)
        if cfg["decoder_spatial_up_sample"][i]:
            # modeling_resnet.py:609-617
            x = causal_conv3d(sd, p + "upsamplers.0.conv", x)
            b, c4, t, h, w = x.shape
            c = c4 // 4
            x = x.reshape(b, c, 2, 2, t, h, w).permute(0, 1, 4, 5, 2, 6, 3).reshape(b, c, t, 2 * h, 2 * w)
        if cfg["decoder_temporal_up_sample"][i]:
            # modeling_resnet.py:716-729
            x = causal_conv3d(sd, p + "temporal_upsamplers.0.conv", x)
            b, c2, t, h, w = x.shape
            c = c2 // 2
            x = x.reshape(b, c, 2, t, h, w).permute(0, 1, 3, 2, 4, 5).reshape(b, c, 2 * t, h, w)
            if is_init_image:
                x = x[:, :, 1:]
    x = F.silu(group_norm_per_frame(sd, "decoder.conv_norm_out", x))
    return causal_conv3d(sd, "decoder.conv_out", x)


def _blend_v(a, b, e):
    e = min(a.shape[3], b.shape[3], e)
    for y in range(e):
        b[:, :, :, y, :] = a[:, :, :, -e + y, :] * (1 - y / e) + b[:, :, :, y, :] * (y / e)
    return b


def _blend_h(a, b, e):
    e = min(a.shape[4], b.shape[4], e)
    for x in range(e):
        b[:, :, :, :, x] = a[:, :, :, :, -e + x] * (1 - x / e) + b[:, :, :, :, x] * (x / e)
    return b


def vae_decode(sd, cfg, z, use_tiling=False, tile_sample_min_size=256):
    """CausalVideoVAE.decode (causal_vae.py:376-395, 468-519). z [B,C,T,h,w] fp32."""
    sd = {k: v.float() for k, v in sd.items()}
    z = z.float()
    tl = int(tile_sample_min_size / 8)
    if not (use_tiling and (z.shape[-1] > tl or z.shape[-2] > tl)):
        return decoder_forward(sd, cfg, z)
    overlap = int(tl * 0.75)
    blend = int(tile_sample_min_size * 0.25)
    limit = tile_sample_min_size - blend
    rows = []
    for i in range(0, z.shape[3], overlap):
        rows.append([decoder_forward(sd, cfg, z[:, :, :, i:i + tl, j:j + tl])
                     for j in range(0, z.shape[4], overlap)])
    out_rows = []
    for i, row in enumerate(rows):
        res = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend_v(rows[i - 1][j], tile, blend)
            if j > 0:
                tile = _blend_h(row[j - 1], tile, blend)
            res.append(tile[:, :, :, :limit, :limit])
        out_rows.append(torch.cat(res, dim=4))
    return torch.cat(out_rows, dim=3)


def to_uint8_frames(image):
    # pipeline.py:1238-1240
    image = image.mul(127.5).add(127.5).clamp(0, 255).byte()
    b, c, t, h, w = image.shape
    return image.permute(0, 2, 3, 4, 1).reshape(b * t, h, w, c)


# ---------------------------------------------------------------------------------------------------------------
# encode side (modeling_causal_vae.py:274-308, 409-466; modeling_enc_dec.py:154-198; modeling_block.py:528-541;
# modeling_resnet.py:291-336, 458-502) -- un-chunked pass (what generate_i2v uses for its single frame)
def causal_conv3d_strided(sd, name, x, stride=(1, 1, 1)):
    # causal_conv.py:116-146 with a (t, h, w) stride: pad 2 zero frames in front + 1 px each side, then strided Conv3d
    w = sd[name + ".conv.weight"]
    b = sd.get(name + ".conv.bias")
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, w, b, stride=stride)


def encoder_forward(sd, cfg, x):
    """CausalVaeEncoder.forward + quant_conv on a full clip [B,3,T,H,W] -> moments [B,2*latent,T',H/8,W/8]"""
    x = causal_conv3d(sd, "encoder.conv_in", x)
    boc = cfg["encoder_block_out_channels"]
    for i in range(len(boc)):
        p = f"encoder.down_blocks.{i}."
        for j in range(cfg["encoder_layers_per_block"][i]):
            x = resnet(sd, p + f"resnets.{j}.", x)
        if cfg["encoder_spatial_down_sample"][i]:
            x = causal_conv3d_strided(sd, p + "downsamplers.0.conv", x, (1, 2, 2))
        if cfg["encoder_temporal_down_sample"][i]:
            x = causal_conv3d_strided(sd, p + "temporal_downsamplers.0.conv", x, (2, 1, 1))
    x = resnet(sd, "encoder.mid_block.resnets.0.", x)
    x = mid_attention(sd, "encoder.mid_block.attentions.0.", x)
    x = resnet(sd, "encoder.mid_block.resnets.1.", x)
    x = F.silu(group_norm_per_frame(sd, "encoder.conv_norm_out", x))
    x = causal_conv3d(sd, "encoder.conv_out", x)
    return causal_conv3d(sd, "quant_conv", x)


def vae_encode_moments(sd, cfg, x, use_tiling=False, tile_sample_min_size=256):
    """CausalVideoVAE.encode up to the posterior parameters (causal_vae.py:274-308, tiled_encode :409-466).
    x [B,3,T,H,W] in [-1,1] -> moments [B,2*latent,T',h,w]; mean = first half, logvar = second half."""
    sd = {k: v.float() for k, v in sd.items()}
    x = x.float()
    ts = tile_sample_min_size
    tl = int(ts / 8)
    if not (use_tiling and (x.shape[-1] > ts or x.shape[-2] > ts)):
        return encoder_forward(sd, cfg, x)
    overlap = int(ts * 0.75)
    blend = int(tl * 0.25)
    limit = tl - blend
    rows = []
    for i in range(0, x.shape[3], overlap):
        rows.append([encoder_forward(sd, cfg, x[:, :, :, i:i + ts, j:j + ts]) for j in range(0, x.shape[4], overlap)])
    out_rows = []
    for i, row in enumerate(rows):
        res = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend_v(rows[i - 1][j], tile, blend)
            if j > 0:
                tile = _blend_h(row[j - 1], tile, blend)
            res.append(tile[:, :, :, :limit, :limit])
        out_rows.append(torch.cat(res, dim=4))
    return torch.cat(out_rows, dim=3)


def posterior_sample(moments, eps=None):
    """DiagonalGaussianDistribution (modeling_enc_dec.py:369-391): mean + exp(0.5*clamp(logvar,-30,20)) * eps;
    eps None -> mode()."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if eps is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * eps
