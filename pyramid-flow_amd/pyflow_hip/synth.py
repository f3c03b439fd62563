"""Synthetic (random-init) weights with the reference's state-dict key layout.

No checkpoints exist in this environment: the benchmark and the tests use random weights of the
released architectures.  Key names/shapes are checked against the instantiated reference modules in
tests/test_oracle_vs_reference.py.
"""
from collections import OrderedDict

import torch

MINIFLUX = dict(num_layers=8, num_single_layers=16, num_attention_heads=30, attention_head_dim=64,
                in_channels=64, joint_attention_dim=4096, pooled_projection_dim=768,
                axes_dims_rope=[16, 24, 24])
TINY_FLUX = dict(num_layers=2, num_single_layers=2, num_attention_heads=4, attention_head_dim=64,
                 in_channels=64, joint_attention_dim=32, pooled_projection_dim=16,
                 axes_dims_rope=[16, 24, 24])
VAE_DEFAULT = dict(latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=(3, 3, 3, 3),
                   spatial_up_sample=(True, True, True, False), temporal_up_sample=(True, True, True, False),
                   out_channels=3, norm_num_groups=32)
TINY_VAE = dict(latent_channels=16, block_out_channels=(32, 32, 64, 64), layers_per_block=(2, 2, 2, 2),
                spatial_up_sample=(True, True, True, False), temporal_up_sample=(True, True, True, False),
                out_channels=3, norm_num_groups=32)


def flux_param_shapes(cfg):
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    hd = cfg["attention_head_dim"]
    s = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    lin("time_text_embed.timestep_embedder.linear_1", d, 256)
    lin("time_text_embed.timestep_embedder.linear_2", d, d)
    lin("time_text_embed.text_embedder.linear_1", d, cfg["pooled_projection_dim"])
    lin("time_text_embed.text_embedder.linear_2", d, d)
    lin("context_embedder", d, cfg["joint_attention_dim"])
    lin("x_embedder", d, cfg["in_channels"])
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}."
        lin(p + "norm1.linear", 6 * d, d)
        lin(p + "norm1_context.linear", 6 * d, d)
        s[p + "attn.norm_q.weight"] = (hd,)
        s[p + "attn.norm_k.weight"] = (hd,)
        for n in ("to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj"):
            lin(p + "attn." + n, d, d)
        lin(p + "attn.to_out.0", d, d)
        lin(p + "attn.to_add_out", d, d)
        s[p + "attn.norm_added_q.weight"] = (hd,)
        s[p + "attn.norm_added_k.weight"] = (hd,)
        lin(p + "ff.net.0.proj", 4 * d, d)
        lin(p + "ff.net.2", d, 4 * d)
        lin(p + "ff_context.net.0.proj", 4 * d, d)
        lin(p + "ff_context.net.2", d, 4 * d)
    for i in range(cfg["num_single_layers"]):
        p = f"single_transformer_blocks.{i}."
        lin(p + "norm.linear", 3 * d, d)
        lin(p + "proj_mlp", 4 * d, d)
        lin(p + "proj_out", d, 5 * d)
        s[p + "attn.norm_q.weight"] = (hd,)
        s[p + "attn.norm_k.weight"] = (hd,)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + "attn." + n, d, d)
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg["in_channels"], d)
    return s


SD3_MMDIT = dict(sample_size=128, patch_size=2, in_channels=16, num_layers=24, attention_head_dim=64,
                 num_attention_heads=24, caption_projection_dim=1536, pooled_projection_dim=2048, pos_embed_max_size=192,
                 joint_attention_dim=4096)


def tiny_mmdit_cfg():
    return dict(sample_size=32, patch_size=2, in_channels=16, num_layers=3, attention_head_dim=64, num_attention_heads=4,
                caption_projection_dim=256, pooled_projection_dim=16, pos_embed_max_size=48, joint_attention_dim=32)


def mmdit_param_shapes(cfg):
    """state-dict layout of PyramidDiffusionMMDiT (mmdit_modules/modeling_pyramid_mmdit.py:60-149) incl. the
    persistent sincos buffer `pos_embed.pos_embed`."""
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    hd = cfg["attention_head_dim"]
    assert cfg["caption_projection_dim"] == d
    s = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    s["pos_embed.pos_embed"] = (1, cfg["pos_embed_max_size"] ** 2, d)
    s["pos_embed.proj.weight"] = (d, cfg["in_channels"], cfg["patch_size"], cfg["patch_size"])
    s["pos_embed.proj.bias"] = (d,)
    lin("time_text_embed.timestep_embedder.linear_1", d, 256)
    lin("time_text_embed.timestep_embedder.linear_2", d, d)
    lin("time_text_embed.text_embedder.linear_1", d, cfg["pooled_projection_dim"])
    lin("time_text_embed.text_embedder.linear_2", d, d)
    lin("context_embedder", d, cfg["joint_attention_dim"])
    n = cfg["num_layers"]
    for i in range(n):
        p = f"transformer_blocks.{i}."
        last = i == n - 1
        lin(p + "norm1.linear", 6 * d, d)
        lin(p + "norm1_context.linear", (2 if last else 6) * d, d)
        s[p + "attn.norm_q.weight"] = (hd,)
        s[p + "attn.norm_k.weight"] = (hd,)
        for nme in ("to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj"):
            lin(p + "attn." + nme, d, d)
        s[p + "attn.norm_add_q.weight"] = (hd,)
        s[p + "attn.norm_add_k.weight"] = (hd,)
        lin(p + "attn.to_out.0", d, d)
        if not last:
            lin(p + "attn.to_add_out", d, d)
        lin(p + "ff.net.0.proj", 4 * d, d)
        lin(p + "ff.net.2", d, 4 * d)
        if not last:
            lin(p + "ff_context.net.0.proj", 4 * d, d)
            lin(p + "ff_context.net.2", d, 4 * d)
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg["patch_size"] ** 2 * cfg["in_channels"], d)
    return s


def sincos_2d_table(embed_dim, grid_size, base_size):
    """the persistent `pos_embed.pos_embed` buffer: get_2d_sincos_pos_embed(embed_dim, grid_size, base_size=...)
    (mmdit_modules/modeling_embedding.py:23-74) -> [grid*grid, D] fp32; host-side table construction."""
    import numpy as np

    def one_d(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    g = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size)
    grid = np.stack(np.meshgrid(g, g), axis=0).reshape([2, 1, grid_size, grid_size])
    return torch.from_numpy(np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)).float()


def mmdit_state_dict(cfg, seed=1234, std=0.02, lively=False):
    """random weights in the MMDiT key layout with the REAL sincos table in the `pos_embed.pos_embed` buffer"""
    sd = random_state_dict(mmdit_param_shapes(cfg), seed=seed, std=std, lively=lively)
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    sd["pos_embed.pos_embed"] = sincos_2d_table(d, cfg["pos_embed_max_size"], cfg["sample_size"] // cfg["patch_size"])[None]
    return sd


def vae_decoder_param_shapes(cfg):
    """decoder + post_quant_conv of CausalVideoVAE (video_vae/modeling_causal_vae.py:137-153)."""
    s = OrderedDict()
    boc = cfg["block_out_channels"]
    lat = cfg["latent_channels"]

    def conv(name, co, ci, k):
        s[name + ".conv.weight"] = (co, ci, k, k, k)
        s[name + ".conv.bias"] = (co,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(p, ci, co):
        norm(p + "norm1", ci)
        conv(p + "conv1", co, ci, 3)
        norm(p + "norm2", co)
        conv(p + "conv2", co, co, 3)
        if ci != co:
            conv(p + "conv_shortcut", co, ci, 1)

    top = boc[-1]
    conv("decoder.conv_in", top, lat, 3)
    resnet("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    norm(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"] = (top, top)
        s[a + n + ".bias"] = (top,)
    resnet("decoder.mid_block.resnets.1.", top, top)
    rev = list(reversed(boc))
    prev = rev[0]
    for i, co in enumerate(rev):
        p = f"decoder.up_blocks.{i}."
        for j in range(cfg["layers_per_block"][i]):
            resnet(p + f"resnets.{j}.", prev if j == 0 else co, co)
        if cfg["spatial_up_sample"][i]:
            conv(p + "upsamplers.0.conv", co * 4, co, 3)
        if cfg["temporal_up_sample"][i]:
            conv(p + "temporal_upsamplers.0.conv", co * 2, co, 3)
        prev = co
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", cfg["out_channels"], boc[0], 3)
    conv("post_quant_conv", lat, lat, 1)
    return s


def vae_encoder_param_shapes(cfg):
    """encoder + quant_conv of CausalVideoVAE (video_vae/modeling_causal_vae.py:120-136, modeling_enc_dec.py:55-152);
    cfg: latent_channels, encoder_block_out_channels, encoder_layers_per_block, encoder_spatial/temporal_down_sample."""
    s = OrderedDict()
    boc = cfg["encoder_block_out_channels"]
    lat = cfg["latent_channels"]

    def conv(name, co, ci, k):
        s[name + ".conv.weight"] = (co, ci, k, k, k)
        s[name + ".conv.bias"] = (co,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(p, ci, co):
        norm(p + "norm1", ci)
        conv(p + "conv1", co, ci, 3)
        norm(p + "norm2", co)
        conv(p + "conv2", co, co, 3)
        if ci != co:
            conv(p + "conv_shortcut", co, ci, 1)

    conv("encoder.conv_in", boc[0], 3, 3)
    prev = boc[0]
    for i, co in enumerate(boc):
        p = f"encoder.down_blocks.{i}."
        for j in range(cfg["encoder_layers_per_block"][i]):
            resnet(p + f"resnets.{j}.", prev if j == 0 else co, co)
        if cfg["encoder_spatial_down_sample"][i]:
            conv(p + "downsamplers.0.conv", co, co, 3)
        if cfg["encoder_temporal_down_sample"][i]:
            conv(p + "temporal_downsamplers.0.conv", co, co, 3)
        prev = co
    top = boc[-1]
    resnet("encoder.mid_block.resnets.0.", top, top)
    a = "encoder.mid_block.attentions.0."
    norm(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"] = (top, top)
        s[a + n + ".bias"] = (top,)
    resnet("encoder.mid_block.resnets.1.", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out", 2 * lat, top, 3)
    conv("quant_conv", 2 * lat, 2 * lat, 1)
    return s


TINY_VAE_ENC = dict(latent_channels=16, encoder_block_out_channels=(32, 32, 64, 64), encoder_layers_per_block=(1, 1, 1, 1),
                    encoder_spatial_down_sample=(True, True, True, False), encoder_temporal_down_sample=(True, True, True, False))
VAE_ENC_DEFAULT = dict(latent_channels=16, encoder_block_out_channels=(128, 256, 512, 512), encoder_layers_per_block=(2, 2, 2, 2),
                       encoder_spatial_down_sample=(True, True, True, False), encoder_temporal_down_sample=(True, True, True, False))


def random_state_dict(shapes, seed=1234, std=0.02, lively=False, dtype=torch.float32):
    """N(0, std^2) matrices, norm gains 1, biases 0.  lively=True perturbs gains and
    biases too (tests: exercises every term)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in shapes.items():
        is_gain = len(shp) == 1 and k.endswith(".weight")
        if is_gain:
            t = torch.ones(shp)
            if lively:
                t = t + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = torch.zeros(shp)
            if lively:
                t = std * torch.randn(shp, generator=g)
        else:
            t = std * torch.randn(shp, generator=g)
        sd[k] = t.to(dtype)
    return sd
