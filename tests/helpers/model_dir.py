"""Builds a diffusers-style checkpoint directory of a TINY model in the layout the reference's constructor reads
(pyramid_dit_for_video_gen_pipeline.py:137-160, flux_modules/modeling_text_encoder.py:15-40):

    <root>/diffusion_transformer_384p/{config.json, diffusion_pytorch_model.safetensors}
    <root>/causal_video_vae/{config.json, diffusion_pytorch_model.safetensors}
    <root>/text_encoder (CLIP-L), text_encoder_2 (T5), tokenizer (CLIPTokenizer files), tokenizer_2 (T5TokenizerFast files)

Weights are seeded random (synth), tokenizers are REAL transformers tokenizers over generated vocabularies (byte-level
CLIP BPE with an empty merge table, a unigram T5 tokenizer): nothing is downloaded."""
import json
import os

import torch


def _save(path, sd, cfg):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "diffusion_pytorch_model.safetensors"))


def _bytes_to_unicode():
    """the byte -> printable unicode table of byte-level BPE (GPT-2 / CLIP): printable bytes map to themselves"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def clip_tokenizer_files(path):
    os.makedirs(path, exist_ok=True)
    chars = list(_bytes_to_unicode().values())
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"model_max_length": 77, "tokenizer_class": "CLIPTokenizer", "bos_token": "<|startoftext|>",
                   "eos_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "unk_token": "<|endoftext|>"}, f)
    return len(vocab)


def t5_tokenizer_files(path):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import T5TokenizerFast
    pieces = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), ("▁", -2.0)]
    pieces += [(c, -3.0) for c in "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789,.;:!?'-"]
    tok = Tokenizer(models.Unigram(pieces, unk_id=2))
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    fast = T5TokenizerFast(tokenizer_object=tok, eos_token="</s>", unk_token="<unk>", pad_token="<pad>", extra_ids=0)
    fast.save_pretrained(path)
    return len(pieces)


def build(root, variant="diffusion_transformer_384p", seed=3, with_text=True):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pyflow_hip import synth
    # prompt-encoder widths the device kernels support (head_dim 64): T5 d_model 256, CLIP hidden 128
    dcfg = dict(synth.TINY_FLUX, joint_attention_dim=256, pooled_projection_dim=128)
    dsd = synth.random_state_dict(synth.flux_param_shapes(dcfg), seed=seed, std=0.05, lively=True)
    _save(os.path.join(root, variant), {k: v.to(torch.bfloat16) for k, v in dsd.items()}, dcfg)
    vcfg = synth.TINY_VAE
    vsd = synth.random_state_dict(synth.vae_decoder_param_shapes(vcfg), seed=seed + 1, std=0.05, lively=True)
    ref_cfg = dict(decoder_in_channels=vcfg["latent_channels"], decoder_out_channels=3,
                   decoder_block_out_channels=list(vcfg["block_out_channels"]),
                   decoder_layers_per_block=list(vcfg["layers_per_block"]),
                   decoder_spatial_up_sample=list(vcfg["spatial_up_sample"]),
                   decoder_temporal_up_sample=list(vcfg["temporal_up_sample"]), decoder_norm_num_groups=32)
    _save(os.path.join(root, "causal_video_vae"), {k: v.to(torch.bfloat16) for k, v in vsd.items()}, ref_cfg)
    if with_text:
        from hf_text import tiny_clip, tiny_t5
        nv_clip = clip_tokenizer_files(os.path.join(root, "tokenizer"))
        nv_t5 = t5_tokenizer_files(os.path.join(root, "tokenizer_2"))
        clip, _ = tiny_clip(seed=seed + 2, hidden=dcfg["pooled_projection_dim"], heads=2, inter=256, layers=2,
                            vocab=nv_clip)
        clip.save_pretrained(os.path.join(root, "text_encoder"), safe_serialization=True)
        t5, _ = tiny_t5(seed=seed + 3, d_model=dcfg["joint_attention_dim"], heads=4, d_ff=512, layers=2,
                        vocab=max(nv_t5, 128))
        t5.save_pretrained(os.path.join(root, "text_encoder_2"), safe_serialization=True)
    return dcfg, dsd, vcfg, vsd
