"""The reference's own caller (inference_multigpu.py) runs against the drop-in: same imports, same constructor call,
`.to(device)` on the three sub-models, `enable_tiling()`, `generate(...)` with the script's keyword set, frames to a
video file, `torch.distributed.barrier()` -- on a tiny checkpoint in the diffusers directory layout, world size 1 over
RCCL.  Also `from_pretrained` on such a directory gives bitwise the engine built from the same tensors."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "helpers"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_reference_caller_sequence_t2v(tmp_path):
    import model_dir
    root = tmp_path / "ckpt"
    model_dir.build(str(root))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "reference_caller.py"), str(root), str(tmp_path), "t2v"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    out = tmp_path / "text_to_video_sample.y4m"
    assert out.exists() and out.stat().st_size > 9 * 384 * 640
    print(p.stdout.decode()[-200:])


def test_from_pretrained_directory_equals_state_dict_engine(tmp_path):
    """diffusers directory -> packed MFMA layouts == the same tensors handed over as a state dict, bit for bit: DiT
    forward and VAE decode (closes the checkpoint-ingestion row)"""
    import model_dir
    from pyramid_dit.flux_modules import PyramidFluxTransformer
    from video_vae import CausalVideoVAE
    root = tmp_path / "ckpt"
    dcfg, dsd, vcfg, vsd = model_dir.build(str(root), with_text=False)
    a = PyramidFluxTransformer.from_pretrained(os.path.join(root, "diffusion_transformer_384p"), torch_dtype=torch.bfloat16,
                                               use_flash_attn=False, use_temporal_causal=True, interp_condition_pos=True,
                                               axes_dims_rope=[16, 24, 24])
    b = PyramidFluxTransformer({k: v.to(torch.bfloat16) for k, v in dsd.items()}, dcfg, "cuda")
    assert a.config.in_channels == dcfg["in_channels"] and a.device.type == "cuda" and a.dtype == torch.bfloat16
    g = torch.Generator().manual_seed(0)
    clips = [torch.randn(2, 16, *s, generator=g).cuda() for s in [(1, 8, 16), (1, 16, 32), (1, 16, 32)]]
    enc = torch.randn(2, 16, dcfg["joint_attention_dim"], generator=g)
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, dcfg["pooled_projection_dim"], generator=g)
    t = torch.tensor([500.0, 500.0])
    # the reference's call form: sample = [[clips...]] (one stage), keyword arguments, list result
    oa = a(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled, timestep_ratio=t)
    ob = b(sample=[clips], encoder_hidden_states=enc, encoder_attention_mask=mask, pooled_projections=pooled, timestep_ratio=t)
    assert isinstance(oa, list) and len(oa) == 1 and oa[0].shape == (2, 16, 1, 16, 32)
    assert torch.equal(oa[0], ob[0])
    # ... and both are the ORACLE's forward on the tensors the directory holds (bf16-rounded), not merely each other
    from oracle.flux_oracle import flux_forward
    from oracle.vae_oracle import vae_decode
    from util import rel_l2
    sd_r = {k: v.to(torch.bfloat16).float() for k, v in dsd.items()}
    ref = flux_forward(sd_r, dcfg, [c.cpu().to(torch.bfloat16).float() for c in clips], enc.to(torch.bfloat16).float(), mask,
                       pooled, t)
    clips_r = [c.to(torch.bfloat16).float() for c in clips]
    oa_r = a(sample=[clips_r], encoder_hidden_states=enc.to(torch.bfloat16).float(), encoder_attention_mask=mask,
             pooled_projections=pooled, timestep_ratio=t)
    err = rel_l2(oa_r[0].float().cpu(), ref)
    print(f"from_pretrained DiT forward rel-L2 vs oracle {err:.3e}")
    assert err < 2e-2
    va = CausalVideoVAE.from_pretrained(os.path.join(root, "causal_video_vae"), torch_dtype=torch.bfloat16, interpolate=False)
    vb = CausalVideoVAE({k: v.to(torch.bfloat16) for k, v in vsd.items()}, vcfg, "cuda")
    z = torch.randn(1, 16, 3, 6, 10, generator=g).cuda()
    xa = va.decode(z, temporal_chunk=True, window_size=1).sample
    xb = vb.decode(z, temporal_chunk=True, window_size=1).sample
    assert torch.equal(xa, xb)
    ocfg = dict(decoder_block_out_channels=vcfg["block_out_channels"], decoder_layers_per_block=vcfg["layers_per_block"],
                decoder_spatial_up_sample=vcfg["spatial_up_sample"], decoder_temporal_up_sample=vcfg["temporal_up_sample"])
    zr = z.cpu().to(torch.bfloat16).float()
    vref = vae_decode({k: v.to(torch.bfloat16).float() for k, v in vsd.items()}, ocfg, zr)
    xr = va.decode(zr.cuda(), temporal_chunk=True, window_size=1).sample
    verr = rel_l2(xr.float().cpu(), vref)
    print(f"from_pretrained VAE decode rel-L2 vs oracle {verr:.3e}")
    assert verr < 2e-2


def test_load_checkpoint_dit_prefixed_pth_vs_oracle(tmp_path):
    """`load_checkpoint` on the training format (pipeline.py:213-227: a flat dict whose DiT keys carry a `dit.` prefix
    next to `vae.*` / `text_encoder.*` entries that must be ignored) re-packs the weights on the GPU: the forward of the
    loaded engine equals the oracle's on the checkpoint's tensors, and differs from the engine it replaced."""
    import model_dir
    from pyflow_hip import synth
    from pyflow_hip.pipeline import PyramidDiTForVideoGeneration
    from oracle.flux_oracle import flux_forward
    from util import rel_l2, round_sd
    root = tmp_path / "ckpt"
    dcfg, dsd, vcfg, vsd = model_dir.build(str(root), with_text=False)
    pipe = PyramidDiTForVideoGeneration(str(root), model_name="pyramid_flux", model_variant="diffusion_transformer_384p",
                                        load_text_encoder=False, load_vae=False)
    new = round_sd(synth.random_state_dict(synth.flux_param_shapes(dcfg), seed=77, std=0.05, lively=True))
    ckpt = {"dit." + k: v.to(torch.bfloat16) for k, v in new.items()}
    ckpt["vae.decoder.conv_in.conv.weight"] = torch.zeros(3)            # must be skipped (:217-218)
    ckpt["text_encoder.dummy"] = torch.zeros(1)
    path = tmp_path / "model.pth"
    torch.save(ckpt, str(path))
    g = torch.Generator().manual_seed(1)
    clips = [torch.randn(2, 16, *s, generator=g).to(torch.bfloat16).float() for s in [(1, 8, 16), (1, 16, 32), (1, 16, 32)]]
    enc = torch.randn(2, 16, dcfg["joint_attention_dim"], generator=g).to(torch.bfloat16).float()
    mask = torch.zeros(2, 16, dtype=torch.long)
    mask[0, :5] = 1
    mask[1, :12] = 1
    pooled = torch.randn(2, dcfg["pooled_projection_dim"], generator=g)
    t = torch.tensor([500.0, 500.0])
    before = pipe.dit.forward([c.cuda() for c in clips], enc, mask, pooled, t).cpu()
    pipe.load_checkpoint(str(path))
    after = pipe.dit.forward([c.cuda() for c in clips], enc, mask, pooled, t).cpu()
    ref = flux_forward(new, dcfg, clips, enc, mask, pooled, t)
    err = rel_l2(after, ref)
    print(f"load_checkpoint(.pth) DiT forward rel-L2 vs oracle {err:.3e}")
    assert err < 2e-2
    assert rel_l2(before, ref) > 0.5
