"""worker of tests/test_sp_gpu.py::test_sp_generate: P ranks (sharing cuda:0, gloo transport) run the whole
sequence-parallel generate() incl. the tile-parallel VAE decode; rank 0 compares with the single-process pipeline.
argv[2] == "cfg": two ranks in GUIDANCE-parallel mode instead (pyflow_hip/flux_cfg.py: one CFG branch per rank)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def build(g, sp):
    from pyflow_hip import synth
    from pyflow_hip.pipeline import PyramidDiTForVideoGeneration
    from util import round_sd
    dsd = round_sd(synth.random_state_dict(synth.flux_param_shapes(g["dit_cfg"]), seed=g["dit_weight_seed"], std=0.05, lively=True))
    vsd = round_sd(synth.random_state_dict(synth.vae_decoder_param_shapes(g["vae_cfg"]), seed=g["vae_weight_seed"], std=0.05, lively=True))
    pipe = PyramidDiTForVideoGeneration(dit_state_dict=dsd, dit_config=g["dit_cfg"], vae_state_dict=vsd,
                                        vae_config=g["vae_cfg"], model_name="pyramid_flux")
    assert (pipe.sp is not None) == bool(sp)
    if sp == "cfg":
        from pyflow_hip.flux_cfg import FluxEngineCFG
        assert isinstance(pipe.dit, FluxEngineCFG)
    elif sp:
        # the bitwise comparison below needs the single-process summation order: a rank's small GEMMs must not split K
        pipe.dit.split_small = False
    pipe.vae.enable_tiling()
    return pipe


def run(pipe, g, tile):
    e, m, p = g["prompt_embeds"], g["prompt_mask"], g["pooled"]
    emb = (e[1:2], m[1:2], p[1:2], e[0:1], m[0:1], p[0:1])
    torch.manual_seed(1234)         # block noise comes from the global CPU generator
    lat = pipe.generate(prompt_embeds=emb, height=g["height"], width=g["width"], temp=g["temp"],
                        num_inference_steps=g["steps"], video_num_inference_steps=g["video_steps"],
                        guidance_scale=g["guidance"], video_guidance_scale=g["video_guidance"],
                        generator=torch.Generator().manual_seed(g["latent_seed"]), output_type="latent")
    z = lat.float()
    aff = (1.0 / pipe.vae_scale_factor, pipe.vae_shift_factor, 1.0 / pipe.vae_video_scale_factor, pipe.vae_video_shift_factor)
    u8 = pipe.vae.decode_to_uint8(z, window_size=1, tile_sample_min_size=tile, affine=aff, comm=pipe.sp)
    return lat, u8


def main():
    out_path = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "sp"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = torch.load(os.path.join(ROOT, "tests", "golden", "generate_tiny_latents.pt"))
    from pyflow_hip import sp as sp_mod
    ok = True
    ref = None
    if rank == 0:                    # single-process result first (before the SP group exists)
        ref = run(build(g, False), g, 32)
    sp_mod.init_sequence_parallel_group(sp_group_size=world, guidance_parallel=(mode == "cfg"))
    lat, u8 = run(build(g, "cfg" if mode == "cfg" else True), g, 32)
    torch.cuda.synchronize()
    if rank == 0:
        same_lat = torch.equal(lat.cpu(), ref[0].cpu())
        same_u8 = u8 is not None and torch.equal(u8.cpu(), ref[1].cpu())
        ok = same_lat and same_u8
        rel = ((lat.float().cpu() - ref[0].float().cpu()).norm() / ref[0].float().cpu().norm()).item()
        if mode == "cfg":
            # one branch per rank = the batch-of-2 forward up to the GEMMs' fp32 summation order (the tile walk depends on the
            # row count); frames: a uint8 step here and there
            du8 = (u8.cpu().int() - ref[1].cpu().int()).abs().max().item() if u8 is not None else 999
            ok = rel < 2e-3 and du8 <= 2
        with open(out_path, "w") as f:
            f.write(f"mode={mode} latents_equal={same_lat} rel={rel:.2e} frames_equal={same_u8} frames={tuple(ref[1].shape)} world={world}\n")
    else:
        ok = u8 is None
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
