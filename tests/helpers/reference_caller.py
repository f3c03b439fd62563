"""The call sequence of the reference's multi-GPU inference script (inference_multigpu.py:30-117), statement by
statement, against the drop-in packages -- only `sys.path` points here instead of at the reference checkout, and
`diffusers.utils.export_to_video` (third-party, absent in this image) is this repository's `export_to_video`.
World size 1 over RCCL (the GPU test box has one GPU); the model directory is a tiny seeded checkpoint in the
diffusers layout (tests/helpers/model_dir.py).  argv: <model_dir> <out_dir> <task>"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))

import torch                                                                                   # noqa: E402
from pyflow_hip.video_io import export_to_video as _export                                      # noqa: E402
_du = types.ModuleType("diffusers.utils")
_du.export_to_video = _export
sys.modules.setdefault("diffusers", types.ModuleType("diffusers"))
sys.modules["diffusers.utils"] = _du

# ---- from here on: the reference caller's own statements (inference_multigpu.py) ------------------------------------
from diffusers.utils import export_to_video                                                     # noqa: E402  (:9)
from pyramid_dit import PyramidDiTForVideoGeneration                                             # noqa: E402  (:10)
from trainer_misc import init_distributed_mode, init_sequence_parallel_group                    # noqa: E402  (:11)
from PIL import Image                                                                           # noqa: E402  (:13)


def main():
    model_dir, out_dir, task = sys.argv[1], sys.argv[2], sys.argv[3]
    args = argparse.Namespace(model_name="pyramid_flux", model_dtype="bf16", model_path=model_dir,
                              variant="diffusion_transformer_384p", task=task, temp=2, sp_group_size=1, sp_proc_num=-1)
    init_distributed_mode(args)                                                                  # :34
    assert args.world_size == args.sp_group_size, "The sequence parallel size should be DDP world size"   # :36
    init_sequence_parallel_group(args)                                                           # :39
    device = torch.device("cuda")
    rank = args.rank
    model_dtype = args.model_dtype
    model = PyramidDiTForVideoGeneration(args.model_path, model_dtype, model_name=args.model_name,
                                         model_variant=args.variant)                            # :45-50
    model.vae.to(device)                                                                         # :52
    model.dit.to(device)
    model.text_encoder.to(device)
    model.vae.enable_tiling()                                                                    # :55
    torch_dtype = torch.bfloat16 if model_dtype == "bf16" else torch.float32
    width, height = 640, 384                                                                     # :68-70 (384p variant)
    if args.task == "t2v":
        prompt = "A movie trailer featuring the adventures of the 30 year old space man"
        with torch.no_grad(), torch.autocast("cuda", enabled=model_dtype != "fp32", dtype=torch_dtype):
            frames = model.generate(prompt=prompt, num_inference_steps=[2, 2, 2], video_num_inference_steps=[1, 1, 1],
                                    height=height, width=width, temp=args.temp, guidance_scale=7.0,
                                    video_guidance_scale=5.0, output_type="pil", save_memory=True, cpu_offloading=False,
                                    inference_multigpu=True)                                      # :75-90
        if rank == 0:
            export_to_video(frames, os.path.join(out_dir, "text_to_video_sample.y4m"), fps=24)   # :91-92
    else:
        image = Image.new("RGB", (100, 60), (120, 80, 40)).resize((width, height))              # :97-99
        prompt = "FPV flying over the Great Wall"
        with torch.no_grad(), torch.autocast("cuda", enabled=model_dtype != "fp32", dtype=torch_dtype):
            frames = model.generate_i2v(prompt=prompt, input_image=image, num_inference_steps=[1, 1, 1], temp=args.temp,
                                        video_guidance_scale=4.0, output_type="pil", save_memory=True,
                                        cpu_offloading=False, inference_multigpu=True)           # :103-113
        if rank == 0:
            export_to_video(frames, os.path.join(out_dir, "image_to_video_sample.y4m"), fps=24)
    torch.distributed.barrier()                                                                  # :119
    n = 1 + 8 * (args.temp - 1)
    assert len(frames) == n and frames[0].size == (width, height), (len(frames), frames[0].size)
    print(f"reference caller sequence ok: {len(frames)} frames {frames[0].size}")


if __name__ == "__main__":
    main()
