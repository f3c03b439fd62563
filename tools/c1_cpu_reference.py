"""Config C1 timed FOR REAL on the CPU: the UNMODIFIED reference (imported from /root/reference through
oracle/shims.py, dev container only) runs `generate()` for one 1024 x 1024 image with ONE pyramid stage, 20 steps,
guidance 9 -- miniFLUX at the released dimensions (1.97 B parameters, random init), default CausalVideoVAE, fp32,
all host cores.  Prints seconds per phase; the output goes to profiles/r02_c1_cpu_reference.log."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from oracle import ref_harness as rh                                                         # noqa: E402
from oracle import shims                                                                     # noqa: E402

MINIFLUX = dict(num_layers=8, num_single_layers=16, num_attention_heads=30, attention_head_dim=64, in_channels=64,
                joint_attention_dim=4096, pooled_projection_dim=768, axes_dims_rope=[16, 24, 24])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    ref = shims.load_reference()
    dit = ref.PyramidFluxTransformer(**MINIFLUX).eval()
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in dit.named_parameters():
            p.copy_(torch.ones(p.shape) if (p.ndim == 1 and n.endswith("weight")) else
                    (torch.zeros(p.shape) if p.ndim == 1 else 0.02 * torch.randn(p.shape, generator=g)))
    vae = ref.CausalVideoVAE(encoder_out_channels=16, decoder_in_channels=16).eval()
    rh.seed_weights(vae, 4321, std=0.02)
    vae.enable_tiling()
    pipe = rh.build_ref_pipeline(dit, vae, stages=(1,), text_encoder=rh.StubTextEncoder(Lt=128, C=4096, Cp=768))
    print(f"build: {time.time() - t0:.1f} s, threads {torch.get_num_threads()}, "
          f"dit params {sum(p.numel() for p in dit.parameters()) / 1e9:.3f} B", flush=True)
    calls = []
    orig = dit.forward

    def timed_forward(*a, **k):
        t = time.time()
        out = orig(*a, **k)
        calls.append(time.time() - t)
        print(f"  DiT forward {len(calls)}: {calls[-1]:.1f} s", flush=True)
        return out
    dit.forward = timed_forward
    t0 = time.time()
    with torch.no_grad():
        frames = pipe.generate(prompt="a photo of a lighthouse", height=1024, width=1024, temp=1,
                               num_inference_steps=[steps], guidance_scale=9.0, video_guidance_scale=5.0,
                               output_type="pil", save_memory=True, generator=torch.Generator().manual_seed(0))
    total = time.time() - t0
    print(f"C1 on the CPU reference: {len(frames)} image {frames[0].size}, total {total:.1f} s = "
          f"{sum(calls):.1f} s DiT ({len(calls)} forwards, mean {sum(calls) / len(calls):.1f} s, L = 4224, B = 2) + "
          f"{total - sum(calls):.1f} s VAE tiled decode / host; {1.0 / total:.5f} images/s on {torch.get_num_threads()} cores",
          flush=True)


if __name__ == "__main__":
    main()
