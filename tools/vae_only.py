"""one 256 x 256-pixel tile of the tiled / chunked CausalVideoVAE decode at the launch shapes of the timed decode
(32 x 32 latent, 5 latent frames = the first coalesced chunk window -> 33 frames), one lane, repeated argv[1] times -- the
target of the rocprofv3 --pmc passes over the VAE conv / GroupNorm kernels (tools/gpu_pmc.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import synth
from pyflow_hip.vae import CausalVideoVAE
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
sd = {}
for k, shp in synth.vae_decoder_param_shapes(synth.VAE_DEFAULT).items():
    sd[k] = (torch.ones(shp, device=dev) if k.endswith(".weight") else torch.zeros(shp, device=dev)) if len(shp) == 1 \
        else torch.randn(shp, generator=g, device=dev) * 0.02
vae = CausalVideoVAE(sd, synth.VAE_DEFAULT, dev)
vae.enable_tiling()
vae.n_streams = 1
z = torch.randn(1, 16, 5, 32, 32, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = vae.decode_to_uint8(z, window_size=1, tile_sample_min_size=256)
torch.cuda.synchronize()
print("vae tile done", tuple(out.shape))
