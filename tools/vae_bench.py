"""CausalVideoVAE decode benchmark at the C3 / C5 shape: latent [1,16,31,96,160] -> 241 x 768 x 1280 uint8 frames,
reference mode tiled(256) / chunked(1).  usage: vae_bench.py [streams:coalesce[:nofuse] ...]
(":nofuse" = GroupNorm statistics by the separate pf_gn_stats pass instead of the conv epilogues: the round-3 A/B)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as _lib
if os.environ.get("PF_BENCH_LIB"):          # measurement only: A/B against another build of the library
    _lib.LIB_PATH = os.path.join(ROOT, "pyramid-flow_amd", os.environ["PF_BENCH_LIB"])
from pyflow_hip import synth, ops
from pyflow_hip.vae import CausalVideoVAE
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
sd = {}
for k, shp in synth.vae_decoder_param_shapes(synth.VAE_DEFAULT).items():
    sd[k] = (torch.ones(shp, device=dev) if k.endswith(".weight") else torch.zeros(shp, device=dev)) if len(shp) == 1 \
        else torch.randn(shp, generator=g, device=dev) * 0.02
vae = CausalVideoVAE(sd, synth.VAE_DEFAULT, dev)
vae.enable_tiling()
T = int(os.environ.get("VAE_T", 31))
z = torch.randn(1, 16, T, 96, 160, device=dev)
# args: "streams:coalesce" pairs, e.g. 4:1 4:4 4:8
for arg in sys.argv[1:] or ["1:4", "4:4"]:
    ns, co = (int(a) for a in (arg.split(":") + ["4"])[:2])
    vae.n_streams, vae.chunk_coalesce = ns, co
    vae.fuse_gn_stats = not arg.endswith(":nofuse")
    vae._programs.clear()
    vae._lane_pools.clear()
    import gc
    gc.collect()                 # programs <-> buffers hold reference cycles: without this the previous setting's pools stay allocated
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    vae.decode_to_uint8(z[:, :, :2], window_size=1, tile_sample_min_size=256)      # warm-up / allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vae.decode_to_uint8(z, window_size=1, tile_sample_min_size=256)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n_streams={ns} chunk_coalesce={co} gn statistics {'in the conv epilogues' if vae.fuse_gn_stats else 'by pf_gn_stats'}: peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  {dt:.2f} s for {out.shape[0]} frames ({7.51e15 * (out.shape[0] / 241) / dt / 1e12:.0f} TFLOP/s conv-equivalent) "
          f"checksum {int(out[::16, ::64, ::64].sum())}", flush=True)
