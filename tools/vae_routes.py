"""Which kernel serves every convolution of the tiled / chunked decode, WITHOUT a GPU: the layer programs of vae.py are walked on the
CPU with the compute entry points of the library stubbed out, and every pf_conv_desc is handed to the host-side routing functions
(pf_conv3d_which, pf_conv3d_fuses_gn_stats -- the decisions pf_conv3d_bf16 itself takes).  Prints, per tile geometry of a 768 x 1280
decode (32 x 32, 24 x 32, 32 x 16, 24 x 16 latents) and chunk (first / later), one line per conv: shape, route, FLOP, share.

    python tools/vae_routes.py            # config C5 / C3 decode: 768 x 1280, tiled(256), chunk windows of 4 latent frames
"""
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as L, ops, synth          # noqa: E402
from pyflow_hip import vae as V                      # noqa: E402

ROUTE = {-2: "conv_halo128", -1: "conv_narrow", 8: "gemm8p<conv>", 0: "gemm_kernel<true>"}


def decode_routes(verbose=False, frames=31, H=96, W=160):
    """-> {route name: FLOP per video}, walking the tile programs of a tiled(256) / chunked decode of `frames` latent frames of H x W on
    the CPU.  The library's compute entry points are stubbed for the duration of the call; pf_conv3d_which is the real one."""
    real = L.load()
    rows = []

    class Stub:
        """every entry point returns 0 without touching a device; the routing queries go to the real library"""

        def __getattr__(self, name):
            if name in ("pf_conv3d_which", "pf_conv3d_fuses_gn_stats", "pf_gemm_which", "pf_gemm_which_desc", "pf_last_error",
                        "pf_gemm_workspace_bytes", "pf_abi_version"):
                return getattr(real, name)
            if name == "pf_conv3d_bf16":
                def rec(dref, _stream):
                    d = dref._obj
                    route = int(real.pf_conv3d_which(dref))
                    fl = 2.0 * d.T * d.H * d.W_ * (d.n_valid or d.N) * d.kt * d.kh * d.kw * d.Cin
                    rows.append((d.T, d.H, d.W_, d.Cin, d.N, d.kt * d.kh * d.kw, (d.st, d.sh, d.sw), bool(d.flags & L.GEMM_GATE_RES),
                                 ROUTE.get(route, f"gemm256<{route},conv>"), fl / 1e9))
                    return 0
                return rec
            return lambda *a, **k: 0

    stub = Stub()
    saved = (L.load, L.stream, V.stream, ops.stream, ops.gemm)
    L.load = lambda: stub                       # vae.py / ops.py reach the library through lib.load()
    L.stream = V.stream = ops.stream = lambda: C.c_void_p(0)
    ops.gemm = lambda *a, **k: None
    try:
        sd = {k: torch.zeros(s_) for k, s_ in synth.vae_decoder_param_shapes(synth.VAE_DEFAULT).items()}
        vae = V.CausalVideoVAE(sd, synth.VAE_DEFAULT, "cpu")
        vae.enable_tiling()
        sizes = tuple(vae.chunk_sizes(frames, 1 * max(1, vae.chunk_coalesce), True))
        if verbose:
            print("chunk sizes (latent frames):", sizes)
        z = torch.zeros(16, frames, H, W)
        tl, ov = 32, 24
        count = defaultdict(int)
        for i in range(0, H, ov):
            for j in range(0, W, ov):
                count[(min(tl, H - i), min(tl, W - j))] += 1
        ntiles = sum(count.values())
        tot = defaultdict(float)
        for (th, tw), ntile in count.items():
            prog = vae._program(th, tw, sizes)
            prog.reset()
            for ci, nt in enumerate(sizes[:2]):
                del rows[:]
                prog.run_chunk(z, 0, nt, 0, 0, ci == 0, torch.zeros(1), 0, (1.0, 0.0, 1.0, 0.0))
                mult = ntile * (1 if ci == 0 else len(sizes) - 1)
                if verbose:
                    print(f"\n== tile {th} x {tw} latents ({ntile} of {ntiles}), chunk {ci} ({nt} latent frames; x{mult} per video)")
                for r in rows:
                    if verbose:
                        print("   T=%3d %3dx%-3d Cin=%3d N=%3d taps=%2d up=%s res=%d  %-22s %7.2f GFLOP" % r)
                    tot[r[8]] += r[9] * 1e9 * mult
        return dict(tot)
    finally:
        L.load, L.stream, V.stream, ops.stream, ops.gemm = saved


if __name__ == "__main__":
    tot = decode_routes(verbose=True)
    s = sum(tot.values())
    print("\nFLOP per video by route:")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"   {k:24s} {v / 1e15:7.3f} PFLOP  {100 * v / s:5.1f} %")
