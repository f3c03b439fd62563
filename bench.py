#!/usr/bin/env python
"""bench.py -- headline benchmark of BASELINE.json: video frames/s (whole node) for 768p (768x1280) 241-frame
text-to-video sampling with the miniFLUX pyramid DiT + CausalVideoVAE decode, synthetic prompts and random-init
weights of the released architecture.

A "step" = one complete video: generate() -> uint8 frames resident on the device (text encoding excluded, as in
SURVEY 8d).  python bench.py --gpus N --steps K --warmup W ; prints ONE JSON line on rank 0.
N > 1: ONE video per step sampled by all N GPUs together -- sequence-parallel DiT (Ulysses all-to-all over RCCL,
pyflow_hip/flux_sp.py) + tile-parallel VAE decode, frames assembled on rank 0 ("scaling": "strong");
`--parallelism replicas` instead runs N independent videos (no data-path collective, "weak").
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (height, width, temp, steps_first, steps_video)
    "c3_768p_241f": (768, 1280, 31, [20, 20, 20], [10, 10, 10]),
    # config C4: SD3-style MMDiT image-to-video, VAE encode + decode in the loop (generate_i2v, temp 16 -> 121 frames)
    "c4_i2v_768p_121f": (768, 1280, 16, [10, 10, 10], [10, 10, 10]),
    "c2_384p_121f": (384, 640, 16, [20, 20, 20], [10, 10, 10]),
    # config C1: miniFLUX 1024 x 1024 single image, ONE pyramid stage, 20 steps, guidance 9
    "c1_1024p_image": (1024, 1024, 1, [20], [20]),
    # config C5: standalone CausalVideoVAE decode of a 768p 241-frame latent (no DiT): the reference's tiled(256) /
    # chunked(1) schedule on one GPU; with N GPUs the temporal context-parallel decode (un-tiled, halo exchange per
    # causal conv) when a rank's frame range fits in HBM, otherwise the tile-parallel form of the tiled decode
    "c5_vae_768p_241f": (768, 1280, 31, None, None),
    "smoke_128p_17f": (128, 192, 3, [4, 4, 4], [2, 2, 2]),
    # the headline GEOMETRY (768 x 1280: 240 / 960 / 3 840 tokens per latent frame, 28 decode tiles) on a short schedule --
    # 3 units, 2 steps per stage = 18 forwards at L = 368 ... 8 768: plumbing runs of the N > 1 paths with many ranks on one
    # test GPU, where the 960 forwards of the real schedule take a quarter of an hour through gloo
    "c3geom_768p_17f": (768, 1280, 3, [2, 2, 2], [2, 2, 2]),
}
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0          # HBM3E (same guide)
PMC_PROFILE = "r06_pmc_forward_maxL.json"      # committed rocprofv3 --pmc passes the DiT kernels' `traffic` fields are replayed from
PMC_PROFILE_VAE = "r06_pmc_vae_tile.json"      # ... and the VAE kernels' (one tile-chunk window at the timed launch shapes)


def build_pipeline(device, tiny=False, mmdit=False, stages=None):
    from pyflow_hip import synth
    from pyflow_hip.pipeline import PyramidDiTForVideoGeneration
    if mmdit:
        dcfg = synth.tiny_mmdit_cfg() if tiny else synth.SD3_MMDIT
    else:
        dcfg = synth.TINY_FLUX if tiny else synth.MINIFLUX
        if tiny and int(os.environ.get("WORLD_SIZE", 1)) > 4:
            # plumbing runs on more ranks than the tiny model has heads: 10 heads = the 2|2|1|1|1|1|1|1 analogue of 30 over 8
            dcfg = dict(dcfg, num_attention_heads=10)
    vcfg = dict(synth.TINY_VAE if tiny else synth.VAE_DEFAULT)
    g = torch.Generator(device=device).manual_seed(1234)

    def rand_sd(shapes):
        # N(0, 0.02^2) matrices, norm gains 1, biases 0, generated on the device
        sd = {}
        for k, shp in shapes.items():
            if len(shp) == 1:
                sd[k] = torch.ones(shp, device=device) if k.endswith(".weight") else torch.zeros(shp, device=device)
            else:
                sd[k] = torch.randn(shp, generator=g, device=device) * 0.02
        return sd
    if mmdit:
        dsd = rand_sd({k: v for k, v in synth.mmdit_param_shapes(dcfg).items() if k != "pos_embed.pos_embed"})
        d = dcfg["num_attention_heads"] * dcfg["attention_head_dim"]
        dsd["pos_embed.pos_embed"] = synth.sincos_2d_table(d, dcfg["pos_embed_max_size"], dcfg["sample_size"] // dcfg["patch_size"])[None]
    else:
        dsd = rand_sd(synth.flux_param_shapes(dcfg))
    vsd = rand_sd(synth.vae_decoder_param_shapes(vcfg))
    if mmdit:        # image-to-video needs the encoder
        ecfg = synth.TINY_VAE_ENC if tiny else synth.VAE_ENC_DEFAULT
        vsd.update(rand_sd(synth.vae_encoder_param_shapes(ecfg)))
        vcfg.update(ecfg)
    pipe = PyramidDiTForVideoGeneration(dit_state_dict=dsd, dit_config=dcfg, vae_state_dict=vsd, vae_config=vcfg,
                                        model_name="pyramid_mmdit" if mmdit else "pyramid_flux", model_dtype="bf16",
                                        device=device, **(dict(stages=[1], stage_range=[0, 1], sample_ratios=[1])
                                                          if stages == 1 else {}))
    pipe.vae.enable_tiling()                      # reference inference setup (inference_multigpu.py:52-55)
    return pipe, dcfg, dsd


def synthetic_prompt(dcfg, device):
    g = torch.Generator().manual_seed(1235)
    Lt = 128
    e = torch.randn(2, Lt, dcfg["joint_attention_dim"], generator=g).to(torch.bfloat16)
    p = torch.randn(2, dcfg["pooled_projection_dim"], generator=g).to(torch.bfloat16)
    m = torch.zeros(2, Lt, dtype=torch.long)
    m[0, :40] = 1       # negative prompt
    m[1, :96] = 1       # positive prompt
    return (e[1:2], m[1:2], p[1:2], e[0:1], m[0:1], p[0:1])


class SampledProfiler:
    """turns the per-launch HIP-event profiler on for every `period`-th DiT forward of the timed region."""

    def __init__(self, pipe, period):
        from pyflow_hip import ops
        self.prof = ops.PROFILER
        self.count = 0
        self.period = period
        eng = pipe.dit
        orig = eng.forward_tokens

        def wrapped(*a, **k):
            self.prof.enabled = self.active and (self.count % self.period == 0)
            self.count += 1
            # a sampled forward runs its launches back-to-back on ONE stream, so that each HIP-event pair brackets one
            # kernel running alone (the side-stream text path would overlap two kernels inside the bracket)
            keep = eng.overlap_text
            eng.overlap_text = keep and not self.prof.enabled
            try:
                return orig(*a, **k)
            finally:
                self.prof.enabled = False
                eng.overlap_text = keep
        eng.forward_tokens = wrapped
        self.active = False


def c3_schedule():
    """(L, forwards) of every (unit, stage) of the headline job: 768p, temp 31, steps [20]*3 / [10]*3, text 128
    (sequence = [text | history clips | current frame])."""
    tok = [240, 960, 3840]                       # tokens per latent frame at stage 0, 1, 2
    out = []
    for u in range(31):
        for s_ in range(3):
            # frame u-1 at the stage's resolution, frame u-2 one stage lower, everything older at stage 0
            hist = (tok[s_] if u >= 1 else 0) + (tok[max(s_ - 1, 0)] if u >= 2 else 0) + max(u - 2, 0) * tok[0]
            out.append((128 + hist + tok[s_], 20 if u == 0 else 10))
    return out


def usable_cpus():
    """CPUs this process may actually use: the smaller of the machine's count, the affinity mask and the cgroup CPU quota (the GPU
    boxes of the pool show 256 logical CPUs and grant a container 16 of them: cpu.max = "1600000 100000").  Threads beyond the
    quota are throttled by the scheduler, which made the round-5 / 6 CPU baselines -- taken with 64 threads -- move by 30-50 %
    from box to box and run to run."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:                                                   # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(dcfg, dsd, threads):
    """The reference path on the host cores, SURVEY 8d recipe, kind "port": the CPU fp32 oracle restatement (oracle/,
    pinned to the imported reference in the dev container) -- /root/reference itself does not exist on the GPU box, so
    the reference cannot be imported where this runs.  Bounded sample (~85 s of CPU work on the CPUs the container may use; every timing is the fastest of three
    samples, their spread is reported):
      * one double-stream + one single-stream miniFLUX block at full width (d = 1920, 30 heads, CFG batch 2) inside a
        complete oracle forward at six sequence lengths of the schedule (unit 0 stages 0-2, unit 1 stages 0-2: up to L = 7 808,
        half of the longest sequence, so that the quadratic term is measured and not extrapolated from L <= 3 968);
        t(L) = a L + b L^2 is fitted (GEMM / attention terms), residuals reported, and summed over the 960 forwards of
        the job x 12 (24 blocks = 12 x the sampled pair);
      * one VAE tile-chunk: a 32 x 32 latent tile, one latent frame, full channel widths (fastest of 3) -> scaled to 28 tiles
        x 241 output frames (first-chunk cost per output frame);
      * the reference's per-block Python sampling loop of the block noise (pipeline.py:697-703): 2 000 draws timed,
        scaled to the 76 800 draws of each of the 30 video units."""
    import numpy as np
    from oracle.flux_oracle import flux_forward
    from oracle.vae_oracle import vae_decode
    from pyflow_hip import synth
    threads = max(1, min(threads, 64))          # beyond ~64 threads the CPU GEMMs of this size slow down
    torch.set_num_threads(threads)
    cfg = dict(dcfg, num_layers=1, num_single_layers=1)
    sd = {k: v.float().cpu() for k, v in dsd.items()
          if not (k.startswith("transformer_blocks.") and not k.startswith("transformer_blocks.0."))
          and not (k.startswith("single_transformer_blocks.") and not k.startswith("single_transformer_blocks.0."))}
    g = torch.Generator().manual_seed(9)
    enc = torch.randn(2, 128, dcfg["joint_attention_dim"], generator=g)
    mask = torch.zeros(2, 128, dtype=torch.long)
    mask[0, :40] = 1
    mask[1, :96] = 1
    pooled = torch.randn(2, dcfg["pooled_projection_dim"], generator=g)
    points = {368: [(1, 24, 40)], 608: [(1, 24, 40), (1, 24, 40)], 1088: [(1, 48, 80)],
              2048: [(1, 48, 80), (1, 48, 80)], 3968: [(1, 96, 160)], 7808: [(1, 96, 160), (1, 96, 160)]}
    meas, spread = [], []
    t_budget = time.time()
    with torch.no_grad():
        for L, shapes in points.items():
            clips = [torch.randn(2, 16, *s_, generator=g) for s_ in shapes]
            ts_ = torch.tensor([900.0, 900.0])
            flux_forward(sd, cfg, clips, enc, mask, pooled, ts_)          # warm-up (allocator, threads)
            samples = []                                                  # three samples of >= ~3.5 s each; the MEDIAN is used
            for _ in range(3):
                reps, t0 = 0, time.time()
                while reps < 12 and (reps == 0 or time.time() - t0 < 3.5):
                    flux_forward(sd, cfg, clips, enc, mask, pooled, ts_)
                    reps += 1
                samples.append((time.time() - t0) / reps)
            samples.sort()
            spread.append((samples[2] - samples[0]) / samples[1])
            # the FASTEST sample: on a shared host (load average ~20 from other tenants on the pool's boxes) disturbances only ever
            # add time, so the minimum is the reproducible estimate of the undisturbed run; the spread of the three is reported
            meas.append((L, samples[0]))
    Ls = np.array([m[0] for m in meas], dtype=np.float64)
    tt = np.array([m[1] for m in meas], dtype=np.float64)
    A = np.stack([Ls, Ls * Ls], axis=1)
    coef, *_ = np.linalg.lstsq(A / tt[:, None], np.ones_like(tt), rcond=None)      # least squares on RELATIVE error
    resid = (A @ coef - tt) / tt
    dit_s = sum(n * max(coef[0] * L + coef[1] * L * L, 0.0) for L, n in c3_schedule()) * 12.0
    # VAE tile-chunk
    vcfg = synth.VAE_DEFAULT
    vshapes = synth.vae_decoder_param_shapes(vcfg)
    gv = torch.Generator().manual_seed(10)
    vsd = {k: (torch.ones(sh) if k.endswith(".weight") else torch.zeros(sh)) if len(sh) == 1
           else torch.randn(sh, generator=gv) * 0.02 for k, sh in vshapes.items()}
    ocfg = dict(decoder_block_out_channels=vcfg["block_out_channels"], decoder_layers_per_block=vcfg["layers_per_block"],
                decoder_spatial_up_sample=vcfg["spatial_up_sample"], decoder_temporal_up_sample=vcfg["temporal_up_sample"])
    z = torch.randn(1, 16, 1, 32, 32, generator=gv)
    with torch.no_grad():
        tl = []
        for _ in range(3):
            t0 = time.time()
            vae_decode(vsd, ocfg, z)
            tl.append(time.time() - t0)
        t_tile = sorted(tl)[0]
    vae_s = t_tile * 28 * 241
    # block-noise loop of the reference (per-block MultivariateNormal.sample() in Python)
    # (the covariance is singular at gamma = 1/3: whether torch's Cholesky accepts it depends on the CPU, so the factor is
    # handed over -- the cost being measured is the per-block Python call, not the factorisation)
    from oracle.pipeline_oracle import block_noise_cholesky
    dist_ = torch.distributions.multivariate_normal.MultivariateNormal(
        torch.zeros(4), scale_tril=block_noise_cholesky(1.0 / 3.0), validate_args=False)
    t0 = time.time()
    for _ in range(2000):
        dist_.sample()
    noise_s = (time.time() - t0) / 2000 * (15360 + 61440) * 30
    est = dit_s + vae_s + noise_s
    return dict(value=float(241.0 / est), unit="frames/s", cores=threads, kind="port", fit_residuals=[round(float(r), 3) for r in resid],
                sample_spread=[round(x, 3) for x in spread],
                sample=("oracle (CPU fp32 restatement of the reference; /root/reference is absent on the GPU box) on "
                        f"{threads} threads (= the CPUs this container may use: machine {os.cpu_count()}, cgroup quota / affinity "
                        f"{usable_cpus()}), {time.time() - t_budget:.0f} s of CPU work: 1 double + 1 single miniFLUX block at "
                        f"full width inside a complete forward at L = {[m[0] for m in meas]} -> {[round(m[1], 3) for m in meas]} s; "
                        f"(fastest of 3 samples per length, (max - min) / median {[round(x, 3) for x in spread]}); "
                        f"fit t = {coef[0]:.3e} L + {coef[1]:.3e} L^2 (relative residuals {[round(float(r), 3) for r in resid]}), "
                        f"summed over the 960 forwards x 12 = {dit_s:.0f} s DiT; one 32x32x1-latent VAE tile-chunk {t_tile:.2f} s "
                        f"x 28 tiles x 241 frames = {vae_s:.0f} s; reference block-noise Python loop {noise_s:.0f} s; "
                        f"total {est:.0f} s per 241-frame video"))


def pmc_traffic_in_run(timeout_s=300):
    """HBM-side traffic of the DiT's kernels MEASURED in this run (rank 0, N = 1): two child passes of
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE` (separate passes, nothing else traced: MI355X_MICROARCH.md, HBM section)
    over tools/forward_only.py -- two full-width forwards at the headline sequence L = 15 488, the launch shapes of the timed
    region's heaviest (unit, stage).  Returns ({kernel name: {launches, fetch_kb, write_kb, hbm_bytes_per_launch}}, note) in the
    format of profiles/rNN_pmc_forward_maxL.json, or (None, reason): the caller then replays the committed profile and says so.
    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 reports half of wide coalesced reads in FETCH_SIZE)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"pf_pmc_{ctr}_", dir="/tmp")
        try:
            pr = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--output-format", "csv", "--",
                                 sys.executable, os.path.join(ROOT, "tools", "forward_only.py"), "2"], cwd="/tmp",
                                env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"{ctr} pass wrote no counter file (rc {pr.returncode}): {pr.stderr.decode()[-200:]}"
            per = collections.OrderedDict()           # dispatch -> (kernel, summed counter over its rows)
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] != ctr:
                        continue
                    e = per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
                    e[1] += float(r["Counter_Value"])
            acc = collections.defaultdict(list)
            for name, v in per.values():
                acc[name].append(v)
            for name, lst in acc.items():
                vals.setdefault(name, {})[ctr] = (len(lst), sum(lst) / len(lst))
        except subprocess.TimeoutExpired:
            return None, f"{ctr} pass exceeded {timeout_s} s"
        except Exception as e:          # noqa: BLE001
            return None, f"{ctr} pass failed: {e!r}"[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for name, dct in vals.items():
        if "FETCH_SIZE" in dct and "WRITE_SIZE" in dct:
            f_, w_ = dct["FETCH_SIZE"][1], dct["WRITE_SIZE"][1]
            out[name] = dict(launches=dct["FETCH_SIZE"][0], fetch_kb=f_, write_kb=w_, hbm_bytes_per_launch=(2 * f_ + w_) * 1024)
    if not out:
        return None, "no kernel carried both counters"
    return out, ("(2*FETCH_SIZE + WRITE_SIZE) KB per launch, MEASURED IN THIS RUN: two rocprofv3 --kernel-trace --pmc child passes "
                 "(FETCH_SIZE, WRITE_SIZE) over tools/forward_only.py = 2 full-width forwards at L=15488 on this GPU, after the "
                 "timed region; mean over this kernel's launches; counted at the L2<->fabric interface incl. Infinity-Cache hits")


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: start one rank per GPU of this node (the reference's launcher is
    torchrun, scripts/inference_multigpu.sh:15-23; the contract each rank sees is the same -- RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT, trainer_misc/utils.py:72-88) and wait for them.  Rank 0 owns stdout (the one
    JSON line); the other ranks' stdout goes to stderr.  With fewer GPUs than ranks (a one-GPU test box) the ranks
    share the GPUs and the transport falls to gloo: plumbing only, flagged in the JSON line."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PF_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:
        env.setdefault("PF_DIST_BACKEND", "gloo")
        # ranks sharing one box's cores (plumbing runs): without a cap every rank's OpenMP pool spins on all of them
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    def die_with_launcher():
        # a launcher that is killed outright (a test harness's timeout, the driver's clock) must not leave rank processes
        # behind on the GPU: the kernel delivers SIGKILL to every rank when this process goes away (PR_SET_PDEATHSIG = 1)
        import ctypes
        import signal
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else sys.stderr, preexec_fn=die_with_launcher))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in alive:               # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--workload", default="c3_768p_241f", choices=list(WORKLOADS))
    ap.add_argument("--tiny-model", action="store_true", help="tiny random model (plumbing check, not a valid bench)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic in the run (N = 1, headline "
                         "workload); the committed profile is replayed instead and the line says so")
    ap.add_argument("--profile-period", type=int, default=21,
                    help="every N-th DiT forward of the timed region runs eagerly with a HIP-event pair around each launch (the "
                         "roofline sample).  21 is coprime with the schedule's 10 / 20 steps per stage, so the sample walks every stage "
                         "and length; the sampled forwards cost the step ~0.1 s at N = 7 (same-box A/B: 43.70 / 43.67 s vs 43.60 s with "
                         "no sample, 43.58 s at N = 28: profiles/r06_bench_sampling_period_ab.log)")
    ap.add_argument("--group-text", default="auto", choices=["auto", "on", "off"],
                    help="double blocks: the text stream's GEMMs inside the image stream's persistent launches (pf_gemm_desc.A2 ..., "
                         "FluxEngine.group_text) -- auto = the engine's default (A/B switch)")
    ap.add_argument("--no-overlap-text", action="store_true",
                    help="run the double blocks' text stream on the compute stream instead of the side stream (A/B switch)")
    ap.add_argument("--launch-mode", default="graph", choices=["eager", "list", "graph"],
                    help="how the DiT's launches reach the device (pyflow_hip/cmdlist.py); default = the engine's default")
    ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                    help="N > 1, sp: communicator -- auto = the C-ABI RCCL communicator (pf_comm_*) when its self-test passes on "
                         "every rank, else torch.distributed")
    ap.add_argument("--comm-windows", type=int, default=0, metavar="MIB",
                    help="N > 1, --comm native / auto: attach exchange windows with slots of MIB MiB to the C-ABI communicator: chunks "
                         "up to that size travel by the COPY ENGINES (IPC-mapped windows, stream memory ops) instead of RCCL "
                         "kernels.  0 (default) = RCCL only: the window transport has not run between two GPUs yet (DESIGN.md 5)")
    ap.add_argument("--gemm-policy", type=int, action="append", default=[],
                    help="A/B switch: pf_gemm_set_policy(value) before the model is built (e.g. -5 = no LDS-halo conv, -4 = no "
                         "tail split); repeatable.  Not for the headline line")
    ap.add_argument("--parallelism", default="auto", choices=["auto", "sp", "guidance", "replicas"],
                    help="N > 1: one video across all GPUs -- sp = sequence-parallel DiT (Ulysses all-to-all), guidance = the two "
                         "ranks of N = 2 take one classifier-free-guidance branch each (no all-to-all), auto (default) = guidance "
                         "at N = 2, sp otherwise; replicas = one video per GPU")
    args = ap.parse_args()
    # torch sizes its intra-op pool by the machine (128 threads on the pool's boxes), the container may use 16 CPUs: every
    # OpenMP region of a host-side op (the start noise's dtype conversion, small copies) then waits for throttled threads
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N` (no torchrun on the command line): this process becomes the launcher
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    use_sp = world > 1 and args.parallelism != "replicas"
    guidance = use_sp and world == 2 and args.parallelism in ("auto", "guidance")
    comm_used = None
    comm_errors = []
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # PF_DIST_BACKEND=gloo: plumbing test of the N > 1 path on a box with fewer GPUs than ranks (tests only)
        dist.init_process_group(os.environ.get("PF_DIST_BACKEND", "nccl"))
        if use_sp:      # must exist before the model is built (reference contract, inference_multigpu.py:34-39)
            from pyflow_hip import sp as sp_mod

            def agreed(ok_local):
                """True iff EVERY rank succeeded (control plane: the default process group)"""
                f = torch.tensor([0.0 if ok_local else 1.0], dtype=torch.float32,
                                 device=device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(f)
                return float(f.item()) == 0.0

            def try_comm(native):
                """build the communicator and run every collective of the path once with known values"""
                try:
                    c = sp_mod.init_sequence_parallel_group(sp_group_size=world, native=native, guidance_parallel=guidance,
                                                            window_mib=args.comm_windows if native else 0)
                    c.selftest(device)
                    return c, True
                except Exception as e:          # noqa: BLE001
                    comm_errors.append(f"{'pf_comm' if native else 'torch.distributed'}: {e!r}"[:300])
                    print(f"[bench] rank {rank}: sequence-parallel self-test ({'pf_comm' if native else 'torch.distributed'}) "
                          f"failed: {e!r}", file=sys.stderr, flush=True)
                    return None, False
            comm = None
            # the C-ABI communicator first (its exchanges are launch-list entries: one C call per forward instead of ~350
            # Python launches + ~50 c10d calls); needs RCCL, i.e. one GPU per rank
            if args.comm in ("auto", "native") and dist.get_backend() == "nccl":
                comm, ok = try_comm(True)
                if not agreed(ok):
                    if ok:
                        comm_errors.append("pf_comm: another rank failed its self-test")
                    comm = None
                    sp_mod._SP = None
            elif args.comm in ("auto", "native"):
                # ranks sharing a GPU (test boxes): RCCL refuses duplicate devices, its init is not even attempted
                comm_errors.append("pf_comm: not attempted (backend gloo: the ranks share GPUs, RCCL needs one GPU per rank)")
            if comm is None and args.comm == "native":
                # an explicit request is strict: a line labelled "native" is never measured on another transport
                raise SystemExit("[bench] --comm native: the C-ABI communicator is not available on every rank: " +
                                 " | ".join(comm_errors) + "  (use --comm auto for the torch.distributed fallback)")
            if comm is None:          # torch.distributed: the agreed fallback of --comm auto, the choice of --comm torch
                comm, ok = try_comm(False)
                if not agreed(ok):
                    comm = None
            if comm is None:        # all ranks agree: independent replicas instead (reported as such in the JSON line)
                print(f"[bench] rank {rank}: no working sequence-parallel communicator; falling back to replicas",
                      file=sys.stderr, flush=True)
                use_sp = guidance = False
                sp_mod._SP = None
            comm_used = comm

    if args.gemm_policy:
        from pyflow_hip import ops as ops_
        for pol in args.gemm_policy:
            ops_.gemm_set_policy(pol)
    H, W, temp, steps1, stepsv = WORKLOADS[args.workload]
    i2v = args.workload.startswith("c4")
    image_only = args.workload.startswith("c1")
    vae_only = args.workload.startswith("c5")
    frames_per_video = 1 + 8 * (temp - 1)
    from pyflow_hip import video_io
    pinned = torch.empty(frames_per_video * H * W * 3, dtype=torch.uint8).pin_memory() if (rank == 0 or not use_sp) else None

    def to_host(u8):
        # the metric ends with the uint8 frames in HOST memory: pinned-buffer D2H inside the timed region
        return None if u8 is None else video_io.frames_to_host(u8, pinned)

    if vae_only:
        # ---- config C5: standalone decode of a synthetic 768p latent
        from pyflow_hip import synth
        from pyflow_hip.vae import CausalVideoVAE
        g = torch.Generator(device=device).manual_seed(1234)
        vcfg = synth.TINY_VAE if args.tiny_model else synth.VAE_DEFAULT
        vsd = {k: (torch.ones(sh, device=device) if k.endswith(".weight") else torch.zeros(sh, device=device)) if len(sh) == 1
               else torch.randn(sh, generator=g, device=device) * 0.02 for k, sh in synth.vae_decoder_param_shapes(vcfg).items()}
        vae = CausalVideoVAE(vsd, vcfg, device)
        vae.enable_tiling()
        z = torch.randn(1, 16, temp, H // 8, W // 8, generator=torch.Generator().manual_seed(5)).to(device)
        comm = sp_mod.get_sequence_parallel_comm() if use_sp else None
        # un-tiled temporal context parallelism: the live set of the one-pass decode is its widest layer (the first resnet of
        # the full-resolution level: input 256 ch + normalised input 256 ch + conv1 output 128 ch of 8 x frames; vae.py hands
        # every dead activation back) + 30 % for the allocator and the lower levels
        cp_bytes = 1.3 * (8 * -(-temp // max(world, 1))) * (H + 2) * (W + 2) * (256 + 256 + 128) * 2
        use_cp = use_sp and -(-temp // world) >= 2 and temp // world >= 2 and cp_bytes < 0.85 * torch.cuda.get_device_properties(0).total_memory
        dcfg, dsd, pipe, sp = None, None, None, None

        def one_video(seed):
            if use_cp:
                return to_host(vae.decode_context_parallel(z, comm))
            return to_host(vae.decode_to_uint8(z, window_size=1, tile_sample_min_size=256, comm=comm))
        vae.decode_to_uint8(torch.randn(1, 16, 2, 8, 8, device=device), window_size=1, tile_sample_min_size=256)
    else:
        pipe, dcfg, dsd = build_pipeline(device, tiny=args.tiny_model, mmdit=i2v, stages=1 if image_only else None)
        embeds = synthetic_prompt(dcfg, device)
        image = torch.randn(3, H, W, generator=torch.Generator().manual_seed(77)).clamp(-1, 1)
        if hasattr(pipe.dit, "launch_mode"):
            pipe.dit.launch_mode = args.launch_mode
        if args.no_overlap_text:
            pipe.dit.overlap_text = False
        if args.group_text != "auto" and hasattr(pipe.dit, "group_text"):
            pipe.dit.group_text = args.group_text == "on"
        sp = SampledProfiler(pipe, args.profile_period)
        pipe.phase_times = None

        def one_video(seed):
            if i2v:
                u8 = pipe.generate_i2v(prompt_embeds=embeds, input_image=image, temp=temp, num_inference_steps=stepsv,
                                       guidance_scale=7.0, video_guidance_scale=4.0,
                                       generator=torch.Generator().manual_seed(seed), output_type="uint8", save_memory=True)
            elif image_only:
                u8 = pipe.generate(prompt_embeds=embeds, height=H, width=W, temp=1, num_inference_steps=steps1,
                                   video_num_inference_steps=stepsv, guidance_scale=9.0, video_guidance_scale=5.0,
                                   generator=torch.Generator().manual_seed(seed), output_type="uint8", save_memory=True)
            else:
                u8 = pipe.generate(prompt_embeds=embeds, height=H, width=W, temp=temp, num_inference_steps=steps1,
                                   video_num_inference_steps=stepsv, guidance_scale=7.0, video_guidance_scale=5.0,
                                   generator=torch.Generator().manual_seed(seed), output_type="uint8", save_memory=True)
            return to_host(u8)

        # lazy code-object loading / first allocations are initialisation, not a step
        pipe.generate(prompt_embeds=embeds, height=64, width=64, temp=1 if image_only else 2,
                      num_inference_steps=[1] * len(pipe.stages), video_num_inference_steps=[1] * len(pipe.stages),
                      guidance_scale=7.0, video_guidance_scale=5.0, generator=torch.Generator().manual_seed(0),
                      output_type="uint8")
    for i in range(args.warmup):
        one_video(100 + i)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    ls0 = dict(getattr(getattr(pipe, "dit", None), "list_stats", None) or {})
    if sp is not None:
        sp.active = True
    if pipe is not None:
        pipe.phase_times = {}          # wall seconds of the sampling loop / the decode inside the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_video(i)
    barrier()
    dt = time.perf_counter() - t0
    if sp is not None:
        sp.active = False
    if rank == 0 or not use_sp:
        assert out.shape == (frames_per_video, H, W, 3) and out.dtype == torch.uint8 and not out.is_cuda
    else:
        assert out is None
    peak_gib = torch.cuda.max_memory_allocated() / 2 ** 30
    ph = dict(pipe.phase_times) if (pipe is not None and pipe.phase_times) else {}
    if pipe is not None:
        pipe.phase_times = None
    if world > 1:
        t = torch.tensor([dt, peak_gib, ph.get("sampling_s", 0.0), ph.get("decode_s", 0.0)], dtype=torch.float64,
                         device=device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, peak_gib = t[0].item(), t[1].item()
        if ph:
            ph = dict(sampling_s=t[2].item(), decode_s=t[3].item())
    if rank != 0:
        return
    from pyflow_hip import ops
    summ = ops.PROFILER.summary()
    roof = None
    extra = {}
    recs = {}
    for name, s in summ.items():
        if s["ms_total"] <= 0:
            continue
        tf = s["work_total"] / (s["ms_total"] * 1e-3) / 1e12
        recs[name] = dict(bound="mfma", achieved=round(tf, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                          frac=round(tf / PEAK_BF16_TFLOPS, 4), traffic=None, kernel=name, launches_timed=s["launches"],
                          avg_launch_ms=round(s["ms_total"] / s["launches"], 4), ms_timed=round(s["ms_total"], 1))
    # Kernel FAMILIES: template flavours of one kernel (gemm8p_kernel<false, 0 / 1 / 4>: plain / residual / GELU epilogue) are
    # one family; the dominant family = the one with the most device time among the sampled launches.  `roofline` is that
    # family (summed work / summed time); every flavour keeps its own entry under roofline_other_kernels.
    def family_of(name):
        return name.split("<")[0].split("(")[0].strip()
    fams = {}
    for name, r_ in recs.items():
        f_ = fams.setdefault(family_of(name), dict(members=[], work=0.0, ms=0.0, launches=0))
        f_["members"].append(name)
        f_["work"] += summ[name]["work_total"]
        f_["ms"] += summ[name]["ms_total"]
        f_["launches"] += summ[name]["launches"]
    families = {k: dict(bound="mfma", achieved=round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                        frac=round(v["work"] / (v["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), kernel=k, members=sorted(v["members"]),
                        launches_timed=v["launches"], avg_launch_ms=round(v["ms"] / v["launches"], 4), ms_timed=round(v["ms"], 1),
                        pflop_timed=round(v["work"] / 1e15, 3)) for k, v in fams.items()}
    dom = max(families, key=lambda n: families[n]["ms_timed"]) if families else None
    roof = dict(families[dom], traffic=None) if dom else None
    extra = recs
    if vae_only:
        roof, extra = None, {}
    # VAE decode kernels: one full 256 x 256 tile (5 latent frames -> 33 frames, the launch shapes of the timed decode) is
    # decoded once more AFTER the timed region on a single stream with the per-launch profiler on: implicit-GEMM convs
    # against the MFMA peak, GroupNorm + SiLU passes against the HBM peak (SURVEY 8d asks for both)
    try:
        the_vae = vae if vae_only else pipe.vae
        if the_vae is not None:
            ops.PROFILER.records = {}
            keep_ns, the_vae.n_streams = the_vae.n_streams, 1
            zt = torch.randn(1, 16, 5, 32, 32, device=device)
            the_vae.decode_to_uint8(zt, window_size=1, tile_sample_min_size=256)       # allocations / first launches
            ops.PROFILER.enabled = True
            the_vae.decode_to_uint8(zt, window_size=1, tile_sample_min_size=256)
            ops.PROFILER.enabled = False
            the_vae.n_streams = keep_ns
            torch.cuda.synchronize()
            vsum = ops.PROFILER.summary()
            conv_all = dict(launches=0, ms_total=0.0, work_total=0.0)       # the decode's conv family = all conv3d:* kernels
            for name, sv in vsum.items():
                if name.startswith("conv3d"):
                    for k_ in conv_all:
                        conv_all[k_] += sv[k_]
            vsum = dict(vsum, conv3d=conv_all)
            for name, sv in vsum.items():
                if sv["ms_total"] <= 0 or not (name.startswith("conv3d") or name in ("gn_stats", "gn_apply")):
                    continue
                rate = sv["work_total"] / (sv["ms_total"] * 1e-3)
                hbm = not name.startswith("conv3d")
                extra["vae:" + name] = dict(
                    bound="hbm" if hbm else "mfma", achieved=round(rate / (1e9 if hbm else 1e12), 1),
                    peak=PEAK_HBM_GBS if hbm else PEAK_BF16_TFLOPS, unit="GB/s" if hbm else "TFLOP/s",
                    frac=round(rate / ((PEAK_HBM_GBS * 1e9) if hbm else (PEAK_BF16_TFLOPS * 1e12)), 4), traffic=None,
                    kernel=name, launches_timed=sv["launches"], avg_launch_ms=round(sv["ms_total"] / sv["launches"], 4),
                    ms_timed=round(sv["ms_total"], 1), sampled="one 256x256-px tile, 33 frames, after the timed region")
    except Exception as e:          # noqa: BLE001  (extra information only: never let it break the result line)
        print(f"[bench] VAE kernel sample skipped: {e!r}", file=sys.stderr, flush=True)
    finally:
        ops.PROFILER.enabled = False
    # L2 <-> fabric traffic per launch from the committed rocprofv3 --pmc passes (one full-width forward at L = 15 488)
    pmc_path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    try:
        with open(pmc_path, "rb") as f:
            raw = f.read()
        import hashlib           # git blob id of the file the numbers are replayed from: a stale replay is visible
        pm = json.loads(raw)["kernels"]
        pmc_blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()[:12]
    except Exception:
        pm, pmc_blob = {}, None
    pmc_live_note, pmc_live_fail = None, None
    if world == 1 and not args.no_pmc and not args.tiny_model and args.workload.startswith("c3_"):
        live, why = pmc_traffic_in_run()
        if live is not None:
            pm, pmc_live_note = live, why
        else:
            pmc_live_fail = why

    try:
        with open(os.path.join(ROOT, "profiles", PMC_PROFILE_VAE), "rb") as f:
            raw_v = f.read()
        pmv = json.loads(raw_v)["kernels"]
        pmc_blob_vae = hashlib.sha1(b"blob %d\0" % len(raw_v) + raw_v).hexdigest()[:12]
    except Exception:
        pmv, pmc_blob_vae = {}, None

    def vae_traffic(kind):
        """HBM-side bytes per launch of the decode's kernels at the sampled tile's launch shapes (the PMC passes ran the same
        tile): GroupNorm passes directly; `conv3d` = launch-weighted mean over the convolution kernels"""
        if kind in ("gn_stats", "gn_apply"):
            for n, v in pmv.items():
                if kind in n:
                    return round(v["hbm_bytes_per_launch"])
            return None
        conv = [v for n, v in pmv.items() if ("true" in n and "gemm" in n) or "conv_narrow" in n]
        nl = sum(v["launches"] for v in conv)
        return round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in conv) / nl) if nl else None

    def pmc_traffic(name):
        if name in ("conv3d", "gn_stats", "gn_apply"):
            return vae_traffic(name)
        if name.startswith("conv3d:"):
            for n, v in pmv.items():
                if name[7:].split("<")[0] in n and ("true" in n or "conv_" in n):
                    return round(v["hbm_bytes_per_launch"])
            return None
        # ("attention" = the fast pass of the pair: attn64_kernel<2, 33, 4, true> since round 4, <2, 1> in the round-3 profile)
        key = {"attention": "attn64_kernel<2, 33", "gemm_kernel(128x128)": "gemm_kernel<false>"}.get(name, name)
        key = key.replace("gemm256_kernel<128>", "gemm256_kernel<128, false").replace("gemm256_kernel<192>", "gemm256_kernel<192, false") \
                 .replace("gemm256_kernel<256>", "gemm256_kernel<256, false")
        for n, v in pm.items():
            if key in n:
                return round(v["hbm_bytes_per_launch"])
        if name == "attention":
            for n, v in pm.items():
                if "attn64_kernel<2, 1" in n:
                    return round(v["hbm_bytes_per_launch"])
        return None
    for r in list(extra.values()):
        if r is not None and r.get("traffic") is None:
            r["traffic"] = pmc_traffic(r["kernel"])
    if roof is not None:      # family: launch-weighted mean of its members' per-launch traffic
        mem = [extra[m_] for m_ in roof["members"] if m_ in extra and extra[m_].get("traffic") is not None]
        nl_ = sum(m_["launches_timed"] for m_ in mem)
        roof["traffic"] = round(sum(m_["traffic"] * m_["launches_timed"] for m_ in mem) / nl_) if nl_ else None
    for k_, r_ in extra.items():
        if k_.startswith("vae:") and r_.get("traffic") is not None:
            r_["traffic_note"] = (f"(2*FETCH_SIZE + WRITE_SIZE) per launch from rocprofv3 --pmc passes over the same tile-chunk "
                                  f"window (profiles/{PMC_PROFILE_VAE}, git blob {pmc_blob_vae}); REPLAYED, not measured in this run")
    if roof is not None and roof["traffic"] is not None and pmc_live_note is not None:
        roof["traffic_note"] = pmc_live_note
    elif roof is not None and roof["traffic"] is not None:
        if pmc_live_fail:
            roof["traffic_in_run_failed"] = pmc_live_fail
        roof["traffic_note"] = ("(2*FETCH_SIZE + WRITE_SIZE) KB per launch of this kernel (gfx950 FETCH correction), mean over "
                                f"the launches of one full-width forward at L=15488; REPLAYED from profiles/{PMC_PROFILE} "
                                f"(git blob {pmc_blob}), not measured in this run; "
                                "counted at the L2<->fabric interface incl. Infinity-Cache hits")
    value = frames_per_video * args.steps * (1 if use_sp else world) / dt
    if dcfg is not None:          # what was actually built (tiny plumbing models and the MMDiT variant included)
        d_model = dcfg["num_attention_heads"] * dcfg["attention_head_dim"]
        n_par = sum(v.numel() for v in dsd.values()) / 1e9
        model_desc = (f"SD3-style MMDiT ({n_par:.2f} B params, {dcfg['num_layers']} joint blocks, d={d_model})" if i2v else
                      f"miniFLUX pyramid DiT ({n_par:.2f} B params, {dcfg['num_layers']}+{dcfg['num_single_layers']} blocks, d={d_model})")
    else:
        model_desc = "no DiT"
    res = {
        "metric": "video frames/sec (whole node) for 768p 241-frame T2V sampling" if args.workload.startswith("c3_")
        else f"video frames/sec (whole node) for {H}x{W} {frames_per_video}-frame T2V sampling ({args.workload}, not the headline metric)",
        "value": round(value, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 1), "higher_is_better": True,
        # N > 1 default: ONE video over all GPUs (total work fixed) = strong scaling; the N = 1 line of the same sweep says
        # the same; `--parallelism replicas` is the weak-scaling form
        "scaling": "strong" if (args.parallelism != "replicas" and (use_sp or world == 1)) else "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {model_desc} + CausalVideoVAE "
                               f"tiled(256)/chunked(1) decode (the reference's save_memory schedule; four chunk windows per launch set), {H}x{W}, temp={temp} ({frames_per_video} frames), steps {steps1}/{stepsv}, "
                               "CFG 7.0/5.0, random-init weights, synthetic prompt embeddings"
                               + (" [TINY MODEL: plumbing only]" if args.tiny_model else ""),
                   "parallelism": "single GPU" if world == 1 else (
                       "guidance2: one video over 2 GPUs, one classifier-free-guidance branch per GPU (velocity tokens "
                       "all-reduced per step, no all-to-all) + tile-parallel VAE decode" if guidance else
                       f"sp{world}: one video over {world} GPUs, sequence-parallel DiT (all-to-all heads<->rows over RCCL, "
                       "uneven 30-head map) + tile-parallel VAE decode" if use_sp
                       else f"{world} independent replicas (one video per GPU)")},
        "roofline": roof,
        "roofline_family": families,
        "roofline_other_kernels": extra,
    }
    if args.workload.startswith("c3_") and not args.tiny_model:
        # whole-step MFMA utilisation: ALGORITHMIC matmul work of one video (DiT GEMMs 27.98 + useful
        # attention 13.35 + un-tiled VAE decode 4.61 PFLOP) / step time / (GPUs x dense bf16 peak)
        res["whole_step_mfma_frac"] = round(45.94e15 * (1 if use_sp or world == 1 else world) / (dt / args.steps) / (world * PEAK_BF16_TFLOPS * 1e12), 4)
    if pipe is not None:
        res["config"]["launch_mode"] = getattr(pipe.dit, "launch_mode", "eager")
        ls1 = getattr(pipe.dit, "list_stats", None)
        if ls1:
            # host-side cost of the launch lists / hipGraphs: recorded and captured once per (unit, stage) plan and kept across
            # videos (pipeline.py: _plan, LRU over a whole schedule) -- inside the timed region only plans not seen in the
            # warm-up are recorded (with --warmup 0: all of them)
            d_ = {k: ls1[k] - ls0.get(k, 0) for k in ls1}
            res["launch_lists"] = dict(
                plans_recorded_total=ls1["records"], record_ms_per_plan=round(1e3 * ls1["record_s"] / max(ls1["records"], 1), 2),
                graphs_instantiated_total=ls1["instantiates"],
                instantiate_ms_per_plan=round(1e3 * ls1["instantiate_s"] / max(ls1["instantiates"], 1), 2),
                recorded_in_timed_region=d_["records"], instantiated_in_timed_region=d_["instantiates"],
                host_s_in_timed_region=round(d_["record_s"] + d_["instantiate_s"], 3), replays_in_timed_region=d_["replays"])
    # device memory the timed region needed (torch allocator high-water mark, max over ranks): the tiled decode runs four
    # tile lanes x four coalesced chunk windows, a hidden requirement of the headline number on a 288 GB part
    res["peak_mem_gib"] = round(peak_gib, 1)
    if ph:      # where the step went (wall seconds per step, max over ranks; device idle at both boundaries)
        res["phases"] = {k: round(v / args.steps, 3) for k, v in ph.items()}
    if world > 1:
        backend = torch.distributed.get_backend()
        res["launcher"] = "bench.py self-launch" if os.environ.get("PF_BENCH_LAUNCHER") == "self" else "external (torchrun)"
        res["rccl_ranks"] = world if backend == "nccl" else 0
        res["communicator"] = (("pf_comm (C-ABI RCCL communicator" + (f" + copy-engine windows of {args.comm_windows} MiB)"
                                                                     if getattr(comm_used, "transport", "") == "windows" else ")"))
                               if getattr(comm_used, "backend", "") == "pf_comm" else
                               f"torch.distributed ({backend}" + (" = RCCL)" if backend == "nccl" else
                               "; ranks share GPUs, transport through the host: PLUMBING ONLY, not a measurement)"))
        res["requested_parallelism"] = args.parallelism
        if comm_errors:      # why a communicator was not used (rank 0's view; every rank agreed on the outcome)
            res["communicator_fallback_reason"] = comm_errors
    if i2v:
        res["metric"] = "video frames/sec for 768p image-to-video sampling (config C4, not the headline metric)"
        res["config"]["workload"] = (f"{args.workload}: {model_desc} generate_i2v + CausalVideoVAE "
                                     f"tiled encode / decode, {H}x{W}, temp={temp} ({frames_per_video} frames), steps {stepsv}, "
                                     "CFG 7.0/4.0, random-init weights, synthetic prompt embeddings and image")
    if image_only:
        res["metric"] = "images/sec for 1024x1024 one-stage text-to-image sampling (config C1, not the headline metric)"
        res["unit"] = "images/s"
        res["config"]["workload"] = (f"{args.workload}: {model_desc}, ONE pyramid stage (stages=[1], stage_range=[0,1]), {H}x{W}, "
                                     f"{steps1[0]} steps, CFG 9.0, L = 4 224 tokens per forward, tiled VAE decode, random-init "
                                     "weights, synthetic prompt embeddings")
    if vae_only:
        res["metric"] = "video frames/sec for standalone 768p 241-frame CausalVideoVAE decode (config C5, not the headline metric)"
        res["config"]["workload"] = (f"{args.workload}: CausalVideoVAE decode of a synthetic latent [1,16,{temp},{H // 8},{W // 8}] -> "
                                     f"{frames_per_video} uint8 frames in host memory; "
                                     + ("temporal context-parallel un-tiled decode (halo exchange per causal conv, uneven frame ranges)"
                                        if (vae_only and use_cp) else "tiled(256) / chunked(1) decode, four chunk windows per launch set"
                                        + (", tile columns split over the ranks" if world > 1 else "")))
        if world > 1:
            res["config"]["parallelism"] = (f"cp{world}: temporal context-parallel decode over {world} GPUs" if use_cp
                                            else f"{world} GPUs, tile columns of the tiled decode split over the ranks")
        conv = extra.get("vae:conv3d") or {}
        res["roofline"] = dict(conv, note="dominant kernel family of the decode: implicit-GEMM CausalConv3d (MFMA-bound for C >= 128); "
                                          "GroupNorm passes against the HBM peak in roofline_other_kernels") if conv else None
        res["roofline_other_kernels"] = {k: v for k, v in extra.items() if k != "vae:conv3d"}
    if args.tiny_model:           # a plumbing run must never read like the headline measurement
        if "[TINY MODEL" not in res["config"]["workload"]:
            res["config"]["workload"] += " [TINY MODEL: plumbing only]"
        res["metric"] = "PLUMBING RUN with a tiny random model (not a measurement): " + res["metric"]
    if not args.no_cpu_baseline and world == 1 and not args.tiny_model and not i2v and not image_only and not vae_only:
        res["cpu_baseline"] = cpu_baseline(dcfg, dsd, usable_cpus())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
