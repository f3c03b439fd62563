"""Predict the 1 -> 2 / 4 / 8 GPU curve of the headline job (C3: 768p, 241 frames) on ONE GPU.

The sequence-parallel DiT engine (pyflow_hip/flux_sp.py; replaces flux_block.py:266-325, 519-565 and
trainer_misc/communicate.py:7-66) gives every rank a contiguous chunk of the merged rows and a share of the 30 heads.  A
rank's kernels therefore have shapes that can be launched on a single device: this tool runs ONE rank's recorded launch
list (rows L / P, uneven head map, both exchanges replaced by device copies of the same size: `PhantomComm`) at
P = 2 / 4 / 8 for rank 0 (owns the text rows) and rank P - 1 (owns the current frame's rows) -- and the PRODUCTION single-GPU
engine for the N = 1 row and, with batch 1, for the guidance-parallel rank of N = 2 -- over a sample of the
93 (unit, stage) sequences of one video, interpolates over the units, weights with the schedule (20 / 10 steps), and adds
  * the exchange time from the bytes a rank sends per forward and the xGMI figures of the task statement
    (7 links x 153 GB/s per GPU, one link per peer pair; `--link-eff` of that is assumed achievable) -- reported both as
    fully exposed (upper bound) and with the single-stream blocks' exchanges partly hidden under the MLP-branch GEMM they
    overlap with (flux_sp.py): the hidden fraction is the one MEASURED for an RCCL kernel beside the persistent GEMM on one
    GPU (`--overlap-hidden`, 0.35: tools/comm_overlap_bench.py), and
  * the host time per forward of the launch-list replay (measured here, with the device idle),
  * the tile-parallel VAE decode: the measured single-GPU decode x ceil(7 / N) / 7 tile columns (vae.py: decode_tiles).
Output: a table (per-kernel-family seconds per rank are in the JSON) + predicted frames/s and efficiency.

    python tools/rank_shape_bench.py [--units 0,1,2,3,5,8,12,16,20,24,28,30] [--reps 2] [--decode-s 7.8] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
sys.path.insert(0, ROOT)


class PhantomComm:
    """rank `rank` of `world` with no peers: an all-to-all becomes a device copy of min(send, recv) elements (the received
    rows then hold projection outputs of the right statistics: RMS-normed q / k keep the attention's fast pass valid),
    all-reduce / broadcast are no-ops.  Recordable like LocalComm (pf_copy_rows)."""
    recordable = True

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.sent = 0            # elements handed to all_to_all since the last reset (bytes model)

    def all_to_all(self, recv, send, recv_splits, send_splits, async_op=False):
        from pyflow_hip import ops
        ns, nr = sum(send_splits), sum(recv_splits)
        self.sent += ns - send_splits[self.rank]
        n = min(ns, nr) // 8 * 8
        if n:
            ops.copy_rows(send, recv, 1, n, n, n, 0, 0, 1)
        return None

    def all_reduce(self, t):
        return t

    def broadcast(self, t, src=0):
        return t

    def barrier(self):
        pass


def clips_for(u, s):
    """clip list [oldest .. current] of (unit u, stage s) as pipeline.py builds it (pipeline.py:1112-1190 of the reference):
    frames older than u - 2 at stage 0, frame u - 2 one stage lower, frame u - 1 and the current frame at stage s"""
    res = lambda k: (24 << k, 40 << k)      # noqa: E731  latent rows / cols of a 768 x 1280 frame at stage k
    clips, n_old, s2 = [], max(u - 2, 0), max(s - 1, 0)
    if u >= 2:
        if s2 == 0:
            clips.append((n_old + 1,) + res(0))
        else:
            if n_old:
                clips.append((n_old,) + res(0))
            clips.append((1,) + res(s2))
    if u >= 1:
        clips.append((1,) + res(s))
    clips.append((1,) + res(s))
    return clips


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", default="0,1,2,3,5,8,12,16,20,24,28,30")
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--decode-s", type=float, default=6.4, help="measured single-GPU tiled decode of the 241 frames (s)")
    ap.add_argument("--link-gbs", type=float, default=153.0)
    ap.add_argument("--link-eff", type=float, default=0.8)
    ap.add_argument("--overlap-hidden", type=float, default=0.35,
                    help="fraction of an exchange that disappears under the GEMM queued beside it.  MEASURED on one GPU (round 5, "
                         "tools/comm_overlap_bench.py, profiles/r05_comm_overlap_bench.log): an RCCL send / recv kernel and the persistent "
                         "GEMM that owns every CU share the chip badly -- 0.31-0.36 of a world-1 self exchange is hidden, with or "
                         "without reserved CUs; a blit copy of the same bytes hides 0.7-0.9 (1.0 = the round-4 assumption)")
    ap.add_argument("--latency-us", type=float, default=15.0, help="fixed cost of one grouped send/recv exchange")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from pyflow_hip import ops, synth
    from pyflow_hip.flux_sp import FluxEngineSP
    dev = "cuda"
    cfg = synth.MINIFLUX
    g = torch.Generator(device=dev).manual_seed(1234)
    sd = {}
    for k, shp in synth.flux_param_shapes(cfg).items():
        sd[k] = (torch.ones(shp, device=dev) if k.endswith(".weight") else torch.zeros(shp, device=dev)) if len(shp) == 1 \
            else torch.randn(shp, generator=g, device=dev) * 0.02
    eng = FluxEngineSP(sd, cfg, dev, comm=PhantomComm(0, 1))
    # the PRODUCTION single-GPU engine (QK epilogue, side-stream text path, hipGraph replay): the N = 1 row every efficiency is
    # quoted against, and -- with batch 1 -- the guidance-parallel rank of N = 2 (pyflow_hip/flux_cfg.py runs exactly this engine)
    from pyflow_hip.flux import FluxEngine
    eng1 = FluxEngine(sd, cfg, dev)
    del sd
    mask = torch.zeros(2, 128, dtype=torch.long)
    mask[0, :40] = 1
    mask[1, :96] = 1
    enc = torch.randn(2, 128, 4096).to(torch.bfloat16)
    pooled = torch.randn(2, 768)
    pooled1 = pooled[1:2].contiguous()
    eng.encode_context(enc)
    eng1.encode_context(enc)
    units = sorted({int(x) for x in args.units.split(",")} | {0, 30})
    Ps = [int(x) for x in args.ranks.split(",")]
    d, B = 1920, 2
    link = args.link_gbs * 1e9 * args.link_eff

    state = {"B": 2, "B1": 2}

    def measure_single(u, s, nb=2):
        """the production engine at (u, s): batch 2 = the single-GPU job, batch 1 = one guidance branch"""
        if state["B1"] != nb:
            eng1.encode_context(enc if nb == 2 else enc[1:2])
            state["B1"] = nb
        shapes = clips_for(u, s)
        clips = [torch.randn(1, 16, *c, device=dev) for c in shapes]
        plan = eng1.make_plan(shapes, mask if nb == 2 else mask[1:2])
        ts_, pooled_ = ([500.0, 500.0], pooled) if nb == 2 else ([500.0], pooled1)
        for _ in range(3):                                   # record, replay, graph replay
            eng1.forward_tokens(plan, clips, ts_, pooled_, shared_clips=True)
        torch.cuda.synchronize()
        # one event pair per repetition, the MINIMUM counts: a single hiccup (allocator, a module load: 15-20 ms in two of the 288
        # samples of the first round-5 run) otherwise ends up, interpolated over four units x ten steps, as 0.5 s of a video
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        h0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            eng1.forward_tokens(plan, clips, ts_, pooled_, shared_clips=True)
            e1.record()
        host_ms = (time.perf_counter() - h0) / args.reps * 1e3
        torch.cuda.synchronize()
        dev_ms = min(e0.elapsed_time(e1) for e0, e1 in evs)
        ops.PROFILER.records = {}
        ops.PROFILER.enabled = True
        eng1.forward_tokens(plan, clips, ts_, pooled_, shared_clips=True)
        ops.PROFILER.enabled = False
        torch.cuda.synchronize()
        fam = {}
        for name, sv in ops.PROFILER.summary().items():
            key = name.split("<")[0].split("(")[0]
            fam[key] = fam.get(key, 0.0) + sv["ms_total"]
        ops.PROFILER.records = {}
        del plan, clips
        return dev_ms, host_ms, fam, 0, 30

    def measure(P, r, u, s, nb=2):
        """device ms, host ms of one forward of rank r of P at (u, s), and its kernel-family split.  nb = 1: ONE branch of the
        guidance pair (the guidance-parallel engine of N = 2, pyflow_hip/flux_cfg.py: all rows, all heads, batch 1)"""
        eng.comm = PhantomComm(r, P)
        eng._layouts = {}
        if state["B"] != nb:
            eng.encode_context(enc if nb == 2 else enc[1:2])
            state["B"] = nb
        shapes = clips_for(u, s)
        clips = [torch.randn(1, 16, *c, device=dev) for c in shapes]
        plan = eng.make_plan(shapes, mask if nb == 2 else mask[1:2])
        ts_, pooled_ = ([500.0, 500.0], pooled) if nb == 2 else ([500.0], pooled1)
        for _ in range(2):                                   # records the launch list, then one replay
            eng.forward_tokens(plan, clips, ts_, pooled_, shared_clips=True)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        h0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            eng.forward_tokens(plan, clips, ts_, pooled_, shared_clips=True)
            e1.record()
        host_ms = (time.perf_counter() - h0) / args.reps * 1e3          # launch loop only (the device runs behind)
        torch.cuda.synchronize()
        dev_ms = min(e0.elapsed_time(e1) for e0, e1 in evs)               # (minimum over the repetitions: see measure_single)
        # one profiled (eager, per-launch events) forward for the family split
        ops.PROFILER.records = {}
        ops.PROFILER.enabled = True
        eng.forward_tokens(plan, clips, ts_, pooled_, shared_clips=True)
        ops.PROFILER.enabled = False
        torch.cuda.synchronize()
        fam = {}
        for name, sv in ops.PROFILER.summary().items():
            key = name.split("<")[0].split("(")[0]
            fam[key] = fam.get(key, 0.0) + sv["ms_total"]
        ops.PROFILER.records = {}
        lay = eng.layout(plan)
        del plan, clips
        return dev_ms, host_ms, fam, lay.nloc, lay.my_heads

    def comm_ms(P, L, nloc):
        """(fully exposed, single-block K|V|Q exchange hidden under the MLP-branch GEMM) ms per forward"""
        if P == 1:
            return 0.0, 0.0
        peer_qkv = nloc * B * 3 * d * 2 / P          # bytes one peer receives from this rank per block (even head split)
        peer_out = nloc * B * d * 2 / P
        t_qkv = args.latency_us * 1e-3 + peer_qkv / link * 1e3
        t_out = args.latency_us * 1e-3 + peer_out / link * 1e3
        full = 24 * (t_qkv + t_out)
        n1 = (4 * d * 2 // 3) // 256 * 256
        t_mlp1 = 2.0 * nloc * B * n1 * d / 1.0e15 * 1e3           # MLP-branch part 1 at 1.0 PFLOP/s
        t_mlp2 = 2.0 * nloc * B * (4 * d - n1) * d / 1.0e15 * 1e3
        f = args.overlap_hidden
        hidden = 8 * (t_qkv + t_out) + 16 * ((t_qkv - f * min(t_qkv, t_mlp1)) + (t_out - f * min(t_out, t_mlp2)))
        return full, hidden

    table = {}
    if 2 in Ps:          # guidance-parallel N = 2: a rank = the whole sequence, all heads, ONE branch of the CFG pair
        for s in range(3):
            for u in units:
                dev_ms, host_ms, fam, nloc, mh = measure_single(u, s, nb=1)
                L = 128 + sum(c[0] * (c[1] // 2) * (c[2] // 2) for c in clips_for(u, s))
                table[("g2", 0, u, s)] = dict(dev_ms=dev_ms, host_ms=host_ms, fam=fam, L=L, nloc=L, heads=mh)
                print(f"guidance2 (production engine, batch 1) u={u:2d} s={s} L={L:5d}: {dev_ms:8.3f} ms device, {host_ms:6.3f} ms host", flush=True)
    for P in Ps:
        for r in sorted({0, P - 1}):
            for s in range(3):
                for u in units:
                    dev_ms, host_ms, fam, nloc, mh = measure_single(u, s) if P == 1 else measure(P, r, u, s)
                    L = 128 + sum(c[0] * (c[1] // 2) * (c[2] // 2) for c in clips_for(u, s))
                    table[(P, r, u, s)] = dict(dev_ms=dev_ms, host_ms=host_ms, fam=fam, L=L, nloc=nloc, heads=mh)
                    print(f"P={P} rank={r} u={u:2d} s={s} L={L:5d} rows={nloc:5d} heads={mh:2d}: {dev_ms:8.3f} ms device, "
                          f"{host_ms:6.3f} ms host", flush=True)

    def interp(P, r, s, u, key):
        us = units
        if u in us:
            return key(table[(P, r, u, s)])
        lo = max(x for x in us if x < u)
        hi = min(x for x in us if x > u)
        a, b = key(table[(P, r, lo, s)]), key(table[(P, r, hi, s)])
        return a + (b - a) * (u - lo) / (hi - lo)

    result = {"assumptions": dict(link_GBs=args.link_gbs, link_eff=args.link_eff, latency_us=args.latency_us,
                                  decode_s_single_gpu=args.decode_s, units_sampled=units, reps=args.reps), "P": {}}
    base = None
    modes = list(Ps) + (["g2"] if 2 in Ps else [])
    for P in modes:
        guidance = P == "g2"
        Pn = 2 if guidance else P
        per_rank = {}
        for r in ([0] if guidance else sorted({0, P - 1})):
            dev_s = host_s = full_s = hid_s = 0.0
            fam_s = {}
            for u in range(31):
                for s in range(3):
                    steps = 20 if u == 0 else 10
                    dm = interp(P, r, s, u, lambda e: e["dev_ms"])
                    hm = interp(P, r, s, u, lambda e: e["host_ms"])
                    L = 128 + sum(c[0] * (c[1] // 2) * (c[2] // 2) for c in clips_for(u, s))
                    nloc = -(-L // Pn)
                    cf, ch = (0.0, 0.0) if guidance else comm_ms(P, L, nloc)
                    dev_s += steps * dm * 1e-3
                    host_s += steps * hm * 1e-3
                    full_s += steps * cf * 1e-3
                    hid_s += steps * ch * 1e-3
                    for k in set().union(*[table[(P, r, x, s)]["fam"].keys() for x in units]):
                        fam_s[k] = fam_s.get(k, 0.0) + steps * interp(P, r, s, u, lambda e, k=k: e["fam"].get(k, 0.0)) * 1e-3
            per_rank[r] = dict(device_s=round(dev_s, 3), host_s=round(host_s, 3), exchange_s_exposed=round(full_s, 3),
                               exchange_s_overlapped=round(hid_s, 3), kernel_family_s={k: round(v, 3) for k, v in sorted(fam_s.items())})
        worst = max(per_rank.values(), key=lambda e: e["device_s"])
        dec = args.decode_s * (-(-7 // Pn)) / 7.0
        t_lo = max(worst["device_s"], worst["host_s"]) + worst["exchange_s_overlapped"] + dec
        t_hi = max(worst["device_s"], worst["host_s"]) + worst["exchange_s_exposed"] + dec
        if P == 1:
            base = t_lo
        result["P"][P] = dict(ranks=per_rank, decode_s=round(dec, 3), video_s=[round(t_lo, 2), round(t_hi, 2)],
                              frames_per_s=[round(241 / t_hi, 2), round(241 / t_lo, 2)],
                              compute_efficiency=round(per_rank[0]["device_s"] and (result["P"][Ps[0]]["ranks"][0]["device_s"] / (Pn * worst["device_s"])) if P != Ps[0] else 1.0, 3),
                              efficiency=[round(base / (Pn * t_hi), 3), round(base / (Pn * t_lo), 3)] if base else None)
    print("\nP   rank  device s  host s  exch exposed / overlapped s  decode s  video s (lo..hi)  frames/s  efficiency  compute eff.")
    for P in modes:
        e = result["P"][P]
        for r, pr in e["ranks"].items():
            print(f"{str(P):<3s} {r:<5d} {pr['device_s']:8.2f} {pr['host_s']:7.2f}   {pr['exchange_s_exposed']:7.2f} / {pr['exchange_s_overlapped']:<7.2f}"
                  f"          {e['decode_s']:6.2f}   {e['video_s'][0]:6.2f}..{e['video_s'][1]:<6.2f}  {e['frames_per_s'][0]:5.2f}..{e['frames_per_s'][1]:<5.2f} "
                  f" {e['efficiency']}  {e['compute_efficiency']}")
    for P in modes:
        print(f"P={P} kernel-family seconds per video, slowest rank:",
              max(result["P"][P]["ranks"].values(), key=lambda e: e["device_s"])["kernel_family_s"])
    if args.out:
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
