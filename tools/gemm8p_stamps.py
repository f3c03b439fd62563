"""Cycle stamps of the persistent GEMM's main loop (lab builds of csrc/gemm8p.hip with -DPF_G8_STAMP=1 | 2).

    make -C pyramid-flow_amd/csrc variant NAME=stamp1 DEFS="-DPF_LAB_HOOKS -DPF_G8_STAMP=1"
    make -C pyramid-flow_amd/csrc variant NAME=stamp2 DEFS="-DPF_LAB_HOOKS -DPF_G8_STAMP=2"
    python tools/gemm8p_stamps.py stamp1        # cycles per K-tile in steady state and across the tile boundary
    python tools/gemm8p_stamps.py stamp2        # per phase: load slot, wait at its barrier, MFMA phase + second barrier

Mode 1 records one s_memtime tick per K-tile (release from phase 0's barrier) for 64 consecutive K-tiles of every wave;
mode 2 three ticks per phase (slot start, arrival at the slot's barrier, release from it) for 4 consecutive K-tiles.  The
matrix pipe needs 2 048 cycles per K-tile (2 waves per SIMD x 64 MFMAs x 16 cycles).  Shapes: the DiT's plain / GELU / QK
flavours at M = 2 x 15 488 (the residual flavour's stamped build spills and is not measured)."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as L                                                                  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "stamp1"
L.use_lab_library(variant)
from pyflow_hip import ops                                                                       # noqa: E402

so = L.load()
mode = 3 if variant.endswith("3") else 1
D, Lseq = 1920, 15488
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)


def run(N, K, gelu_from, window, qk=False):
    A = (torch.randn(2 * Lseq, K, generator=g, device=dev) * 1.0).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    Cc = torch.empty(2 * Lseq, N, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(N, device=dev)
    if os.environ.get("G8_POLICY"):          # e.g. 9 = desynchronised start (mode 1, window 0 only: the window rides in the same field)
        assert window == 0
        ops.gemm_set_policy(int(os.environ["G8_POLICY"]))
    else:
        ops.gemm_set_policy(100000 + window)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        if it == 2:
            ev[0].record()
        qkd = None
        if qk:          # the double blocks' K | V | Q projection: QK-RMSNorm + RoPE on the K and Q column blocks
            rope = torch.randn(Lseq, 32, 2, generator=g, device=dev)
            qkd = dict(rope=rope, wq=torch.ones(64, device=dev), wk=torch.ones(64, device=dev), d=D, k_col0=0, q_col0=2 * D, row0=0,
                       eps=1e-6, q_scale=0.18)
        ops.gemm(A, W, Cc, Lseq, N, K, K, K, N, bias=bias, batch=2, strideA=Lseq * K, strideC=Lseq * N,
                 gelu_from=gelu_from, qk=qkd)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    buf = (C.c_uint * (256 * 8 * 64))()
    rc = so.pf_lab_gemm8p_stamps(buf)
    assert rc == 0, rc
    st = torch.tensor(list(buf), dtype=torch.int64).view(256, 8, 64)
    return st, ms, 2.0 * 2 * Lseq * N * K / (ms * 1e-3) / 1e12


def d32(a, b):
    return (b - a) & 0xFFFFFFFF


shapes = [("K|V|Q-like plain N=5760 K=1920", 3 * D, D, -1, False), ("K|V|Q with the QK epilogue N=5760 K=1920", 3 * D, D, -1, True),
          ("MLP up GELU N=7680 K=1920", 4 * D, D, 0, False),
          ("K|V|Q|MLP plain N=13440 K=1920", 7 * D, D, -1, False), ("plain N=1920 K=9600", D, 5 * D, -1, False)]
for name, N, K, gf, qk_ in shapes:
    nk = K // 64
    if mode == 1:
        st, ms, tf = run(N, K, gf, 0, qk_)
        print(f"== {name}: {ms:.3f} ms = {tf:.0f} TFLOP/s (stamped build); nk = {nk} K-tiles per tile")
        steady, bound = [], []
        for wg in range(0, 256, 8):
            for w in (0, 4):
                t = st[wg, w]
                dl = [d32(int(t[i]), int(t[i + 1])) for i in range(min(63, 4 * nk))]
                for i, dv in enumerate(dl):
                    kt = i + 1                      # delta i = tick(kt) - tick(kt - 1); tick(kt) belongs to K-tile kt
                    (bound if kt % nk == 0 else steady).append(dv)
        print(f"   cycles per K-tile, steady state: median {statistics.median(steady):.0f}  p10 {sorted(steady)[len(steady) // 10]}  "
              f"p90 {sorted(steady)[9 * len(steady) // 10]}   (matrix pipe: 2048)")
        if bound:
            print(f"   across a tile boundary (last K-tile's phases 1-3 + epilogue + first load slot): median {statistics.median(bound):.0f}  "
                  f"-> boundary cost ~ {statistics.median(bound) - statistics.median(steady):.0f} cycles per tile "
                  f"= {100 * (statistics.median(bound) - statistics.median(steady)) / (nk * statistics.median(steady)):.1f} % of a tile")
        t = st[0, 0]
        print("   workgroup 0 wave 0, first 40 K-tile periods:", [d32(int(t[i]), int(t[i + 1])) for i in range(40)])
        t = st[0, 4]
        print("   workgroup 0 wave 4, first 40 K-tile periods:", [d32(int(t[i]), int(t[i + 1])) for i in range(40)])
    elif mode == 3:
        st, ms, tf = run(N, K, gf, 0, qk_)
        print(f"== {name}: {ms:.3f} ms = {tf:.0f} TFLOP/s (stamped build); epilogue of tiles 1..8 of each wave, cycles: "
              "[entry -> conversions done | -> queue drained | -> stores issued | -> accumulators re-initialised]; "
              "and from this tile's exit to the next tile's entry (= the main loop of a tile)")
        for w in (0, 1, 2, 4, 5, 6):
            seg = [[] for _ in range(5)]
            for wg in range(0, 256, 4):
                t = st[wg, w]
                for tl in range(1, 8):
                    e = [int(t[tl * 5 + k]) for k in range(5)]
                    if 0 in e or int(t[(tl + 1) * 5]) == 0:
                        continue
                    for k in range(4):
                        seg[k].append(d32(e[k], e[k + 1]))
                    seg[4].append(d32(e[4], int(t[(tl + 1) * 5])))
            print(f"   wave {w} (group {w // 4}): " + " | ".join(f"{statistics.median(x):.0f}" for x in seg[:4]) +
                  f"   total {sum(statistics.median(x) for x in seg[:4]):.0f};  main loop between epilogues {statistics.median(seg[4]):.0f}")
    else:
        for window in (8, nk - 2):
            st, ms, tf = run(N, K, gf, window, qk_)
            print(f"== {name}: {ms:.3f} ms = {tf:.0f} TFLOP/s (stamped build); window = K-tiles {window}..{window + 3} of nk = {nk}")
            for w in (0, 4):
                rows = []
                for kt in range(4):
                    for ph in range(4):
                        slot, wait, rest = [], [], []
                        for wg in range(0, 256, 4):
                            t = st[wg, w]
                            ix = kt * 16 + ph * 4
                            a, b, c = int(t[ix]), int(t[ix + 1]), int(t[ix + 2])
                            nxt = ix + 4 if ph < 3 else (kt + 1) * 16
                            if a == 0 or c == 0:
                                continue
                            slot.append(d32(a, b))
                            wait.append(d32(b, c))
                            if nxt < 64 and int(t[nxt]) != 0:
                                rest.append(d32(c, int(t[nxt])))
                        if slot:
                            rows.append((kt, ph, statistics.median(slot), statistics.median(wait),
                                         statistics.median(rest) if rest else float("nan")))
                print(f"   wave {w} (group {w // 4}): per phase [load slot | wait at its barrier | MFMA phase + second barrier] cycles, median over 64 workgroups")
                for kt in range(4):
                    rr = [r for r in rows if r[0] == kt]
                    tot = sum(r[2] + r[3] + (0 if r[4] != r[4] else r[4]) for r in rr)
                    print(f"     K-tile {window + kt}: " + "   ".join(f"ph{r[1]}: {r[2]:.0f} | {r[3]:.0f} | {r[4]:.0f}" for r in rr) + f"   sum {tot:.0f}")
ops.gemm_set_policy(0)
