"""A/B of gemm8p's tail split (pf_gemm_desc.workspace) on the six large DiT projections at EVERY sequence length of one
768p / 241-frame video (31 units x 3 stages), interleaved in one process; prints per-shape and whole-video GEMM time
(weighted with the steps and launches per forward) without the scratch and with it at several values of the split's
assumed fixed cost (tail_plan's `ov`, the measurement hook pf_gemm_set_policy(400 + ov)).
Usage: python tools/gemm_tail_ab.py [iters]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import lib as _labsel; _labsel.use_lab_library()      # needs `make -C pyramid-flow_amd/csrc lab`
from pyflow_hip import ops                                                                    # noqa: E402

TOK = [240, 960, 3840]
d = 1920
# name: (N, K, gelu_from, gate_res, image rows only, launches per forward)
SHAPES = {"qkv_img": (3 * d, d, -1, False, True, 8), "ff1_img": (4 * d, d, 0, False, True, 8),
          "o_img": (d, d, -1, True, True, 8), "ff2_img": (d, 4 * d, -1, True, True, 8),
          "kvqm": (7 * d, d, 3 * d, False, False, 16), "out_sgl": (d, 5 * d, -1, True, False, 16)}


def seq_len(u, s):
    n = 128 + TOK[s]
    if u >= 1:
        n += TOK[s]
        cur, ptx = s, 1
        while ptx < u:
            cur = max(cur - 1, 0)
            if cur == 0:
                break
            ptx += 1
            n += TOK[cur]
        if cur == 0 and ptx < u:
            n += (u - ptx) * TOK[0]
    return n


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    g = torch.Generator(device="cuda").manual_seed(0)
    Lmax, B = 15488, 2
    ws = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    A = torch.randn(B, Lmax, 5 * d, device="cuda", generator=g).bfloat16()
    Wbig = (torch.randn(7 * d, 5 * d, device="cuda", generator=g) * 0.02).bfloat16()
    bias = torch.randn(7 * d, device="cuda", generator=g)
    gate = torch.randn(B, d, device="cuda", generator=g)
    res = torch.randn(B, Lmax, d, device="cuda", generator=g).bfloat16()
    out = torch.empty(B, Lmax, 7 * d, device="cuda", dtype=torch.bfloat16)
    lens = {}
    for u in range(31):
        for s in range(3):
            lens.setdefault(seq_len(u, s), 0)
            lens[seq_len(u, s)] += 20 if u == 0 else 10
    tot = {k: [0.0] * len(ARMS) for k in SHAPES}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    for L, steps in sorted(lens.items()):
        for name, (N, K, gf, gr, img, cnt) in SHAPES.items():
            M = L - 128 if img else L
            W = Wbig[:N, :K]
            kw = dict(bias=bias[:N], batch=B, strideA=Lmax * 5 * d, strideC=Lmax * 7 * d, gelu_from=gf)
            if gr:
                kw.update(res=res, gate=gate, ldr=d, strideR=Lmax * d, gate_stride=d, flags=ops.GEMM_GATE_RES)
            which = ops.L.load().pf_gemm_which(M, B, N, K)
            # arms: no scratch | scratch with tail_plan's fixed cost = OVS[i] K-tile periods (measurement hook 400 + ov);
            # the arms run in forward / reversed / reversed / forward order and the four samples are averaged (the chip is
            # power-limited: what runs later in a burst runs warmer, a min over passes would favour the first position)
            best = [0.0] * len(ARMS)
            fwd = list(range(len(ARMS)))
            for order in (fwd, fwd[::-1], fwd[::-1], fwd):
                for ai in order:
                    ov = ARMS[ai]
                    if ov is not None:
                        ops.gemm_set_policy(400 + ov)
                    arm = None if ov is None else ws
                    ops.gemm(A, W, out, M, N, K, 5 * d, 5 * d, 7 * d, workspace=arm, **kw)        # warm
                    e0.record()
                    for _ in range(iters):
                        ops.gemm(A, W, out, M, N, K, 5 * d, 5 * d, 7 * d, workspace=arm, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    best[ai] += e0.elapsed_time(e1) / iters / 4          # mean over the ABBA passes: linear drift cancels
            for ai in range(len(ARMS)):
                tot[name][ai] += best[ai] * steps * cnt
            rows.append((L, name, which, best))
    ops.gemm_set_policy(400 + 4)
    print("arms: " + ", ".join("no scratch" if a is None else f"ov={a}" for a in ARMS))
    print("per (L, shape): kernel family, ms per arm")
    for L, name, which, best in rows:
        if name in ("ff2_img", "out_sgl", "kvqm") and L % 7 in (0, 1, 2):          # a readable sample of the table
            print(f"  L={L:6d} {name:8s} which={which:3d}  " + " ".join(f"{b:8.4f}" for b in best))
    print("one video (960 forwards), seconds of GEMM time per arm")
    for k, v in tot.items():
        print(f"  {k:8s} " + " ".join(f"{x / 1e3:7.3f}" for x in v))
    sums = [sum(v[i] for v in tot.values()) for i in range(len(ARMS))]
    print("  total    " + " ".join(f"{x / 1e3:7.3f}" for x in sums))
    print("  saved vs no scratch: " + " ".join(f"{(sums[0] - x) / 1e3:7.3f}" for x in sums))


ARMS = [None, 4, 8, 14, 199]


if __name__ == "__main__":
    main()
