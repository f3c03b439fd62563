"""GPU idle time inside the sampling loop, from a rocprofv3 --kernel-trace CSV of `bench.py` (one video): the window from the first to
the last cfg_euler_kernel (the Euler update that ends every DiT step), the union of all kernel intervals inside it, and the largest
gaps with the kernels on either side -- is the host ever late with the next forward?
    rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-pmc
    python tools/gap_analysis.py /tmp/kt/.../kt_kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
eul = [i for i, r in enumerate(rows) if "cfg_euler" in r[2]]
if len(sys.argv) > 2:          # argv[2] = Euler steps per video: analyse the LAST video of the trace only (steady state: plans cached)
    eul = eul[-int(sys.argv[2]):]
# the warm-up generate() (64 x 64, one step per stage) has Euler steps too: the video's loop starts after the longest pause between them
cut = 0
if len(sys.argv) <= 2 and len(eul) > 1:
    cut = max(range(1, len(eul)), key=lambda k: rows[eul[k]][0] - rows[eul[k - 1]][0])
    if rows[eul[cut]][0] - rows[eul[cut - 1]][0] < 50e6:          # no pause of 50 ms: there was no separate warm-up
        cut = 0
first, last = eul[cut], eul[-1]
w0, w1 = rows[first][0], rows[last][1]
busy, cur_end, gaps = 0, w0, []
prev = rows[first][2]
for s, e, n in rows[first:last + 1]:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        prev = n
span = w1 - w0
idle = span - busy
print(f"sampling window {span / 1e9:.3f} s, {last - first + 1} kernels, {len(eul) - cut} Euler steps; device busy {busy / 1e9:.3f} s, idle {idle / 1e9:.3f} s = {100 * idle / span:.2f} %")
gaps.sort(reverse=True)
import collections
h = collections.Counter()
for g, _, _ in gaps:
    h["< 2 us" if g < 2000 else "2-10 us" if g < 10000 else "10-100 us" if g < 100000 else "0.1-1 ms" if g < 1000000 else ">= 1 ms"] += g
for k in ("< 2 us", "2-10 us", "10-100 us", "0.1-1 ms", ">= 1 ms"):
    n = sum(1 for g, _, _ in gaps if (k == "< 2 us" and g < 2000) or (k == "2-10 us" and 2000 <= g < 10000) or (k == "10-100 us" and 10000 <= g < 100000) or (k == "0.1-1 ms" and 100000 <= g < 1000000) or (k == ">= 1 ms" and g >= 1000000))
    print(f"   gaps {k:9s}: {n:7d}  total {h[k] / 1e6:9.1f} ms")
pairs = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g >= 100000:
        k = (a.replace("(anonymous namespace)::", "").replace("void ", "")[:34], b.replace("(anonymous namespace)::", "").replace("void ", "")[:34])
        pairs[k][0] += 1
        pairs[k][1] += g
print("gaps >= 0.1 ms by (kernel before -> kernel after): count, total ms")
for k, (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   {n:5d} {t / 1e6:9.1f}   {k[0]:34s} -> {k[1]}")
pairs2 = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if 10000 <= g < 100000:
        k = (a.replace("(anonymous namespace)::", "").replace("void ", "")[:34], b.replace("(anonymous namespace)::", "").replace("void ", "")[:34])
        pairs2[k][0] += 1
        pairs2[k][1] += g
print("gaps of 10-100 us by (kernel before -> kernel after): count, total ms, mean us")
for k, (n, t) in sorted(pairs2.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"   {n:5d} {t / 1e6:9.1f} {t / n / 1e3:7.1f}   {k[0]:34s} -> {k[1]}")
print("largest gaps (ms): between <kernel before> and <kernel after>")
for g, a, b in gaps[:8]:
    print(f"   {g / 1e6:8.3f}  {a[:60]:60s} -> {b[:60]}")
# ---- the decode that follows the loop: from the last Euler step to the last kernel of the trace (lanes overlap: union of intervals)
tail = rows[last + 1:]
if tail:
    d0, cur_end, busy2, gaps2, prev = rows[last][1], rows[last][1], 0, [], rows[last][2]
    for s_, e_, n_ in tail:
        if s_ > cur_end:
            gaps2.append((s_ - cur_end, prev, n_))
        if e_ > cur_end:
            busy2 += e_ - max(s_, cur_end)
            cur_end = e_
            prev = n_
    span2 = cur_end - d0
    print(f"after the loop (decode + frame copies): {span2 / 1e9:.3f} s, {len(tail)} kernels, busy {busy2 / 1e9:.3f} s, idle {(span2 - busy2) / 1e9:.3f} s = {100 * (span2 - busy2) / max(span2, 1):.2f} %")
    gaps2.sort(reverse=True)
    for g, a, b in gaps2[:12]:
        print(f"   {g / 1e6:8.3f}  {a[:60]:60s} -> {b[:60]}")
