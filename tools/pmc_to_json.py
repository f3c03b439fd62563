"""gpurun_out/pmc/<tag>.txt (tools/gpu_pmc.sh ... traffic: FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum passes)
-> the per-kernel traffic table bench.py reads (profiles/rNN_pmc_forward_maxL.json).  hbm_bytes_per_launch =
(2 * FETCH_SIZE + WRITE_SIZE) KB: gfx950 reports HALF of wide coalesced reads in FETCH_SIZE (MI355X_MICROARCH.md, HBM
section); counted at the L2 <-> fabric interface, Infinity-Cache hits included."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
kern, cur = {}, None
for ln in open(src):
    m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", ln)
    if m and cur is not None:
        kern.setdefault(cur, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    elif ln.strip() and not ln.startswith(" ") and not ln.startswith("pass "):
        cur = ln.strip()
out = {}
for k, d in kern.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        continue
    f, w = d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]
    e = dict(launches=d["FETCH_SIZE"][0], fetch_kb=f, write_kb=w, hbm_bytes_per_launch=(2 * f + w) * 1024)
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        h, mi = d["TCC_HIT_sum"][1], d["TCC_MISS_sum"][1]
        e["l2_hit"] = h / max(h + mi, 1.0)
    out[k] = e
target = ("tools/vae_only.py (one 256 x 256-pixel tile, chunk windows of the tiled decode at the released channel widths)"
          if "vae" in src.lower() else "tools/forward_only.py (2 full-width miniFLUX forwards at L=15488, B=2)")
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes over "
                   + target + ". hbm_bytes_per_launch = "
                   "(2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                   "coalesced reads); counted at the L2<->fabric interface, Infinity-Cache hits included.",
           "kernels": out}, open(dst, "w"), indent=1)
print(f"{len(out)} kernels -> {dst}")
