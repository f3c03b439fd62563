"""per RUN of consecutive dispatches of one kernel name (= one problem shape in tools/blas_pmc.py): mean counter values from a
rocprofv3 counter_collection.csv, or mean durations from a kernel_trace.csv.  usage: pmc_runs.py <file.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("empty csv")
skip = ("vectorized", "elementwise", "distribution", "fill", "checksum")
if "Counter_Name" in rows[0]:
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(int(r["Dispatch_Id"]), (r["Kernel_Name"], {}))[1].setdefault(r["Counter_Name"], 0.0)
        per[int(r["Dispatch_Id"])][1][r["Counter_Name"]] += float(r["Counter_Value"])
    seq = [(per[k][0], per[k][1]) for k in sorted(per)]
else:
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(r["Kernel_Name"], {"duration_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3}) for r in rows]
runs = []
for name, vals in seq:
    if any(s in name for s in skip):
        continue
    if runs and runs[-1][0] == name:
        runs[-1][1].append(vals)
    else:
        runs.append((name, [vals]))
for name, lst in runs:
    keys = sorted({k for v in lst for k in v})
    print(f"{name[:90]:90s} n={len(lst):3d} " + "  ".join(f"{k}={sum(v.get(k, 0.0) for v in lst) / len(lst):.5g}" for k in keys))
