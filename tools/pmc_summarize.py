"""mean counter value per kernel name from a rocprofv3 counter_collection.csv"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "vectorized" in k or "elementwise" in k or "distribution" in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
