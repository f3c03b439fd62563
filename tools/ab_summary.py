"""summarise tools/box_gemm_variants.sh output: per shape, per variant the medians of its processes and whether all bits agree"""
import collections, re, statistics, sys
cur = None
data = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    m = re.match(r"### rep (\d+) variant (\S+)", l)
    if m:
        cur = m.group(2)
        continue
    m = re.match(r"M=\S+ N=(\d+) K=(\d+) (.*?)\s+which=\d+:\s+\[0\] (\d+) \((\d+) \.\. (\d+)\).*checksum (\w+)", l)
    if m:
        data[(m.group(1), m.group(2), m.group(3).strip())][cur].append((int(m.group(4)), m.group(7)))
for k, v in data.items():
    same = len(set(c for lst in v.values() for _, c in lst)) == 1
    print(f"N={k[0]:>5} K={k[1]:>4} {k[2]:<36}", "  ".join(f"{var} {statistics.median([a for a, _ in lst]):.0f}" for var, lst in v.items()),
          " same bits" if same else " BITS DIFFER")
