"""Per-kernel durations and idle gaps between consecutive kernels from a rocprofv3 --kernel-trace csv.
   python3 tools/trace_gaps.py <dir> [min_calls]   (prints a table; gap = start of this kernel - end of the previous one on the device)"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur, gap = defaultdict(list), defaultdict(list)
for i, (s, e, k) in enumerate(rows):
    k = k.split("(")[0][:90]
    dur[k].append((e - s) / 1e3)
    if i:
        gap[k].append((s - rows[i - 1][1]) / 1e3)
def q(v, p):
    v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))]
print(f"{'kernel':90s} {'calls':>6s} {'dur p50':>8s} {'dur p90':>8s} {'gap-before p50':>14s} {'p90':>7s}   (us)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= minc:
        g = gap.get(k, [0.0])
        print(f"{k:90s} {len(v):6d} {q(v, .5):8.1f} {q(v, .9):8.1f} {q(g, .5):14.1f} {q(g, .9):7.1f}")
