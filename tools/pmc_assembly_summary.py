"""Per-kernel averages of the counter passes of tools/pmc_assembly.sh (k_assemble_v4 kernels only)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for sub in "abcdefg":
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_assemble_v4" not in k: continue
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_assemble_v4" in k: dur[k.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    d = sorted(dur[k]); med = d[len(d) // 2] if d else float("nan")
    print(k, f"  median launch {med:.1f} us (profiled)")
    for c, v in sorted(cs.items()):
        v = v[len(v) // 3:]  # drop the first launches
        print(f"   {c:28s} {sum(v) / len(v):16.0f}")
