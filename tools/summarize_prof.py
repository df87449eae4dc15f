"""Condenses the rocprofv3 output of tools/profile_round.sh into the two files kept under profiles/:
<tag>_kernel_stats.csv (the --stats table) and <tag>_pmc_summary.json (per-kernel mean of every counter,
per launch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them, uncorrected)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
keep = ("k_assemble", "k_spmv", "k_cg_", "k_amg", "k_apply_dirichlet", "k_sens")
summ = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches_sampled": max(v[1] for v in cs.values())}
        for k, cs in acc.items() if any(s in k for s in keep)}
json.dump(summ, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: {c: round(x) for c, x in v.items() if c in ("FETCH_SIZE", "WRITE_SIZE")} for k, v in summ.items()}, indent=1))
