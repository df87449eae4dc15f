# Runs ON THE GPU BOX: rocprofv3 kernel trace of tools/c5_parts.py (the pieces of one topology-optimisation iteration, five
# repetitions each) and a per-kernel table of launch-time quantiles.   gpurun -- bash tools/trace_c5_parts.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_c5; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/tools/c5_parts.py > $O/log.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n in sorted(d, key=lambda k: -sum(d[k]))[:45]:
    v = sorted(d[n])
    # distinct duration clusters: print quantiles
    q = lambda p: v[min(len(v) - 1, int(p * len(v)))]
    print(f"{sum(v)/1e3:9.2f} ms {len(v):6d} x  p10 {q(.1):8.1f} p50 {q(.5):8.1f} p90 {q(.9):8.1f} max {v[-1]:8.1f} us  {n[:80]}")
PY
