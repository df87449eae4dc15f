#!/bin/bash
# Runs ON THE GPU BOX: counters of the big kernels of the AMG numeric setup and cycle on the finest 2-D elasticity level
# over tools/c5_parts.py, separate passes.   gpurun -- bash tools/pmc_amg_setup.sh
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_amg; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $REPO/tools/c5_parts.py"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o p -- $P > $OUT/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -o p -- $P > $OUT/b.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/c -o p -- $P > $OUT/c.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES --kernel-trace --output-format csv -d $OUT/e -o p -- $P > $OUT/e.log 2>&1 || echo "pass e failed"
timeout -k 10 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES --kernel-trace --output-format csv -d $OUT/g -o p -- $P > $OUT/g.log 2>&1 || echo "pass g failed"
python3 - <<PY
import csv, glob, collections
WANT = ("k_sa_galerkin_wave<2, 3, 1>", "k_sa_smooth_P<2, 2, 3>", "k_sa_prolong<2, 3>", "k_sa_restrict_stream<2, 3>", "k_spmv<2, 1, 3, 640, true, float>", "k_cg_update_xr")
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for sub in "abceg":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if any(w in k for w in WANT): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k in acc: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    d = sorted(dur[k]); print(k, "median us", d[len(d) // 2] if d else None)
    for c, v in sorted(cs.items()): print(f"   {c:30s} {sum(v) / len(v):16.0f}")
PY
