#!/bin/bash
# Runs ON THE GPU BOX: the assembly kernels (quads 2048^2, 1415^2 split triangles, arm5; round 3: quads with a density field,
# Helmholtz K + R) under separate counter passes; tools/pmc_assembly_summary.py prints per-kernel averages.
#   gpurun -- 'bash tools/pmc_assembly.sh r03'
set -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_asm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $REPO/tools/pmc_assembly.py"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o p -- $P > $OUT/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -o p -- $P > $OUT/b.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/c -o p -- $P > $OUT/c.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES --kernel-trace --output-format csv -d $OUT/d -o p -- $P > $OUT/d.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES --kernel-trace --output-format csv -d $OUT/e -o p -- $P > $OUT/e.log 2>&1 || echo "pass e failed"
timeout -k 10 300 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES --kernel-trace --output-format csv -d $OUT/f -o p -- $P > $OUT/f.log 2>&1 || echo "pass f failed"
timeout -k 10 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES --kernel-trace --output-format csv -d $OUT/g -o p -- $P > $OUT/g.log 2>&1 || echo "pass g failed"
python3 $REPO/tools/pmc_assembly_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
