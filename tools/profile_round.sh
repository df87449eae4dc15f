#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace + separate PMC passes of the default bench
# command, then tools/summarize_prof.py writes the files that get copied to profiles/.
#   gpurun -- 'bash tools/profile_round.sh r01'
set -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --cg-iters 100 --no-cpu-baseline --legs rho,c3,c4"
# the driver's exact command (BENCH_rNN.json: "python3 bench.py --gpus 1 --steps 20 --warmup 5")
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/trace.log 2>&1 || exit 1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
echo "pmc write done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 1
echo "pmc sq done"
python3 $REPO/tools/summarize_prof.py $OUT $TAG
