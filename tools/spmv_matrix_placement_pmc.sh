#!/bin/bash
# Runs ON THE GPU BOX: counter passes of tools/spmv_matrix_placement_pmc.py, each repeated until the process holds a fast AND
# a slow matrix buffer (at most 5 tries per pass); summaries -> gpurun_out/<tag>/pmc_<pass>_summary.json
TAG=${1:-r3m}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  for try in 1 2 3 4 5; do
    rm -rf $OUT/pmc_$name
    timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python3 $R/tools/spmv_matrix_placement_pmc.py > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || return 1
    spread=$(python3 -c "import json;d=json.load(open('$OUT/pmc_$name.json'));t=d['ms_second'];print(int(100*(max(t)/min(t)-1)))")
    echo "pass $name try $try spread ${spread}%"
    if [ "$spread" -ge 4 ]; then
      python3 $R/tools/spmv_matrix_placement_pmc.py --summarise $OUT/pmc_$name > $OUT/pmc_${name}_summary.json
      rm -rf $OUT/pmc_$name
      return 0
    fi
  done
  rm -rf $OUT/pmc_$name
  echo "pass $name: no process with both speeds in 5 tries"
}
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_REQ_sum
pass tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
