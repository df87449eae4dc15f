L=$PWD/pyfem_gpu_testflight_amd/libpyfem_hip_exp.so
# Iteration counts of the AMG-PCG solves of six other systems against the number of power-iteration steps of the lambda_max
# estimates (PF_AMG_POW, experiments library).  Runs ON THE GPU BOX:  gpurun -- bash tools/pow_robust.sh
mkdir -p gpurun_out/pow
for cfg in 20 10 8 6 5; do set -- $cfg 0; echo "== PF_AMG_POW $1"
PF_AMG_POW=$1 PF_AMG_VERBOSE=1 PF_LIB_PATH=$L timeout -k 10 100 python tools/pow_probe.py 2> gpurun_out/pow/err_$1.txt | cut -c1-400 || exit 1
grep "lambda_max" gpurun_out/pow/err_$1.txt | sort | uniq -c | sort -rn | head -2 | cut -c90-200
PF_AMG_POW=$1 PF_LIB_PATH=$L timeout -k 10 150 python tools/bench3d.py 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('3d block', d['n'], d['solve_rtol1e-8']['amg'])" || exit 1
PF_AMG_POW=$1 PF_LIB_PATH=$L timeout -k 10 150 python tools/bench_tab3d.py 33 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['element'], d['amg_pcg_iters'])" || exit 1
PF_AMG_POW=$1 PF_LIB_PATH=$L timeout -k 10 200 python tools/bench_configs.py c3 c4arm 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l[0]!='{': continue
    d=json.loads(l); print(d.get('config')[:40], {k:v for k,v in d.items() if 'iters' in k})" || exit 1
done
