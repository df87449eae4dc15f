#!/bin/bash
# A/B variant of the product library: recompiles pf_pattern.hip, pf_assemble.hip, pf_solve.hip, pf_halo.hip and pf_amg.hip with extra -D flags and links them
# with the product's other objects into pyfem_gpu_testflight_amd/libpyfem_hip_<name>.so (select with PF_LIB_PATH).
#   tools/build_variant.sh p1 -DPF_TRI_PASSES=1
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/pyfem_gpu_testflight_amd
python -m pyfem_gpu_testflight_amd.build > /dev/null
mkdir -p /tmp/pfvar_$name
for f in pf_pattern pf_assemble pf_solve pf_halo pf_amg; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -I $R/include -I $P/csrc -c $P/csrc/$f.hip -o /tmp/pfvar_$name/$f.o &
done
wait
B=$P/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libpyfem_hip_$name.so /tmp/pfvar_$name/pf_pattern.o /tmp/pfvar_$name/pf_assemble.o /tmp/pfvar_$name/pf_solve.o /tmp/pfvar_$name/pf_halo.o \
  /tmp/pfvar_$name/pf_amg.o $B/pf_assemble3d.o $B/pf_tabular.o $B/pf_dist.o $B/pf_bench.o
echo $P/libpyfem_hip_$name.so
