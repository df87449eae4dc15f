#!/usr/bin/env python
"""The SpMV placement effect under counters.  One process: the metric-mesh matrix, NC candidate output
vectors (separate allocations, optionally with gaps between them); each candidate is timed (HIP events, 40 launches),
then the FASTEST and the SLOWEST are each run REP times back to back so that a rocprofv3 --pmc pass of this script can
tell them apart by dispatch order: the k_spmv dispatches come as  [timing passes ...][fast x REP][slow x REP]  and the
script prints the dispatch counts.   Also reports virtual addresses and which candidates are fast.
   python tools/spmv_placement_pmc.py [NC] [REP]
   rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -o p -- python3 tools/spmv_placement_pmc.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 20
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); h = m.hctx
K = m.compute_jacobian().vals
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda")
ys, pads = [], []
for i in range(NC):
    ys.append(torch.empty_like(x))
    pads.append(torch.empty((i * 37 + 11) * 4099, dtype=torch.float64, device="cuda"))  # odd-sized gaps between candidates
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K, in1=x, out0=ys[0])  # steady clocks
times = [h.time_kernel(_lib.PF_K_SPMV, 40, in0=K, in1=x, out0=y) for y in ys]
times2 = [h.time_kernel(_lib.PF_K_SPMV, 40, in0=K, in1=x, out0=y) for y in ys]
fast, slow = min(range(NC), key=lambda i: times2[i]), max(range(NC), key=lambda i: times2[i])
torch.cuda.synchronize()
for i in (fast, slow):
    for _ in range(REP):
        h.spmv(K, x, out=ys[i])
    torch.cuda.synchronize()
print(json.dumps({"ms_first_pass": [round(t, 4) for t in times], "ms_second_pass": [round(t, 4) for t in times2], "fast": fast, "slow": slow,
                  "rep": REP, "spmv_dispatches_before_tagged": 300 + 1 + 2 * NC * 41,
                  "vals_ptr": hex(K.data_ptr()), "x_ptr": hex(x.data_ptr()), "y_ptrs": [hex(y.data_ptr()) for y in ys],
                  "y_ptr_mod_2MiB": [y.data_ptr() % (2 << 20) for y in ys], "y_ptr_mod_1GiB_MiB": [(y.data_ptr() % (1 << 30)) >> 20 for y in ys]}))
