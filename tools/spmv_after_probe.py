#!/usr/bin/env python
"""bench.py measured the 2x2 SpMV at 0.22-0.24 ms in its own process against 0.2117 ms in a fresh one
(tools/spmv_placement_vs_ramp.py).  What bench.py does before: pf_hbm_probe allocates and frees 1.2 - 3.6 GB with
hipMalloc / hipFree, THEN the matrix and the vectors are allocated.  This script times the product (a) in a fresh process,
(b) with matrix and vectors allocated after the probes' hipMalloc / hipFree churn, (c) the same vectors again later.
   python tools/spmv_after_probe.py [probe|noprobe]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "probe"
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); h = m.hctx
out = {"mode": mode}
if mode == "probe":
    nb = 8 * h.nnz
    h.hbm_probe(_lib.PF_HBM_MIX, nb, reps=150)
    for md in (_lib.PF_HBM_FILL, _lib.PF_HBM_READ, _lib.PF_HBM_MIX):
        h.hbm_probe(md, nb, reps=60); h.hbm_probe(md, nb, reps=60, nt=1)
Kc = m.compute_jacobian(); K = Kc.vals
for _ in range(300): m.compute_jacobian(out=Kc)
torch.cuda.synchronize()
x = torch.rand(h.nrows, dtype=torch.float64).to("cuda")
ys = [torch.empty_like(x) for _ in range(3)]
sw = lambda reps: [round(h.time_kernel(_lib.PF_K_SPMV, reps, in0=K, in1=x, out0=y), 4) for y in ys]
out["first_sweep_50"] = sw(50); out["second_sweep_50"] = sw(50)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K, in1=x, out0=ys[0]); out["after_300_warm"] = sw(50)
K2 = K.clone(); out["clone_of_K"] = [round(h.time_kernel(_lib.PF_K_SPMV, 50, in0=K2, in1=x, out0=y), 4) for y in ys]
out["ptrs"] = {"K": hex(K.data_ptr()), "K2": hex(K2.data_ptr()), "x": hex(x.data_ptr()), "ys": [hex(y.data_ptr()) for y in ys]}
print(json.dumps(out))
