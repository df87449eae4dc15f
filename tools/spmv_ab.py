#!/usr/bin/env python
"""SpMV / CG A/B for one library (PF_LIB_PATH): d = 2 on the 4M-quad plane-stress matrix, d = 1 on the Helmholtz 2048^2 matrix,
200 Jacobi-PCG iterations, one AMG-PCG solve on 2048 x 1024.   python tools/spmv_ab.py <tag>"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
tag = sys.argv[1] if len(sys.argv) > 1 else "product"
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); K = m.compute_jacobian(); h = m.hctx
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
rhs = h.to_device(m.compute_rhs(), torch.float64); Kbc, b = m.apply_dirichlet_bcs(K, rhs, True)
c3 = pyfem.ProblemCreator(2049, 2049); conn3, X3, x3 = c3.create_helmhotz_problem()
mh = pyfem.Helmholtz(0.05, X3, conn3, qq, bq); hh = mh.hctx; mh.reuse_forward = False  # every timed apply() is a solve
xh = torch.rand(hh.nrows, dtype=torch.float64, device="cuda"); yh = torch.empty_like(xh)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K.vals, in1=x, out0=y)  # steady clocks first (~100 launches after idle)
for rep in range(3):
    out = {"lib": tag, "spmv_d2_ms": round(h.time_kernel(_lib.PF_K_SPMV, 200, in0=K.vals, in1=x, out0=y), 4),
           "spmv_d1_ms": round(hh.time_kernel(_lib.PF_K_SPMV, 200, in0=mh.K.vals, in1=xh, out0=yh), 4)}
    h.cg(Kbc.vals, b, rtol=0.0, maxiter=10, allow_maxiter=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); h.cg(Kbc.vals, b, rtol=0.0, maxiter=200, check_every=200, allow_maxiter=True); torch.cuda.synchronize()
    out["cg_it_per_s"] = round(200 / (time.perf_counter() - t0), 1)
    torch.cuda.synchronize(); t0 = time.perf_counter(); rho = mh.apply(xh); torch.cuda.synchronize()
    out["helmholtz_apply_ms"] = round(1e3 * (time.perf_counter() - t0), 3); out["helmholtz_iters"] = mh.last_iters
    torch.cuda.synchronize(); t0 = time.perf_counter(); _, it, _ = h.cg(Kbc.vals, b, rtol=1e-8, precond=_lib.PF_PRECOND_AMG); torch.cuda.synchronize()
    out["amg_pcg_2048sq_ms"] = round(1e3 * (time.perf_counter() - t0), 2); out["amg_iters"] = it
    print(json.dumps(out), flush=True)
