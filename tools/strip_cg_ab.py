#!/usr/bin/env python
"""One rank's share of the metric mesh at N = 8 (2048 x 256 quads) for one library (PF_LIB_PATH): product time, pf_cg and
pf_dist_cg iterations/s -- the numbers the 8-GPU CG estimate of DESIGN 7.1 is made of.   python tools/strip_cg_ab.py <tag>"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
from pyfem_gpu_testflight_amd.dist import PartitionedElasticity, dist_cg
tag = sys.argv[1] if len(sys.argv) > 1 else "product"
nx, ny = (int(a) for a in (sys.argv[2:4] + ["2048", "256"][len(sys.argv) - 2:])) if len(sys.argv) > 2 else (2048, 256)
c = pyfem.ProblemCreator(nx + 1, ny + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
pm = PartitionedElasticity(X, conn, dof_fixed, None, force, q, b, rank=0, world=1, transport="peer")
h = pm.model.hctx
K = pm.compute_jacobian()
rhs = pm.ops.from_numpy(pm.model.compute_rhs(), torch.float64)
Kbc, bb = pm.apply_dirichlet_bcs(K, rhs)
dinv = 1.0 / pm.iface.sum(pm.ops.diag(Kbc.vals))
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def rate(fn, n=600):
    fn(100); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
def run_dist(n):
    try: dist_cg(pm.ops, Kbc.vals, bb, dinv, pm.own_w, pm.iface, rtol=0.0, maxiter=n, check_every=10**9, native=True)
    except RuntimeError: pass
def run_cg(n): h.cg(Kbc.vals, bb, rtol=0.0, maxiter=n, check_every=n, allow_maxiter=True)
for rep in range(2):
    out = {"lib": tag, "mesh": f"{nx}x{ny}", "spmv_ms": round(h.time_kernel(_lib.PF_K_SPMV, 300, in0=Kbc.vals, in1=x, out0=y), 5),
           "pf_cg_it_per_s": round(rate(run_cg)), "pf_dist_cg_it_per_s": round(rate(run_dist))}
    out["dist_us_per_it"] = round(1e6 / out["pf_dist_cg_it_per_s"], 1)
    print(json.dumps(out), flush=True)
