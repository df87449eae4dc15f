#!/usr/bin/env python
"""A/B of the V-cycle tail kernel (k_amg_tail; NOT KEPT: see profiles/r03_vcycle_tail_ab.jsonl and the note in amg_cycle) on configuration 5's two systems: PF_AMG_TAIL=0/1 with the experiments
library (PF_LIB_PATH=.../libpyfem_hip_exp.so).  Prints ms per Helmholtz filter solve and per elasticity AMG-PCG solve."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 1025)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
filtr = pyfem.Helmholtz(0.05, X, conn, qq, bq); filtr.reuse_forward = False
model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, qq, bq, p=5.0)
h = model.hctx
xs = [h.to_device(np.random.default_rng(k).uniform(1e-3, 1, X.shape[0]), torch.float64) for k in range(3)]
asm = pyfem.Assembler(model)
def filt(): return filtr.apply(xs[0])
rho = filt().clamp(1e-3, 1.0)
def solve():
    K = model.compute_jacobian(rho); rhs = model.compute_rhs_device(); K2, r2 = model.apply_dirichlet_bcs(K, rhs, True)
    return asm._solve_linear_system(K2, r2, "cg")
out = {"PF_AMG_TAIL": os.environ.get("PF_AMG_TAIL", "default")}
for name, fn, reps in (("filter_ms", filt, 10), ("elasticity_ms", solve, 5)):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); out[name] = round(1e3 * (time.perf_counter() - t0) / reps, 3)
out["filter_iters"], out["elasticity_iters"] = filtr.last_iters, asm.last_iters
print(json.dumps(out))
