#!/usr/bin/env python
"""Filter solve and elasticity AMG-PCG solve of configuration 5 (2048 x 1024, random nodal design) for ONE library
(PF_LIB_PATH selects it; experiment knobs through the environment): ms per Helmholtz apply and per solve incl. the numeric
setup, iteration counts.  The A/B driver of round 3's solver changes (first use: the V-cycle tail kernel that was NOT kept,
profiles/r03_vcycle_tail_ab.jsonl -- hence the PF_AMG_TAIL field of the output line)."""
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
