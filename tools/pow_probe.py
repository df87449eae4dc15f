#!/usr/bin/env python
"""How many power-iteration steps do the lambda_max(D^-1 A) estimates of the AMG setup need?  Configuration 5's elasticity system
(random nodal design, RAMP p = 5) and the Helmholtz filter, solved with PF_AMG_POW = 3..20 steps (experiments library):
the estimates per level (PF_AMG_VERBOSE), PCG iterations and solve time.
   for k in 3 4 5 6 8 10 20; do PF_AMG_POW=$k PF_LIB_PATH=.../libpyfem_hip_exp.so python tools/pow_probe.py; done"""
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
asm = pyfem.Assembler(model)
# SWEEP="pow:omega_scale:psmooth,..." runs every setting in this process (the knobs are read at each numeric setup)
settings = [tuple(t.split(":")) for t in os.environ.get("SWEEP", "").split(",") if t] or [None]
designs = []
for seed in range(3):
    x0 = h.to_device(np.random.default_rng(seed).uniform(1e-3, 1, X.shape[0]), torch.float64)
    designs.append(filtr.apply(x0).clamp(1e-3, 1.0) if seed < 2 else x0)   # third design: unfiltered noise (contrast 1e3 node to node)
for st in settings:
  if st:
    os.environ["PF_AMG_POW"], os.environ["PF_AMG_OMEGA_SCALE"], os.environ["PF_AMG_PSMOOTH"] = st
  out = {"PF_AMG_POW": os.environ.get("PF_AMG_POW", "default (10)"), "PF_AMG_OMEGA_SCALE": os.environ.get("PF_AMG_OMEGA_SCALE", "1"),
         "PF_AMG_PSMOOTH": os.environ.get("PF_AMG_PSMOOTH", "1"), "designs": []}
  for seed in range(3):
    rho = designs[seed]
    def solve():
        K = model.compute_jacobian(rho); rhs = model.compute_rhs_device(); K2, r2 = model.apply_dirichlet_bcs(K, rhs, True)
        return asm._solve_linear_system(K2, r2, "cg")
    solve(); solve(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): solve()
    torch.cuda.synchronize()
    out["designs"].append({"seed": seed, "ms": round(1e3 * (time.perf_counter() - t0) / 3, 2), "iters": asm.last_iters})
  print(json.dumps(out), flush=True)
