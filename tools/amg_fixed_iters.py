#!/usr/bin/env python
"""Five AMG-PCG iterations of configuration 5's elasticity system with a fixed count (rtol = 0, maxiter = 5): a workload for
kernel traces of libraries whose results are wrong on purpose (tools/build_variant.sh nogather -DPF_ABLATE_GATHER)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 1025)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, qq, bq, p=5.0)
h = model.hctx
rho = h.to_device(np.random.default_rng(0).uniform(0.2, 1, X.shape[0]), torch.float64)
K = model.compute_jacobian(rho); rhs = model.compute_rhs_device(); K2, r2 = model.apply_dirichlet_bcs(K, rhs, True)
for _ in range(4):
    h.cg(K2.vals, r2, rtol=0.0, maxiter=5, precond=_lib.PF_PRECOND_AMG, allow_maxiter=True)
torch.cuda.synchronize()
