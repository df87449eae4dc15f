"""Timing/profiling driver for the AMG-preconditioned solves of the C5 inner iteration (2048x1024).
   python tools/prof_amg.py            (wrap in rocprofv3 --kernel-trace --stats for the kernel breakdown)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem, _lib
c = pyfem.ProblemCreator(2049, 1025)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
f = pyfem.Helmholtz(0.05, X, conn, q, b); f.rtol = 1e-8
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=5.0)
h = m.hctx
x = h.to_device(np.random.default_rng(0).uniform(1e-3, 1, X.shape[0]), torch.float64)
f.apply(x); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); rho = f.apply(x); torch.cuda.synchronize(); print("filter apply", time.perf_counter() - t0, f.last_iters)
K = m.compute_jacobian(rho.clamp(1e-3, 1)); rhs = h.to_device(m.compute_rhs(), torch.float64)
K, rhs = m.apply_dirichlet_bcs(K, rhs, True)
for _ in range(2):
    t0 = time.perf_counter(); u, it, rr = h.cg(K.vals, rhs, rtol=1e-8, precond=_lib.PF_PRECOND_AMG); torch.cuda.synchronize(); print("elasticity solve", time.perf_counter() - t0, it)
for _ in range(3):
    t0 = time.perf_counter(); rho = f.apply(x); torch.cuda.synchronize(); print("filter apply after elasticity", time.perf_counter() - t0, f.last_iters)
g = torch.ones_like(x)
for _ in range(3):
    t0 = time.perf_counter(); f.apply_gradient(g); torch.cuda.synchronize(); print("filter gradient", time.perf_counter() - t0, f.last_iters)
t0 = time.perf_counter(); y = h.spmv(K.vals, rhs); torch.cuda.synchronize(); print("one spmv call", time.perf_counter() - t0)
