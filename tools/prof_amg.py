"""Timing/profiling driver for the AMG-preconditioned solves of the C5 inner iteration (2048x1024).
   python tools/prof_amg.py [3d]       (wrap in rocprofv3 --kernel-trace for the kernel breakdown: tools/amg_breakdown.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem, _lib
if len(sys.argv) > 1 and sys.argv[1] == "3d":  # 96^3 blocks, three AMG-PCG solves (the third is the one to look at)
    n = 96
    c = pyfem.ProblemCreator(n + 1, n + 1, n + 1, element_type="block")
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q = pyfem.QuadratureBlock3D(); b = pyfem.BasisBlock3D(q)
    m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=3.0)
    h = m.hctx
    K = m.compute_jacobian(); rhs = h.to_device(m.compute_rhs(), torch.float64)
    K, rhs = m.apply_dirichlet_bcs(K, rhs, True)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, its, rr = h.cg(K.vals, rhs, rtol=1e-8, precond=_lib.PF_PRECOND_AMG)
        torch.cuda.synchronize(); print("solve", its, time.perf_counter() - t0, flush=True)
    sys.exit(0)
c = pyfem.ProblemCreator(2049, 1025)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
f = pyfem.Helmholtz(0.05, X, conn, q, b); f.rtol = 1e-8; f.reuse_forward = False  # every timed apply() is a solve
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
