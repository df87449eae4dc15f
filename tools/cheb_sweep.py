"""Two-factor (Chebyshev degree-2) smoother: PF_AMG_CHEB sweep on plane stress 2048x1024, Helmholtz 2049x1025, 64^3 blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem, _lib
def solve(h, K, rhs):
    h.cg(K, rhs, rtol=1e-8, precond=_lib.PF_PRECOND_AMG)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, it, _ = h.cg(K, rhs, rtol=1e-8, precond=_lib.PF_PRECOND_AMG)
    torch.cuda.synchronize(); return it, time.perf_counter() - t0
c = pyfem.ProblemCreator(2049, 1025)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=3.0)
rho = m.hctx.to_device(0.3 + 0.7 * np.random.RandomState(1).rand(X.shape[0]), torch.float64)
K = m.compute_jacobian(rho); rhs = m.hctx.to_device(m.compute_rhs(), torch.float64)
K, rhs = m.apply_dirichlet_bcs(K, rhs, True)
f = pyfem.Helmholtz(0.05, X, conn, q, b)
xf = f.hctx.to_device(np.random.RandomState(0).uniform(1e-3, 1, X.shape[0]), torch.float64)
bf = f.hctx.spmv(f.R.vals, xf)
c3 = pyfem.ProblemCreator(65, 65, 65, element_type="block")
conn3, X3, fix3, force3 = c3.create_linear_elasticity_problem()
q3 = pyfem.QuadratureBlock3D(); b3 = pyfem.BasisBlock3D(q3)
m3 = pyfem.LinearElasticity(X3, conn3, fix3, None, force3, q3, b3)
K3 = m3.compute_jacobian(); rhs3 = m3.hctx.to_device(m3.compute_rhs(), torch.float64)
K3, rhs3 = m3.apply_dirichlet_bcs(K3, rhs3, True)
for cheb in os.environ.get("CHEB", "0 3 4.4 6 10").split():
    os.environ["PF_AMG_CHEB"] = cheb
    a = solve(m.hctx, K.vals, rhs); bq = solve(f.hctx, f.K.vals, bf); c_ = solve(m3.hctx, K3.vals, rhs3)
    print(f"cheb={cheb:>4s}: plane stress {a[0]} it {1e3*a[1]:.1f} ms | helmholtz {bq[0]} it {1e3*bq[1]:.1f} ms | blocks 64^3 {c_[0]} it {1e3*c_[1]:.1f} ms", flush=True)
