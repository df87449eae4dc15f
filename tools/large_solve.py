"""One AMG-PCG solve of the clamped plane-stress problem on an NX x NX quad mesh (default 5120: 26 M elements,
52 M dofs, 7.6 GB of K values): capacity check of the symbolic and numeric AMG phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PF_AMG_VERBOSE", "1")
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
nx = int(os.environ.get("NX", 5120))
t0 = time.perf_counter()
c = pyfem.ProblemCreator(nx + 1, nx + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
h = m.hctx
torch.cuda.synchronize(); print(f"mesh + pattern {time.perf_counter() - t0:.2f} s, ndof {h.nrows}", flush=True)
K = m.compute_jacobian(); rhs = h.to_device(m.compute_rhs(), torch.float64)
K, rhs = m.apply_dirichlet_bcs(K, rhs, True)
for k in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u, it, rr = h.cg(K.vals, rhs, rtol=1e-8, precond=_lib.PF_PRECOND_AMG)
    torch.cuda.synchronize(); print(f"solve {k}: {it} iterations, relres {rr:.2e}, {time.perf_counter() - t0:.3f} s", flush=True)
res = rhs - h.spmv(K.vals, u)
print("true relative residual", float(res.norm() / rhs.norm()), "levels", h.amg_info())
print("GPU memory allocated by torch: %.1f GB" % (torch.cuda.memory_allocated() / 1e9))
