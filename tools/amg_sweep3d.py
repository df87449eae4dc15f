"""3-D elasticity AMG sweep: over-correction of the (unsmoothed) finest-level transfer and smoothing sweeps, 64^3 blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem, _lib
n = 64
c = pyfem.ProblemCreator(n + 1, n + 1, n + 1, element_type="block")
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBlock3D(); b = pyfem.BasisBlock3D(q)
for over in ["1.5", "2.0", "2.5", "3.0"]:
    for ns in ["1", "2", "3"]:
        os.environ["PF_AMG_OVER"] = over; os.environ["PF_AMG_NSMOOTH"] = ns
        m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=3.0)
        h = m.hctx
        rho = h.to_device(0.3 + 0.7 * np.random.RandomState(4).rand(X.shape[0]), torch.float64)
        K = m.compute_jacobian(rho); rhs = h.to_device(m.compute_rhs(), torch.float64)
        K, rhs = m.apply_dirichlet_bcs(K, rhs, True)
        h.cg(K.vals, rhs, rtol=1e-1, precond=_lib.PF_PRECOND_AMG)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, it, rr = h.cg(K.vals, rhs, rtol=1e-8, precond=_lib.PF_PRECOND_AMG)
        torch.cuda.synchronize(); print(f"over={over} nsmooth={ns}: {it} its {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
        del m, K, h
