"""dist_cg (world 1) against the native pf_cg on a 1/8 strip of the 4M-quad mesh: host overhead of the distributed driver."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem
from pyfem_gpu_testflight_amd.dist import PartitionedElasticity, dist_cg
for nx, ny in ((2048, 2048), (2048, 256)):
    c = pyfem.ProblemCreator(nx + 1, ny + 1)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
    pm = PartitionedElasticity(X, conn, dof_fixed, None, force, q, b, rank=0, world=1)
    K = pm.compute_jacobian()
    rhs = pm.ops.from_numpy(pm.model.compute_rhs(), torch.float64)
    Kbc, bb = pm.apply_dirichlet_bcs(K, rhs)
    dinv = 1.0 / pm.iface.sum(pm.ops.diag(Kbc.vals))
    def run(n):
        try:
            dist_cg(pm.ops, Kbc.vals, bb, dinv, pm.own_w, pm.iface, rtol=0.0, maxiter=n, check_every=10**9)
        except RuntimeError:
            pass
    run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # host-only cost: time to enqueue (no sync inside run except final .item())
    h = pm.model.hctx
    x, it, rr = h.cg(Kbc.vals, bb, rtol=0.0, maxiter=400, check_every=400, allow_maxiter=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.cg(Kbc.vals, bb, rtol=0.0, maxiter=400, check_every=400, allow_maxiter=True)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(f"mesh {nx}x{ny}: dist_cg(world=1) {dt/400*1e6:.1f} us/iter ({400/dt:.0f} it/s); native pf_cg {dt2/400*1e6:.1f} us/iter ({400/dt2:.0f} it/s)")
