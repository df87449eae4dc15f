"""The native distributed Jacobi-PCG loop (pf_dist_cg: product + send, update) and pf_cg on one rank's share of the 4M-quad
mesh at N = 8 (2048 x 256 quads), for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/dist_cg_trace.py [nx ny iters]
   python3 tools/trace_gaps.py <dir>     -> per-kernel durations and the idle gaps between consecutive launches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem
from pyfem_gpu_testflight_amd.dist import PartitionedElasticity, dist_cg
nx, ny, iters = (int(a) for a in (sys.argv[1:4] + ["2048", "256", "300"][len(sys.argv) - 1:]))
c = pyfem.ProblemCreator(nx + 1, ny + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
pm = PartitionedElasticity(X, conn, dof_fixed, None, force, q, b, rank=0, world=1, transport="peer")
K = pm.compute_jacobian()
rhs = pm.ops.from_numpy(pm.model.compute_rhs(), torch.float64)
Kbc, bb = pm.apply_dirichlet_bcs(K, rhs)
dinv = 1.0 / pm.iface.sum(pm.ops.diag(Kbc.vals))
for name, fn in (("pf_dist_cg", lambda n: dist_cg(pm.ops, Kbc.vals, bb, dinv, pm.own_w, pm.iface, rtol=0.0, maxiter=n, check_every=10**9, native=True)),
                 ("pf_cg", lambda n: pm.model.hctx.cg(Kbc.vals, bb, rtol=0.0, maxiter=n, check_every=n, allow_maxiter=True))):
    for n in (50, iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            fn(n)
        except RuntimeError:
            pass
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {iters / dt:.0f} it/s ({1e6 * dt / iters:.1f} us per iteration)", flush=True)
