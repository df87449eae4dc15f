"""Host / launch overhead of the distributed CG drivers against the native single-GPU pf_cg, one rank, on the 4M-quad
mesh and on a 1/8 strip of it (the share of one GPU at N = 8: 2048 x 256 quads).  Three drivers of the SAME
single-reduction iteration: pf_cg (3 launches per iteration), pf_dist_cg (the native loop over the peer-store exchange:
6 launches, the exchange kernels run with a window that has no peers) and dist.py's Python loop (ctypes launches +
torch for the scalars).   python tools/dist_cg_overhead.py  ->  one JSON line per mesh (profiles/r03_dist_cg_overhead.jsonl)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem
from pyfem_gpu_testflight_amd.dist import PartitionedElasticity, dist_cg
ITERS = 400
for nx, ny in ((2048, 2048), (2048, 256)):
    c = pyfem.ProblemCreator(nx + 1, ny + 1)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
    out = {"mesh": f"{nx}x{ny} quads", "iterations_timed": ITERS}
    for tag, transport, native in (("pf_dist_cg_native", "peer", True), ("python_loop_peer_exchange", "peer", False), ("python_loop_torch", "p2p", False)):
        pm = PartitionedElasticity(X, conn, dof_fixed, None, force, q, b, rank=0, world=1, transport=transport)
        assert pm.iface.transport == transport
        K = pm.compute_jacobian()
        rhs = pm.ops.from_numpy(pm.model.compute_rhs(), torch.float64)
        Kbc, bb = pm.apply_dirichlet_bcs(K, rhs)
        dinv = 1.0 / pm.iface.sum(pm.ops.diag(Kbc.vals))
        def run(n):
            try:
                dist_cg(pm.ops, Kbc.vals, bb, dinv, pm.own_w, pm.iface, rtol=0.0, maxiter=n, check_every=10**9, native=native)
            except RuntimeError:
                pass
        run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(ITERS); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[tag + "_it_per_s"] = ITERS / dt
    h = pm.model.hctx
    h.cg(Kbc.vals, bb, rtol=0.0, maxiter=20, check_every=20, allow_maxiter=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.cg(Kbc.vals, bb, rtol=0.0, maxiter=ITERS, check_every=ITERS, allow_maxiter=True)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    out["pf_cg_it_per_s"] = ITERS / dt2
    out["native_dist_over_pf_cg"] = out["pf_dist_cg_native_it_per_s"] / out["pf_cg_it_per_s"]
    print(json.dumps(out), flush=True)
    del pm
