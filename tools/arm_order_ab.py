#!/usr/bin/env python
"""Triangle assembly on the refined torque arm in three node numberings: refinement order (what red_refine_tri
produces), reverse Cuthill-McKee, and sorted along a Morton curve.  Same mesh, same K up to a permutation: what changes
is the locality of the coordinate gathers and whether the compact 8-byte records fit.  One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from scipy.sparse.csgraph import reverse_cuthill_mckee
import bench
from pyfem_gpu_testflight_amd import pyfem

def timer(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def morton(X):
    q = ((X - X.min(0)) / (X.max(0) - X.min(0)).max() * 65535).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x0000FFFF0000FFFF; v = (v | (v << 8)) & 0x00FF00FF00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F; v = (v | (v << 2)) & 0x3333333333333333
        return (v | (v << 1)) & 0x5555555555555555
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)), kind="stable")

conn, X, df, f = bench.arm_mesh(int(os.environ.get("LEVELS", "5")))
conn = np.asarray(conn); n = X.shape[0]
qt = pyfem.QuadratureTriangle2D(); bt = pyfem.BasisTriangle2D(qt)
out = {"nelems": int(conn.shape[0]), "nnodes": int(n)}
orders = {"refinement": None}
i = np.concatenate([conn[:, [0, 1]], conn[:, [1, 2]], conn[:, [2, 0]]])
G = sp.coo_matrix((np.ones(i.shape[0]), (i[:, 0], i[:, 1])), shape=(n, n)).tocsr(); G = G + G.T
orders["rcm"] = np.asarray(reverse_cuthill_mckee(G.tocsr(), symmetric_mode=True))
orders["morton"] = morton(X)
for name, perm in orders.items():
    if perm is None: Xp, cp = X, conn
    else:
        inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
        Xp, cp = X[perm], inv[conn]
        cp = cp[np.argsort(cp.min(1), kind="stable")]  # elements follow their lowest node
    m = pyfem.LinearElasticity(Xp, cp, [0, 1], None, {int(n - 1): [0.0, 1.0]}, qt, bt)
    K = m.compute_jacobian()
    maps = m.hctx.assembly_maps()
    out[name] = {"ms": round(timer(lambda: m.compute_jacobian(out=K)), 4), "rec_bytes": 8 if maps["rec8"] is not None else 12}
    m.hctx.close()
print(json.dumps(out))
