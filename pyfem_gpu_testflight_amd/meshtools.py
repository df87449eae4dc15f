"""Mesh preprocessing helpers for the callers of the hot path (host side, numpy).

The reference has nothing of the kind: its unstructured case is the 2494-triangle `arm.inp`
(examples/torque_arm), and gmsh is not available to re-mesh it.  BASELINE.json's configuration 4 asks for a
~4M-triangle unstructured mesh, so the way to get one is uniform red refinement of that mesh, and
a refined mesh numbered level by level needs a locality-restoring renumbering before it is a sensible input
for any assembly kernel (tiles are runs of consecutive node ids)."""
import numpy as np


def red_refine_tri(X, conn, node_sets=None):
    """One level of red refinement: every triangle -> 4 (corner triangles keep the parent's orientation, so the
    sign of detJ is preserved -- arm.inp has detJ < 0 throughout).  New nodes are edge midpoints,
    appended after the old ones in order of the sorted (lo, hi) edge key.  node_sets: dict name -> node ids; a
    midpoint joins a set when both ends of its edge are in it (boundary groups such as "fixed"/"load")."""
    X = np.asarray(X, dtype=float)
    conn = np.asarray(conn, dtype=np.int64)
    n = X.shape[0]
    e = np.concatenate([conn[:, [0, 1]], conn[:, [1, 2]], conn[:, [2, 0]]], axis=0)
    lo, hi = e.min(axis=1), e.max(axis=1)
    key = lo * n + hi
    ukey, inv = np.unique(key, return_inverse=True)
    ulo, uhi = ukey // n, ukey % n
    mid = n + inv.reshape(3, -1).T  # (E,3): midpoints of edges 01, 12, 20
    X2 = np.concatenate([X, 0.5 * (X[ulo] + X[uhi])], axis=0)
    a, b, c = conn[:, 0], conn[:, 1], conn[:, 2]
    m01, m12, m20 = mid[:, 0], mid[:, 1], mid[:, 2]
    conn2 = np.empty((4 * conn.shape[0], 3), dtype=np.int64)
    conn2[0::4] = np.stack([a, m01, m20], axis=1)
    conn2[1::4] = np.stack([m01, b, m12], axis=1)
    conn2[2::4] = np.stack([m20, m12, c], axis=1)
    conn2[3::4] = np.stack([m01, m12, m20], axis=1)
    sets2 = None
    if node_sets is not None:
        sets2 = {}
        for name, ids in node_sets.items():
            mask = np.zeros(n, dtype=bool)
            mask[np.asarray(ids, dtype=np.int64)] = True
            new = n + np.flatnonzero(mask[ulo] & mask[uhi])
            sets2[name] = np.concatenate([np.asarray(ids, dtype=np.int64), new])
    return X2, conn2, sets2


def morton_node_order(X):
    """new_id[old_id]: nodes numbered along a Morton (Z-order) curve of their coordinates (16 bits per axis)."""
    X = np.asarray(X, dtype=float)
    lo, hi = X.min(axis=0), X.max(axis=0)
    q = ((X - lo) / np.where(hi > lo, hi - lo, 1.0) * 65535.0).astype(np.uint64)
    nd = X.shape[1]
    code = np.zeros(X.shape[0], dtype=np.uint64)
    for bit in range(16):
        for ax in range(nd):
            code |= ((q[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(bit * nd + ax)
    order = np.argsort(code, kind="stable")  # order[k] = old id of the k-th node
    new_id = np.empty(X.shape[0], dtype=np.int64)
    new_id[order] = np.arange(X.shape[0])
    return new_id


def renumber_nodes(X, conn, new_id, node_sets=None):
    """Apply new_id (old -> new) to the mesh; elements are then sorted by their smallest node so that element
    ids follow the node order too (element-block partitions stay compact)."""
    X = np.asarray(X)
    X2 = np.empty_like(X)
    X2[new_id] = X
    conn2 = new_id[np.asarray(conn, dtype=np.int64)]
    conn2 = conn2[np.argsort(conn2.min(axis=1), kind="stable")]
    sets2 = None if node_sets is None else {k: np.sort(new_id[np.asarray(v, dtype=np.int64)]) for k, v in node_sets.items()}
    return X2, conn2, sets2
