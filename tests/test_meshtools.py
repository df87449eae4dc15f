"""Host-side mesh helpers (pyfem_gpu_testflight_amd/meshtools.py): refinement and renumbering keep the
discretisation intact -- checked through the oracle's element matrices and assembled operator."""
import numpy as np

from oracle import pyfem_oracle as orc
from pyfem_gpu_testflight_amd import meshtools
from tests._golden import load


def _area(X, conn):
    a, b, c = X[conn[:, 0]], X[conn[:, 1]], X[conn[:, 2]]
    return 0.5 * ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))


def test_red_refinement_of_the_arm_mesh():
    g = load("arm_tri")
    X, conn = g["X"], g["conn"].astype(np.int64)
    fixed_nodes = np.unique(g["dof_fixed"] // 2)
    X2, conn2, sets = meshtools.red_refine_tri(X, conn, {"fixed": fixed_nodes, "load": g["force_nodes"]})
    assert conn2.shape[0] == 4 * conn.shape[0] and conn2.min() == 0 and conn2.max() == X2.shape[0] - 1
    A, A2 = _area(X, conn), _area(X2, conn2)
    assert np.all(np.sign(A2.reshape(-1, 4)) == np.sign(A)[:, None])  # orientation kept (detJ < 0 on arm.inp)
    assert np.allclose(A2.reshape(-1, 4).sum(axis=1), A, rtol=1e-12) and np.allclose(A2, np.repeat(A / 4, 4), rtol=1e-9)
    # conforming: every interior edge is shared by exactly two triangles, boundary edges by one
    e = np.sort(np.concatenate([conn2[:, [0, 1]], conn2[:, [1, 2]], conn2[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e[:, 0] * X2.shape[0] + e[:, 1], return_counts=True)
    assert set(np.unique(cnt)) <= {1, 2}
    assert sets["fixed"].size > fixed_nodes.size and np.all(np.isin(fixed_nodes, sets["fixed"]))
    # rigid-body modes stay in the null space of the refined operator
    K = orc.assemble_csr(orc.elasticity_Ke(X2, conn2, "tri"), orc.create_conn_dof(conn2, 2))
    u = np.zeros(2 * X2.shape[0]); u[0::2] = 1.0
    assert np.abs(K.dot(u)).max() < 1e-9 * np.abs(K.data).max()


def test_morton_renumbering_is_a_similarity_transform():
    conn, X = orc.structured_mesh(9, 7, "tri")
    rng = np.random.RandomState(0)
    X = X + 0.01 * rng.rand(*X.shape)
    new_id = meshtools.morton_node_order(X)
    assert np.array_equal(np.sort(new_id), np.arange(X.shape[0]))
    X2, conn2, sets = meshtools.renumber_nodes(X, conn, new_id, {"s": np.array([0, 5, 7])})
    assert np.allclose(X2[new_id], X) and np.array_equal(sets["s"], np.sort(new_id[[0, 5, 7]]))
    K = orc.assemble_csr(orc.elasticity_Ke(X, conn, "tri"), orc.create_conn_dof(conn, 2)).toarray()
    K2 = orc.assemble_csr(orc.elasticity_Ke(X2, conn2, "tri"), orc.create_conn_dof(conn2, 2)).toarray()
    perm = np.stack([2 * new_id, 2 * new_id + 1], axis=1).ravel()
    assert np.abs(K2[np.ix_(perm, perm)] - K).max() < 1e-12 * np.abs(K).max()
