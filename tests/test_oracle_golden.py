"""Pins oracle/pyfem_oracle.py (the CPU restatement) against fixtures produced by the
reference itself (oracle/gen_golden.py) and the known answers SURVEY.md 8c records."""
import numpy as np
import pytest

from oracle import pyfem_oracle as orc
from tests._golden import ELASTICITY_CASES, HELMHOLTZ_CASES, csr_of, force_dict, load, relmax

K_TOL = 1e-13  # fp reordering noise; the product bar is 1e-10 (BASELINE.json north_star)
U_TOL = 1e-10


@pytest.mark.parametrize("name", list(ELASTICITY_CASES))
def test_elasticity_matches_reference(name):
    g = load(name)
    etype = ELASTICITY_CASES[name]
    X, conn = g["X"], g["conn"].astype(np.int64)
    rho = g.get("rho")
    Ke = orc.elasticity_Ke(X, conn, etype, rho=rho, p=float(g["p"]))
    assert relmax(Ke[0], g["Ke0"]) < K_TOL
    assert relmax(Ke[-1], g["Ke_last"]) < K_TOL
    K = orc.assemble_csr(Ke, orc.create_conn_dof(conn, X.shape[1]))
    assert K.nnz == int(g["K_nnz"])
    assert abs(np.linalg.norm(K.data) - float(g["K_norm"])) < 1e-12 * float(g["K_norm"])
    assert relmax(K.diagonal(), g["K_diag"]) < K_TOL
    assert relmax(K.dot(g["v"]), g["Kv"]) < K_TOL
    if "K_data" in g:
        Kref = csr_of(g, "K", K.shape[0])
        assert np.array_equal(K.indptr, Kref.indptr) and np.array_equal(K.indices, Kref.indices)
        assert np.abs(K.data - Kref.data).max() < K_TOL * np.abs(Kref.data).max()
    rhs = orc.elasticity_rhs(K.shape[0], force_dict(g))
    assert np.array_equal(rhs, g["rhs"])
    vals = g.get("fixed_vals")
    Kbc, rhs_bc = orc.apply_dirichlet_bcs(K, rhs, g["dof_fixed"], vals, True)
    # post-BC nnz is data dependent (eliminate_zeros drops analytically-zero entries only when
    # they round to exactly 0.0), so compare as operators, not by count
    assert relmax(Kbc.dot(g["v"]), g["Kbc_v"]) < K_TOL
    assert np.abs(rhs_bc - g["rhs_bc"]).max() <= 1e-13 * max(1.0, np.abs(g["rhs_bc"]).max())
    Kns, rhs_ns = orc.apply_dirichlet_bcs(K, rhs, g["dof_fixed"], vals, False)
    assert relmax(Kns.dot(g["v"]), g["Kns_v"]) < K_TOL
    assert np.abs(rhs_ns - g["rhs_ns"]).max() <= 1e-13 * max(1.0, np.abs(g["rhs_ns"]).max())
    from scipy.sparse.linalg import spsolve

    u = spsolve(Kbc.tocsc(), rhs_bc)
    assert relmax(u, g["u"]) < U_TOL


def test_survey_known_answers():
    """SURVEY.md 8c probe values (reference imported unmodified)."""
    conn, X = orc.structured_mesh(32, 32, "quad")
    dof_fixed, force = orc.elasticity_problem(32, 32)
    u, K, _, _ = orc.solve_elasticity(X, conn, "quad", dof_fixed, force)
    np.random.seed(123)
    p = np.random.rand(u.shape[0])
    assert abs(p.dot(u) / -164.42713355728523 - 1) < 1e-10  # tests/test_elasticity.py:42-48 bar
    assert K.nnz == 35344
    assert abs(np.linalg.norm(K.data) / 970.6724718974741 - 1) < 1e-13
    Ke = orc.elasticity_Ke(X, conn, "quad")
    row0 = [4.945054945055, 1.785714285714, -3.021978021978, -0.137362637363,
            -2.472527472527, -1.785714285714, 0.549450549451, 0.137362637363]
    assert np.allclose(Ke[0, 0], row0, atol=1e-11)
    # 2-element unit-square triangle mesh
    Xt = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    ct = np.array([[0, 1, 3], [3, 2, 0]])
    Kt = orc.elasticity_Ke(Xt, ct, "tri")
    assert np.allclose(Kt[0, 0], [5.494505494505, 0, -5.494505494505, 1.648351648352, 0, -1.648351648352], atol=1e-11)


def test_jacobi_pcg_matches_direct():
    g = load("ps13x9_quad")
    u, *_ = orc.solve_elasticity(g["X"], g["conn"].astype(np.int64), "quad", g["dof_fixed"], force_dict(g),
                                 method="cg", rtol=1e-12)
    assert relmax(u, g["u"]) < 1e-8


@pytest.mark.parametrize("name", list(HELMHOLTZ_CASES))
def test_helmholtz_matches_reference(name):
    g = load(name)
    etype = HELMHOLTZ_CASES[name]
    X, conn = g["X"], g["conn"].astype(np.int64)
    Ke, Re = orc.helmholtz_Ke_Re(X, conn, etype, float(g["r0"]))
    assert relmax(Ke[0], g["Ke0"]) < K_TOL and relmax(Re[0], g["Re0"]) < K_TOL
    f = orc.HelmholtzFilter(float(g["r0"]), X, conn, etype)
    n = X.shape[0]
    for M, pre in ((f.K, "K"), (f.R, "R")):
        ref = csr_of(g, pre, n)
        assert np.array_equal(M.indptr, ref.indptr) and np.array_equal(M.indices, ref.indices)
        assert np.abs(M.data - ref.data).max() < K_TOL * np.abs(ref.data).max()
    assert relmax(f.R.dot(g["x"]), g["Rx"]) < K_TOL
    assert relmax(f.apply(g["x"]), g["rho"]) < 1e-10
    assert relmax(f.apply_gradient(g["g"]), g["gradx"]) < 1e-10
    if name == "helm32_quad":
        np.random.seed(123)
        p = np.random.rand(n)
        assert abs(p.dot(f.apply(g["x"])) / 116.98194101446458 - 1) < 1e-8  # tests/test_helmholtz.py:42


def test_mesh_generator_matches_reference():
    g = load("mesh6x4")
    for etype in ("quad", "tri"):
        conn, X = orc.structured_mesh(6, 4, etype)
        assert np.array_equal(conn, g[f"{etype}_conn"])
        assert np.allclose(X, g[f"{etype}_X"], rtol=0, atol=1e-15)
        dof_fixed, force = orc.elasticity_problem(6, 4)
        assert np.array_equal(dof_fixed, g[f"{etype}_dof_fixed"])
        assert list(force.keys()) == list(g[f"{etype}_force_nodes"]) and np.array_equal(
            np.array(list(force.values())), g[f"{etype}_force_vals"])
        assert np.array_equal(orc.helmholtz_input(6, 4), g[f"{etype}_helm_x"])


@pytest.mark.parametrize("name,etype", [("sens24x17_quad", "quad"), ("sens21x16_tri", "tri"),
                                        ("sens3d_5x4x4_block", "block")])
def test_sensitivity_matches_reference(name, etype):
    """_compute_K_dv_sens and compliance_grad as the reference itself computed them (p = 5)."""
    g = load(name)
    X, conn = g["X"], g["conn"].astype(np.int64)
    s = orc.elasticity_K_dv_sens(X, conn, etype, g["rho"], g["phi"], g["psi"], p=float(g["p"]))
    assert relmax(s, g["sens"]) < 1e-13
    grad = -orc.elasticity_K_dv_sens(X, conn, etype, g["rho"], g["u"], g["u"], p=float(g["p"]))
    assert relmax(grad, g["compliance_grad"]) < 1e-12


def _gfunc(x):  # the source term of the reference's test (tests/test_linear_poisson.py:11-14)
    _x = x[..., 0]
    _y = x[..., 1]
    return _x * (_x - 5.0) * (_x - 10.0) * _y * (_y - 4.0)


@pytest.mark.parametrize("name", ["poi32_quad", "poi17x12_quad_ramp"])
def test_poisson_matches_reference(name):
    g = load(name)
    X, conn = g["X"], g["conn"].astype(np.int64)
    p = float(g["p"])
    Ke = orc.poisson_Ke(X, conn, "quad", g["rho"], p)
    assert relmax(Ke[0], g["Ke0"]) < K_TOL
    K = orc.assemble_csr(Ke, conn)
    ref = csr_of(g, "K", X.shape[0])
    assert np.array_equal(K.indptr, ref.indptr) and np.array_equal(K.indices, ref.indices)
    assert np.abs(K.data - ref.data).max() < K_TOL * np.abs(ref.data).max()
    rhs = orc.poisson_rhs(X, conn, "quad", _gfunc)
    assert relmax(rhs, g["rhs"]) < K_TOL
    Kbc, rhs_bc = orc.apply_dirichlet_bcs(K, rhs, g["dof_fixed"], g.get("fixed_vals"), True)
    assert relmax(rhs_bc, g["rhs_bc"]) < 1e-12
    from scipy.sparse.linalg import spsolve

    u = spsolve(Kbc.tocsc(), rhs_bc)
    assert relmax(u, g["u"]) < U_TOL
    assert relmax(orc.poisson_K_dv_sens(X, conn, "quad", g["rho"], g["phi"], g["psi"], p), g["sens"]) < 1e-13
    if name == "poi32_quad":
        np.random.seed(123)
        pr = np.random.rand(u.shape[0])
        assert abs(pr.dot(u) / -7082.301628403809 - 1) < 1e-10
        assert np.allclose(Ke[0, 0], [2 / 3, -1 / 6, -1 / 3, -1 / 6], atol=1e-13)


# ---- 3-D 8-node block element ------------------------------------------------
def _gfunc3d(x):  # source term used by oracle/gen_golden.py main_3d
    return x[..., 0] * (x[..., 1] - 2.0) + x[..., 2]


def test_block_tables_and_mesh_match_reference():
    g = load("mesh3d_4x3x3")
    conn, X = orc.structured_mesh3d(4, 3, 3)
    assert np.array_equal(conn, g["conn"]) and np.allclose(X, g["X"], rtol=0, atol=1e-15)
    dof_fixed, force = orc.elasticity_problem3d(4, 3, 3)
    assert np.array_equal(dof_fixed, g["dof_fixed"])
    assert list(force.keys()) == list(g["force_nodes"]) and np.array_equal(np.array(list(force.values())), g["force_vals"])
    assert np.array_equal(orc.helmholtz_input3d(4, 3, 3), g["helm_x"])
    assert np.array_equal(orc.poisson_fixed3d(4, 3, 3), g["poisson_fixed"])
    N, dN = orc.shape_tables("block")
    assert np.allclose(N.sum(axis=1), 1.0) and np.allclose(dN.sum(axis=1), 0.0)


def test_block_helmholtz_matches_reference():
    g = load("helm3d_5x4x3_block")
    X, conn = g["X"], g["conn"].astype(np.int64)
    f = orc.HelmholtzFilter(float(g["r0"]), X, conn, "block")
    n = X.shape[0]
    for M, pre in ((f.K, "K"), (f.R, "R")):
        ref = csr_of(g, pre, n)
        assert np.array_equal(M.indptr, ref.indptr) and np.array_equal(M.indices, ref.indices)
        assert np.abs(M.data - ref.data).max() < K_TOL * np.abs(ref.data).max()
    assert relmax(f.apply(g["x"]), g["rho"]) < 1e-10
    assert relmax(f.apply_gradient(g["g"]), g["gx"]) < 1e-10


def test_block_poisson_matches_reference():
    g = load("poi3d_5x4x3_block")
    X, conn = g["X"], g["conn"].astype(np.int64)
    p = float(g["p"])
    Ke = orc.poisson_Ke(X, conn, "block", g["rho"], p)
    assert relmax(Ke[0], g["Ke0"]) < K_TOL
    K = orc.assemble_csr(Ke, conn)
    ref = csr_of(g, "K", X.shape[0])
    assert np.array_equal(K.indptr, ref.indptr) and np.array_equal(K.indices, ref.indices)
    assert np.abs(K.data - ref.data).max() < K_TOL * np.abs(ref.data).max()
    rhs = orc.poisson_rhs(X, conn, "block", _gfunc3d)  # N[j,k] quirk: NOT the consistent load vector in 3-D
    assert relmax(rhs, g["rhs"]) < K_TOL
    Kbc, rhs_bc = orc.apply_dirichlet_bcs(K, rhs, g["dof_fixed"], g["fixed_vals"], True)
    assert relmax(rhs_bc, g["rhs_bc"]) < 1e-12
    from scipy.sparse.linalg import spsolve

    assert relmax(spsolve(Kbc.tocsc(), rhs_bc), g["u"]) < U_TOL
    assert relmax(orc.poisson_K_dv_sens(X, conn, "block", g["rho"], g["phi"], g["psi"], p), g["sens"]) < 1e-13


@pytest.mark.parametrize("name,etype,shape", [("ho_tet_5x5x3", "tet", (5, 5, 3)), ("ho_brick20_5x5x5", "brick20", (5, 5, 5))])
def test_quadratic_elements_match_reference(name, etype, shape):
    """10-node tetrahedra / 20-node bricks (pyfem.py:115-172, 380-632, 2538-2700): tables, mesh generator, K with
    nodal RAMP density on a distorted mesh, sensitivities."""
    g = load(name)
    pts, w = orc.quadrature(etype)
    N, dN = orc.shape_tables(etype)
    assert np.abs(pts - g["qpts"]).max() < 1e-15 and np.abs(w - g["qw"]).max() == 0.0
    assert np.abs(N - g["N"]).max() < 1e-15 and np.abs(dN - g["dN"]).max() < 1e-15
    conn, X = orc.structured_mesh3d_ho(*shape, etype)
    assert np.array_equal(conn, g["conn"]) and np.array_equal(X, g["X0"])
    dof_fixed, force = orc.elasticity_problem3d(*shape)
    assert np.array_equal(dof_fixed, g["dof_fixed"]) and list(force) == list(g["force_nodes"])
    conn = g["conn"].astype(np.int64)
    Ke = orc.elasticity_Ke(g["X"], conn, etype, rho=g["rho"], p=float(g["p"]))
    assert relmax(Ke[0], g["Ke0"]) < K_TOL
    K = orc.assemble_csr(Ke, orc.create_conn_dof(conn, 3))
    Kref = csr_of(g, "K", K.shape[0])
    assert np.array_equal(K.indptr, Kref.indptr) and np.array_equal(K.indices, Kref.indices)
    assert np.abs(K.data - Kref.data).max() < K_TOL * np.abs(Kref.data).max()
    s = orc.elasticity_K_dv_sens(g["X"], conn, etype, g["rho"], g["phi"], g["psi"], p=float(g["p"]))
    assert relmax(s, g["sens"]) < 1e-12


def test_nonlinear_poisson_matches_reference():
    """NonlinearPoisson2D tangent / residual at a random state on a distorted mesh, and the Newton solution of the
    reference's own test case (tests/test_nonlinear_poisson.py) with the direct solver."""
    g = load("nlpoisson32_quad")
    conn = g["conn"].astype(np.int64)
    Ke, _ = orc.nlpoisson_element(g["Xd"], conn, g["xdv2"], g["us"])
    assert relmax(Ke[0], g["Ke0"]) < K_TOL
    K = orc.assemble_csr(Ke, conn)
    Kref = csr_of(g, "K", K.shape[0])
    assert np.array_equal(K.indptr, Kref.indptr) and np.array_equal(K.indices, Kref.indices)
    assert np.abs(K.data - Kref.data).max() < K_TOL * np.abs(Kref.data).max()
    assert relmax(orc.nlpoisson_residual(g["Xd"], conn, g["xdv2"], g["us"]), g["res"]) < K_TOL
    u = orc.nlpoisson_solve(g["X"], conn, g["dof_fixed"], g["xdv"])
    assert relmax(u, g["u"]) < U_TOL
    np.random.seed(123)
    assert abs(np.random.rand(u.shape[0]).dot(u) / float(g["pTu"]) - 1) < 1e-10
