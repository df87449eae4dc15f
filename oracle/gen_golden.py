"""
Golden-vector generator  --  TEST INFRASTRUCTURE.

Imports the *reference* (aaronyicongfu/pyfem_gpu_testflight, expected read-only at
/root/reference) exactly as its own tests do and records inputs/expected outputs of
the hot path as small .npz fixtures under tests/golden/.  Runs only in the build
container; the GPU box never sees the reference -- tests read the committed .npz.

pyamg is not installed here.  `import pyfem` needs the module name to exist, and
Helmholtz.__init__ calls pyamg.ruge_stuben_solver(K): a functional stand-in backed by
a sparse LU (what the reference's own oracle tests/ref_helmholtz.py:106 uses) is
injected.  No reference source is copied; only numeric arrays are stored.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("PYFEM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference():
    from scipy.sparse.linalg import factorized

    class _LU:
        def __init__(self, K):
            self._lu = factorized(K.tocsc())

        def solve(self, b, tol=None):
            return self._lu(b)

    stub = types.ModuleType("pyamg")
    stub.ruge_stuben_solver = _LU
    sys.modules["pyamg"] = stub
    import matplotlib

    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "tests"))
    os.chdir("/tmp")  # the reference's profiler appends ./profiler.log
    import pyfem
    import parse_inp

    return pyfem, parse_inp


def _csr(K, prefix):
    K = K.tocsr()
    K.sort_indices()
    return {
        f"{prefix}_indptr": K.indptr.astype(np.int32),
        f"{prefix}_indices": K.indices.astype(np.int32),
        f"{prefix}_data": K.data.copy(),
    }


def _qb(pyfem, etype):
    if etype == "quad":
        q = pyfem.QuadratureBilinear2D()
        return q, pyfem.BasisBilinear2D(q)
    if etype == "block":
        q = pyfem.QuadratureBlock3D()
        return q, pyfem.BasisBlock3D(q)
    q = pyfem.QuadratureTriangle2D()
    return q, pyfem.BasisTriangle2D(q)


def elasticity_case(pyfem, name, X, conn, dof_fixed, nodal_force, etype, rho=None, p=0.0, full_K=True,
                    fixed_vals=None):
    q, b = _qb(pyfem, etype)
    model = pyfem.LinearElasticity(X, conn, dof_fixed, fixed_vals, nodal_force, q, b, p=p)
    K = model.compute_jacobian() if rho is None else model.compute_jacobian(rho)
    out = dict(
        X=np.asarray(X, dtype=float),
        conn=np.asarray(conn, dtype=np.int32),
        dof_fixed=np.asarray(dof_fixed, dtype=np.int32),
        force_nodes=np.array(list(nodal_force.keys()), dtype=np.int32),
        force_vals=np.array(list(nodal_force.values()), dtype=float),
        p=np.float64(p),
        Ke0=model.Ke_mat[0].copy(),
        Ke_last=model.Ke_mat[-1].copy(),
        K_norm=np.float64(np.linalg.norm(K.data)),
        K_diag=K.diagonal(),
        K_nnz=np.int64(K.nnz),
    )
    if rho is not None:
        out["rho"] = rho
    if fixed_vals is not None:
        out["fixed_vals"] = np.asarray(fixed_vals, dtype=float)
    if full_K:
        out.update(_csr(K, "K"))
    rng = np.random.RandomState(7)
    v = rng.rand(K.shape[0])
    out["v"] = v
    out["Kv"] = K.dot(v)
    rhs = model.compute_rhs().copy()
    out["rhs"] = rhs.copy()
    Kbc, rhs_bc = model.apply_dirichlet_bcs(K.copy(), rhs.copy(), enforce_symmetric_K=True)
    out["rhs_bc"] = rhs_bc.copy()
    out["Kbc_nnz"] = np.int64(Kbc.nnz)
    out["Kbc_v"] = Kbc.dot(v)
    Kns, rhs_ns = model.apply_dirichlet_bcs(K.copy(), rhs.copy(), enforce_symmetric_K=False)
    out["Kns_v"] = Kns.dot(v)
    out["rhs_ns"] = rhs_ns.copy()
    from scipy.sparse.linalg import spsolve

    out["u"] = spsolve(Kbc.tocsc(), rhs_bc)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: nelems={conn.shape[0]} ndof={K.shape[0]} nnz={K.nnz} |K|={out['K_norm']:.13e}")
    return out


def main_3d(pyfem):
    """3-D 8-node block element (row f4): elasticity, Helmholtz, LinearPoisson K, mesh generator."""
    from scipy.sparse.linalg import spsolve

    c = pyfem.ProblemCreator(6, 4, 3, element_type="block")
    conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
    elasticity_case(pyfem, "ps3d_6x4x3_block", X, conn, dof_fixed, nodal_force, "block")
    c = pyfem.ProblemCreator(5, 4, 4, element_type="block")
    conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
    rng = np.random.RandomState(11)
    Xd = X + 0.05 * (rng.rand(*X.shape) - 0.5)
    rho = 0.1 + 0.9 * rng.rand(X.shape[0])
    o = elasticity_case(pyfem, "ps3d_5x4x4_block_ramp_distorted", Xd, conn, dof_fixed, nodal_force, "block", rho=rho, p=5.0)
    # sensitivities of the same case (tests/test_elasticity.py:68-104 pattern)
    q, b = _qb(pyfem, "block")
    model = pyfem.LinearElasticity(Xd, conn, dof_fixed, None, nodal_force, q, b, p=5.0)
    np.random.seed(0)
    phi, psi = np.random.rand(3 * Xd.shape[0]), np.random.rand(3 * Xd.shape[0])
    sens = model._compute_K_dv_sens(rho, phi, psi).copy()
    np.savez_compressed(os.path.join(OUT, "sens3d_5x4x4_block.npz"), X=Xd, conn=conn.astype(np.int32), rho=rho, phi=phi,
                        psi=psi, sens=sens, p=np.float64(5.0), u=o["u"],
                        compliance_grad=model.compliance_grad(rho, o["u"]).copy())
    print(f"sens3d_5x4x4_block: |sens|={np.linalg.norm(sens):.13e}")

    # Helmholtz filter on blocks (Helmholtz is dimension-generic, pyfem.py:2085-2177)
    c = pyfem.ProblemCreator(5, 4, 3, element_type="block")
    conn, X, x = c.create_helmhotz_problem()
    m = pyfem.Helmholtz(0.1, X, conn, q, b)
    g = np.random.RandomState(5).rand(X.shape[0])
    out = dict(X=X, conn=conn.astype(np.int32), x=x, r0=np.float64(0.1), rho=m.apply(x), g=g, gx=m.apply_gradient(g),
               Ke0=m.Ke_mat[0].copy() if hasattr(m, "Ke_mat") else np.zeros(1))
    out.update(_csr(m.K, "K"))
    out.update(_csr(m.R, "R"))
    np.savez_compressed(os.path.join(OUT, "helm3d_5x4x3_block.npz"), **out)
    print(f"helm3d_5x4x3_block: nnz={m.K.nnz}")

    # LinearPoisson on blocks (examples/heat_conduction.py shape): K, rhs (with the reference's quirks), solve
    def gfunc(x):
        return x[..., 0] * (x[..., 1] - 2.0) + x[..., 2]

    c = pyfem.ProblemCreator(5, 4, 3, element_type="block")
    conn, X, dof_fixed = c.create_poisson_problem()
    rng = np.random.RandomState(12)
    X = X + 0.04 * (rng.rand(*X.shape) - 0.5)
    rho = 0.1 + 0.9 * rng.rand(X.shape[0])
    vals = 0.1 * rng.rand(len(dof_fixed))
    model = pyfem.LinearPoisson(X, conn, dof_fixed, vals, q, b, gfunc, p=5.0)
    K = model.compute_jacobian(rho)
    rhs = model.compute_rhs().copy()
    out = dict(X=X, conn=conn.astype(np.int32), dof_fixed=np.asarray(dof_fixed, dtype=np.int32), rho=rho, p=np.float64(5.0),
               Ke0=model.Ke_mat[0].copy(), rhs=rhs.copy(), fixed_vals=vals)
    out.update(_csr(K, "K"))
    Kbc, rhs_bc = model.apply_dirichlet_bcs(K.copy(), rhs.copy(), enforce_symmetric_K=True)
    out["rhs_bc"] = rhs_bc.copy()
    out["u"] = spsolve(Kbc.tocsc(), rhs_bc)
    np.random.seed(0)
    phi, psi = np.random.rand(X.shape[0]), np.random.rand(X.shape[0])
    out["phi"], out["psi"] = phi, psi
    out["sens"] = model._compute_K_dv_sens(rho, phi, psi).copy()
    np.savez_compressed(os.path.join(OUT, "poi3d_5x4x3_block.npz"), **out)
    print(f"poi3d_5x4x3_block: nnz={K.nnz}")

    # mesh generator pins
    c = pyfem.ProblemCreator(4, 3, 3, element_type="block")
    conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
    _, _, x = c.create_helmhotz_problem()
    _, _, pfix = c.create_poisson_problem()
    np.savez_compressed(os.path.join(OUT, "mesh3d_4x3x3.npz"), conn=conn.astype(np.int32), X=X,
                        dof_fixed=np.asarray(dof_fixed, dtype=np.int32),
                        force_nodes=np.array(list(nodal_force.keys()), dtype=np.int32),
                        force_vals=np.array(list(nodal_force.values()), dtype=float), helm_x=x,
                        poisson_fixed=np.asarray(pfix, dtype=np.int32))
    print("mesh3d_4x3x3 written")


def main_highorder(pyfem):
    """10-node tetrahedra and 20-node bricks (pyfem.py:115-172, 380-632, 2538-2700; tests/test_elements.py): mesh
    generator, K with nodal RAMP density on a distorted mesh, first element matrix, sensitivities.  The generator
    leaves grid nodes that no element uses (K is singular on them), so no solution is stored."""
    for etype, shape, Qc, Bc in (("tet", (5, 5, 3), "QuadratureTetrahedron5Point", "BasisTetrahedron10node"),
                                 ("brick20", (5, 5, 5), "QuadratureBrick333Point", "BasisBrick20Nodes")):
        c = pyfem.ProblemCreator(*shape, element_type=etype)
        conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
        rng = np.random.RandomState(21)
        Xd = X + 0.03 * (rng.rand(*X.shape) - 0.5)
        rho = 0.1 + 0.9 * rng.rand(X.shape[0])
        q = getattr(pyfem, Qc)()
        b = getattr(pyfem, Bc)(q)
        model = pyfem.LinearElasticity(Xd, conn, dof_fixed, None, nodal_force, q, b, p=5.0)
        K = model.compute_jacobian(rho)
        np.random.seed(0)
        phi, psi = np.random.rand(3 * Xd.shape[0]), np.random.rand(3 * Xd.shape[0])
        out = dict(X0=X, X=Xd, conn=conn.astype(np.int32), dof_fixed=np.asarray(dof_fixed, dtype=np.int32),
                   force_nodes=np.array(list(nodal_force.keys()), dtype=np.int32),
                   force_vals=np.array(list(nodal_force.values()), dtype=float), rho=rho, p=np.float64(5.0),
                   Ke0=model.Ke_mat[0].copy(), phi=phi, psi=psi, sens=model._compute_K_dv_sens(rho, phi, psi).copy(),
                   qpts=np.asarray(q.get_pt()), qw=np.asarray(q.get_weight()),
                   N=np.asarray(b.eval_shape_fun()), dN=np.asarray(b.eval_shape_fun_deriv()))
        out.update(_csr(K, "K"))
        name = f"ho_{etype}_{shape[0]}x{shape[1]}x{shape[2]}"
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(f"{name}: nelems={conn.shape[0]} nnz={K.nnz} |K|={np.linalg.norm(K.data):.13e}")


def main_nonlinear(pyfem):
    """NonlinearPoisson2D (pyfem.py:1332-1665) and Assembler.solve_nonlinear (pyfem.py:2319-2355) on the reference's own
    test case (tests/test_nonlinear_poisson.py: 32x32 quads, xdv = ones(10)/10): tangent and residual at a random state
    on a distorted mesh, and the Newton solution with the direct solver."""
    c = pyfem.ProblemCreator(nnodes_x=32, nnodes_y=32)
    conn, X, dof_fixed = c.create_poisson_problem()
    q = pyfem.QuadratureBilinear2D()
    b = pyfem.BasisBilinear2D(q)
    xdv = np.ones(10) / 10
    model = pyfem.NonlinearPoisson2D(X, conn, dof_fixed, None, q, b)
    u = pyfem.Assembler(model).solve_nonlinear(method="direct", xdv=xdv).copy()
    np.random.seed(123)
    pr = np.random.rand(u.shape[0])
    out = dict(X=X, conn=conn.astype(np.int32), dof_fixed=np.asarray(dof_fixed, dtype=np.int32), xdv=xdv, u=u,
               pTu=np.float64(pr.dot(u)))
    # tangent / residual at a random state on a distorted mesh with a non-uniform design vector
    rng = np.random.RandomState(7)
    Xd = X + 0.2 / 31 * (rng.rand(*X.shape) - 0.5)
    xdv2 = rng.rand(7)
    us = 0.3 * rng.rand(X.shape[0])
    m2 = pyfem.NonlinearPoisson2D(Xd, conn, dof_fixed, None, q, b)
    K = m2.compute_jacobian(xdv2, us)
    out.update(Xd=Xd, xdv2=xdv2, us=us, res=m2.compute_rhs(xdv2, us).copy(), Ke0=m2.Ke_mat[0].copy())
    out.update(_csr(K, "K"))
    np.savez_compressed(os.path.join(OUT, "nlpoisson32_quad.npz"), **out)
    print(f"nlpoisson32_quad: p.u={out['pTu']!r} |K|={np.linalg.norm(K.data):.13e} |res|={np.linalg.norm(out['res']):.13e}")


def main():
    pyfem, parse_inp = _import_reference()
    if "--only-nl" in sys.argv:
        return main_nonlinear(pyfem)
    os.makedirs(OUT, exist_ok=True)
    if "--only-ho" in sys.argv:
        return main_highorder(pyfem)
    if "--only-3d" in sys.argv:
        return main_3d(pyfem)
    main_3d(pyfem)
    main_highorder(pyfem)
    main_nonlinear(pyfem)

    # --- PS32: the reference's own test case (tests/test_elasticity.py:23-51) ------------
    c = pyfem.ProblemCreator(nnodes_x=32, nnodes_y=32)
    conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
    o = elasticity_case(pyfem, "ps32_quad", X, conn, dof_fixed, nodal_force, "quad", full_K=False)
    np.random.seed(123)
    pr = np.random.rand(o["u"].shape[0])
    print("  PS32 p.u =", repr(pr.dot(o["u"])), "(survey: -164.42713355728523)")
    # and the independent hand-written oracle the reference's test compares against
    from ref_plane_stress import PlaneStress

    bcs = {int(d) // 2: [0, 1] for d in dof_fixed}
    u_ref = PlaneStress(conn, X, bcs, nodal_force).solve()
    print("  max|u - u_ref_plane_stress| =", np.abs(o["u"] - u_ref).max())

    # --- PS32 RAMP p=5, random nodal rho (tests/test_elasticity.py:82-88) ----------------
    np.random.seed(0)
    rho = np.random.rand(X.shape[0])
    elasticity_case(pyfem, "ps32_quad_ramp", X, conn, dof_fixed, nodal_force, "quad", rho=rho, p=5.0, full_K=False)

    # --- small structured quad, full K, rectangular domain ------------------------------
    c = pyfem.ProblemCreator(nnodes_x=13, nnodes_y=9)
    conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
    elasticity_case(pyfem, "ps13x9_quad", X, conn, dof_fixed, nodal_force, "quad")

    # --- distorted quads + RAMP + non-zero prescribed displacements ---------------------
    rng = np.random.RandomState(42)
    Xd = X + 0.02 * (rng.rand(*X.shape) - 0.5)
    rho = 0.2 + 0.8 * rng.rand(X.shape[0])
    vals = 0.01 * (rng.rand(len(dof_fixed)) - 0.5)
    elasticity_case(pyfem, "ps13x9_quad_distorted", Xd, conn, dof_fixed, nodal_force, "quad", rho=rho, p=3.0,
                    fixed_vals=vals)

    # --- structured tri -----------------------------------------------------------------
    c = pyfem.ProblemCreator(nnodes_x=11, nnodes_y=8, element_type="tri")
    conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
    elasticity_case(pyfem, "ps11x8_tri", X, conn, dof_fixed, nodal_force, "tri")
    rng = np.random.RandomState(5)
    rho = 0.1 + 0.9 * rng.rand(X.shape[0])
    elasticity_case(pyfem, "ps11x8_tri_ramp", X, conn, dof_fixed, nodal_force, "tri", rho=rho, p=5.0)

    # --- torque arm: unstructured CPS3 mesh (examples/torque_arm/arm.py) ----------------
    parser = parse_inp.InpParser(os.path.join(REF, "examples", "torque_arm", "arm.inp"))
    conn_d, X3, groups = parser.parse()
    X = X3[:, 0:2]
    conn = conn_d["CPS3"]
    dof_fixed = np.concatenate((2 * groups["fixed"], 2 * groups["fixed"] + 1))
    nodal_force = {int(n): [0.0, -1.0] for n in groups["load"]}
    o = elasticity_case(pyfem, "arm_tri", X, conn, dof_fixed, nodal_force, "tri")
    np.random.seed(123)
    pr = np.random.rand(o["u"].shape[0])
    print("  ARM p.u =", repr(pr.dot(o["u"])), "(survey: 21177.45599189116)")

    # --- Helmholtz filter (tests/test_helmholtz.py:11-44) --------------------------------
    for name, nx, ny, etype, r0 in [("helm32_quad", 32, 32, "quad", 0.1), ("helm12x9_tri", 12, 9, "tri", 0.07)]:
        c = pyfem.ProblemCreator(nnodes_x=nx, nnodes_y=ny, element_type=etype)
        conn, X, x = c.create_helmhotz_problem()
        q, b = _qb(pyfem, etype)
        model = pyfem.Helmholtz(r0, X, conn, q, b)
        np.random.seed(0)
        g = np.random.rand(x.shape[0])
        out = dict(
            X=X, conn=conn.astype(np.int32), x=x, g=g, r0=np.float64(r0),
            Ke0=model.Ke_mat[0].copy(), Re0=model.Re[0].copy(),
            rho=model.apply(x).copy(), gradx=model.apply_gradient(g).copy(),
            Rx=model.R.dot(x),
        )
        out.update(_csr(model.K, "K"))
        out.update(_csr(model.R, "R"))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        np.random.seed(123)
        pr = np.random.rand(x.shape[0])
        print(f"{name}: nnz={model.K.nnz}  p.rho = {pr.dot(out['rho'])!r}")

    # --- sensitivity pass (tests/test_elasticity.py:68-104: phi^T dK/drho psi, p = 5) --------------
    for name, nx, ny, etype in [("sens24x17_quad", 24, 17, "quad"), ("sens21x16_tri", 21, 16, "tri")]:
        c = pyfem.ProblemCreator(nnodes_x=nx, nnodes_y=ny, element_type=etype)
        conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
        rng = np.random.RandomState(3)
        X = X + (0.01 * (rng.rand(*X.shape) - 0.5) if etype == "quad" else 0.0)
        q, b = _qb(pyfem, etype)
        model = pyfem.LinearElasticity(X, conn, dof_fixed, None, nodal_force, q, b, p=5.0)
        np.random.seed(0)
        nnodes = X.shape[0]
        phi = np.random.rand(2 * nnodes)
        psi = np.random.rand(2 * nnodes)
        rho = np.random.rand(nnodes)
        sens = model._compute_K_dv_sens(rho, phi, psi).copy()
        K = model.compute_jacobian(rho)
        rhs = model.compute_rhs().copy()
        Kbc, rhs_bc = model.apply_dirichlet_bcs(K, rhs, enforce_symmetric_K=True)
        from scipy.sparse.linalg import spsolve
        u = spsolve(Kbc.tocsc(), rhs_bc)
        grad = model.compliance_grad(rho, u).copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, conn=conn.astype(np.int32),
                            dof_fixed=np.asarray(dof_fixed, dtype=np.int32),
                            force_nodes=np.array(list(nodal_force.keys()), dtype=np.int32),
                            force_vals=np.array(list(nodal_force.values()), dtype=float),
                            rho=rho, phi=phi, psi=psi, sens=sens, u=u, compliance=np.float64(rhs_bc.dot(u)),
                            compliance_grad=grad, p=np.float64(5.0))
        print(f"{name}: |sens|={np.linalg.norm(sens):.13e} c={rhs_bc.dot(u):.13e}")

    # --- LinearPoisson (tests/test_linear_poisson.py:11-40 and :57-113) -----------------------------
    def gfunc(x):
        _x = x[..., 0]
        _y = x[..., 1]
        return _x * (_x - 5.0) * (_x - 10.0) * _y * (_y - 4.0)

    from scipy.sparse.linalg import spsolve
    for name, nx, ny, distort, p in [("poi32_quad", 32, 32, 0.0, 0.0), ("poi17x12_quad_ramp", 17, 12, 0.02, 5.0)]:
        c = pyfem.ProblemCreator(nnodes_x=nx, nnodes_y=ny)
        conn, X, dof_fixed = c.create_poisson_problem()
        rng = np.random.RandomState(8)
        X = X + distort * (rng.rand(*X.shape) - 0.5)
        q, b = _qb(pyfem, "quad")
        vals = None if p == 0.0 else 0.1 * rng.rand(len(dof_fixed))
        model = pyfem.LinearPoisson(X, conn, dof_fixed, vals, q, b, gfunc, p=p)
        rho = np.ones(X.shape[0]) if p == 0.0 else 0.1 + 0.9 * rng.rand(X.shape[0])
        K = model.compute_jacobian(rho)
        rhs = model.compute_rhs().copy()
        out = dict(X=X, conn=conn.astype(np.int32), dof_fixed=np.asarray(dof_fixed, dtype=np.int32), rho=rho,
                   p=np.float64(p), Ke0=model.Ke_mat[0].copy(), rhs=rhs.copy())
        if vals is not None:
            out["fixed_vals"] = vals
        out.update(_csr(K, "K"))
        Kbc, rhs_bc = model.apply_dirichlet_bcs(K.copy(), rhs.copy(), enforce_symmetric_K=True)
        out["rhs_bc"] = rhs_bc.copy()
        u = spsolve(Kbc.tocsc(), rhs_bc)
        out["u"] = u
        np.random.seed(0)
        phi, psi = np.random.rand(X.shape[0]), np.random.rand(X.shape[0])
        out["phi"], out["psi"] = phi, psi
        out["sens"] = model._compute_K_dv_sens(rho, phi, psi).copy()
        out["compliance"] = np.float64(rhs_bc.dot(u))
        out["compliance_grad"] = model.compliance_grad(rho, u).copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        np.random.seed(123)
        pr = np.random.rand(u.shape[0])
        print(f"{name}: nnz={K.nnz} p.u = {pr.dot(u)!r}" + (" (survey: -7082.301628403809)" if name == "poi32_quad" else ""))

    # --- mesh generator pins (ProblemCreator, pyfem.py:2469-2520, 2728-2773) -------------
    out = {}
    for etype in ("quad", "tri"):
        c = pyfem.ProblemCreator(nnodes_x=6, nnodes_y=4, element_type=etype)
        conn, X, dof_fixed, nodal_force = c.create_linear_elasticity_problem()
        _, _, x = c.create_helmhotz_problem()
        _, _, pfix = c.create_poisson_problem()
        out[f"{etype}_conn"] = conn.astype(np.int32)
        out[f"{etype}_X"] = X
        out[f"{etype}_dof_fixed"] = np.asarray(dof_fixed, dtype=np.int32)
        out[f"{etype}_force_nodes"] = np.array(list(nodal_force.keys()), dtype=np.int32)
        out[f"{etype}_force_vals"] = np.array(list(nodal_force.values()), dtype=float)
        out[f"{etype}_helm_x"] = x
        out[f"{etype}_poisson_fixed"] = np.asarray(pfix, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "mesh6x4.npz"), **out)
    print("mesh6x4 written")


if __name__ == "__main__":
    main()
