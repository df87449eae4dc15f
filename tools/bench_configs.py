#!/usr/bin/env python
"""Secondary configurations of BASELINE.json (parity-test cases, not the headline bench line):
C2 plane stress 1024^2 quads, C3 Helmholtz 2048^2 quads (SpMV/CG roofline at d=1), C4 plane stress on
4M triangles (structured split, node order shuffled in 64-node pages to break the structure), C5 one
topology-optimisation inner iteration on 2048x1024 (1 elasticity assemble+BC+solve, 4 Helmholtz solves).
c4arm: the reference's own unstructured mesh (arm.inp) red-refined five times.
Prints one JSON object per configuration.   python tools/bench_configs.py [c2 c3 c4 c4arm c5]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyfem_gpu_testflight_amd import _lib, pyfem  # noqa: E402

PEAK = 8000.0


def sync():
    torch.cuda.synchronize()


def timed(fn, reps):
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def qb(etype):
    if etype == "quad":
        q = pyfem.QuadratureBilinear2D()
        return q, pyfem.BasisBilinear2D(q)
    q = pyfem.QuadratureTriangle2D()
    return q, pyfem.BasisTriangle2D(q)


def arm_mesh(levels, renumber):
    """examples/torque_arm arm.inp (tests/golden/arm_tri.npz holds its nodes, triangles and groups) after `levels`
    uniform red refinements; BASELINE.json config 4 (5 levels = 2.55 M triangles)."""
    from pyfem_gpu_testflight_amd import meshtools

    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "arm_tri.npz"))
    X, conn = g["X"], g["conn"].astype(np.int64)
    sets = {"fixed": np.unique(g["dof_fixed"] // 2), "load": g["force_nodes"].astype(np.int64)}
    load_val = g["force_vals"][0]
    for _ in range(levels):
        X, conn, sets = meshtools.red_refine_tri(X, conn, sets)
    if renumber:
        X, conn, sets = meshtools.renumber_nodes(X, conn, meshtools.morton_node_order(X), sets)
    dof_fixed = list(np.stack([2 * sets["fixed"], 2 * sets["fixed"] + 1], axis=1).ravel())
    force = {int(n): [float(load_val[0]), float(load_val[1])] for n in sets["load"]}
    return conn, X, dof_fixed, force


def elasticity(name, nx, ny, etype, shuffle=False, p=0.0, cg_iters=200, mesh=None, solve=False):
    if mesh is not None:
        conn, X, dof_fixed, force = mesh
    else:
        c = pyfem.ProblemCreator(nx + 1, ny + 1, element_type=etype)
        conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    if shuffle:  # permute nodes inside 64-node pages and elements inside 4096-element pages
        rng = np.random.RandomState(0)
        nn = X.shape[0]
        perm = np.arange(nn)
        for s in range(0, nn - 63, 64):
            perm[s:s + 64] = s + rng.permutation(64)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(nn)
        X = X[perm]
        conn = inv[conn]
        dof_fixed = list(np.stack([2 * inv[np.asarray(dof_fixed[0::2]) // 2], 2 * inv[np.asarray(dof_fixed[0::2]) // 2] + 1], 1).ravel())
        force = {int(inv[k]): v for k, v in force.items()}
    q, b = qb(etype)
    t0 = time.perf_counter()
    m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=p)
    sync()
    setup = time.perf_counter() - t0
    h = m.hctx
    rho = None
    if p:
        rho = torch.rand(h.nnodes, dtype=torch.float64, device="cuda") * 0.7 + 0.3
    K = m.compute_jacobian(1.0 if rho is None else rho)
    t_asm = timed(lambda: m.compute_jacobian(1.0 if rho is None else rho, out=K), 10)
    P = conn.shape[1]
    b_asm = 4 * P * h.nelems + 16 * h.nnodes + (8 * h.nnodes if p else 0) + 8 * h.nnz
    x = torch.rand(h.nrows, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    ms_spmv = h.time_kernel(_lib.PF_K_SPMV, 30, in0=K.vals, in1=x, out0=y)
    b_spmv = 12 * h.nnz + 20 * h.nrows
    rhs = h.to_device(m.compute_rhs(), torch.float64)
    Kbc, bb = m.apply_dirichlet_bcs(K, rhs, True)
    h.cg(Kbc.vals, bb, rtol=0.0, maxiter=10, allow_maxiter=True)
    sync()
    t0 = time.perf_counter()
    h.cg(Kbc.vals, bb, rtol=0.0, maxiter=cg_iters, check_every=cg_iters, allow_maxiter=True)
    sync()
    t_cg = (time.perf_counter() - t0) / cg_iters
    extra = {}
    if solve:  # full AMG-PCG solve to rtol 1e-8: first call (symbolic + numeric setup) and a second one (numeric only)
        for tag in ("first", "second"):
            sync()
            t0 = time.perf_counter()
            _, it, rr = h.cg(Kbc.vals, bb, rtol=1e-8, precond=_lib.PF_PRECOND_AMG)
            sync()
            extra[f"amg_solve_{tag}_s"] = time.perf_counter() - t0
        extra["amg_iters"] = it
        extra["amg_levels"] = [a for a, _, _ in h.amg_info()]
    return {"config": name, "nelems": h.nelems, "ndof": h.nrows, "nnz": h.nnz, "max_row_blocks": h.max_row_blocks, **extra,
            "setup_s": setup, "assemble_ms": t_asm * 1e3, "elements_per_s": h.nelems / t_asm,
            "assemble_GBps_algorithmic": b_asm / t_asm / 1e9, "assemble_frac": b_asm / t_asm / 1e9 / PEAK,
            "spmv_ms": ms_spmv, "spmv_GBps_algorithmic": b_spmv / ms_spmv / 1e6, "spmv_frac": b_spmv / ms_spmv / 1e6 / PEAK,
            "cg_iter_per_s": 1.0 / t_cg}


def helmholtz(name, nx, ny, r0=0.05):
    c = pyfem.ProblemCreator(nx + 1, ny + 1)
    conn, X, x = c.create_helmhotz_problem()
    q, b = qb("quad")
    t0 = time.perf_counter()
    m = pyfem.Helmholtz(r0, X, conn, q, b)
    sync()
    ctor = time.perf_counter() - t0
    h = m.hctx
    t_asm = timed(lambda: h.assemble_helmholtz(r0), 10)
    b_asm = 16 * h.nelems + 16 * h.nnodes + 2 * 8 * h.nnz
    xd = h.to_device(x, torch.float64)
    y = torch.empty_like(xd)
    ms_spmv = h.time_kernel(_lib.PF_K_SPMV, 50, in0=m.K.vals, in1=xd, out0=y)
    b_spmv = 12 * h.nnz + 20 * h.nrows
    m.reuse_forward = False  # time a solve, not the kept result of the warm-up call
    m.apply(xd)
    sync()
    t0 = time.perf_counter()
    rho = m.apply(xd)
    sync()
    t_apply = time.perf_counter() - t0
    r = m.hctx.spmv(m.R.vals, xd) - m.K.dot(rho)
    relres = float(r.norm() / m.hctx.spmv(m.R.vals, xd).norm())
    return {"config": name, "nelems": h.nelems, "nrows": h.nrows, "nnz": h.nnz, "ctor_s": ctor,
            "assemble_K_and_R_ms": t_asm * 1e3, "assemble_GBps_algorithmic": b_asm / t_asm / 1e9,
            "spmv_ms": ms_spmv, "spmv_GBps_algorithmic": b_spmv / ms_spmv / 1e6, "spmv_frac": b_spmv / ms_spmv / 1e6 / PEAK,
            "apply_s": t_apply, "apply_iters": m.last_iters, "apply_iter_per_s": m.last_iters / t_apply,
            "apply_true_relres": relres}


def topo_iteration(name, nx, ny):
    """examples/topology_optimization/topo.py:50-75 without the ParOpt driver and the sensitivity pass
    (row f2): x -> rho = filter(x); (c, u) = compliance(rho); 3 more filter solves stand in for the
    gradient filtering.  Density is nodal."""
    c = pyfem.ProblemCreator(nx + 1, ny + 1)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q, b = qb("quad")
    filtr = pyfem.Helmholtz(0.05, X, conn, q, b); filtr.reuse_forward = False  # every timed apply() is a solve
    model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=5.0)
    filtr.rtol = 1e-8
    h = model.hctx
    xk = h.to_device(np.random.default_rng(0).uniform(1e-3, 1, X.shape[0]), torch.float64)
    asm = pyfem.Assembler(model)
    asm.rtol = 1e-8
    g = torch.ones_like(xk)

    def iteration():
        sync()
        t0 = time.perf_counter()
        rho = filtr.apply(xk).clamp(1e-3, 1.0)
        sync()
        t1 = time.perf_counter()
        K = model.compute_jacobian(rho)
        rhs = h.to_device(model.compute_rhs(), torch.float64)
        K, rhs = model.apply_dirichlet_bcs(K, rhs, True)
        sync()
        t2 = time.perf_counter()
        u = asm._solve_linear_system(K, rhs, "cg")
        comp = float(rhs.dot(u))
        sync()
        t3 = time.perf_counter()
        for _ in range(3):
            filtr.apply_gradient(g)
        sync()
        t4 = time.perf_counter()
        return t0, t1, t2, t3, t4, comp

    iteration()  # the first pass builds the (graph-only) AMG hierarchies once per mesh and loads every kernel
    t0, t1, t2, t3, t4, comp = iteration()
    return {"config": name, "nelems": h.nelems, "filter_apply_s": t1 - t0, "filter_iters": filtr.last_iters,
            "assemble_bc_s": t2 - t1, "elasticity_solve_s": t3 - t2, "elasticity_cg_iters": asm.last_iters,
            "compliance": comp, "three_filter_gradients_s": t4 - t3, "iteration_total_s": t4 - t0}


if __name__ == "__main__":
    which = [a.lower() for a in sys.argv[1:]] or ["c2", "c3", "c4", "c4arm", "c5"]
    if "c2" in which:
        print(json.dumps(elasticity("C2 plane stress 1024x1024 quads", 1024, 1024, "quad")), flush=True)
    if "c3" in which:
        print(json.dumps(helmholtz("C3 Helmholtz 2048x2048 quads r0=0.05", 2048, 2048)), flush=True)
    if "c4" in which:
        print(json.dumps(elasticity("C4 plane stress 4.0M triangles (1415x1415 split quads), structured order", 1415, 1415, "tri")), flush=True)
        print(json.dumps(elasticity("C4 plane stress 4.0M triangles, node order shuffled in 64-node pages", 1415, 1415, "tri", shuffle=True)), flush=True)
        print(json.dumps(elasticity("plane stress 2048x2048 quads with nodal density (RAMP p=5)", 2048, 2048, "quad", p=5.0)), flush=True)
    if "c4arm" in which:
        for renumber in (False, True):
            tag = "Morton node order" if renumber else "refinement order (nodes appended level by level)"
            print(json.dumps(elasticity(f"C4 arm.inp after 5 red refinements (2.55 M triangles, all detJ < 0), {tag}",
                                        0, 0, "tri", mesh=arm_mesh(5, renumber), solve=True)), flush=True)
    if "c5" in which:
        print(json.dumps(topo_iteration("C5 topology-optimisation inner iteration 2048x1024, AMG-PCG rtol 1e-8 (second iteration: hierarchy cached)", 2048, 1024)), flush=True)
