"""3-D block-element throughput (SciTech2023/performance_test.py shape: n^3 elements, plane x=0 clamped).
   python tools/bench3d.py [n ...]      prints one JSON object per size"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib

for n in [int(a) for a in sys.argv[1:]] or [64, 96, 128]:
    c = pyfem.ProblemCreator(n + 1, n + 1, n + 1, element_type="block")
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q = pyfem.QuadratureBlock3D(); b = pyfem.BasisBlock3D(q)
    t0 = time.perf_counter()
    m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=3.0)
    torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
    h = m.hctx
    K = m.compute_jacobian()
    h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 30, in0=None, out0=K.vals, E=10.0, nu=0.3, p=0.0)  # steady clocks first
    ms_asm = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 20, in0=None, out0=K.vals, E=10.0, nu=0.3, p=0.0)
    rho = torch.rand(h.nnodes, dtype=torch.float64, device="cuda") * 0.5 + 0.5
    ms_asm_rho = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 20, in0=rho, out0=K.vals, E=10.0, nu=0.3, p=3.0)
    x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    h.time_kernel(_lib.PF_K_SPMV, 50, in0=K.vals, in1=x, out0=y)
    ms_spmv = h.time_kernel(_lib.PF_K_SPMV, 50, in0=K.vals, in1=x, out0=y)
    K = m.compute_jacobian()
    rhs = h.to_device(m.compute_rhs(), torch.float64)
    K, rhs = m.apply_dirichlet_bcs(K, rhs, True)
    h.cg(K.vals, rhs, rtol=0.0, maxiter=10, allow_maxiter=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, it, _ = h.cg(K.vals, rhs, rtol=0.0, maxiter=200, check_every=200, allow_maxiter=True)
    torch.cuda.synchronize(); t_cg = time.perf_counter() - t0
    solve = {}
    for name, pc in (("amg", _lib.PF_PRECOND_AMG), ("jacobi", _lib.PF_PRECOND_JACOBI)):
        if name == "jacobi" and n > 96:
            continue
        h.cg(K.vals, rhs, rtol=1e-1, precond=pc)  # hierarchy / kernels warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, its, rr = h.cg(K.vals, rhs, rtol=1e-8, precond=pc)
        torch.cuda.synchronize()
        solve[name] = {"iters": its, "s": time.perf_counter() - t0}
    if "amg" in solve:
        solve["amg_levels"] = h.amg_info()
    b_asm = 4 * 8 * h.nelems + 24 * h.nnodes + 8 * h.nnz
    b_spmv = 8 * h.nnz + 4 * h.nblocks + 52 * h.nnodes
    print(json.dumps({"n": n, "nelems": h.nelems, "ndof": h.nrows, "nnz": h.nnz, "setup_s": t_setup,
                      "assemble_ms": ms_asm, "assemble_Gelem_s": h.nelems / ms_asm / 1e6,
                      "assemble_GBps_algorithmic": b_asm / ms_asm / 1e6, "assemble_rho_ms": ms_asm_rho,
                      "fp64_TFLOPs_useful": h.nelems * 8 * 2 * 2000 / ms_asm / 1e9,
                      "spmv_ms": ms_spmv, "spmv_GBps_algorithmic": b_spmv / ms_spmv / 1e6,
                      "cg_iter_per_s": it / t_cg, "solve_rtol1e-8": solve}), flush=True)
    del m, K, h
    torch.cuda.empty_cache()
