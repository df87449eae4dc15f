"""Table-driven 3-D elements (10-node tetrahedra, 20-node bricks): assembly and AMG-PCG solve on an n^3-node grid.
   python tools/bench_tab3d.py [n ...]      (n odd; default 33 65)   one JSON object per element type and size"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem

for n in [int(a) for a in sys.argv[1:]] or [33, 65]:
    for etype, Qc, Bc in (("tet", pyfem.QuadratureTetrahedron5Point, pyfem.BasisTetrahedron10node),
                          ("brick20", pyfem.QuadratureBrick333Point, pyfem.BasisBrick20Nodes)):
        c = pyfem.ProblemCreator(n, n, n, element_type=etype)
        conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
        q = Qc(); b = Bc(q)
        t0 = time.perf_counter()
        m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=5.0)
        torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
        h = m.hctx
        rho = torch.rand(h.nnodes, dtype=torch.float64, device="cuda") * 0.5 + 0.5
        K = m.compute_jacobian(rho)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            m.compute_jacobian(rho, out=K)
        torch.cuda.synchronize(); t_asm = (time.perf_counter() - t0) / 10
        asm = pyfem.Assembler(m)
        asm.rtol = 1e-8
        asm.solve(method="cg")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        asm.solve(method="cg")
        torch.cuda.synchronize(); t_solve = time.perf_counter() - t0
        print(json.dumps({"element": etype, "grid_nodes": n, "nelems": h.nelems, "ndof": h.nrows, "nnz": h.nnz,
                          "max_row_blocks": h.max_row_blocks, "setup_s": t_setup, "assemble_ms": 1e3 * t_asm,
                          "elements_per_s": h.nelems / t_asm, "solve_rtol1e-8_s": t_solve, "amg_pcg_iters": asm.last_iters,
                          "amg_levels": [a for a, _, _ in h.amg_info()]}), flush=True)
        del m, K, h, asm
        torch.cuda.empty_cache()
