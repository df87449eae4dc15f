"""Cost of the partitioned AMG-PCG driver on ONE rank against the single-GPU pf_cg with the same hierarchy, on the
2048 x 1024 plane-stress system with a RAMP density field (configuration 5's matrix): what the Python-orchestrated V-cycle
and the whole-mesh-length restriction / prolongation cost before any communication.
   python tools/dist_amg_overhead.py  ->  one JSON line (profiles/r02_dist_amg_overhead.jsonl)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem
from pyfem_gpu_testflight_amd.dist import PartitionedElasticity
nx, ny = 2048, 1024
c = pyfem.ProblemCreator(nx + 1, ny + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
xn = (X - X.min(axis=0)) / (X.max(axis=0) - X.min(axis=0))
rho = np.clip(0.5 + 0.6 * np.sin(11 * xn[:, 0]) * np.cos(7 * xn[:, 1]), 1e-3, 1.0)
model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=5.0)
rho_d = model.hctx.to_device(rho, torch.float64)
def single():
    K = model.compute_jacobian(rho_d)
    rhs = model.hctx.to_device(model.compute_rhs(), torch.float64)
    K, rhs = model.apply_dirichlet_bcs(K, rhs, True)
    asm = pyfem.Assembler(model); asm.rtol = 1e-8
    u = asm._solve_linear_system(K, rhs, "cg")
    return asm.last_iters
pm = PartitionedElasticity(X, conn, dof_fixed, None, force, q, b, p=5.0, rank=0, world=1)
def parted():
    pm.solve(rho=rho, rtol=1e-8, precond="amg")
    return pm.last_iters
out = {"mesh": f"{nx}x{ny} quads, RAMP p=5"}
for name, fn in (("single_gpu_pf_cg_amg", single), ("partitioned_driver_one_rank", parted)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); it = fn(); torch.cuda.synchronize()
    out[name + "_s"] = time.perf_counter() - t0; out[name + "_iters"] = it
print(json.dumps(out), flush=True)
