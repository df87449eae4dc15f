"""SpMV shape sweep (build with PF_SPMV_SWEEP=1; PF_SPMV_SHAPE selects a candidate) on the 4M-quad matrices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
nx = 2048
c = pyfem.ProblemCreator(nx + 1, nx + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
for d, model in ((1, pyfem.Helmholtz(0.05, X, conn, q, b)), (2, pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b))):
    h = model.hctx
    vals = model.K.vals if d == 1 else model.compute_jacobian().vals
    x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    os.environ["PF_SPMV_SHAPE"] = "0"; yref = h.spmv(vals, x).clone()
    balg = 12 * h.nnz + 20 * h.nrows
    for shape in range(6):
        os.environ["PF_SPMV_SHAPE"] = str(shape)
        ok = bool((h.spmv(vals, x) == yref).all())
        best = min(h.time_kernel(_lib.PF_K_SPMV, 30, in0=vals, in1=x, out0=y) for _ in range(3))
        print(f"d={d} shape={shape} same={ok} {best*1e3:7.1f} us  {balg/best/1e6:7.0f} GB/s alg", flush=True)
