import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 1025); conn, X, x0 = c.create_helmhotz_problem()
m = pyfem.Helmholtz(0.05, X, conn, qq, bq); h = m.hctx
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=m.K.vals, in1=x, out0=y)
out = {"d1_fp64_ms": h.time_kernel(_lib.PF_K_SPMV, 300, in0=m.K.vals, in1=x, out0=y)}
c = pyfem.ProblemCreator(2049, 1025); conn, X, df, f = c.create_linear_elasticity_problem()
me = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); he = me.hctx; K = me.compute_jacobian()
x2 = torch.rand(he.nrows, dtype=torch.float64, device="cuda"); y2 = torch.empty_like(x2)
he.time_kernel(_lib.PF_K_SPMV, 300, in0=K.vals, in1=x2, out0=y2)
out["d2_fp64_ms"] = he.time_kernel(_lib.PF_K_SPMV, 300, in0=K.vals, in1=x2, out0=y2)
print(json.dumps(out))
