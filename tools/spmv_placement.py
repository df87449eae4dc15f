import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); K = m.compute_jacobian(); h = m.hctx
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K.vals, in1=x, out0=y)
res = []
keep = []
for i in range(8):
    v = K.vals.clone(); xx = x.clone(); yy = torch.empty_like(x)
    keep.append((v, xx, yy))
    t = h.time_kernel(_lib.PF_K_SPMV, 100, in0=v, in1=xx, out0=yy)
    t0 = h.time_kernel(_lib.PF_K_SPMV, 100, in0=K.vals, in1=x, out0=y)
    res.append({"copy_ms": round(t, 4), "v": hex(v.data_ptr()), "x": hex(xx.data_ptr()), "y": hex(yy.data_ptr())})
    # same values, other vectors: which buffer's placement matters?
    res[-1]["copyvals_origxy_ms"] = round(h.time_kernel(_lib.PF_K_SPMV, 100, in0=v, in1=x, out0=y), 4)
    res[-1]["origvals_copyxy_ms"] = round(h.time_kernel(_lib.PF_K_SPMV, 100, in0=K.vals, in1=xx, out0=yy), 4)
    res[-1]["orig_v_copy_x_orig_y_ms"] = round(h.time_kernel(_lib.PF_K_SPMV, 100, in0=K.vals, in1=xx, out0=y), 4)
    res[-1]["orig_v_orig_x_copy_y_ms"] = round(h.time_kernel(_lib.PF_K_SPMV, 100, in0=K.vals, in1=x, out0=yy), 4)
print(json.dumps({"orig": {"v": hex(K.vals.data_ptr()), "x": hex(x.data_ptr()), "y": hex(y.data_ptr())}}))
for r in res: print(json.dumps(r))
