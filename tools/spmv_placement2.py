#!/usr/bin/env python
"""SpMV d = 2 with y at different OFFSETS inside one allocation (the placement effect of DESIGN 3.2: which granularity
switches between the fast and the slow time?)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); K = m.compute_jacobian(); h = m.hctx
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K.vals, in1=x, out0=y)
pad = 64 << 20
big = torch.empty(h.nrows + pad // 8, dtype=torch.float64, device="cuda")
out = {"orig_ms": round(h.time_kernel(_lib.PF_K_SPMV, 100, in0=K.vals, in1=x, out0=y), 4), "v": hex(K.vals.data_ptr()), "y": hex(y.data_ptr()), "big": hex(big.data_ptr())}
res = {}
for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 4 << 20, 6 << 20, 8 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20, 63 << 20):
    yy = big[off // 8: off // 8 + h.nrows]
    res[off] = round(h.time_kernel(_lib.PF_K_SPMV, 60, in0=K.vals, in1=x, out0=yy), 4)
out["y_offset_bytes_ms"] = res
# and several separate allocations of y
sep = []
keep = []
for i in range(6):
    yy = torch.empty_like(x); keep.append(yy)
    sep.append((hex(yy.data_ptr()), round(h.time_kernel(_lib.PF_K_SPMV, 60, in0=K.vals, in1=x, out0=yy), 4)))
out["separate_y"] = sep
print(json.dumps(out))
