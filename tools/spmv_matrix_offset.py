#!/usr/bin/env python
"""Follow-up to tools/spmv_matrix_placement_pmc.py: is the slow / fast state of a matrix value buffer a matter of its
ALIGNMENT (offset inside one allocation) or of the ALLOCATION (physical placement)?  One allocation B of nnz + 4 Mi doubles;
the values are copied to offsets 0, 4 KiB, 64 KiB, 1 MiB, 2 MiB, 16 MiB inside it and the product is timed on each; then
the same for fresh allocations of different sizes (exact, +2 MiB, +64 MiB, 2x)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); h = m.hctx
K = m.compute_jacobian().vals
n = K.numel()
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
tm = lambda v: round(h.time_kernel(_lib.PF_K_SPMV, 40, in0=v, in1=x, out0=y), 4)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K, in1=x, out0=y)
out = {"K_itself": tm(K)}
B = torch.empty(n + (4 << 20), dtype=torch.float64, device="cuda")
offs = {"0": 0, "4KiB": 512, "64KiB": 8192, "1MiB": 1 << 17, "2MiB": 1 << 18, "16MiB": 1 << 21}
res = {}
for name, o in offs.items():
    v = B[o:o + n]; v.copy_(K); res[name] = tm(v)
out["offsets_inside_one_allocation"] = res
res = {}
for name, extra in (("exact", 0), ("+2MiB", 1 << 18), ("+64MiB", 1 << 23), ("2x", n)):
    Bn = torch.empty(n + extra, dtype=torch.float64, device="cuda"); v = Bn[:n]; v.copy_(K); res[name] = tm(v)
    res[name + "_tail"] = None
    if extra >= n:
        v2 = Bn[n:2 * n]; v2.copy_(K); res[name + "_tail"] = tm(v2)
out["fresh_allocations"] = res
out["K_again"] = tm(K)
print(json.dumps(out))
