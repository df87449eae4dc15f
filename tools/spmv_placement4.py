#!/usr/bin/env python
"""Is the two-speed SpMV a property of the BUFFER (placement), of who wrote it last (non-temporal assembly stores against
plain copies), or of time (clock / thermal state)?  Alternating measurements on K and a clone, with assembly bursts into
one or the other in between."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); h = m.hctx
K = m.compute_jacobian().vals
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
K2 = K.clone(); K3 = K.clone()
sp = lambda v: round(h.time_kernel(_lib.PF_K_SPMV, 60, in0=v, in1=x, out0=y), 4)
def asm(v, n=300):
    for _ in range(n): h.assemble_elasticity(10.0, 0.3, 0.0, None, out=v)
    torch.cuda.synchronize()
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K, in1=x, out0=y)
log = []
log.append(("start K K2 K3 K K2 K3", [sp(K), sp(K2), sp(K3), sp(K), sp(K2), sp(K3)]))
asm(K2); log.append(("after 300 assemblies into K2", [sp(K), sp(K2), sp(K3), sp(K), sp(K2), sp(K3)]))
K3.copy_(K); torch.cuda.synchronize(); log.append(("after plain copy into K3", [sp(K), sp(K2), sp(K3)]))
asm(K3, 1); log.append(("after ONE assembly into K3", [sp(K), sp(K2), sp(K3)]))
K2.copy_(K); torch.cuda.synchronize(); log.append(("after plain copy into K2", [sp(K), sp(K2), sp(K3)]))
K4 = K2.clone(); log.append(("fresh clone K4 of K2", [sp(K4), sp(K2)]))
for name, v in log: print(json.dumps({name: v}))
