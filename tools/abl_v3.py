"""v3 assembly ablations (PF_ABLATE bits: 1 no quadrature, 4 no write-out, 8 plain LDS stores instead of
atomic adds, 16 no record/coordinate loads).  Ablated kernels give wrong sums on purpose; only the time
is of interest.   python tools/abl_v3.py [abl ...]   env WPC="12 24 32" sweeps persistent waves per CU."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
nx = int(os.environ.get("NX", 2048))
c = pyfem.ProblemCreator(nx + 1, nx + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
h = m.hctx
K = m.compute_jacobian()
abls = [int(a) for a in sys.argv[1:]] or [0, 1, 4, 5, 8, 9, 12, 13, 16, 25, 29]
for wpc in os.environ.get("WPC", "24").split():
    os.environ["PF_GATHER_WAVES_PER_CU"] = wpc
    for order in os.environ.get("ORDER", "s").split():
        os.environ["PF_TILE_ORDER"] = order
        for abl in abls:
            os.environ["PF_ABLATE"] = str(abl)
            h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 5, in0=None, out0=K.vals, E=10.0, nu=0.3, p=0.0)
            ms = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 20, in0=None, out0=K.vals, E=10.0, nu=0.3, p=0.0)
            print(f"wpc={wpc:>3s} order={order} ABL={abl:2d}: {ms*1e3:8.1f} us", flush=True)
