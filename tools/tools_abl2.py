import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
nx = 2048
c = pyfem.ProblemCreator(nx + 1, nx + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
h = m.hctx
K = m.compute_jacobian()
for nw in ["1", "2", "4"]:
    os.environ["PF_GATHER_NW"] = nw
    for abl in [0, 1, 4]:
        os.environ["PF_ABLATE"] = str(abl)
        ms = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 5, out0=K.vals, E=10.0, nu=0.3, p=0.0)
        print(f"NW={nw} ABL={abl}: {ms*1e3:8.1f} us")
