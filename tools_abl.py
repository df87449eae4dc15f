import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
nx = 2048
c = pyfem.ProblemCreator(nx + 1, nx + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
h = m.hctx
os.environ["PF_GATHER_IMPL"] = "legacy"
ref = m.compute_jacobian().vals.clone()
os.environ["PF_GATHER_IMPL"] = "v2"
K = m.compute_jacobian()
print("v2 bitwise equal to legacy:", bool((K.vals == ref).all()), float((K.vals - ref).abs().max()))
for wpc in ["0", "4", "5", "6", "8"]:
    os.environ["PF_GATHER_WAVES_PER_CU"] = wpc
    for abl in [0, 1, 4, 5]:
        os.environ["PF_ABLATE"] = str(abl)
        ms = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 20, out0=K.vals, E=10.0, nu=0.3, p=0.0)
        print(f"  v2 waves/CU={wpc} ABL={abl} (noquad={abl&1} nowrite={(abl>>2)&1}): {ms*1e3:8.1f} us")
os.environ["PF_ABLATE"] = "0"; os.environ["PF_GATHER_WAVES_PER_CU"] = "0"
rho = torch.rand(h.nnodes, dtype=torch.float64, device="cuda") * 0.5 + 0.5
ms = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 20, in0=rho, out0=K.vals, E=10.0, nu=0.3, p=3.0)
print(f"  v2 with rho: {ms*1e3:8.1f} us")
