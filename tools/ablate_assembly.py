import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
nx = int(os.environ.get("NX", 2048))
c = pyfem.ProblemCreator(nx + 1, nx + 1)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
h = m.hctx
os.environ["PF_GATHER_IMPL"] = "legacy"
ref = m.compute_jacobian().vals.clone()
os.environ["PF_GATHER_IMPL"] = "v3"
rho = torch.rand(h.nnodes, dtype=torch.float64, device="cuda") * 0.5 + 0.5
K = m.compute_jacobian()
def t(label, in0=None, p=0.0):
    ms = h.time_kernel(_lib.PF_K_ASSEMBLE_ELASTICITY, 20, in0=in0, out0=K.vals, E=10.0, nu=0.3, p=p)
    print(f"  {label}: {ms*1e3:8.1f} us", flush=True)
for order in ["contig", "strided"]:
    os.environ["PF_TILE_ORDER"] = order
    K2 = m.compute_jacobian()
    print(order, "max abs diff vs legacy:", float((K2.vals - ref).abs().max()))
    for wpc in ["12", "16", "24", "32"]:
        os.environ["PF_GATHER_WAVES_PER_CU"] = wpc
        for abl in [0, 1, 4]:
            os.environ["PF_ABLATE"] = str(abl)
            t(f"{order} waves/CU={wpc} ABL={abl}")
        os.environ["PF_ABLATE"] = "0"
        t(f"{order} waves/CU={wpc} with rho", in0=rho, p=3.0)
