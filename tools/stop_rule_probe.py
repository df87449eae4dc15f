"""What stopping rule does the u <= 1e-8 bar need?  Configuration 5's system (2048 x 1024, RAMP density field, p = 5) with a
MANUFACTURED solution: after k AMG-PCG iterations (k = 6 .. 22, each a fresh deterministic run with maxiter = k) the true
error max|u_k - u*| / max|u*|, the residual ||b - K u_k|| / ||b|| the product's rule tests, and the energy estimate
sqrt(r.M^-1 r / b.M^-1 b) the recurrence carries anyway.   python tools/stop_rule_probe.py  ->  table"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from oracle import pyfem_oracle as orc   # the manufactured right-hand side only
from pyfem_gpu_testflight_amd import pyfem, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_scale import _oracle_matvec, _smooth_field
case = sys.argv[1] if len(sys.argv) > 1 else "c5"
if case == "c5":
    conn, X = orc.structured_mesh(2049, 1025, "quad"); dof_fixed, _ = orc.elasticity_problem(2049, 1025)
    xn = (X - X.min(axis=0)) / (X.max(axis=0) - X.min(axis=0))
    rho = np.clip(0.5 + 0.6 * np.sin(11 * xn[:, 0]) * np.cos(7 * xn[:, 1]), 1e-3, 1.0); p = 5.0
else:
    conn, X = orc.structured_mesh(1025, 1025, "quad"); dof_fixed, _ = orc.elasticity_problem(1025, 1025); rho = None; p = 0.0
dof_fixed = np.asarray(dof_fixed, dtype=np.int64)
ustar = _smooth_field(X, 5)
b = _oracle_matvec(orc, X, conn, "quad", ustar, rho=rho, p=p)
q = pyfem.QuadratureBilinear2D(); bs = pyfem.BasisBilinear2D(q)
model = pyfem.LinearElasticity(X, conn, dof_fixed, ustar[dof_fixed], {0: [0.0, 0.0]}, q, bs, p=p)
h = model.hctx
K = model.compute_jacobian(1.0 if rho is None else rho)
K, rhs = model.apply_dirichlet_bcs(K, b.copy(), True)
bd = h.to_device(rhs, torch.float64); us = h.to_device(ustar, torch.float64)
def vcycle(r):
    z = torch.empty_like(r); h._sync_stream(); h._call("pf_amg_vcycle", C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr())); return z
h.cg(K.vals, bd, rtol=1e-3, precond=_lib.PF_PRECOND_AMG)  # sets the hierarchy up
bMb = float(bd.dot(vcycle(bd)))
print(f"{'k':>3} {'max err / max u*':>18} {'||r||/||b||':>14} {'sqrt(r.z / b.M^-1 b)':>22}")
for k in range(6, 24):
    x, it, rr = h.cg(K.vals, bd, rtol=0.0, maxiter=k, precond=_lib.PF_PRECOND_AMG_REUSE, allow_maxiter=True)
    r = bd - h.spmv(K.vals, x)
    print(f"{k:3d} {float((x - us).abs().max() / us.abs().max()):18.3e} {float(r.norm() / bd.norm()):14.3e} {abs(float(r.dot(vcycle(r))) / bMb) ** 0.5:22.3e}", flush=True)
