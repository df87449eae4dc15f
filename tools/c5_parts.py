import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 1025)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
filtr = pyfem.Helmholtz(0.05, X, conn, qq, bq); filtr.rtol = 1e-8; filtr.reuse_forward = False  # every timed apply() is a solve
model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, qq, bq, p=5.0)
h = model.hctx
xk = h.to_device(np.random.default_rng(0).uniform(1e-3, 1, X.shape[0]), torch.float64)
asm = pyfem.Assembler(model); asm.rtol = 1e-8
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
out = {}
out["filter_apply_ms"], rho = T(lambda: filtr.apply(xk).clamp(1e-3, 1.0))
def asm_bc():
    K = model.compute_jacobian(rho); rhs = h.to_device(model.compute_rhs(), torch.float64)
    return model.apply_dirichlet_bcs(K, rhs, True)
out["assemble_bc_ms"], (K, rhs) = T(asm_bc)
out["compute_rhs_to_device_ms"], _ = T(lambda: h.to_device(model.compute_rhs(), torch.float64))
out["solve_ms"], u = T(lambda: asm._solve_linear_system(K, rhs, "cg"), 3)
out["iters"] = asm.last_iters
out["compliance_dot_ms"], _ = T(lambda: float(rhs.dot(u)))
out["compliance_grad_ms"], g = T(lambda: model.compliance_grad(rho, u))
out["apply_gradient_ms"], _ = T(lambda: filtr.apply_gradient(g))
out["apply_gradient_iters"] = filtr.last_iters
print(json.dumps(out))
