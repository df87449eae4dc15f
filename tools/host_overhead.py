import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem
c = pyfem.ProblemCreator(65, 65)
conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
K = m.compute_jacobian()
for _ in range(200): m.compute_jacobian(out=K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): m.compute_jacobian(out=K)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host us per compute_jacobian(out=K) call:", (t1 - t0) / 2000 * 1e6, "incl. drain:", (t2 - t0) / 2000 * 1e6)
