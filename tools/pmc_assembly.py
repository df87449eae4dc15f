"""Assembly kernel under the counters for quads, split-quad triangles and the refined arm (tools/pmc_assembly.sh; r01 form:
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS \
             SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d out -o p -- python3 tools/pmc_assembly.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
import bench
for etype, n in (("quad", 2049), ("tri", 1416), ("tri", 0)):
    if n:
        c = pyfem.ProblemCreator(n, n, element_type=etype)
        conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    else:
        conn, X, dof_fixed, force = bench.arm_mesh(5)  # refined torque arm, refinement order
    q = pyfem.QuadratureBilinear2D() if etype == "quad" else pyfem.QuadratureTriangle2D()
    b = pyfem.BasisBilinear2D(q) if etype == "quad" else pyfem.BasisTriangle2D(q)
    m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b)
    K = m.compute_jacobian()
    for _ in range(10):
        m.compute_jacobian(out=K)
    torch.cuda.synchronize()
    print(etype, m.hctx.nelems, flush=True)
# round 3: the two kernels VERDICT r2 asked to re-record -- quads with a nodal density field (RAMP) and Helmholtz K + R
import ctypes as C
c = pyfem.ProblemCreator(2049, 2049); conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
m = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b, p=5.0)
rho = torch.rand(m.hctx.nnodes, dtype=torch.float64, device="cuda") * 0.7 + 0.3
K = m.compute_jacobian(rho)
for _ in range(10):
    m.compute_jacobian(rho, out=K)
torch.cuda.synchronize()
print("quad rho", m.hctx.nelems, flush=True)
c = pyfem.ProblemCreator(2049, 2049); conn, X, x0 = c.create_helmhotz_problem()
mh = pyfem.Helmholtz(0.05, X, conn, q, b)
Kv, Rv = mh.hctx.assemble_helmholtz(0.05)
for _ in range(10):
    mh.hctx._call("pf_assemble_helmholtz2d", 0.05, C.c_void_p(Kv.data_ptr()), C.c_void_p(Rv.data_ptr()))
torch.cuda.synchronize()
print("helmholtz K+R", mh.hctx.nelems, flush=True)
