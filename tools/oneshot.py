"""Wall time of a one-shot script (the shape of the reference's examples): mesh, model, Assembler.solve, numpy out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
import numpy as np, torch
from pyfem_gpu_testflight_amd import pyfem
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t_import = time.perf_counter() - t00
for etype, shape in (("quad", (1025, 1025)), ("block", (65, 65, 65))):
    T = [time.perf_counter()]
    if etype == "quad":
        c = pyfem.ProblemCreator(*shape); q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
    else:
        c = pyfem.ProblemCreator(*shape, element_type="block"); q = pyfem.QuadratureBlock3D(); b = pyfem.BasisBlock3D(q)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem(); T.append(time.perf_counter())
    model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, b); torch.cuda.synchronize(); T.append(time.perf_counter())
    u = pyfem.Assembler(model).solve(method="cg"); T.append(time.perf_counter())
    u2 = pyfem.Assembler(model).solve(method="cg"); T.append(time.perf_counter())
    print(f"{etype} {shape}: ndof {u.size}: mesh {T[1]-T[0]:.3f} s, model (upload + pattern) {T[2]-T[1]:.3f} s, first solve {T[3]-T[2]:.3f} s, "
          f"second solve {T[4]-T[3]:.3f} s  (import + GPU init {t_import:.2f} s)", flush=True)
