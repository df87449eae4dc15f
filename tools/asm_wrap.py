#!/usr/bin/env python
"""Ablation (experiments library): the assembly kernels with their copy-out wrapped into the first N MB of K
(PF_ABL_WRAP_MB) -- wrong results, timing only.   PF_LIB_PATH=.../libpyfem_hip_exp.so python tools/asm_wrap.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem
def timer(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
qt = pyfem.QuadratureTriangle2D(); bt = pyfem.BasisTriangle2D(qt)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); K = m.compute_jacobian()
c = pyfem.ProblemCreator(1416, 1416, element_type="tri"); conn, X, df, f = c.create_linear_elasticity_problem()
mt = pyfem.LinearElasticity(X, conn, df, None, f, qt, bt); Kt = mt.compute_jacobian()
timer(lambda: m.compute_jacobian(out=K))
for mb in (0, 4):
    for mask in (0, 1, 2, 4, 8, 3, 5, 7, 15, 0):
        if mb: os.environ["PF_ABL_WRAP_MB"] = str(mb)
        else: os.environ.pop("PF_ABL_WRAP_MB", None)
        os.environ["PF_ABL_MASK"] = str(mask)
        print(json.dumps({"wrap_MB": mb or "off", "mask(1 adds,2 clear,4 quadrature,8 copy-out)": mask,
                          "quad": round(timer(lambda: m.compute_jacobian(out=K)), 4),
                          "tri": round(timer(lambda: mt.compute_jacobian(out=Kt)), 4)}), flush=True)
