#!/usr/bin/env python
"""Tile walk of the assembly kernel (experiments library): round-robin over the grid (default), contiguous run per
wave (PF_TILE_ORDER=c), XCD-aware round-robin (x).   PF_LIB_PATH=.../libpyfem_hip_exp.so python tools/tile_order_ab.py"""
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
for order in ("r", "c", "x", "r", "c", "x"):
    os.environ["PF_TILE_ORDER"] = order
    print(json.dumps({"tile_order": {"r": "round-robin", "c": "contiguous run per wave", "x": "XCD-aware round-robin"}[order],
                      "quad": round(timer(lambda: m.compute_jacobian(out=K)), 4), "tri": round(timer(lambda: mt.compute_jacobian(out=Kt)), 4)}), flush=True)
