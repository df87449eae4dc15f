import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pyfem_gpu_testflight_amd import pyfem
def timer(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
qt = pyfem.QuadratureTriangle2D(); bt = pyfem.BasisTriangle2D(qt)
c = pyfem.ProblemCreator(1416, 1416, element_type="tri"); conn, X, df, f = c.create_linear_elasticity_problem()
for fixed in (0, 1, 0, 1):
    if fixed: os.environ["PF_TILE_FIXED"] = "1"
    else: os.environ.pop("PF_TILE_FIXED", None)
    mt = pyfem.LinearElasticity(X, conn, df, None, f, qt, bt); Kt = mt.compute_jacobian()
    maps = mt.hctx.assembly_maps()
    td = maps["tile_desc"]
    print(json.dumps({"PF_TILE_FIXED": fixed, "ntiles": int(td.shape[0]), "mean_nodes_per_tile": float((td[:,3] & 63).mean()), "mean_pairs": float(((td[:,3]>>8)&255).mean()),
                      "tri_ms": round(timer(lambda: mt.compute_jacobian(out=Kt)), 4)}), flush=True)
    del mt, Kt
