#!/usr/bin/env python
"""Assembly-kernel sweep with an experiments library (PF_LIB_PATH=.../libpyfem_hip_exp.so): persistent waves per CU
for the four 2-D kernels of the bench legs.   python tools/asm_sweep.py [tag]   prints one JSON line per setting."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("PF_LIB_PATH", "product"))

def timer(fn, reps=100, warm=150):
    # the device needs ~100 back-to-back launches (30 ms) after an idle period to reach its steady clocks
    # (tools/launch_series.py: 0.346 -> 0.262 ms for the quad kernel): warm up that long, then time
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
qt = pyfem.QuadratureTriangle2D(); bt = pyfem.BasisTriangle2D(qt)
cases = {}
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq, p=5.0); K = m.compute_jacobian()
rho = torch.rand(m.hctx.nnodes, dtype=torch.float64, device="cuda") * 0.7 + 0.3
cases["quad"] = lambda: m.compute_jacobian(out=K)
cases["quad_rho"] = lambda: m.compute_jacobian(rho, out=K)
c = pyfem.ProblemCreator(1416, 1416, element_type="tri"); conn, X, df, f = c.create_linear_elasticity_problem()
mt = pyfem.LinearElasticity(X, conn, df, None, f, qt, bt); Kt = mt.compute_jacobian()
cases["tri"] = lambda: mt.compute_jacobian(out=Kt)
import bench
conn, X, df, f = bench.arm_mesh(5)
ma = pyfem.LinearElasticity(X, conn, df, None, f, qt, bt); Ka = ma.compute_jacobian()
cases["arm5"] = lambda: ma.compute_jacobian(out=Ka)
c = pyfem.ProblemCreator(2049, 2049); conn, X, x = c.create_helmhotz_problem()
mh = pyfem.Helmholtz(0.05, X, conn, qq, bq)
Kv, Rv = mh.hctx.assemble_helmholtz(0.05)
import ctypes as C
cases["helm_KR"] = lambda: mh.hctx._call("pf_assemble_helmholtz2d", 0.05, C.c_void_p(Kv.data_ptr()), C.c_void_p(Rv.data_ptr()))
for wpc in [int(a) for a in os.environ.get("WPC", "0").split()]:
    if wpc: os.environ["PF_GATHER_WAVES_PER_CU"] = str(wpc)
    else: os.environ.pop("PF_GATHER_WAVES_PER_CU", None)
    print(json.dumps({"lib": tag, "wpc": wpc or "auto", **{k: round(timer(fn), 4) for k, fn in cases.items()}}), flush=True)
