import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from pyfem_gpu_testflight_amd import pyfem
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, x = c.create_helmhotz_problem()
mh = pyfem.Helmholtz(0.05, X, conn, qq, bq)
Kv, Rv = mh.hctx.assemble_helmholtz(0.05)
fn = lambda: mh.hctx._call("pf_assemble_helmholtz2d", 0.05, C.c_void_p(Kv.data_ptr()), C.c_void_p(Rv.data_ptr()))
def timer(fn, reps=200, warm=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(json.dumps({"lib": os.environ.get("PF_LIB_PATH", "product")[-12:], "helm_KR_ms": [round(timer(fn), 4) for _ in range(3)]}))
