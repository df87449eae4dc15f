import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
def timer(fn, reps=300, warm=600):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
for ny in (2048, 1024, 512, 256):
    c = pyfem.ProblemCreator(2049, ny + 1); conn, X, df, f = c.create_linear_elasticity_problem()
    m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); K = m.compute_jacobian()
    out[f"2048x{ny}"] = round(timer(lambda: m.compute_jacobian(out=K)), 4)
    m.hctx.close()
print(json.dumps(out))
# host side of one step on the 8-GPU share: wall clock of 3000 back-to-back calls against the device time above
import time
c = pyfem.ProblemCreator(2049, 257); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); K = m.compute_jacobian()
for _ in range(600): m.compute_jacobian(out=K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3000): m.compute_jacobian(out=K)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps({"2048x256 wall_ms_per_step": round((t2 - t0) / 3, 4), "host_issue_ms_per_step": round((t1 - t0) / 3, 4)}))
