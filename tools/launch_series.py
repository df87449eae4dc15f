#!/usr/bin/env python
"""Per-launch device times of the quad elasticity assembly over a long back-to-back series (HIP events around every
launch): does the kernel speed up as the series goes on (clock ramp / cache state)?  Prints the mean of every bin of 25."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem
N = int(os.environ.get("N", "500"))
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq, p=5.0); K = m.compute_jacobian()
torch.cuda.synchronize()
import time; time.sleep(float(os.environ.get("IDLE", "2")))  # let the device go idle first
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    m.compute_jacobian(out=K); ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print(json.dumps({"bins_of_25_ms": [round(sum(t[i:i + 25]) / 25, 4) for i in range(0, N, 25)], "first5": [round(x, 4) for x in t[:5]]}))
