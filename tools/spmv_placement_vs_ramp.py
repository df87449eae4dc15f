#!/usr/bin/env python
"""Is the 'two-speed' SpMV of round 2 (0.208 / 0.235 ms "depending on where y lies") a property of the OUTPUT BUFFER or of
WHEN it is measured?  bench.py's SpMV leg timed three candidate output vectors one after the other, 1 + 50 launches each,
right after the assembly leg.  Here: six candidates, timed in that style (a) right after a burst of 300 assembly launches,
in order 0..5, (b) again after another burst, in REVERSE order, (c) after 300 SpMV launches of warm-up, in order, (d) in
reverse.  A buffer effect follows the buffer; a ramp follows the position in the sequence.
   python tools/spmv_placement_vs_ramp.py  ->  JSON lines (profiles/r03_spmv_placement_vs_ramp.jsonl)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); h = m.hctx
Kc = m.compute_jacobian(); K = Kc.vals
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda")
ys = [torch.empty_like(x) for _ in range(6)]
def burst(n=300):
    for _ in range(n): m.compute_jacobian(out=Kc)
    torch.cuda.synchronize()
def sweep(order, reps=50):
    return {i: round(h.time_kernel(_lib.PF_K_SPMV, reps, in0=K, in1=x, out0=ys[i]), 4) for i in order}
burst(); print(json.dumps({"a_after_assembly_burst_in_order": sweep(range(6))}), flush=True)
burst(); print(json.dumps({"b_after_assembly_burst_reverse_order": sweep(reversed(range(6)))}), flush=True)
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K, in1=x, out0=ys[0])
print(json.dumps({"c_after_300_spmv_in_order": sweep(range(6))}), flush=True)
print(json.dumps({"d_after_300_spmv_reverse_order": sweep(reversed(range(6)))}), flush=True)
torch.cuda.synchronize()
import time; time.sleep(0.5)  # idle
print(json.dumps({"e_after_half_a_second_idle_in_order_6_launches_each": sweep(range(6), reps=6)}), flush=True)
print(json.dumps({"f_again_6_launches_each": sweep(range(6), reps=6)}), flush=True)
