#!/usr/bin/env python
"""Kernel timeline of ONE warm configuration-5 elasticity solve and ONE warm Helmholtz filter solve.
   workload (under rocprofv3 --kernel-trace):  python3 tools/solve_timeline.py run
   table from the csv:                         python3 tools/solve_timeline.py table <trace dir>
The workload separates its phases by 60 ms sleeps; the table splits the trace at idle gaps > 30 ms and prints, for each of the
last two segments, per-kernel launch counts and summed durations, the wall time and the idle time between launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run():
    import numpy as np, torch
    from pyfem_gpu_testflight_amd import pyfem
    qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
    c = pyfem.ProblemCreator(2049, 1025)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    filtr = pyfem.Helmholtz(0.05, X, conn, qq, bq); filtr.reuse_forward = False
    model = pyfem.LinearElasticity(X, conn, dof_fixed, None, force, qq, bq, p=5.0)
    h = model.hctx
    x0 = h.to_device(np.random.default_rng(0).uniform(1e-3, 1, X.shape[0]), torch.float64)
    asm = pyfem.Assembler(model)
    rho = filtr.apply(x0).clamp(1e-3, 1.0)
    def solve():
        K = model.compute_jacobian(rho); rhs = model.compute_rhs_device(); K2, r2 = model.apply_dirichlet_bcs(K, rhs, True)
        return asm._solve_linear_system(K2, r2, "cg")
    for _ in range(3): solve(); filtr.apply(x0)
    torch.cuda.synchronize(); time.sleep(0.06)
    solve(); torch.cuda.synchronize(); time.sleep(0.06)
    filtr.apply(x0); torch.cuda.synchronize()
    print("iters", asm.last_iters, filtr.last_iters)

def table(d):
    import csv, glob, collections
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:100])
                   for r in csv.DictReader(open(f))))
    segs, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - cur[-1][1] > 30e6: segs.append(cur); cur = []
        cur.append(r)
    segs.append(cur)
    for title, seg in zip(("elasticity solve (assembly, BCs, numeric setup, PCG)", "filter apply (R x, PCG with the kept hierarchy)"), segs[-2:]):
        wall = (seg[-1][1] - seg[0][0]) / 1e3; busy = sum(e - s for s, e, _ in seg) / 1e3
        print(f"== {title}: {len(seg)} launches, wall {wall:.0f} us, kernels {busy:.0f} us, idle between launches {wall - busy:.0f} us")
        agg = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in seg: agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{t:9.1f} us {c:5d} x {t / c:8.1f}  {n}")
        print()

if __name__ == "__main__":
    run() if sys.argv[1] == "run" else table(sys.argv[2])
