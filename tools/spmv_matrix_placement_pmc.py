#!/usr/bin/env python
"""The SpMV placement effect is a property of the MATRIX VALUE buffer (tools/spmv_after_probe.py: a clone of K ran at
0.2345 ms against 0.2119 for K, the same for every output vector).  This script makes NC clones of the metric mesh's value
array (1.2 GB each), times the product on each, and then runs the FASTEST and the SLOWEST REP times each, in that order,
as the last 2*REP k_spmv dispatches of the process -- so that a rocprofv3 --pmc pass can compare them counter by counter:
   rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace \\
             --output-format csv -d <dir> -o p -- python3 tools/spmv_matrix_placement_pmc.py
   python3 tools/spmv_matrix_placement_pmc.py --summarise <dir> [REP]   -> mean counter values of the two groups"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    import csv, glob
    from collections import defaultdict
    d, rep = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void k_spmv<2, 1, 3")]
    byd = defaultdict(dict)
    for r in rows:
        byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(byd)[-2 * rep:]
    out = {}
    for name, part in (("fast", ids[:rep]), ("slow", ids[rep:])):
        acc = defaultdict(float)
        for i in part:
            for k, v in byd[i].items():
                acc[k] += v / len(part)
        out[name] = dict(acc)
    out["slow_over_fast"] = {k: (out["slow"][k] / out["fast"][k] if out["fast"].get(k) else None) for k in out["slow"]}
    print(json.dumps(out, indent=1))
    sys.exit(0)
import torch
from pyfem_gpu_testflight_amd import pyfem, _lib
NC = int(os.environ.get("NC", 6)); REP = int(os.environ.get("REP", 20))
qq = pyfem.QuadratureBilinear2D(); bq = pyfem.BasisBilinear2D(qq)
c = pyfem.ProblemCreator(2049, 2049); conn, X, df, f = c.create_linear_elasticity_problem()
m = pyfem.LinearElasticity(X, conn, df, None, f, qq, bq); h = m.hctx
K = m.compute_jacobian().vals
x = torch.rand(h.nrows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
Ks = [K] + [K.clone() for _ in range(NC - 1)]
h.time_kernel(_lib.PF_K_SPMV, 300, in0=K, in1=x, out0=y)
t1 = [h.time_kernel(_lib.PF_K_SPMV, 40, in0=v, in1=x, out0=y) for v in Ks]
t2 = [h.time_kernel(_lib.PF_K_SPMV, 40, in0=v, in1=x, out0=y) for v in Ks]
fast, slow = min(range(NC), key=lambda i: t2[i]), max(range(NC), key=lambda i: t2[i])
torch.cuda.synchronize()
for i in (fast, slow):
    for _ in range(REP):
        h.spmv(Ks[i], x, out=y)
    torch.cuda.synchronize()
print(json.dumps({"ms_first": [round(t, 4) for t in t1], "ms_second": [round(t, 4) for t in t2], "fast": fast, "slow": slow, "rep": REP,
                  "ptrs": [hex(v.data_ptr()) for v in Ks]}))
