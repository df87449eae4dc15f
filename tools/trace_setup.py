"""Timeline of the numeric AMG setup of ONE elasticity solve from the kernel trace csv of tools/trace_c5_parts.sh:
every launch between the assembly of K and k_cg_init, with its start offset, duration and the idle gap before it."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
inits = [i for i, r in enumerate(rows) if name(r).startswith("k_cg_init")]
# the last solve whose setup contains a Galerkin product of the finest level
cands = [i for i in inits if any(name(rows[j]).startswith(("k_sa_galerkin_wave<2, 3", "k_sa_ap<2, 3")) for j in range(max(0, i - 400), i))]
i1 = cands[-1]
i0 = max(j for j in range(i1) if name(rows[j]).startswith(("k_cg_update_p", "k_apply_dirichlet")))  # end of the previous solve
t0 = int(rows[i0]["End_Timestamp"]); prev = t0; busy = 0
for r in rows[i0 + 1:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {name(r)}")
    busy += e - s; prev = e
print("setup wall", (prev - t0) / 1e6, "ms, busy", busy / 1e6, "ms, launches", i1 - i0)
