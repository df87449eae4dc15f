"""Per-kernel breakdown of one AMG-PCG solve from a rocprofv3 --kernel-trace database (rocpd sqlite, the default output
of this image's rocprofv3).   python tools/amg_breakdown.py <results.db> [which]
`which` counts the pf_cg calls of the traced run (k_cg_init launches), 0-based; the segment runs from the end of the
previous solve to the last k_cg_update_p of this one, so it includes this solve's numeric AMG setup."""
import sqlite3, re, collections, sys
db=sqlite3.connect(sys.argv[1])
cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'info_kernel_symbol' in t][0]
names={r[0]:r[1] for r in cur.execute(f"select id, display_name from {ks}")}
rows=list(cur.execute(f"select kernel_id,start,end,grid_size_x from {kd} order by start"))
def short(n):
    n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n); return n[:70]
seq=[(short(names[k]),s,e,g) for k,s,e,g in rows]
inits=[i for i,(n,*_ ) in enumerate(seq) if n.startswith('k_cg_init')]
print(len(seq), inits[:12])
which=int(sys.argv[2]) if len(sys.argv)>2 else 4
i5=inits[which]
lastp=max(t for t in range(inits[which-1],i5) if seq[t][0].startswith('k_cg_update_p'))
a=lastp+1; b=inits[which+1] if which+1 < len(inits) else len(seq)
endp=max(t for t in range(i5,b) if seq[t][0].startswith('k_cg_update_p'))
seg=seq[a:endp+1]
tot=collections.defaultdict(lambda:[0,0])
for n,s,e,g in seg:
    tot[n][0]+=e-s; tot[n][1]+=1
wall=seg[-1][2]-seg[0][1]
busy=sum(v[0] for v in tot.values())
print("segment kernels",len(seg),"wall ms",wall/1e6,"busy ms",busy/1e6)
for n,(t,c) in sorted(tot.items(), key=lambda kv:-kv[1][0])[:30]:
    print(f"{t/1e6:8.3f} ms {c:5d} x {t/c/1e3:8.1f} us  {n}")
si=[t for t in range(a,endp+1) if seq[t][0].startswith('k_cg_init')][0]
print("setup wall ms", (seq[si][1]-seq[a][1])/1e6, "iter wall ms", (seq[endp][2]-seq[si][1])/1e6, "first", seq[a][0])
