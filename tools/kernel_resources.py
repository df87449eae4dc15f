#!/usr/bin/env python
"""Registers / occupancy / LDS / scratch of every kernel of one .hip file, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.   python tools/kernel_resources.py pf_assemble.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "pyfem_gpu_testflight_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"),
       "-I", os.path.dirname(src), "-c", src, "-o", "/tmp/isa/x.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if flt in name:
        print(f"{name:70s} VGPR {v.get('VGPRs','?'):>4} AGPR {v.get('AGPRs','?'):>3} SGPR {v.get('TotalSGPRs', v.get('SGPRs','?')):>4} occ {v.get('Occupancy','?')} scratch {v.get('ScratchSize','?')} lds {v.get('LDS Size','?')}")
