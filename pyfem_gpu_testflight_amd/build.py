"""Builds libpyfem_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m pyfem_gpu_testflight_amd.build          # incremental
    python -m pyfem_gpu_testflight_amd.build --force
    python -m pyfem_gpu_testflight_amd.build --experiments   # libpyfem_hip_exp.so with -DPF_EXPERIMENTS for tools/:
                                                              # the ablation / tuning knobs exist only there
                                                              # (select it with PF_LIB_PATH=<path> before importing)
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libpyfem_hip.so")
SOURCES = ["pf_pattern.hip", "pf_assemble.hip", "pf_assemble3d.hip", "pf_tabular.hip", "pf_solve.hip", "pf_amg.hip", "pf_halo.hip", "pf_dist.hip", "pf_bench.hip"]
LIB_EXP = os.path.join(PKG, "libpyfem_hip_exp.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpyfem_hip.so cannot be built (no CPU fallback exists)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, experiments=False):
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build_exp" if experiments else "build")
    flags = FLAGS + (["-DPF_EXPERIMENTS", "-DPF_SPMV_SWEEP"] if experiments else [])
    lib = LIB_EXP if experiments else LIB
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "pf_common.h"), os.path.join(CSRC, "pf_hostlogic.h"), os.path.join(INC, "pyfem_hip.h"), os.path.join(INC, "pyfem_hip_bench.h")]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + flags + ["-I", INC, "-I", CSRC, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
