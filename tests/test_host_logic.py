"""Sanitizer build of the host-side logic (VERDICT r2 hygiene): csrc/pf_hostlogic.h -- the greedy
AMG aggregation, the coarse graphs, the host fallback of the prolongator patterns and the tables of the fused halo
exchange -- is plain C++ and is compiled here WITHOUT HIP, with AddressSanitizer and UBSan, against
tests/host/host_logic_check.cpp, then run on the CPU.  A sanitizer report or a failed check fails the test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_under_address_and_ub_sanitizers(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "host_logic_check")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
           "-Wno-unused-function", "-pthread", "-I", os.path.join(ROOT, "pyfem_gpu_testflight_amd", "csrc"),
           os.path.join(ROOT, "tests", "host", "host_logic_check.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "host logic ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
