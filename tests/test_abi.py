"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
exactly the symbols include/pyfem_hip.h declares; the ctypes table covers all of them.  No
compute call is made (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pyfem_gpu_testflight_amd import _lib, build

    build.build(verbose=False)
    return _lib.load()


def _declared(header="pyfem_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from pyfem_gpu_testflight_amd import _lib

    names = _declared()
    bench = _declared("pyfem_hip_bench.h")  # measurement helpers: their own header, not the drop-in boundary
    assert len(names) >= 25 and bench == ["pf_hbm_probe", "pf_time_kernel"] and not set(bench) & set(names)
    for n in names + bench:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported by libpyfem_hip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == sorted(names + bench)


def test_error_path_without_gpu(lib):
    """NULL ctx -> PF_E_INVALID and a message, no crash; pf_ctx_create fails cleanly without a device."""
    import ctypes as C

    from pyfem_gpu_testflight_amd import _lib

    assert lib.pf_version() == 1
    assert lib.pf_pattern_build(None, 2) == _lib.PF_E_INVALID
    assert b"NULL" in lib.pf_last_error(None)
    import torch

    if not torch.cuda.is_available():
        ctx = C.c_void_p()
        assert lib.pf_ctx_create(0, C.byref(ctx)) < 0
        assert not ctx.value


def test_product_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyfem_gpu_testflight_amd import pyfem

    c = pyfem.ProblemCreator(4, 4)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q = pyfem.QuadratureBilinear2D()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pyfem.LinearElasticity(X, conn, dof_fixed, None, force, q, pyfem.BasisBilinear2D(q))


def test_problem_creator_matches_reference_fixture():
    import numpy as np

    from pyfem_gpu_testflight_amd import pyfem
    from tests._golden import load

    g = load("mesh6x4")
    for etype in ("quad", "tri"):
        c = pyfem.ProblemCreator(6, 4, element_type=etype)
        conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
        assert np.array_equal(conn, g[f"{etype}_conn"]) and np.allclose(X, g[f"{etype}_X"], rtol=0, atol=1e-15)
        assert np.array_equal(np.asarray(dof_fixed), g[f"{etype}_dof_fixed"])
        assert list(force.keys()) == list(g[f"{etype}_force_nodes"])
        assert np.array_equal(np.array(list(force.values())), g[f"{etype}_force_vals"])
        assert np.array_equal(c.create_helmhotz_problem()[2], g[f"{etype}_helm_x"])
        assert np.array_equal(np.asarray(c.create_poisson_problem()[2]), g[f"{etype}_poisson_fixed"])
    q = pyfem.QuadratureBilinear2D()
    b = pyfem.BasisBilinear2D(q)
    from oracle import pyfem_oracle as orc

    N, dN = orc.shape_tables("quad")
    assert np.array_equal(b.eval_shape_fun(), N) and np.array_equal(b.eval_shape_fun_deriv(), dN)
    assert repr(float(q.get_pt()[2, 0])) == "0.5773502691896258"  # the constant the kernels carry
    qt = pyfem.QuadratureTriangle2D()
    bt = pyfem.BasisTriangle2D(qt)
    N, dN = orc.shape_tables("tri")
    assert np.array_equal(bt.eval_shape_fun(), N) and np.array_equal(bt.eval_shape_fun_deriv(), dN)
