"""Host-side utilities of the mirror module that need no GPU."""
import numpy as np

from oracle import pyfem_oracle as orc
from pyfem_gpu_testflight_amd import utils


def test_to_vtk_writes_a_readable_legacy_file(tmp_path):
    conn, X = orc.structured_mesh3d(3, 3, 2)
    u = np.arange(X.shape[0], dtype=float)
    path = tmp_path / "block.vtk"
    utils.to_vtk(conn, X, {"ux": u, "uy": 2 * u}, vtk_name=str(path))
    lines = path.read_text().splitlines()
    assert lines[3] == "DATASET UNSTRUCTURED_GRID" and lines[4] == f"POINTS {X.shape[0]} double"
    pts = np.array([[float(v) for v in ln.split()] for ln in lines[5:5 + X.shape[0]]])
    assert np.allclose(pts, X)
    k = 5 + X.shape[0]
    assert lines[k] == f"CELLS {conn.shape[0]} {conn.shape[0] * 9}"
    cells = np.array([[int(v) for v in ln.split()] for ln in lines[k + 1:k + 1 + conn.shape[0]]])
    assert np.all(cells[:, 0] == 8) and np.array_equal(cells[:, 1:], conn)
    k += 1 + conn.shape[0]
    assert lines[k] == f"CELL_TYPES {conn.shape[0]}" and set(lines[k + 1:k + 1 + conn.shape[0]]) == {"12"}
    assert f"POINT_DATA {X.shape[0]}" in lines and "SCALARS ux float 1" in lines and "SCALARS uy float 1" in lines
    # 2-D meshes get a zero z column; mixed meshes come as a dict
    c2, X2 = orc.structured_mesh(3, 3, "quad")
    c3, _ = orc.structured_mesh(3, 3, "tri")
    utils.to_vtk({"quad": c2, "tri": c3[:2]}, X2, vtk_name=str(tmp_path / "mixed.vtk"))
    txt = (tmp_path / "mixed.vtk").read_text().splitlines()
    assert f"CELLS {c2.shape[0] + 2} {c2.shape[0] * 5 + 2 * 4}" in txt
    assert txt[5].split()[2] in ("0", "0.0")


def test_profiler_records_by_qualname():
    class A:
        @utils.time_this
        def work(self):
            return 7

    utils.timer_off()
    assert A().work() == 7 and "test_profiler_records_by_qualname.<locals>.A.work" not in utils.MyProfiler.saved_times
    utils.timer_on()
    try:
        assert A().work() == 7
        assert len(utils.MyProfiler.saved_times["test_profiler_records_by_qualname.<locals>.A.work"]) == 1
    finally:
        utils.timer_off()


_INP = """*Heading
 synthetic
*NODE
3, 1.0, 1.0, 0
1, 0.0, 0.0, 0
2, 1.0, 0.0, 0
4, 0.0, 1.0, 0
** a comment line
*ELEMENT, type=T3D2, ELSET=Line1
1, 1, 2
*ELEMENT, type=CPS3, ELSET=Surface1
3, 3, 4, 1
2, 1, 2, 3
*ELSET,ELSET=load
2,
*NSET,NSET=load
3, 4,
*NSET,NSET=fixed
1,
2
"""


def test_inp_parser_conventions(tmp_path):
    from pyfem_gpu_testflight_amd.parse_inp import InpParser

    path = tmp_path / "mesh.inp"
    path.write_text(_INP)
    conn, X, groups = InpParser(str(path)).parse()
    assert list(conn) == ["CPS3"] and np.array_equal(conn["CPS3"], [[0, 1, 2], [2, 3, 0]])  # sorted by element id
    assert np.array_equal(X, [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])  # sorted by node id, 3 coordinates
    assert np.array_equal(groups["load"], [2, 3]) and np.array_equal(groups["fixed"], [0, 1])
    p = InpParser(str(path))
    p.to_vtk({"t": np.arange(4.0)})
    assert (tmp_path / "mesh.vtk").read_text().count("SCALARS t float 1") == 1


def test_inp_parser_reads_the_reference_arm_mesh_like_the_reference():
    """When the reference checkout is present (build container only): arm.inp through this parser gives the mesh
    and groups the reference's own parser produced for the committed fixture (oracle/gen_golden.py)."""
    import os

    import pytest

    from pyfem_gpu_testflight_amd.parse_inp import InpParser
    from tests._golden import load

    path = "/root/reference/examples/torque_arm/arm.inp"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    conn, X, groups = InpParser(path).parse()
    g = load("arm_tri")
    assert np.array_equal(conn["CPS3"], g["conn"]) and np.array_equal(X[:, :2], g["X"])
    fixed = np.concatenate((2 * groups["fixed"], 2 * groups["fixed"] + 1))
    assert np.array_equal(np.sort(fixed), np.sort(g["dof_fixed"]))
    assert np.array_equal(np.sort(groups["load"]), np.sort(g["force_nodes"]))


def test_compat_modules_shadow_the_reference_names():
    """compat/ on sys.path makes `import pyfem, utils` / `from parse_inp import InpParser` resolve to this package
    (how the reference's own examples and tests are pointed at the MI355X path)."""
    import importlib
    import os
    import sys

    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")
    saved = {k: sys.modules.pop(k) for k in ("pyfem", "utils", "parse_inp") if k in sys.modules}
    sys.path.insert(0, compat)
    try:
        pyfem = importlib.import_module("pyfem")
        utils_mod = importlib.import_module("utils")
        parse_inp = importlib.import_module("parse_inp")
        from pyfem_gpu_testflight_amd import pyfem as ours

        assert pyfem.LinearElasticity is ours.LinearElasticity and pyfem.ProblemCreator is ours.ProblemCreator
        assert pyfem.QuadratureBlock3D is ours.QuadratureBlock3D and pyfem.Assembler is ours.Assembler
        assert callable(utils_mod.time_this) and callable(utils_mod.to_vtk) and callable(utils_mod.timer_on)
        assert parse_inp.InpParser.__module__ == "pyfem_gpu_testflight_amd.parse_inp"
        # the mesh generator needs no GPU: the reference's scripts start with it
        conn, X, dof_fixed, force = pyfem.ProblemCreator(4, 3).create_linear_elasticity_problem()
        assert conn.shape == (6, 4) and X.shape == (12, 2) and len(dof_fixed) == 6 and list(force) == [3]
    finally:
        sys.path.remove(compat)
        for k in ("pyfem", "utils", "parse_inp"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
