"""Mirror of the reference's `utils` profiler surface (utils.py:9-151) -- observability only.

`@time_this` keeps the reference's qualnames in `MyProfiler.saved_times[qualname] -> [ms, ...]`
so drivers that read those keys (examples/SciTech2023/performance/performance_test.py:103-113)
keep working.  `to_vtk` writes the same legacy VTK files.  Differences, all deliberate: no ./profiler.log is written, timings of methods that
only enqueue GPU work cover the enqueue unless MyProfiler.sync_device is set (then the device is
synchronised around each timed call), and on/off is honoured at call time.
"""
import functools
from time import perf_counter_ns


class MyProfiler:
    is_on = False
    print_to_stdout = False
    sync_device = False
    t_min = 1.0  # ms
    saved_times = {}

    @staticmethod
    def timer_on():
        MyProfiler.is_on = True

    @staticmethod
    def timer_off():
        MyProfiler.is_on = False

    @staticmethod
    def timer_to_stdout():
        MyProfiler.print_to_stdout = True

    @staticmethod
    def timer_set_threshold(t):
        MyProfiler.t_min = t

    @staticmethod
    def time_this(func):
        name = func.__qualname__

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            if not MyProfiler.is_on:
                return func(*args, **kwargs)
            if MyProfiler.sync_device:
                _sync()
            t0 = perf_counter_ns()
            out = func(*args, **kwargs)
            if MyProfiler.sync_device:
                _sync()
            ms = (perf_counter_ns() - t0) / 1e6
            MyProfiler.saved_times.setdefault(name, []).append(ms)
            if MyProfiler.print_to_stdout and ms >= MyProfiler.t_min:
                print(f"{name:<60s} {ms:10.3f} ms")
            return out

        return wrapper


def _sync():
    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()


time_this = MyProfiler.time_this
timer_on = MyProfiler.timer_on
timer_off = MyProfiler.timer_off
timer_to_stdout = MyProfiler.timer_to_stdout
timer_set_threshold = MyProfiler.timer_set_threshold


# VTK cell types of the element families ProblemCreator can produce (legacy ASCII format)
_VTK_CELL = {3: 5, 4: 9, 8: 12, 10: 24, 20: 25}


def to_vtk(conn, X, nodal_sol={}, vtk_name="problem.vtk"):
    """Legacy-ASCII VTK unstructured grid with optional nodal scalars -- same call as the reference's utils.to_vtk
    (utils.py:302-428; examples/block/block.py:62 ends with it), written with numpy instead of per-row Python loops.
    conn: (nelems, P) array, or {name: array} for a mixed mesh (the cell type follows from P)."""
    import numpy as np

    blocks = list(conn.values()) if isinstance(conn, dict) else [conn]
    blocks = [np.asarray(c, dtype=np.int64) for c in blocks]
    for c in blocks:
        if c.ndim != 2 or c.shape[1] not in _VTK_CELL:
            raise ValueError(f"to_vtk: no VTK cell type for {c.shape[1] if c.ndim == 2 else '?'}-node elements")
    X = np.asarray(X, dtype=float)
    if X.shape[1] == 2:  # VTK points are three-dimensional
        X = np.concatenate([X, np.zeros((X.shape[0], 1))], axis=1)
    nelems = sum(c.shape[0] for c in blocks)
    with open(vtk_name, "w") as fh:
        fh.write("# vtk DataFile Version 3.0\npyfem mesh\nASCII\nDATASET UNSTRUCTURED_GRID\n")
        fh.write(f"POINTS {X.shape[0]} double\n")
        np.savetxt(fh, X, fmt="%.17g")
        fh.write(f"CELLS {nelems} {sum(c.shape[0] * (1 + c.shape[1]) for c in blocks)}\n")
        for c in blocks:
            np.savetxt(fh, np.concatenate([np.full((c.shape[0], 1), c.shape[1], dtype=np.int64), c], axis=1), fmt="%d")
        fh.write(f"CELL_TYPES {nelems}\n")
        for c in blocks:
            np.savetxt(fh, np.full(c.shape[0], _VTK_CELL[c.shape[1]], dtype=np.int64), fmt="%d")
        if nodal_sol:
            fh.write(f"POINT_DATA {X.shape[0]}\n")
            for name, data in nodal_sol.items():
                data = np.asarray(data, dtype=float).ravel()
                if data.size != X.shape[0]:
                    raise ValueError(f"to_vtk: nodal field {name!r} has {data.size} values for {X.shape[0]} nodes")
                fh.write(f"SCALARS {name} float 1\nLOOKUP_TABLE default\n")
                np.savetxt(fh, data, fmt="%.9g")
    print(f"[Info] Done generating {vtk_name}")
