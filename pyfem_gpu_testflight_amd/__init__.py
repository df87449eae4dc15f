"""MI355X-native implementation of the pyfem hot path (assembly -> scatter -> SpMV/CG).

    from pyfem_gpu_testflight_amd import pyfem, utils, parse_inp   # drop-in for the reference's modules

The compute lives in libpyfem_hip.so (hand-written HIP for gfx950, C ABI in include/pyfem_hip.h);
this package is the host-side mirror of the reference's Model/Assembler surface.
"""
from . import _lib, utils  # noqa: F401

__all__ = ["pyfem", "utils", "parse_inp", "meshtools", "device", "dist"]


def __getattr__(name):
    if name in ("pyfem", "device", "dist", "parse_inp", "meshtools"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
