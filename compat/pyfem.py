"""`import pyfem` for scripts written against the reference: put this directory on PYTHONPATH (it then shadows the
reference's own pyfem.py, which the scripts look for through sys.path.append("..")) and every name resolves to the
MI355X implementation in pyfem_gpu_testflight_amd.pyfem.

    PYTHONPATH=/path/to/this/repo/compat python examples/block/block.py --solver cg
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfem_gpu_testflight_amd.pyfem import *  # noqa: F401,F403,E402
from pyfem_gpu_testflight_amd.pyfem import (  # noqa: F401,E402  (underscore-free names the reference's scripts use)
    Assembler, BasisBilinear2D, BasisBlock3D, BasisTriangle2D, Helmholtz, LinearElasticity, LinearPoisson, ProblemCreator,
    QuadratureBilinear2D, QuadratureBlock3D, QuadratureTriangle2D)
