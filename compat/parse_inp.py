"""`from parse_inp import InpParser` for scripts written against the reference (see compat/pyfem.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfem_gpu_testflight_amd.parse_inp import InpParser  # noqa: F401,E402
