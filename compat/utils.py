"""`import utils` for scripts written against the reference (see compat/pyfem.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfem_gpu_testflight_amd.utils import *  # noqa: F401,F403,E402
from pyfem_gpu_testflight_amd.utils import (  # noqa: F401,E402
    MyProfiler, time_this, timer_off, timer_on, timer_set_threshold, timer_to_stdout, to_vtk)
