"""Mirror of the reference's `utils` profiler surface (utils.py:9-151) -- observability only.

`@time_this` keeps the reference's qualnames in `MyProfiler.saved_times[qualname] -> [ms, ...]`
so drivers that read those keys (examples/SciTech2023/performance/performance_test.py:103-113)
keep working.  Differences, all deliberate: no ./profiler.log is written, timings of methods that
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
