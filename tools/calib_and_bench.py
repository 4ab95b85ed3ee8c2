"""Runs the counter-calibration copy (known byte count, 8 B per lane) and then bench.py in one profiled process."""
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cerberus_amd import api, synth  # noqa: E402

ctx = api.Context(synth.default_config(), 0)
api.lib().vilo_debug_calib_copy.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
api.lib().vilo_debug_calib_copy(ctx.h, 1 << 27, 4)   # 4 x (1 GiB read + 1 GiB written)
ctx.close()
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
