"""bench.py on libmsegk_test.so (the build that contains the host transport): used by tests/test_gpu_dp2.py to run the
multi-rank path of bench.py with two processes on one GPU.  Same argv as bench.py."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicalseg_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmsegk_test.so")
_lib._lib = None
os.environ["MSEGK_LIB"] = _lib.LIB_PATH    # bench.py --gpus N re-executes itself as the supervised workers: they inherit the choice
sys.argv[0] = os.path.join(ROOT, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
