"""Launch helper for tests/test_bench_launch_dryrun.py: runs the real bench.py main() against the
no-compute stand-in library (tests/fake_msegk.c) so the multi-process control flow (env parsing,
TCP rendezvous of the RCCL id, DataParallel wiring, rank-0 JSON line) can be exercised on CPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import medicalseg_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.environ["MSK_FAKE_LIB"]
_lib._lib = None
import bench  # noqa: E402

bench.main()
