"""`--profiler_options` of train.py (reference medicalseg/utils/train_profiler.py:26-112, called once per iteration at
core/train.py:153), on this stack's profilers:

  * inside `batch_range` the library's per-kernel HIP-event profile is on (msk_prof_enable) and written to
    `profile_path` as "tag<TAB>calls<TAB>total_ms" sorted by `sorted_key` when the range ends;
  * every iteration of the range is bracketed by a roctx range "train_iter_<n>" when libroctx64 is present, so a run
    under `rocprofv3 --marker-trace --kernel-trace` shows the iteration boundaries (Paddle's profiler has no equivalent
    here; `state` / `tracer_option` are accepted and ignored);
  * `exit_on_finished` ends the process after the report, as the reference does.

The option string has the reference's format: "batch_range=[50, 60]; profile_path=model.profile; exit_on_finished=False".
"""
import ctypes
import sys

_profiler_step_id = 0
_profiler_options = None
_roctx = None
_range_open = False


class ProfilerOptions(object):
    def __init__(self, options_str):
        assert isinstance(options_str, str)
        self._options = {'batch_range': [10, 20], 'state': 'All', 'sorted_key': 'total', 'tracer_option': 'Default',
                         'profile_path': '/tmp/profile', 'exit_on_finished': True}
        if options_str != "":
            self._parse_from_string(options_str)

    def _parse_from_string(self, options_str):
        for kv in options_str.replace(' ', '').split(';'):
            if not kv:
                continue
            key, value = kv.split('=')
            if key == 'batch_range':
                vals = [int(v) for v in value.replace('[', '').replace(']', '').split(',')]
                if len(vals) >= 2 and vals[0] >= 0 and vals[1] > vals[0]:
                    self._options[key] = vals
            elif key == 'exit_on_finished':
                self._options[key] = value.lower() in ("yes", "true", "t", "1")
            elif key in ('state', 'sorted_key', 'tracer_option', 'profile_path'):
                self._options[key] = value

    def __getitem__(self, name):
        if self._options.get(name, None) is None:
            raise ValueError("ProfilerOptions does not have an option named %s." % name)
        return self._options[name]


def _load_roctx():
    global _roctx
    if _roctx is None:
        _roctx = False
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
    return _roctx


def _write_report(dev, path, sorted_key):
    rep = dev.prof_report()
    key = {"calls": lambda kv: -kv[1][0], "ave": lambda kv: -kv[1][1] / max(kv[1][0], 1)}.get(sorted_key,
                                                                                               lambda kv: -kv[1][1])
    with open(path, "w") as f:
        f.write("# per-kernel HIP-event profile over the profiled iterations\n# tag\tcalls\ttotal_ms\n")
        for tag, (calls, ms) in sorted(rep.items(), key=key):
            f.write("%s\t%d\t%.4f\n" % (tag, calls, ms))


def add_profiler_step(options_str=None, dev=None):
    """One call = one profiler step (call it once per training iteration).  `dev` defaults to the process device."""
    if options_str is None:
        return
    global _profiler_step_id, _profiler_options, _range_open
    if _profiler_options is None:
        _profiler_options = ProfilerOptions(options_str)
    if dev is None:
        from ..device import get_device
        dev = get_device()
    lo, hi = _profiler_options['batch_range'][:2]
    rt = _load_roctx()
    if _range_open and rt:
        rt.roctxRangePop()
        _range_open = False
    if _profiler_step_id == lo:
        dev.set_option("prof_only_halo", 0)
        dev.prof_reset()
        dev.prof_enable(True)
    elif _profiler_step_id == hi:
        dev.sync()
        dev.prof_enable(False)
        _write_report(dev, _profiler_options['profile_path'], _profiler_options['sorted_key'])
        if _profiler_options['exit_on_finished']:
            sys.exit(0)
    if lo <= _profiler_step_id < hi and rt:
        rt.roctxRangePushA(("train_iter_%d" % _profiler_step_id).encode())
        _range_open = True
    _profiler_step_id += 1


def reset():
    """Forget the parsed options and the step counter (tests)."""
    global _profiler_step_id, _profiler_options, _range_open
    _profiler_step_id, _profiler_options, _range_open = 0, None, False
