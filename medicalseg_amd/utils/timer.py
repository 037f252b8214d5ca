"""Throughput bookkeeping for the train loop: mean batch / reader cost and samples per second
(`ips`), same public surface as the reference's utils/timer.py (TimeAverager.record /
get_average / get_ips_average / reset, calculate_eta) so core/train.py's log lines keep
their meaning.  ips is per process, as in the reference."""
from dataclasses import dataclass


@dataclass
class _Window:
    events: int = 0
    seconds: float = 0.0
    samples: int = 0


class TimeAverager:
    """Accumulates (duration, optional sample count) pairs over one logging window."""

    def __init__(self):
        self._w = _Window()

    def reset(self):
        self._w = _Window()

    def record(self, usetime, num_samples=None):
        w = self._w
        w.events += 1
        w.seconds += usetime
        w.samples += int(num_samples) if num_samples else 0

    def get_average(self):
        """Mean seconds per recorded event (0 before the first record)."""
        w = self._w
        return w.seconds / w.events if w.events else 0

    def get_ips_average(self):
        """Samples per second over the window (0 when no sample counts were recorded)."""
        w = self._w
        return w.samples / w.seconds if (w.events and w.samples) else 0


def calculate_eta(remaining_step, speed):
    """'HH:MM:SS' for `remaining_step` steps at `speed` seconds per step."""
    seconds = int(max(remaining_step, 0) * speed)
    hours, rest = divmod(seconds, 3600)
    minutes, secs = divmod(rest, 60)
    return "%02d:%02d:%02d" % (hours, minutes, secs)
