"""Console logger of the train/eval loops: `YYYY-mm-dd HH:MM:SS [LEVEL]<TAB>message`, printed by
the master rank only and filtered by `log_level` -- the line format and the four entry points
(debug/info/warning/error) of the reference's utils/logger.py, which the reference's log
parsers and users' grep patterns rely on.  The rank comes from the launcher's environment
(torch.distributed.run exports RANK) instead of paddle's ParallelEnv."""
import os
import sys
from datetime import datetime

ERROR, WARNING, INFO, DEBUG = range(4)
levels = {ERROR: 'ERROR', WARNING: 'WARNING', INFO: 'INFO', DEBUG: 'DEBUG'}
log_level = INFO  # messages above this verbosity are dropped
stream = None     # None = sys.stdout (the reference prints its log there); bench.py points it at stderr: its stdout carries ONE JSON line


def _master_rank() -> bool:
    env = os.environ
    return int(env.get("RANK", env.get("LOCAL_RANK", "0"))) == 0


def log(level=INFO, message=""):
    if level > log_level or not _master_rank():
        return
    stamp = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    out = stream if stream is not None else sys.stdout
    out.write("%s [%s]\t%s\n" % (stamp, levels[level], message))
    out.flush()


def _at(level):
    def emit(message=""):
        log(level, message)
    emit.__name__ = levels[level].lower()
    return emit


debug, info, warning, error = _at(DEBUG), _at(INFO), _at(WARNING), _at(ERROR)
