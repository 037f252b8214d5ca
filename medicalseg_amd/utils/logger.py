"""Rank-0 level logger (reference medicalseg/utils/logger.py:20-48 behaviour: prints
`time [LEVEL]\tmessage` on the local master only)."""
import os
import sys
import time

levels = {0: 'ERROR', 1: 'WARNING', 2: 'INFO', 3: 'DEBUG'}
log_level = 2


def _is_master():
    return int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0"))) == 0


def log(level=2, message=""):
    if not _is_master() or log_level < level:
        return
    stamp = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime())
    print("{} [{}]\t{}".format(stamp, levels[level], message).encode("utf-8").decode("latin1"))
    sys.stdout.flush()


def debug(message=""):
    log(level=3, message=message)


def info(message=""):
    log(level=2, message=message)


def warning(message=""):
    log(level=1, message=message)


def error(message=""):
    log(level=0, message=message)
