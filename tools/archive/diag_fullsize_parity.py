"""Full-size parity case (tests/test_gpu_fullsize_parity.py) under option overrides: which switch moves which gradient tensor.
Usage: python tools/diag_fullsize_parity.py [case] "opt=val,opt=val" "opt=val" ...   ("-" = product defaults)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import test_gpu_fullsize_parity as T  # noqa: E402
from helpers import dev  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] and sys.argv[1] != "-" else "vnet128"
sets = [a for a in sys.argv[1:] if a != case] or ["-"]
DEFAULTS = {"reduce_vpl": 8, "wbf_tin_groups": -1, "wgrad_renorm": 1, "wgrad_async": 1, "conv_split": 2, "bwd_fuse": -1,
            "wbf_fuse": 1, "late_split": 1}
d = dev()
for s in sets:
    kv = [p.split("=") for p in s.split(",")] if s != "-" else []
    old = {k: (int(v) // 1000 * 1000 if k == "reduce_vpl_site" else DEFAULTS[k]) for k, v in kv}
    for k, v in kv:
        d.set_option(k, int(v))
    try:
        r = T._run_case(case)
    finally:
        for k, v in old.items():
            d.set_option(k, v)
    top = sorted(r["l2s"].items(), key=lambda kv_: -kv_[1])[:6]
    print("[%s] median %.2e | " % (s, r["med"]) + " | ".join("%s %.2e" % kv_ for kv_ in top), flush=True)
