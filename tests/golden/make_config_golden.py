"""Capture the reference's RESOLVED configurations as a fixture.

Runs in the build container only (needs /root/reference).  For every YAML under the reference's configs/ tree it stores
  raw      -- the file as yaml.load parses it (data, not text), so the test can rebuild the tree in a temp directory;
  resolved -- `.dic` of the reference's OWN `Config(path)` (medicalseg/cvlibs/config.py:73-126: `_base_` inheritance
              through `_update_dic`, then the data_root check), or the exception type it raises (the MRI model files
              inherit from a file that was never shipped; files without data_root are rejected).
The reference module is loaded from its file with `paddle` and its package siblings stubbed in sys.modules: only the
YAML-merging code runs, nothing is copied.

    python tests/golden/make_config_golden.py        # writes tests/golden/config_golden.json
"""
import importlib.util
import json
import os
import sys
import types
import warnings

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_golden.json")


def load_reference_config_class():
    for name in ("paddle", "medicalseg", "medicalseg.cvlibs", "medicalseg.cvlibs.manager", "medicalseg.utils",
                 "medicalseg.utils.logger"):
        sys.modules.setdefault(name, types.ModuleType(name))
    pd = sys.modules["paddle"]
    pd.optimizer = types.SimpleNamespace(lr=types.SimpleNamespace(LRScheduler=object), Optimizer=object)
    pd.nn = types.SimpleNamespace(Layer=object)
    pd.io = types.SimpleNamespace(Dataset=object)
    sys.modules["medicalseg.cvlibs"].manager = sys.modules["medicalseg.cvlibs.manager"]
    sys.modules["medicalseg.utils"].logger = sys.modules["medicalseg.utils.logger"]
    spec = importlib.util.spec_from_file_location("ref_config", os.path.join(REF, "medicalseg/cvlibs/config.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Config


def main():
    Config = load_reference_config_class()
    root = os.path.join(REF, "configs")
    out = {}
    for dirpath, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if not f.endswith((".yml", ".yaml")):
                continue
            path = os.path.join(dirpath, f)
            rel = os.path.relpath(path, root)
            with open(path, encoding="utf-8") as fh:
                raw = yaml.load(fh, Loader=yaml.FullLoader)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    resolved = {"ok": Config(path).dic}      # the reference's constructor: parse + data_root check
            except Exception as e:  # noqa: BLE001 -- the exception type is part of the fixture
                resolved = {"error": type(e).__name__}
            out[rel] = {"raw": raw, "resolved": resolved}
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", OUT, {k: ("ok" if "ok" in v["resolved"] else v["resolved"]["error"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
