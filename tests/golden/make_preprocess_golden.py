"""Generate tests/golden/preprocess_golden.npz by running the REFERENCE's own
tools/preprocess_utils numeric code (numpy/scipy backend) in the build container.

Run from anywhere:  python tests/golden/make_preprocess_golden.py
Needs /root/reference (not available on the GPU box -- only the .npz travels).
Recipe: SURVEY.md Appendix F (stub absent I/O libs, path-only `tools` package,
cwd=/root/reference because tools/preprocess_utils/__init__.py:5 opens a relative
path).  Only inputs and expected outputs are stored; no reference source.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    os.chdir(REF)
    for n in ["SimpleITK", "nibabel", "pydicom", "nrrd", "cv2", "visualdl"]:
        try:
            importlib.import_module(n)
        except Exception:
            sys.modules[n] = types.ModuleType(n)
    t = types.ModuleType("tools")
    t.__path__ = [os.path.join(REF, "tools")]
    sys.modules["tools"] = t
    from tools.preprocess_utils import HUnorm, label_remap, normalize, resample
    return resample, HUnorm, normalize, label_remap


def main():
    resample, HUnorm, normalize, label_remap = load_reference()
    rng = np.random.default_rng(20260928)
    out = {}

    # --- resample: float32 order 1, int32 order 0, several shape pairs (incl. up/down,
    # size-1 output axis, identity axis, anisotropic MRI-like slab)
    cases = [((20, 30, 10), (16, 16, 16)), ((40, 40, 6), (20, 20, 6)),
             ((9, 7, 5), (18, 15, 11)), ((12, 12, 12), (12, 5, 1)),
             ((33, 17, 3), (16, 16, 12))]
    for i, (sin, sout) in enumerate(cases):
        img = (rng.standard_normal(sin) * 400 - 300).astype(np.float32)
        lab = rng.integers(0, 4, sin).astype(np.int32)
        o1, sp1 = resample(img.copy(), spacing=[0.7, 0.8, 2.5], new_shape=list(sout), order=1)
        o0, _ = resample(lab.copy(), new_shape=list(sout), order=0)
        of0, _ = resample(img.copy(), new_shape=list(sout), order=0)
        out[f"rs{i}_img"] = img
        out[f"rs{i}_lab"] = lab
        out[f"rs{i}_shape"] = np.array(sout)
        out[f"rs{i}_o1"] = o1
        out[f"rs{i}_o0"] = o0
        out[f"rs{i}_of0"] = of0
        out[f"rs{i}_spacing"] = np.array(sp1, dtype=np.float64)
    # 4-element spacing drops element 0 (geometry.py:59-60)
    img = rng.standard_normal((8, 8, 8)).astype(np.float32)
    _, sp = resample(img, spacing=[9.0, 1.0, 2.0, 3.0], new_shape=[4, 4, 4], order=1)
    out["rs_sp4"] = np.array(sp, dtype=np.float64)

    # --- HUnorm
    hu = np.array([-3000, -1200, 0, 600, np.nan, 5000, -1199.5, 599.9, 1.0], dtype=np.float32)
    out["hu_in"] = hu
    out["hu_out"] = HUnorm(hu.copy())
    vol = (rng.standard_normal((6, 7, 8)) * 900 - 400).astype(np.float32)
    vol[1, 2, 3] = np.nan
    out["hu_vol_in"] = vol
    out["hu_vol_out"] = HUnorm(vol.copy())
    out["hu_vol_out_custom"] = HUnorm(vol.copy(), HU_min=-1000, HU_max=400, HU_nan=-1500)

    # --- normalize
    v = (rng.random((5, 6, 7)) * 3000 - 200).astype(np.float32)
    out["nm_in"] = v
    out["nm_out_auto"] = normalize(v.copy())
    out["nm_out_bounds"] = normalize(v.copy(), min_val=0, max_val=2650)

    # --- label_remap (sequential semantics: chained keys)
    lab = rng.integers(0, 6, (4, 5, 6)).astype(np.int32)
    out["lr_in"] = lab
    out["lr_out"] = label_remap(lab.copy(), {1: 0, 2: 1, 3: 1, 5: 2})
    out["lr_out_chain"] = label_remap(lab.copy(), {1: 2, 2: 3})

    # --- the two shipped image pipelines end to end
    ct = (rng.standard_normal((24, 20, 9)) * 500 - 500).astype(np.float32)
    out["pipe_ct_in"] = ct
    p = HUnorm(ct.copy())
    p, _ = resample(p, new_shape=[16, 16, 16], order=1)
    out["pipe_ct_out"] = p.astype(np.float32)
    mr = (rng.random((30, 30, 6)) * 2650).astype(np.float32)
    out["pipe_mr_in"] = mr
    q = normalize(mr.copy(), min_val=0, max_val=2650)
    q, _ = resample(q, new_shape=[16, 16, 6], order=1)
    out["pipe_mr_out"] = q.astype(np.float32)

    np.savez_compressed(os.path.join(HERE, "preprocess_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "preprocess_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
