#!/usr/bin/env python
"""Preprocessing throughput: the reference's CT pipeline (HUnorm + resample to 128^3, order 1;
tools/prepare_lung_coronavirus.py:81-90) and MRI pipeline (normalize(0,2650) + resample to
512x512x12; tools/prepare_mri_spine_seg.py:71-80) on the device (pinned H2D + HIP kernels) vs the
same arithmetic with scipy.ndimage.zoom on the host (what the reference's numpy backend runs).
The reference's published figure (README.md:55-58: 20 CT scans, 50.7 s numpy / 31.4 s CuPy) includes
file I/O and is not reproducible here; this prints kernel-level and end-to-end-in-memory times."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def host_ct(raw):
    import scipy.ndimage
    x = np.nan_to_num(raw.copy(), nan=-2000)
    x = (x + 1200) / (1800 / 255)
    np.clip(x, 0, 255, out=x)
    return scipy.ndimage.zoom(x, np.array([128, 128, 128]) / np.array(x.shape), mode="nearest", order=1)


def main():
    from medicalseg_amd.device import get_device
    from medicalseg_amd.preprocess import DevicePipeline
    dev = get_device()
    rng = np.random.default_rng(0)
    out = {}
    ct = np.clip(rng.standard_normal((300, 512, 512), dtype=np.float32) * 450 - 600, -2000, 2000)
    pipe = DevicePipeline()
    pipe.image(ct).HUnorm().resample([128, 128, 128], 1).max_normalize().numpy()  # warm-up
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ch = pipe.image(ct).HUnorm().resample([128, 128, 128], 1).max_normalize()
    dev.sync()
    out["ct_512x512x300_to_128^3_device_ms_incl_pinned_h2d"] = (time.perf_counter() - t0) / reps * 1e3
    vol = pipe.image(ct)
    dev.sync()
    dev.timer_start()
    vol.HUnorm().resample([128, 128, 128], 1).max_normalize()
    out["ct_kernels_only_ms"] = dev.timer_stop()
    t0 = time.perf_counter()
    ref = host_ct(ct)
    out["ct_host_scipy_ms"] = (time.perf_counter() - t0) * 1e3
    got = pipe.image(ct).HUnorm().resample([128, 128, 128], 1).numpy()
    out["ct_max_abs_diff_vs_host"] = float(np.abs(got - ref).max())
    mr = (rng.random((1008, 1008, 12)) * 2650).astype(np.float32)
    pipe.image(mr).normalize(0, 2650).resample([512, 512, 12], 1).numpy()
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe.image(mr).normalize(0, 2650).resample([512, 512, 12], 1).max_normalize()
    dev.sync()
    out["mri_1008x1008x12_to_512x512x12_device_ms_incl_pinned_h2d"] = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
