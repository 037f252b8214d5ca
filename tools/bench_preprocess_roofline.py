#!/usr/bin/env python
"""Per-kernel bandwidth of the preprocessing path (reference tools/preprocess_utils/geometry.py:31-69,
values.py:54-87, tools/prepare_mri_spine_seg.py:71-80, tools/prepare_lung_coronavirus.py:81-90) against the HBM roof, and the
pinned host -> device copy against the link:
    python tools/bench_preprocess_roofline.py [--out profiles/r05_preprocess.json]
Every kernel is timed with HIP events (msk_timer_*) over `reps` back-to-back launches on resident data; bytes = what the
algorithm must move (each source voxel read once, each destination voxel written once)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md
HBM_STREAM_GBPS = 6300.0        # what a streaming kernel reaches (same guide)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from medicalseg_amd.device import get_device
    dev = get_device()
    rng = np.random.default_rng(0)
    rows = []

    def timed(fn, reps=a.reps):
        fn()
        dev.sync()
        dev.timer_start()
        for _ in range(reps):
            fn()
        return dev.timer_stop() / reps

    def row(name, ms, nbytes, note=""):
        gbps = nbytes / (ms * 1e-3) / 1e9
        rows.append({"kernel": name, "ms": round(ms, 4), "MB": round(nbytes / 1e6, 1), "GBps": round(gbps, 1),
                     "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 3), "frac_of_streaming_rate": round(gbps / HBM_STREAM_GBPS, 3),
                     "note": note})
        print("%-46s %8.3f ms %9.1f MB %8.1f GB/s  %s" % (name, ms, nbytes / 1e6, gbps, note))

    # --- MRI: 1008 x 1008 x 12 raw -> 512 x 512 x 12 (prepare_mri_spine_seg.py:71-80) -------------------------------------------
    src_shape, dst_shape = (1008, 1008, 12), (512, 512, 12)
    nsrc, ndst = int(np.prod(src_shape)), int(np.prod(dst_shape))
    raw = (rng.random(src_shape, dtype=np.float32) * 2650.0)
    sp, dp = dev.malloc(nsrc * 4), dev.malloc(ndst * 4)
    dev.h2d(sp, raw)
    row("msk_minmax_norm 1008x1008x12 (in place)", timed(lambda: dev.call("msk_minmax_norm", C.c_void_p(sp), C.c_void_p(sp), C.c_size_t(nsrc), 1, C.c_float(0.0), C.c_float(2650.0))), 2 * nsrc * 4)
    row("msk_resample3d order 1 1008x1008x12 -> 512x512x12", timed(lambda: dev.call("msk_resample3d", C.c_void_p(sp), *src_shape, C.c_void_p(dp), *dst_shape, 1, 0)),
        (nsrc + ndst) * 4, "bytes = source once + destination once; coordinates in double (bit-level scipy agreement)")
    row("msk_resample3d order 0 (labels, int32)", timed(lambda: dev.call("msk_resample3d", C.c_void_p(sp), *src_shape, C.c_void_p(dp), *dst_shape, 0, 1)),
        (nsrc // 4 + ndst) * 4, "nearest: one source voxel in 4 is touched (2x down-sampling in-plane) -- whole lines are still fetched")
    row("msk_max_norm 512x512x12 (two passes: max, scale)", timed(lambda: dev.call("msk_max_norm", C.c_void_p(dp), C.c_void_p(dp), C.c_size_t(ndst))), 3 * ndst * 4)
    dev.free(sp)
    dev.free(dp)
    # --- CT: 512 x 512 x 300 raw -> 128^3 (prepare_lung_coronavirus.py:81-90) ---------------------------------------------------
    src_shape, dst_shape = (300, 512, 512), (128, 128, 128)
    nsrc, ndst = int(np.prod(src_shape)), int(np.prod(dst_shape))
    ct = np.clip(rng.standard_normal(src_shape, dtype=np.float32) * 450 - 600, -2000, 2000)
    sp, dp = dev.malloc(nsrc * 4), dev.malloc(ndst * 4)
    dev.h2d(sp, ct)
    row("msk_hu_norm 300x512x512 (in place)", timed(lambda: dev.call("msk_hu_norm", C.c_void_p(sp), C.c_void_p(sp), C.c_size_t(nsrc), C.c_float(-1200), C.c_float(600), C.c_float(-2000))), 2 * nsrc * 4)
    row("msk_resample3d order 1 300x512x512 -> 128^3", timed(lambda: dev.call("msk_resample3d", C.c_void_p(sp), *src_shape, C.c_void_p(dp), *dst_shape, 1, 0)),
        (ndst * 8 + ndst) * 4, "4x / 4x / 2.3x down-sampling: the 8 corners of a destination voxel are all it needs -- bytes = 8 source voxels + 1 destination voxel per output (the whole source would be 315 MB)")
    # --- pinned host -> device ----------------------------------------------------------------------------------------------------
    p = C.c_void_p()
    nb = nsrc * 4
    dev.call("msk_pinned_alloc", C.c_size_t(nb), C.byref(p))
    C.memmove(p.value, ct.ctypes.data, nb)
    dev.sync()

    def h2d():
        dev.call("msk_h2d_async", C.c_void_p(sp), C.c_void_p(p.value), C.c_size_t(nb))
    ms = timed(h2d, reps=5)
    rows.append({"kernel": "pinned H2D 300x512x512 fp32", "ms": round(ms, 3), "MB": round(nb / 1e6, 1), "GBps": round(nb / ms / 1e6, 1),
                 "note": "host link (PCIe 5 x16: 64 GB/s per direction nominal); the in-loop pipeline's floor for raw volumes that arrive from the host"})
    print("%-46s %8.3f ms %9.1f MB %8.1f GB/s" % ("pinned H2D", ms, nb / 1e6, nb / ms / 1e6))
    t0 = time.perf_counter()
    C.memmove(p.value, ct.ctypes.data, nb)
    rows.append({"kernel": "host memmove into the pinned staging buffer", "ms": round((time.perf_counter() - t0) * 1e3, 3), "MB": round(nb / 1e6, 1)})
    dev.call("msk_pinned_free", p)
    dev.free(sp)
    dev.free(dp)
    out = {"hbm_peak_GBps": HBM_PEAK_GBPS, "hbm_streaming_GBps": HBM_STREAM_GBPS, "rows": rows,
           "how": "HIP events around %d back-to-back launches on resident data (tools/bench_preprocess_roofline.py)" % a.reps}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
