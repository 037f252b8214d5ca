"""Round-5 verdict item 3 ("overlap the two halves of the step"): what does the chip give when a matrix-bound chain and an
HBM-bound chain run CONCURRENTLY on two streams, against the same two chains one after the other?
  M = one LUConv forward at 32 channels, 2 x 128^3 (input transform + matrix stage: 1.6-1.8 ms, matrix-bound)
  H = BatchNorm-apply + PReLU passes over a 32-channel 2 x 128^3 tensor (read 537 MB + write 537 MB each: HBM-bound)
Prints wall time of k x M alone, k x H alone (H sized to about M's duration), and both at once (two contexts = two streams on
GPU 0), plus rocm-smi's clock / power while each runs.   python tools/probes/overlap_probe.py"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=10).stdout
        sclk = [l for l in out.splitlines() if "sclk" in l]
        pw = [l for l in out.splitlines() if "Power (W)" in l]
        return (sclk[0].split("(")[-1].split(")")[0] if sclk else "?") + " / " + (pw[0].split(":")[-1].strip() + " W" if pw else "?")
    except Exception as e:
        return repr(e)


def main():
    from medicalseg_amd._lib import MskConvDesc, NULL_TENSOR
    from medicalseg_amd.device import Device, Tensor
    da, db = Device(0), Device(0)
    n, s, c, k = 2, 128, 32, 5
    vox = n * s ** 3
    rng = np.random.default_rng(0)

    def mk(dev, fill=True):
        t = Tensor(dev, dev.malloc(vox * c * 4), n, s, s, s, c, c, None)
        if fill:
            dev.h2d(t.ptr, rng.standard_normal(vox * c, dtype=np.float32))
        return t
    x, y = mk(da), mk(da, False)
    w = da.malloc(c * c * k ** 3 * 4)
    da.h2d(w, (rng.standard_normal(c * c * k ** 3) * 0.01).astype(np.float32))
    b = da.small(c)
    cd = MskConvDesc(k, k, k, 1, 1, 1, 2, 2, 2)
    hx, hy = mk(db), mk(db, False)
    scale, shift, alpha = db.small(c), db.small(c), db.small(c)
    db.h2d(scale, np.ones(c, np.float32)); db.h2d(alpha, np.full(c, 0.25, np.float32))
    vp = C.c_void_p

    def M():
        da.call("msk_conv3d_fwd", cd, x.msk(), vp(w), vp(b), y.msk())

    def H():
        db.call("msk_affine_act_fwd", hx.msk(), vp(scale), vp(shift), NULL_TENSOR, vp(alpha), hy.msk())

    for _ in range(3):
        M(); H()
    da.sync(); db.sync()

    def timed(fn_list, reps, label):
        for d in (da, db):
            d.sync()
        samples = []
        stop = threading.Event()

        def sampler():
            while not stop.is_set():
                samples.append(smi())
                time.sleep(0.3)
        th = threading.Thread(target=sampler)
        t0 = time.perf_counter()
        th.start()
        for _ in range(reps):
            for fn in fn_list:
                fn()
        da.sync(); db.sync()
        dt = time.perf_counter() - t0
        stop.set(); th.join()
        mid = samples[len(samples) // 2] if samples else "?"
        print("%-34s %8.3f ms per iteration   (sclk / power mid-run: %s)" % (label, dt / reps * 1e3, mid), flush=True)
        return dt / reps * 1e3

    # size H to M's duration
    tm = timed([M], 200, "M alone (LUConv fwd 32ch 2x128^3)")
    th1 = timed([H], 200, "H alone (one BN-apply+PReLU pass)")
    nh = max(1, int(round(tm / th1)))
    Hn = lambda: [H() for _ in range(nh)]
    th = timed([Hn], 200, "H x %d alone" % nh)
    tb = timed([M, Hn], 200, "M and H x %d concurrently" % nh)
    print("sum of the two alone %.3f ms, concurrent %.3f ms -> overlap hides %.0f %% of the shorter chain"
          % (tm + th, tb, 100.0 * (tm + th - tb) / min(tm, th)))
    try:
        print(subprocess.run(["rocm-smi", "--showmaxpower", "--showsclkrange", "-d", "0"], capture_output=True, text=True, timeout=10).stdout)
    except Exception as e:
        print(repr(e))


if __name__ == "__main__":
    main()
