"""Micro-benchmark of the kernel == stride (2x2x2) down / up convolutions of VNet (vnet.py:67,108) through the C ABI:
every op of the four levels with its algorithmic HBM bytes and the rate they imply.

    python tools/bench_ks.py [--n 2] [--size 128] [--iters 10] [--opt KEY=INT]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--levels", type=int, default=2, help="how many of the four VNet levels (finest first)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT")
    ap.add_argument("--fine-ld", type=int, default=0, help="voxel stride (floats) of the 16- / 32-channel fine tensors: 2x their channels = "
                    "the zero-copy concat slices of the real step")
    ap.add_argument("--sets", type=int, default=1, help="rotate over this many copies of every tensor (cold LLC: 4 sets of the 128^3 level "
                    "exceed the 256 MB MALL)")
    a = ap.parse_args()
    from medicalseg_amd._lib import MskConvDesc
    from medicalseg_amd.device import Tensor, get_device
    dev = get_device()
    dev.set_option("wgrad_async", 0)
    for kv in a.opt:
        key, val = kv.split("=")
        dev.set_option(key, int(val))
    rng = np.random.default_rng(0)
    vp = C.c_void_p
    cd = MskConvDesc(2, 2, 2, 2, 2, 2, 0, 0, 0)
    n = a.n

    class Rot:
        """a.sets copies of one tensor; .msk() hands out the next one"""

        def __init__(self, ts):
            self.ts, self.i = ts, 0
            self.ptr = ts[0].ptr

        def msk(self):
            self.i = (self.i + 1) % len(self.ts)
            return self.ts[self.i].msk()

    def mk(s, ch, fill=True, fine=False):
        ld = a.fine_ld * ch // 16 if (fine and a.fine_ld) else ch
        ts = []
        for _ in range(a.sets):
            t = Tensor(dev, dev.malloc(n * s ** 3 * ld * 4), n, s, s, s, ch, ld, None)
            if fill:
                dev.h2d(t.ptr, rng.standard_normal(n * s ** 3 * ld, dtype=np.float32))
            ts.append(t)
        return Rot(ts)

    def run(name, fn, nbytes):
        fn()
        dev.sync()
        dev.timer_start()
        for _ in range(a.iters):
            fn()
        ms = dev.timer_stop() / a.iters
        dev.set_option("prof_only_halo", 0)
        dev.prof_reset()
        dev.prof_enable(True)
        fn()
        dev.sync()
        dev.prof_enable(False)
        tags = " + ".join(f"{t} {v[1]:.3f}" for t, v in sorted(dev.prof_report().items(), key=lambda kv: -kv[1][1]))
        print(f"{name:44s} {ms:7.3f} ms  {nbytes / 1e6:7.1f} MB  {nbytes / ms / 1e9:6.2f} TB/s   [{tags}]")

    # (fine size, down conv cin -> cout) and (up conv cin -> cout producing the fine size)
    levels = [(a.size, 16, 32, 64, 16), (a.size // 2, 32, 64, 128, 32), (a.size // 4, 64, 128, 256, 64),
              (a.size // 8, 128, 256, 256, 128)][:a.levels]
    for s, dci, dco, uci, uco in levels:
        vf, vc = n * s ** 3, n * (s // 2) ** 3
        # down conv: x[s, dci] -> y[s/2, dco]
        x, y, dx = mk(s, dci, fine=True), mk(s // 2, dco), mk(s, dci, fine=True)
        w = dev.malloc(dci * dco * 8 * 4)
        dev.h2d(w, (rng.standard_normal(dci * dco * 8) * 0.05).astype(np.float32))
        dw, b, db = dev.malloc(dci * dco * 8 * 4), dev.small(dco), dev.small(dco)
        run(f"down fwd   {dci}->{dco} @{s}->{s // 2}", lambda: dev.call("msk_conv3d_fwd", cd, x.msk(), vp(w), vp(b), y.msk()),
            4 * (vf * dci + vc * dco))
        run(f"down dgrad {dco}->{dci} @{s // 2}->{s} acc=1", lambda: dev.call("msk_conv3d_dgrad", cd, y.msk(), vp(w), dx.msk(), 1),
            4 * (vc * dco + 2 * vf * dci))
        run(f"down dgrad {dco}->{dci} @{s // 2}->{s} acc=0", lambda: dev.call("msk_conv3d_dgrad", cd, y.msk(), vp(w), dx.msk(), 0),
            4 * (vc * dco + vf * dci))
        run(f"down wgrad {dci}x{dco}", lambda: dev.call("msk_conv3d_wgrad", cd, x.msk(), y.msk(), vp(dw), vp(db), 0),
            4 * (vf * dci + vc * dco))
        # up conv (transposed): u[s/2, uci] -> v[s, uco]
        u, v, du = mk(s // 2, uci), mk(s, uco, fine=True), mk(s // 2, uci)
        wt = dev.malloc(uci * uco * 8 * 4)
        dev.h2d(wt, (rng.standard_normal(uci * uco * 8) * 0.05).astype(np.float32))
        dwt, bt, dbt = dev.malloc(uci * uco * 8 * 4), dev.small(uco), dev.small(uco)
        run(f"up   fwd   {uci}->{uco} @{s // 2}->{s}", lambda: dev.call("msk_convT3d_fwd", cd, u.msk(), vp(wt), vp(bt), v.msk()),
            4 * (vc * uci + vf * uco))
        run(f"up   dgrad {uco}->{uci} @{s}->{s // 2}", lambda: dev.call("msk_convT3d_dgrad", cd, v.msk(), vp(wt), du.msk(), 0),
            4 * (vf * uco + vc * uci))
        run(f"up   wgrad {uci}x{uco}", lambda: dev.call("msk_convT3d_wgrad", cd, u.msk(), v.msk(), vp(dwt), vp(dbt), 0),
            4 * (vc * uci + vf * uco))
        for r in (x, y, dx, u, v, du):
            for t in r.ts:
                dev.free(t.ptr)


if __name__ == "__main__":
    main()
