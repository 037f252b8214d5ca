"""Eval-mode forward (core/infer.py:62-94) on one GPU: ordinary kernels vs nn.fused_inference() (BN folded into the
convolutions, PReLU in the conv epilogue -- SURVEY 8 f4).   python tools/bench_infer.py [--size 128] [--batch 1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--model", default="VNet")
    a = ap.parse_args()
    from medicalseg_amd import models, nn
    from medicalseg_amd.core import infer
    from medicalseg_amd.device import get_device, to_tensor
    dev = get_device()
    model = getattr(models, a.model)(num_classes=3)
    model.eval()
    x = np.random.default_rng(0).standard_normal((a.batch, 1, a.size, a.size, a.size)).astype(np.float32)
    vox = a.batch * a.size ** 3

    def plain():
        xt = to_tensor(x)
        logits = model(xt)[0]
        ptr = dev.arena.alloc(logits.voxels * 4)
        dev.call("msk_argmax_c", logits.msk(), ptr)

    def fused():
        infer.inference(model, to_tensor(x))

    for name, fn in (("eval forward + argmax", plain), ("fused inference      ", fused)):
        with nn.fused_inference() if fn is fused else open(os.devnull):
            fn()
            dev.sync()
            dev.timer_start()
            for _ in range(a.iters):
                fn()
            ms = dev.timer_stop() / a.iters
        print(f"{a.model} {a.batch}x{a.size}^3 {name}: {ms:7.3f} ms  {vox / ms / 1e3:7.1f} M voxels/s")


if __name__ == "__main__":
    main()
