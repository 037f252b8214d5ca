"""Side measurements of the non-headline workloads (NOT bench.py's contract line): the MRI-slab
VNet and VNetDeepSup training steps at BASELINE's configs[4]-class shapes.  Prints ms/step,
voxels/s and, with --shapes, the per-kernel HIP-event profile.

  python tools/bench_workloads.py --model VNetDeepSup --steps 5
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="VNetDeepSup", choices=["VNet", "VNetDeepSup", "UNet3D"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"], help="UNet3D: fp16 = 3x3x3 convolutions on the fp16 matrix pipe")
    ap.add_argument("--shape", default="512,512,12")
    ap.add_argument("--num-classes", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iso", action="store_true", help="isotropic 2x2x2 kernels/strides (lung) instead of the MRI ones")
    ap.add_argument("--profile-out", default=None)
    a = ap.parse_args()
    from medicalseg_amd import models, optimizer as optim
    from medicalseg_amd.device import get_device, to_tensor
    from medicalseg_amd.utils import loss_computation
    shape = tuple(int(v) for v in a.shape.split(","))
    K = [[2, 2, 2]] * 4 if a.iso else [[2, 2, 4], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
    S = [[2, 2, 2]] * 4 if a.iso else [[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]]
    dev = get_device()
    if a.model == "UNet3D":   # builder-defined (no reference model): python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2
        model = models.UNet3D(num_classes=a.num_classes, base_channels=32, depth=4, precision=a.precision)
    else:
        model = getattr(models, a.model)(num_classes=a.num_classes, kernel_size=K, stride_size=S)
    model.train()
    n_out = getattr(model, "num_outputs", 1)
    losses = {"types": [models.MixedLoss([models.CrossEntropyLoss(), models.DiceLoss()], [1, 1]) for _ in range(n_out)],
              "coef": [1.0 / n_out] * n_out}
    opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(0)
    x = to_tensor(rng.random((a.batch, 1) + shape, dtype=np.float32))
    y = to_tensor(rng.integers(0, a.num_classes, (a.batch,) + shape).astype(np.int32))

    def step():
        ll, _ = loss_computation(model(x), y, losses)
        sum(ll).backward()
        opt.step()
        model.clear_gradients()

    for _ in range(a.warmup):
        step()
    dev.sync()
    if a.profile_out:
        dev.set_option("prof_shapes", 1)
        dev.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dev.sync()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    vox = a.batch * shape[0] * shape[1] * shape[2]
    print(f"{a.model} {shape} ncls={a.num_classes} batch={a.batch}: {ms:.2f} ms/step, {vox / ms / 1e3:.2f} M voxels/s"
          + (" (per-kernel profiling on: serialised)" if a.profile_out else ""))
    if a.profile_out:
        prof = dev.prof_report()
        total = sum(ms_ for _, ms_ in prof.values())
        with open(a.profile_out, "w") as f:
            f.write("# per-kernel HIP-event time (%d steps)\n# tag\tcalls\ttotal_ms\tavg_ms\tshare\n" % a.steps)
            for tag, (c, ms_) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                f.write("%s\t%d\t%.3f\t%.4f\t%.4f\n" % (tag, c, ms_, ms_ / max(c, 1), ms_ / max(total, 1e-9)))


if __name__ == "__main__":
    main()
