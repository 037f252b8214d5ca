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
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="msk_set_option knob for experiments")
    ap.add_argument("--eager-opt", type=int, default=1, choices=(0, 1), help="optimizer update per block during backward (Momentum.enable_eager; models with block hooks)")
    ap.add_argument("--inloop-preprocess", action="store_true",
                    help="BASELINE configs[4] as worded: every step takes a RAW sample from host memory (2x the model's in-plane "
                         "size, e.g. 1008x1008x12 MRI: pinned H2D -> normalize(0, 2650) -> resample(order 1) -> max-normalise; label "
                         "resample(order 0); tools/prepare_mri_spine_seg.py:71-80) on a second stream, one sample ahead of the "
                         "training step; reports ms/step with and without (batch 1)")
    ap.add_argument("--profile-out", default=None)
    ap.add_argument("--json-out", default=None, help="write ms/step, algorithmic FLOP rates and the roofline fraction here")
    a = ap.parse_args()
    from medicalseg_amd import models, optimizer as optim
    from medicalseg_amd.device import get_device, to_tensor
    from medicalseg_amd.utils import loss_computation
    shape = tuple(int(v) for v in a.shape.split(","))
    K = [[2, 2, 2]] * 4 if a.iso else [[2, 2, 4], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
    S = [[2, 2, 2]] * 4 if a.iso else [[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]]
    dev = get_device()
    for kv in a.opt:
        key, val = kv.split("=")
        dev.set_option(key, int(val))
    if a.model == "UNet3D":   # builder-defined (no reference model): python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2
        model = models.UNet3D(num_classes=a.num_classes, base_channels=32, depth=4, precision=a.precision)
    else:
        model = getattr(models, a.model)(num_classes=a.num_classes, kernel_size=K, stride_size=S)
    model.train()
    n_out = getattr(model, "num_outputs", 1)
    losses = {"types": [models.MixedLoss([models.CrossEntropyLoss(), models.DiceLoss()], [1, 1]) for _ in range(n_out)],
              "coef": [1.0 / n_out] * n_out}
    opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    eager = bool(a.eager_opt) and opt.enable_eager(model)     # as core.train() / bench.py at one rank: update + re-pack per block on the weight-gradient stream
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
    inloop = None
    if a.inloop_preprocess:
        inloop = run_inloop(a, dev, model, losses, opt, shape, step)
    from medicalseg_amd import nn as _nn
    _nn.FLOPS.update(on=True, same_k5=0.0, same_k3=0.0, other=0.0)
    step()                                   # one (untimed) step with the host-side FLOP accounting on
    _nn.FLOPS["on"] = False
    dev.sync()
    if a.profile_out:
        dev.set_option("prof_shapes", 1)
        dev.set_option("wgrad_async", 0)      # one stream: a kernel's HIP-event time is its own (as bench.py's serialized pass)
        dev.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dev.sync()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    vox = a.batch * shape[0] * shape[1] * shape[2]
    print(f"{a.model} {shape} ncls={a.num_classes} batch={a.batch}: {ms:.2f} ms/step, {vox / ms / 1e3:.2f} M voxels/s"
          + (" (per-kernel profiling on: serialised)" if a.profile_out else ""))
    if inloop:
        print(inloop["line"])
    if a.json_out:
        import json
        F = _nn.FLOPS
        alg = F["same_k5"] + F["same_k3"] + F["other"]
        # time the EXECUTED matrix work needs at the peaks: 'same' 5^3 / 3^3 convolutions on the 16-bit pipe (Winograd F(4,5):
        # 0.4 of the MACs, F(4,3): 0.5; x3 products per fp32 product with two-piece operands, x1 with fp16 operands), everything
        # else at the fp32 MFMA peak
        prod3 = 1.0 if (a.model == "UNet3D" and a.precision == "fp16") else 3.0
        floor_s = (F["same_k5"] * 0.4 * 3.0 + F["same_k3"] * 0.5 * prod3) / 2500e12 + F["other"] / 157.3e12
        res = {"workload": f"{a.model} {shape} ncls={a.num_classes} batch={a.batch} precision={a.precision}", "eager_optimizer": eager,
               "ms_per_step": round(ms, 3), "voxels_per_s": round(vox / ms * 1e3, 1),
               "algorithmic_flop_per_step": alg, "algorithmic_tflops": round(alg / ms / 1e9, 1),
               "algorithmic_speedup_vs_fp32_mfma_peak": round(alg / ms / 1e9 / 157.3, 3),
               "inloop_preprocess": inloop and inloop["json"],
               "roofline": {"bound": "mfma", "frac": round(floor_s / (ms * 1e-3), 4),
                            "definition": "time the executed matrix work of the step needs at the hardware peaks (16-bit pipe 2500 "
                                          "TFLOP/s for the Winograd-pipeline convolutions, fp32 MFMA 157.3 TFLOP/s for the rest) / "
                                          "measured step time -- the step_executed_frac of bench.py",
                            "flop_same_k5": F["same_k5"], "flop_same_k3": F["same_k3"], "flop_other": F["other"]}}
        os.makedirs(os.path.dirname(os.path.abspath(a.json_out)) or ".", exist_ok=True)
        with open(a.json_out, "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    if a.profile_out:
        prof = dev.prof_report()
        total = sum(ms_ for _, ms_ in prof.values())
        with open(a.profile_out, "w") as f:
            f.write("# per-kernel HIP-event time (%d steps)\n# tag\tcalls\ttotal_ms\tavg_ms\tshare\n" % a.steps)
            for tag, (c, ms_) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                f.write("%s\t%d\t%.3f\t%.4f\t%.4f\n" % (tag, c, ms_, ms_ / max(c, 1), ms_ / max(total, 1e-9)))


def run_inloop(a, dev, model, losses, opt, shape, plain_step):
    """configs[4]: in-loop preprocessing one sample ahead of the step, on a second context (= a second stream) of the same GPU.
    Per iteration i (no host synchronisation with the training stream):
        pre.wait_for(dev)        the buffers sample i+1 is written into were last read by step i-1
        sample i+1: pinned H2D -> normalize -> resample(order 1) -> max-normalise ; label: H2D -> resample(order 0)   [stream 2]
        dev.wait_for(pre) was issued BEFORE step i started for sample i;  step i runs beside the preprocessing       [stream 1]"""
    import numpy as np
    from medicalseg_amd.device import Device
    from medicalseg_amd.preprocess import DevicePipeline
    from medicalseg_amd.utils import loss_computation
    if a.batch != 1:
        raise SystemExit("--inloop-preprocess: batch 1 (the MRI config, vnet_mri_spine_seg_512_512_12_15k.yml)")
    raw_shape = (2 * shape[0] - 16, 2 * shape[1] - 16, shape[2]) if shape[0] >= 64 else (2 * shape[0], 2 * shape[1], shape[2])
    rng = np.random.default_rng(1)
    import ctypes as C
    pre = Device(dev.index)
    pipe_x, pipe_y = DevicePipeline(pre, pooled=True), DevicePipeline(pre, pooled=True)
    # the raw samples live in PINNED host memory (a loader that reads files into pinned buffers): the copy is one DMA, no staging
    nraw = int(np.prod(raw_shape))
    raws = []
    for _ in range(2):
        pi, pl = C.c_void_p(), C.c_void_p()
        pre.call("msk_pinned_alloc", C.c_size_t(nraw * 4), C.byref(pi))
        pre.call("msk_pinned_alloc", C.c_size_t(nraw * 4), C.byref(pl))
        np.ctypeslib.as_array((C.c_float * nraw).from_address(pi.value))[:] = rng.random(nraw, dtype=np.float32) * 2650.0
        np.ctypeslib.as_array((C.c_int32 * nraw).from_address(pl.value))[:] = rng.integers(0, a.num_classes, nraw).astype(np.int32)
        raws.append((pi.value, pl.value))

    def prep(i):
        img, lab = raws[i % 2]
        x = pipe_x.from_pinned(img, raw_shape).normalize(0, 2650).resample(list(shape), 1).max_normalize().tensor()
        y = pipe_y.from_pinned(lab, raw_shape, np.int32).resample(list(shape), 0).int_tensor()
        return x, y

    def run(n):
        cur = prep(0)
        old = None
        for i in range(n):
            dev.wait_for(pre)                      # sample i is ready before step i reads it
            pre.wait_for(dev)                      # step i-1 has finished reading the buffers that go back to the pool now
            if old is not None:
                pipe_x.release(old[0])
                pipe_y.release(old[1])
            nxt = prep(i + 1)                      # enqueued BEFORE step i: runs beside it
            x, y = cur
            x.dev = y.dev = dev                    # same GPU, the training context's stream from here on
            ll, _ = loss_computation(model(x), y, losses)
            sum(ll).backward()
            opt.step()
            model.clear_gradients()
            old, cur = cur, nxt
        dev.sync()
        pre.sync()

    run(a.warmup + 1)
    t0 = time.perf_counter()
    run(a.steps)
    ms_in = (time.perf_counter() - t0) * 1e3 / a.steps
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plain_step()
    dev.sync()
    ms_plain = (time.perf_counter() - t0) * 1e3 / a.steps
    # the preprocessing alone (its own stream, nothing beside it)
    pre.sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        s = prep(i)
        pipe_x.release(s[0])
        pipe_y.release(s[1])
    pre.sync()
    ms_pre = (time.perf_counter() - t0) * 1e3 / a.steps
    raw_mb = (np.prod(raw_shape) * 8) / 1e6
    j = {"raw_shape": list(raw_shape), "model_shape": list(shape), "ms_per_step_with_inloop_preprocess": round(ms_in, 3),
         "ms_per_step_resident_input": round(ms_plain, 3), "preprocess_alone_ms_per_sample": round(ms_pre, 3),
         "raw_MB_per_sample_image_plus_label": round(raw_mb, 1),
         "pipeline": "raw sample in pinned host memory -> H2D (one DMA) -> msk_minmax_norm(0, 2650) -> msk_resample3d(order 1) -> msk_max_norm ; label: pinned H2D -> "
                     "msk_resample3d(order 0); second context/stream, one sample ahead, handed over with msk_ctx_wait"}
    line = ("in-loop preprocessing %s -> %s: %.2f ms/step (resident input: %.2f ms/step; the preprocessing alone: %.2f ms per sample)"
            % (raw_shape, shape, ms_in, ms_plain, ms_pre))
    return {"json": j, "line": line}


if __name__ == "__main__":
    main()
