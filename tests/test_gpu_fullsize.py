"""Full-size (BASELINE.json configs) checks on the GPU through properties that need no full
oracle run: sampled-voxel comparison against direct numpy dot products, linearity of the
convolution, bitwise run-to-run determinism of a whole training step, finite losses and shapes
for VNet 128^3 batch 2 and the anisotropic MRI slab 512x512x12 with 20 classes."""
import ctypes as C

import numpy as np
import pytest

from helpers import dev, t_empty, t_from_ncdhw, vec, vec_back, vp

pytestmark = pytest.mark.gpu


def _desc(k, s, p):
    from medicalseg_amd._lib import MskConvDesc
    return MskConvDesc(*k, *s, *p)


def test_conv5_128cubed_sampled_and_linear():
    d = dev()
    rng = np.random.default_rng(0)
    S, Cc = 128, 32
    x1 = rng.standard_normal((1, Cc, S, S, S), dtype=np.float32)
    x2 = rng.standard_normal((1, Cc, S, S, S), dtype=np.float32)
    w = (rng.standard_normal((Cc, Cc, 5, 5, 5)) / np.sqrt(Cc * 125)).astype(np.float32)
    b = rng.standard_normal(Cc).astype(np.float32)
    wp, bp = vec(w.ravel()), vec(b)
    desc = _desc((5,) * 3, (1,) * 3, (2,) * 3)

    def conv(x, bias=True):
        xt, yt = t_from_ncdhw(x), t_empty(1, Cc, S, S, S)
        d.call("msk_conv3d_fwd", desc, xt.msk(), vp(wp), vp(bp) if bias else None, yt.msk())
        out = yt.numpy()
        d.free(xt.ptr)
        d.free(yt.ptr)
        return out

    y1 = conv(x1)
    # sampled voxels (corners/edges included) against float64 dot products
    xp = np.pad(x1[0].astype(np.float64), ((0, 0), (2, 2), (2, 2), (2, 2)))
    pts = [(0, 0, 0), (127, 127, 127), (0, 127, 5), (64, 0, 127)] + [tuple(rng.integers(0, S, 3)) for _ in range(40)]
    w64 = w.astype(np.float64)
    for (z, yy, xx) in pts:
        patch = xp[:, z:z + 5, yy:yy + 5, xx:xx + 5]
        ref = np.tensordot(w64, patch, axes=([1, 2, 3, 4], [0, 1, 2, 3])) + b
        assert np.abs(y1[0, :, z, yy, xx] - ref).max() < 5e-5 * (np.abs(ref).max() + 1)
    # linearity: conv(2*x1 - 3*x2) == 2*conv(x1) - 3*conv(x2)   (bias-free)
    ya, yb = conv(x1, False), conv(x2, False)
    yc = conv(2 * x1 - 3 * x2, False)
    assert np.abs(yc - (2 * ya - 3 * yb)).max() < 2e-4 * np.abs(yc).max()


def test_wgrad_128cubed_sampled():
    d = dev()
    rng = np.random.default_rng(1)
    S, Cc = 128, 32
    x = rng.standard_normal((1, Cc, S, S, S), dtype=np.float32)
    dy = rng.standard_normal((1, Cc, S, S, S), dtype=np.float32)
    xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
    dw, db = vec(np.zeros(Cc * Cc * 125)), vec(np.zeros(Cc))
    d.call("msk_conv3d_wgrad", _desc((5,) * 3, (1,) * 3, (2,) * 3), xt.msk(), dyt.msk(), vp(dw), vp(db), 0)
    got = vec_back(dw, Cc * Cc * 125).reshape(Cc, Cc, 5, 5, 5)
    xp = np.pad(x[0], ((0, 0), (2, 2), (2, 2), (2, 2)))
    for _ in range(12):
        co, ci = rng.integers(0, Cc, 2)
        a, b_, c = rng.integers(0, 5, 3)
        ref = np.dot(dy[0, co].astype(np.float64).ravel(), xp[ci, a:a + S, b_:b_ + S, c:c + S].astype(np.float64).ravel())
        assert abs(got[co, ci, a, b_, c] - ref) < 2e-4 * np.sqrt(S ** 3), (co, ci, a, b_, c)
    assert np.abs(vec_back(db, Cc) - dy.sum(axis=(0, 2, 3, 4))).max() < 1e-3 * np.sqrt(S ** 3)


def _train_step(model, opt, losses, x, y):
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.utils import loss_computation
    ll, per = loss_computation(model(x), to_tensor(y), losses)
    loss = sum(ll)
    loss.backward()
    opt.step()
    model.clear_gradients()
    return float(loss), np.asarray(per).copy()


def test_vnet_128_batch2_step_is_finite_and_bitwise_deterministic():
    from medicalseg_amd import nn
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.datasets import SyntheticCT
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    ds = SyntheticCT(num_samples=2, shape=(128, 128, 128), num_classes=3)
    items = [ds[i] for i in range(2)]
    x, y = np.stack([i[0] for i in items]), np.stack([i[1] for i in items])
    results = []
    for run in range(2):
        nn.seed(0)
        nn.Dropout3D.step, nn.Dropout3D.seed = 0, 0
        model = VNet(num_classes=3)
        model.train()
        opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
        losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
        sites = {k: l.site for k, l in model.dropout_layers().items()}
        base = min(sites.values())
        for k, l in model.dropout_layers().items():      # same RNG sites in both runs
            l.site = l.site - base + 1
        l1, per1 = _train_step(model, opt, losses, x, y)
        l2, per2 = _train_step(model, opt, losses, x, y)
        sd = model.state_dict()
        results.append((l1, l2, per1, sd["up_tr32.ops.0.conv1.weight"], sd["in_tr.bn1._mean"], sd["down_tr256.ops.1.conv1.weight"]))
        assert np.isfinite(l1) and np.isfinite(l2) and 0 < l1 < 10 and per1.shape == (3,)
        assert np.all(per1 >= 0) and np.all(per1 <= 1)
        assert np.abs(sd["in_tr.bn1._mean"]).max() > 0          # running stats moved
    a, b = results
    assert a[0] == b[0] and a[1] == b[1]
    for u, v in zip(a[2:], b[2:]):
        assert np.array_equal(u, v)                              # bitwise reproducible step


def test_vnet_mri_512x512x12_20_classes_runs():
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    K = [[2, 2, 4], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
    S = [[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]]
    model = VNet(num_classes=20, kernel_size=K, stride_size=S)
    model.train()
    opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    rng = np.random.default_rng(0)
    x = rng.random((1, 1, 512, 512, 12), dtype=np.float32)
    y = rng.integers(0, 20, (1, 512, 512, 12)).astype(np.int32)
    loss, per = _train_step(model, opt, losses, x, y)
    assert [a.shape for a in model._acts] == [(1, 16, 512, 512, 12), (1, 32, 256, 256, 9), (1, 64, 128, 128, 8),
                                              (1, 128, 64, 64, 4), (1, 256, 32, 32, 2)]
    assert np.isfinite(loss) and per.shape == (20,) and np.all(np.isfinite(per))


def test_vnetdeepsup_mri_512x512x12_step_deterministic():
    """BASELINE's VNetDeepSup config at full size (vnetdeepsup_mri_spine_seg_512_512_12_15k.yml):
    four 20-class outputs at 512x512x12; the step must be finite and bitwise reproducible
    (the resize adjoint is a fixed-order gather, no atomics)."""
    from medicalseg_amd import nn
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNetDeepSup
    K = [[2, 2, 4], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
    S = [[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]]
    rng = np.random.default_rng(0)
    x = rng.random((1, 1, 512, 512, 12), dtype=np.float32)
    y = rng.integers(0, 20, (1, 512, 512, 12)).astype(np.int32)
    runs = []
    for run in range(2):
        nn.seed(0)
        nn.Dropout3D.step, nn.Dropout3D.seed = 0, 0
        model = VNetDeepSup(num_classes=20, kernel_size=K, stride_size=S)
        model.train()
        base = min(l.site for l in model.dropout_layers().values())
        for l in model.dropout_layers().values():
            l.site = l.site - base + 1
        opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
        losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1]) for _ in range(4)], "coef": [0.25] * 4}
        l1, per = _train_step(model, opt, losses, x, y)
        l2, _ = _train_step(model, opt, losses, x, y)
        assert [h.shape for h in model._heads] == [(1, 20, 64, 64, 4), (1, 20, 128, 128, 8), (1, 20, 256, 256, 9)]
        assert np.isfinite(l1) and np.isfinite(l2) and per.shape == (20,)
        sd = model.state_dict()
        runs.append((l1, l2, sd["out_tr256.weight"], sd["out_tr64.weight"], sd["up_tr256.ops.0.conv1.weight"]))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    for u, v in zip(runs[0][2:], runs[1][2:]):
        assert np.array_equal(u, v)


def test_conv_beyond_4gib_tensors_is_chunked_per_sample():
    """288 GB of HBM invites batches whose activations exceed 4 GiB (32ch @ 128^3: N >= 17), beyond the 32-bit byte
    offsets some kernels use: the dispatch cuts the batch into whole-sample chunks.  Every sample carries the SAME
    volume here, so every output sample must equal sample 0 bit for bit (also across the chunk boundary), sample 0 is
    spot-checked against numpy, and the weight gradient must be N x the single-sample one."""
    from medicalseg_amd.device import Tensor
    d = dev()
    S, Cc, N = 128, 32, 17
    vox = S ** 3
    assert N * vox * Cc * 4 > (1 << 32)
    rng = np.random.default_rng(7)
    x1 = rng.standard_normal((vox, Cc), dtype=np.float32)          # NDHWC sample
    g1 = rng.standard_normal((vox, Cc), dtype=np.float32)
    w = (rng.standard_normal((Cc, Cc, 5, 5, 5)) / np.sqrt(Cc * 125)).astype(np.float32)
    b = rng.standard_normal(Cc).astype(np.float32)
    mk = lambda n: Tensor(d, d.malloc(n * vox * Cc * 4), n, S, S, S, Cc, Cc, None)
    x, y, dy, dx = mk(N), mk(N), mk(N), mk(N)
    xs, gs = mk(1), mk(1)
    d.h2d(xs.ptr, x1)
    d.h2d(gs.ptr, g1)
    for n in range(N):
        d.d2d(x.ptr + n * vox * Cc * 4, xs.ptr, vox * Cc * 4)
        d.d2d(dy.ptr + n * vox * Cc * 4, gs.ptr, vox * Cc * 4)
    wp, bp = vec(w.ravel()), vec(b)
    cd = _desc((5,) * 3, (1,) * 3, (2,) * 3)
    d.call("msk_conv3d_fwd", cd, x.msk(), vp(wp), vp(bp), y.msk())
    d.call("msk_conv3d_dgrad", cd, dy.msk(), vp(wp), dx.msk(), 0)

    def sample(t, n):
        return d.d2h(t.ptr + n * vox * Cc * 4, (S, S, S, Cc), np.float32)

    y0, dx0 = sample(y, 0), sample(dx, 0)
    for n in (8, 9, 16):
        assert np.array_equal(sample(y, n), y0), n
        assert np.array_equal(sample(dx, n), dx0), n
    xv = x1.reshape(S, S, S, Cc)
    xp = np.pad(xv, ((2, 2), (2, 2), (2, 2), (0, 0))).astype(np.float64)
    for _ in range(8):
        dd, hh, ww = rng.integers(0, S, 3)
        co = int(rng.integers(0, Cc))
        patch = xp[dd:dd + 5, hh:hh + 5, ww:ww + 5, :]                       # [kd,kh,kw,ci]
        ref = float(np.einsum("abci,iabc->", patch, w[co].astype(np.float64))) + float(b[co])
        assert abs(y0[dd, hh, ww, co] - ref) < 2e-4 * max(1.0, abs(ref)), (dd, hh, ww, co)
    # weight gradient: chunks accumulate; N identical samples -> N x the one-sample gradient
    dw1, dwN, db = vec(np.zeros(w.size)), vec(np.zeros(w.size)), vec(np.zeros(Cc))
    d.call("msk_conv3d_wgrad", cd, xs.msk(), gs.msk(), vp(dw1), vp(db), 0)
    d.call("msk_conv3d_wgrad", cd, x.msk(), dy.msk(), vp(dwN), vp(db), 0)
    a, c = vec_back(dw1, w.size).astype(np.float64), vec_back(dwN, w.size).astype(np.float64)
    assert np.abs(c - N * a).max() < 2e-4 * np.abs(N * a).max()
    assert np.abs(vec_back(db, Cc) - N * g1.sum(axis=0, dtype=np.float64)).max() < 1e-3 * np.sqrt(N * vox)
    for t in (x, y, dy, dx, xs, gs):
        d.free(t.ptr)


def test_unet3d_fp16_192x192x64_full_size_step():
    """BASELINE configs[3] at full size: UNet3D (base 32, depth 4), 2 x 192 x 192 x 64, precision fp16.  No reference
    model exists, so properties: the fp16 matrix kernels really run at this size, the step is finite and BITWISE
    reproducible, the loss of a fixed batch goes down over three steps, and the fp16 logits stay within the stated fp16
    tolerance (2e-2 of max|logit|) of the fp32 (exact bf16x3 operands) forward pass of the same weights."""
    from medicalseg_amd import nn
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, UNet3D
    from medicalseg_amd.utils import loss_computation
    rng = np.random.default_rng(0)
    shape = (192, 192, 64)
    x = rng.standard_normal((2, 1) + shape).astype(np.float32)
    y = ((x[:, 0] > 0.3).astype(np.int32) + (x[:, 0] > 1.0).astype(np.int32))
    d = dev()
    runs, logits0 = [], {}
    for prec in ("fp16", "fp16", "fp32"):
        nn.seed(0)
        model = UNet3D(in_channels=1, num_classes=3, base_channels=32, depth=4, precision=prec)
        model.train()
        opt = optim.Momentum(1e-2, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
        losses = {"types": [MixedLoss([CrossEntropyLoss(weight=[1.0, 1.0, 1.0]), DiceLoss()], [1, 1])], "coef": [1]}
        d.prof_reset()
        d.set_option("prof_only_halo", 0)
        d.prof_enable(True)
        hist = []
        for it in range(3 if prec == "fp16" else 1):
            lg = model(to_tensor(x))
            if it == 0:
                logits0[prec] = lg[0].numpy()
            ll, per = loss_computation(lg, to_tensor(y), losses)
            loss = sum(ll)
            loss.backward()
            opt.step()
            model.clear_gradients()
            hist.append(float(loss))
        d.sync()
        d.prof_enable(False)
        tags = d.prof_report()
        if prec == "fp16":
            assert any(k.startswith("wbf_gemm_f16_k") for k in tags) and any(k.startswith("wbf_wgrad_f16_k") for k in tags)
            assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist
            sd = model.state_dict()
            runs.append((hist, sd["enc0.conv2.weight"], sd["up0.ops.conv1.weight"], sd["enc3.conv1.weight"]))
        assert logits0[prec].shape == (2, 3) + shape and np.all(np.isfinite(logits0[prec]))
    assert runs[0][0] == runs[1][0]
    for u, v in zip(runs[0][1:], runs[1][1:]):
        assert np.array_equal(u, v)                                   # bitwise reproducible fp16 step
    e = np.abs(logits0["fp16"] - logits0["fp32"]).max() / np.abs(logits0["fp32"]).max()
    print("UNet3D 2x192x192x64: fp16 vs fp32 logits %.2e; fp16 losses %s" % (e, runs[0][0]))
    assert e < 2e-2


def test_mri_inloop_preprocess_at_full_size_matches_oracle():
    """BASELINE configs[4] at ITS size: raw MRI slab 1008 x 1008 x 12 (prepare_mri_spine_seg.py:76) -> normalize(0, 2650)
    -> resample to 512 x 512 x 12 (order 1; labels order 0) -> max-normalise on the device, against the numpy restatement of
    tools/preprocess_utils (pinned to goldens from the reference's own code at small sizes, tests/test_oracle.py).  Round-1
    verdict: the in-loop HIP preprocessing had only been oracle-checked at 60 x 60 x 12 -> 32 x 32 x 12."""
    from oracle import preprocess_numpy as P
    from medicalseg_amd.preprocess import DevicePipeline
    rng = np.random.default_rng(8)
    raw = (rng.random((1008, 1008, 12)) * 2650).astype(np.float32)
    raw_lab = rng.integers(0, 20, (1008, 1008, 12)).astype(np.int32)
    pipe = DevicePipeline()
    chain = pipe.image(raw).normalize(0, 2650).resample([512, 512, 12], 1).max_normalize()
    lab = pipe.label(raw_lab).resample([512, 512, 12], 0)
    ref, _ = P.resample(P.normalize(raw.copy(), 0, 2650), [512, 512, 12], 1)
    ref = P.max_normalize(ref).astype(np.float32)
    got = chain.numpy()
    assert got.shape == (512, 512, 12)
    assert np.abs(got - ref).max() < 2e-6                       # order 1: within 1 fp32 ulp of values in [0, 1]
    assert np.array_equal(lab.numpy(), P.resample(raw_lab, [512, 512, 12], 0)[0])   # labels: bit-exact


def test_inloop_preprocess_on_a_second_context_feeds_the_training_context():
    """Round 5 (configs[4] as worded): the pipeline of the previous test on a SECOND context (= a second stream), raw samples in
    pinned memory, pooled buffers, handed to the training context with msk_ctx_wait (no host synchronisation between the two) --
    what tools/bench_workloads.py --inloop-preprocess times.  The tensor the training context reads must be the oracle's, for
    several samples in a row through recycled buffers, and a kernel of the training context that consumes it right behind the
    wait must see the finished data."""
    import ctypes as C
    from oracle import preprocess_numpy as P
    from medicalseg_amd.device import Device, Tensor, get_device
    from medicalseg_amd.preprocess import DevicePipeline
    dev = get_device()
    pre = Device(dev.index)
    pipe_x, pipe_y = DevicePipeline(pre, pooled=True), DevicePipeline(pre, pooled=True)
    raw_shape, shape = (240, 240, 12), (128, 128, 12)
    nraw = int(np.prod(raw_shape))
    rng = np.random.default_rng(3)
    raws = []
    for _ in range(2):
        pi, pl = C.c_void_p(), C.c_void_p()
        pre.call("msk_pinned_alloc", C.c_size_t(nraw * 4), C.byref(pi))
        pre.call("msk_pinned_alloc", C.c_size_t(nraw * 4), C.byref(pl))
        img = (rng.random(raw_shape) * 2650).astype(np.float32)
        lab = rng.integers(0, 20, raw_shape).astype(np.int32)
        np.ctypeslib.as_array((C.c_float * nraw).from_address(pi.value))[:] = img.ravel()
        np.ctypeslib.as_array((C.c_int32 * nraw).from_address(pl.value))[:] = lab.ravel()
        ref, _ = P.resample(P.normalize(img.copy(), 0, 2650), list(shape), 1)
        raws.append((pi.value, pl.value, P.max_normalize(ref).astype(np.float32)[0], P.resample(lab, list(shape), 0)[0]))
    old = None
    for i in range(5):
        pi, pl, ref_x, ref_y = raws[i % 2]
        pre.wait_for(dev)                       # the buffers going back to the pool were last read by the training context
        if old is not None:
            pipe_x.release(old[0])
            pipe_y.release(old[1])
        x = pipe_x.from_pinned(pi, raw_shape).normalize(0, 2650).resample(list(shape), 1).max_normalize().tensor()
        y = pipe_y.from_pinned(pl, raw_shape, np.int32).resample(list(shape), 0).int_tensor()
        dev.wait_for(pre)                       # device-side hand-over: NO host synchronisation of `pre`
        x.dev = dev
        # a kernel of the training context right behind the wait: out = 2 * x (msk_copy_scale with a per-(n, c) factor of 2)
        out = Tensor.empty(dev, 1, *shape, 1, arena=False)
        two = dev.malloc(16)
        dev.h2d(two, np.array([2.0], np.float32))
        dev.call("msk_copy_scale", x.msk(), C.c_void_p(two), out.msk(), 0)
        got2 = out.numpy()[0, 0]
        assert np.abs(got2 - 2.0 * ref_x).max() < 4e-6, i
        assert np.array_equal(dev.d2h(y.ptr, shape, np.int32), ref_y), i
        dev.free(two)
        dev.free(out.ptr)
        old = (x, y)
    for pi, pl, _, _ in raws:
        pre.call("msk_pinned_free", C.c_void_p(pi))
        pre.call("msk_pinned_free", C.c_void_p(pl))


def test_training_step_under_the_per_kernel_profile():
    """bench.py --profile-out / train.py --profiler_options bracket EVERY launch with a pair of events (msk_launch_scope); the
    brackets do not nest -- a kernel entry point that calls another bracketed entry point inside its own bracket leaves an event
    unrecorded and the report fails with 'invalid resource handle' (round 5: the statistics merge inside the up-convolution's
    bracket).  One VNet step at a size that reaches every product kernel class (64^3: the LDS-staged scatter / gather kernels of
    the k = s convolutions need >= 16384 source voxels), profile on, report read back."""
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.device import get_device, to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation
    dev = get_device()
    rng = np.random.default_rng(2)
    model = VNet(num_classes=3)
    model.train()
    opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    assert opt.enable_eager(model)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    x = to_tensor(rng.standard_normal((2, 1, 64, 64, 64)).astype(np.float32))
    y = to_tensor(rng.integers(0, 3, (2, 64, 64, 64)).astype(np.int32))
    for on in (False, True, True):
        dev.set_option("prof_only_halo", 0)
        dev.prof_reset()
        dev.prof_enable(on)
        ll, _ = loss_computation(model(x), y, losses)
        loss = sum(ll)
        loss.backward()
        opt.step()
        model.clear_gradients()
        dev.sync()
        dev.prof_enable(False)
        rep = dev.prof_report()
        assert np.isfinite(float(loss))
        if on:
            assert any(t.startswith("convT_scatter") for t in rep) and any(t.startswith("wbf_gemm") for t in rep), sorted(rep)
            assert all(ms >= 0.0 for _, ms in rep.values())


@pytest.mark.gpu
def test_dominant_kernel_events_attached_to_the_dispatch_agree_with_the_brackets():
    """bench.py's live roofline measurement: the events of the matrix stage ride on its dispatch (MSK_LAUNCH_TIMED =
    hipExtLaunchKernelGGL's start / stop events; no marker packets: the brackets idled the packet processor for ~0.2 ms of a
    headline step), and option "prof_paused" switches them off for the steps bench.py does not sample without draining.
    Same launches counted by both forms, the same time within the brackets' own overhead, nothing recorded while paused."""
    from medicalseg_amd.device import get_device, to_tensor
    from medicalseg_amd.models import VNet
    dev = get_device()
    rng = np.random.default_rng(4)
    model = VNet(num_classes=3)
    model.train()
    x = to_tensor(rng.standard_normal((2, 1, 64, 64, 64)).astype(np.float32))
    model(x)                       # packs, workspaces
    dev.sync()
    rep = {}
    for mode in ("attached", "brackets", "paused"):
        dev.set_option("prof_only_halo", 1)
        dev.set_option("prof_attach", 0 if mode == "brackets" else 1)
        dev.set_option("prof_paused", 1 if mode == "paused" else 0)
        dev.prof_reset()
        dev.prof_enable(True)
        for _ in range(3):
            model(x)
        dev.sync()
        dev.prof_enable(False)
        rep[mode] = dev.prof_report()
    dev.set_option("prof_attach", 1)
    dev.set_option("prof_paused", 0)
    dev.set_option("prof_only_halo", 0)
    assert rep["paused"] == {}, rep["paused"]
    assert rep["attached"] and all(t.startswith("wbf_gemm") for t in rep["attached"]), sorted(rep["attached"])
    assert sorted(rep["attached"]) == sorted(rep["brackets"])
    for t, (calls, ms) in rep["attached"].items():
        bc, bms = rep["brackets"][t]
        assert calls == bc and calls % 3 == 0 and ms > 0.0
        # the bracket form includes the markers' own latency (a few us per launch); the kernel time is the same
        assert ms <= bms * 1.10 + 0.02 * calls and bms <= ms * 1.35 + 0.02 * calls, (t, calls, ms, bms)
