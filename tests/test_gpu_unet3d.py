"""Builder-defined UNet3D (no reference model exists) against its torch-CPU float64 restatement:
logits, the CE+Dice loss, every parameter gradient, and a training step through the unchanged optimizer / Config path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _loss_and_grads_torch(tm, x, y, ncls):
    import torch
    from oracle.vnet_torch import torch_mixed_loss
    logits = tm(torch.as_tensor(x, dtype=torch.float64))
    ce, dl, _ = torch_mixed_loss(logits, torch.as_tensor(y.astype(np.int64)), torch.ones(ncls, dtype=torch.float64))
    (ce + dl).backward()
    return logits.detach().numpy(), float(ce + dl)


@pytest.mark.parametrize("shape,depth,base", [((2, 1, 16, 16, 8), 3, 8), ((1, 1, 8, 24, 16), 4, 4)])
def test_unet3d_forward_backward_matches_torch(shape, depth, base):
    import torch
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, UNet3D
    from medicalseg_amd.utils import loss_computation
    from oracle.unet3d_torch import TorchUNet3D
    ncls = 3
    rng = np.random.default_rng(depth)
    from medicalseg_amd import nn
    nn.seed(0)     # the conv weights come from the package's init stream: independent of which tests ran before
    model = UNet3D(in_channels=1, num_classes=ncls, base_channels=base, depth=depth)
    state = model.state_dict()
    for k, v in state.items():                      # non-trivial affine parameters and slopes
        if k.endswith(".scale"):
            state[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith("_weight"):
            state[k] = rng.uniform(0.1, 0.4, v.shape).astype(np.float32)
        elif k.endswith(".bias"):
            state[k] = (rng.standard_normal(v.shape) * 0.1).astype(np.float32)
    model.set_state_dict(state)
    x = rng.standard_normal(shape).astype(np.float32)
    y = rng.integers(0, ncls, (shape[0],) + shape[2:]).astype(np.int32)
    tm = TorchUNet3D(1, ncls, base, depth).double().load_msk_state(state)
    ref_logits, ref_loss = _loss_and_grads_torch(tm, x, y, ncls)

    model.train()
    class_w = {"types": [MixedLoss([CrossEntropyLoss(weight=[1.0] * ncls), DiceLoss()], [1, 1])], "coef": [1]}
    logits = model(to_tensor(x))
    got = logits[0].numpy()
    scale = np.abs(ref_logits).max()
    assert np.abs(got - ref_logits).max() <= 2e-4 * scale, np.abs(got - ref_logits).max() / scale
    ll, _ = loss_computation(logits, to_tensor(y), class_w)
    loss = sum(ll)
    assert abs(float(loss.numpy()[0]) - ref_loss) <= 1e-4 * abs(ref_loss)
    loss.backward()
    ref_g = tm.grads_as_msk([n for n, _ in model.named_parameters()])
    for name, p in model.named_parameters():
        g, r = p.grad_numpy(), ref_g[name]
        if name.endswith("conv1.bias") or name.endswith("conv2.bias") or name.endswith("conv.bias"):
            # a bias in front of an instance-statistics norm has an identically zero gradient; the product skips
            # the pass that would compute the rounding noise torch returns
            assert np.abs(r).max() < 1e-9 and np.abs(g).max() == 0.0, name
            continue
        tol = 3e-4 * max(np.abs(r).max(), 1e-6)
        assert np.abs(g - r).max() <= tol, (name, np.abs(g - r).max() / max(np.abs(r).max(), 1e-6))


def test_unet3d_trains_through_config(tmp_path):
    """YAML `type: UNet3D` through the unchanged Config / train() path: the loss goes down on a fixed batch."""
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.cvlibs import manager
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss
    from medicalseg_amd.utils import loss_computation
    assert "UNet3D" in manager.MODELS.components_dict
    model = manager.MODELS["UNet3D"](in_channels=1, num_classes=3, base_channels=8, depth=3)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 1, 16, 16, 16)).astype(np.float32)
    yy = (x[:, 0] > 0.3).astype(np.int32) + (x[:, 0] > 1.0).astype(np.int32)
    opt = optim.Momentum(0.05, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    hist = []
    for _ in range(12):
        ll, _ = loss_computation(model(to_tensor(x)), to_tensor(yy), losses)
        loss = sum(ll)
        hist.append(float(loss.numpy()[0]))
        loss.backward()
        opt.step()
        model.clear_gradients()
    assert hist[-1] < 0.8 * hist[0], hist


def test_unet3d_fp16_path_matches_torch_at_the_stated_fp16_tolerance():
    """precision="fp16" (BASELINE configs[3]): the 3x3x3 convolutions of the 32-channel level run on the fp16 matrix pipe
    (`wbf_gemm_f16_k` / `wbf_wgrad_f16_k`: fp16 operands in the Winograd F(4,3) domain, fp32 accumulation; activations,
    InstanceNorm statistics, loss and optimizer in fp32).  No reference model exists: the yardstick is the
    torch-CPU float64 restatement.  STATED fp16 TOLERANCE: logits 1.5e-2 of max|logit|, loss 5e-3 relative, parameter
    gradients rel-L2 1e-1 per tensor (per-convolution operand rounding 2^-11, amplified by the InstanceNorm adjoints;
    measured: logits 2.4e-3, loss 1e-5, gradients 4.1e-2 median / 6.7e-2 worst; the fp32 path of the same model: 1.5e-6 / 5.5e-6)."""
    import torch
    from medicalseg_amd import nn
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, UNet3D
    from medicalseg_amd.utils import loss_computation
    from oracle.unet3d_torch import TorchUNet3D
    from helpers import dev
    ncls, shape, depth, base = 3, (2, 1, 16, 16, 16), 2, 32     # batch 2: two instances with their own statistics
    rng = np.random.default_rng(7)
    nn.seed(0)
    x = rng.standard_normal(shape).astype(np.float32)
    x[1] = 2.0 * x[1] - 0.3
    y = rng.integers(0, ncls, (shape[0],) + shape[2:]).astype(np.int32)
    res = {}
    from medicalseg_amd.models import unet3d as U
    for prec in ("fp32-unfused", "fp32", "fp16"):
        # "fp32-unfused": the InstanceNorm backward as separate reduce / apply passes with dy in HBM (A/B of round 4's fused form)
        U.FUSED_IN_BACKWARD = prec != "fp32-unfused"
        unfused, prec = prec == "fp32-unfused", prec.split("-")[0]
        nn.seed(0)
        model = UNet3D(in_channels=1, num_classes=ncls, base_channels=base, depth=depth, precision=prec)
        state = model.state_dict()
        if "tm" not in res:
            res["tm"] = TorchUNet3D(1, ncls, base, depth).double().load_msk_state(state)
            res["ref"] = _loss_and_grads_torch(res["tm"], x, y, ncls)
            res["ref_g"] = res["tm"].grads_as_msk([n for n, _ in model.named_parameters()])
        model.train()
        d = dev()
        d.prof_reset()
        d.set_option("prof_only_halo", 0)
        d.prof_enable(True)
        losses = {"types": [MixedLoss([CrossEntropyLoss(weight=[1.0] * ncls), DiceLoss()], [1, 1])], "coef": [1]}
        logits = model(to_tensor(x))
        got = logits[0].numpy()
        ll, _ = loss_computation(logits, to_tensor(y), losses)
        loss = sum(ll)
        lv = float(loss.numpy()[0])
        loss.backward()
        d.sync()
        d.prof_enable(False)
        tags = d.prof_report()
        ran16 = any(k.startswith("wbf_gemm_f16_k") for k in tags) and any(k.startswith("wbf_wgrad_f16_k") for k in tags)
        assert ran16 == (prec == "fp16"), sorted(tags)
        # round 4: the square 3x3x3 units take the InstanceNorm backward inside the dual transform (msk_conv3d_bwd_inact)
        assert ("wbf_tin_dual_k" in tags) == (not unfused), sorted(tags)
        ref_logits, ref_loss = res["ref"]
        e_lg = np.abs(got - ref_logits).max() / np.abs(ref_logits).max()
        e_ls = abs(lv - ref_loss) / abs(ref_loss)
        l2 = {}
        for name, p in model.named_parameters():
            r = res["ref_g"][name]
            if np.abs(r).max() < 1e-9:
                continue
            g = p.grad_numpy().astype(np.float64)
            l2[name] = float(np.linalg.norm(g - r) / np.linalg.norm(r))
        worst = max(l2, key=l2.get)
        print("\nUNet3D %s%s: logits %.2e loss %.2e grads rel-L2 worst %.2e (%s) median %.2e" %
              (prec, " (unfused backward)" if unfused else "", e_lg, e_ls, l2[worst], worst, float(np.median(list(l2.values())))))
        if unfused:
            res["l2_unfused"] = l2
            continue
        if prec == "fp32":
            assert e_lg < 2e-4 and e_ls < 1e-4 and l2[worst] < 3e-3
            # tensor by tensor no worse than the separate passes (the bias of a norm in front of another instance norm has a
            # nearly vanishing true gradient: its relative error is large in ANY fp32 evaluation)
            for name, e in l2.items():
                assert e <= 3 * res["l2_unfused"][name] + 2e-5, (name, e, res["l2_unfused"][name])
        else:
            assert e_lg < 1.5e-2 and e_ls < 5e-3 and l2[worst] < 1e-1
    U.FUSED_IN_BACKWARD = True


@pytest.mark.parametrize("cin,cout,shape,records", [(32, 32, (2, 64, 128, 32), True), (64, 64, (2, 64, 64, 32), True), (32, 32, (2, 16, 32, 16), False),
                                                    (64, 32, (2, 16, 16, 16), False), (32, 128, (2, 8, 16, 8), False), (8, 8, (2, 6, 8, 8), False)])
def test_instance_statistics_from_the_convolution_call(cin, cout, shape, records):
    """Round 4: msk_conv3d_fwd_in -- convolution + per-sample statistics + their finalisation in one call.  For <= 64 output
    channels the statistics come from the per-tile records of the one-kernel matrix stage (no read of y: no bn_stats_partial
    launch), else from one statistics pass per sample inside the call; either way the unit's output and the per-sample
    coefficients must equal the separate passes (msk_bn_stats + msk_bn_finalize per sample) to fp32 rounding."""
    from helpers import dev
    from medicalseg_amd import nn
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import unet3d as U
    d = dev()
    N, D, H, W = shape
    rng = np.random.default_rng(cin + cout)
    nn.seed(1)
    conv, norm, act = nn.Conv3D(cin, cout, 3, padding=1), U.InstanceNorm3D(cout), nn.PReLU(cout)
    unit = U.ConvINAct(conv, norm, act)
    params = [conv.weight, conv.bias, norm.scale, norm.bias, act._weight]
    arena = nn.ParamArena(d, params)
    norm.scale.set_value(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    norm.bias.set_value((rng.standard_normal(cout) * 0.1).astype(np.float32))
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    x[1] = x[1] * 3.0 + 0.5                      # the two instances have different statistics
    outs, coefs, tags = {}, {}, {}
    for mode in (False, True):
        U.INSTANCE_STATS_IN_CONV = mode
        try:
            d.arena.reset()
            d.prof_reset()
            d.set_option("prof_only_halo", 0)
            d.prof_enable(True)
            out = unit.forward(to_tensor(x))
            d.sync()
            d.prof_enable(False)
            tags[mode] = d.prof_report()
            outs[mode] = out.numpy()
            sc = norm.scratch(d, N)
            coefs[mode] = d.d2h(sc["per"], (N, 4, cout), np.float32)
        finally:
            U.INSTANCE_STATS_IN_CONV = True
    scale = np.abs(outs[False]).max()
    assert np.abs(outs[True] - outs[False]).max() <= 2e-5 * scale
    assert np.abs(coefs[True] - coefs[False]).max() <= 2e-5 * np.abs(coefs[False]).max()
    assert np.abs(coefs[True][0] - coefs[True][1]).max() > 1e-2          # really per sample
    if records:   # enough tiles for the one-kernel matrix stage (>= 2 per CU): the statistics come from its tile records
        assert "bn_stats_partial" not in tags[True], tags[True]
        assert tags[True].get("bn_stats_merge", (0, 0))[0] == N
    del arena
