"""Oracle parity AT THE BASELINE.json SIZES (round-3 verdict, "Next 1"): ONE full training step -- forward, CE + Dice,
backward -- of
  * configs[1]: VNet 2 x 128^3, 3 classes, synthetic CT volumes, dropout on (fixed masks), and
  * configs[4]: VNet 1 x 512 x 512 x 12, 20 classes, the anisotropic MRI kernels (vnet_mri_spine_seg_512_512_12_15k.yml:9-10)
on the HIP path, against the float64 torch-CPU restatement of the reference network (oracle/vnet_torch.py).  The float64
step costs 6 / 4 minutes on 8 host cores, so it was evaluated ONCE in the build container and committed as a fixture
(tests/golden/make_fullsize_golden.py -> tests/golden/fullsize_*_golden.npz: logits at 32768 sampled voxels, class
weights, both losses, per-class dice, every parameter gradient -- whole when <= 8192 entries, else a fixed random subset of
8192 -- and the BatchNorm batch statistics); the inputs are regenerated here from the same seeds and checked by SHA-256.

Bounds (the 32^3-calibrated ones of tests/test_gpu_model.py::test_vnet_32cube_batch2_gradients_calibrated, tightened where
the full-size problem is better conditioned): logits 2e-5 of max|logit|, CE 2e-5 relative, Dice loss 2e-5, per-class dice
1e-5, class weights 1e-5 relative; per parameter tensor rel-L2 <= 8e-3 (or, where two fp32 evaluations of the step differ by
more than that bound allows, 2e-3 + twice their distance: _fp32_spread) with the median <= 4e-3, and the SYSTEMATIC part
separately: least-squares scale of every tensor's gradient against the oracle's within 1e-3 of one; BatchNorm running
statistics 2e-5.  The kernels the smaller parity tests cannot reach are asserted to have run (HIP-event tags with shapes):
the fused matrix + output-transform kernel at 128^3, the 512-way split in_tr weight gradient, out_tr's three kernels."""
import os

import numpy as np
import pytest

import fullsize_cases as FC
from helpers import dev

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def _run_case(name, grads_only=False):
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation
    case = FC.build(name)
    c = case["cfg"]
    gold = np.load(os.path.join(HERE, "golden", c["golden"]))
    assert str(gold["digest"]) == FC.digest(case), "regenerated inputs differ from the ones the fixture was computed on"
    d = dev()
    model = VNet(elu=False, in_channels=1, num_classes=c["ncls"], kernel_size=c["K"], stride_size=c["S"])
    missing, unexpected = model.set_state_dict(case["params"])
    assert not missing and not unexpected
    model.train()
    model.set_dropout_masks(case["masks"])
    ce_l = CrossEntropyLoss()
    losses = {"types": [MixedLoss([ce_l, DiceLoss()], [1, 1])], "coef": [1]}
    d.prof_reset()
    d.set_option("prof_only_halo", 0)
    d.set_option("prof_shapes", 1)
    d.prof_enable(True)
    try:
        logits = model(case["x"])
        lg = logits[0].numpy()
        loss_list, per = loss_computation(logits, to_tensor(case["y"]), losses)
        model.clear_gradients()
        sum(loss_list).backward()
        d.sync()
    finally:
        d.prof_enable(False)
        d.set_option("prof_shapes", 0)
    tags = d.prof_report()
    N, C = lg.shape[:2]
    vox = int(np.prod(lg.shape[2:]))
    assert lg.shape == (c["N"], c["ncls"]) + tuple(c["shape"])
    flat = np.moveaxis(lg.reshape(N, C, vox), 1, 2).reshape(N * vox, C)
    e_lg = float(np.abs(flat[gold["logit_idx"]] - gold["logit_val"]).max() / float(gold["logit_absmax"]))
    e_w = float(np.abs(np.asarray(ce_l.weight, np.float64) / gold["class_weights"] - 1).max())
    e_ce = abs(float(loss_list[0]) / float(gold["ce"]) - 1)
    e_dl = abs(float(loss_list[1]) - float(gold["dice_loss"]))
    e_per = float(np.abs(np.asarray(per, np.float64) - gold["per_channel_dice"]).max())
    if grads_only:   # second evaluation of the same step (another summation order): the parameter gradients only
        return {pname: (p.grad_numpy().astype(np.float64).ravel(), FC.sample_indices(pname, int(np.prod(p.shape))))
                for pname, p in model.named_parameters()}
    l2s, bias, zero, grads = {}, {}, 0, {}
    for pname, p in model.named_parameters():
        ref = gold["g/" + pname]
        idx = FC.sample_indices(pname, int(np.prod(p.shape)))
        g = p.grad_numpy().astype(np.float64).ravel()
        g = g if idx is None else g[idx]
        assert g.shape == ref.shape, pname
        if float(gold["gn/" + pname]) < 1e-9 * max(1.0, np.sqrt(ref.size)):
            # conv bias ahead of a train-mode BatchNorm: exactly 0 in exact arithmetic
            assert np.abs(g).max() < 1e-4, pname
            zero += 1
            continue
        l2s[pname] = _l2(g, ref)
        grads[pname] = (g, float(np.linalg.norm(ref)))
        if ref.size >= 1000:
            bias[pname] = float(np.vdot(g, ref) / np.vdot(ref, ref) - 1.0)
    sd = model.state_dict()
    e_bn = 0.0
    for k in case["params"]:
        if k.endswith("._mean") or k.endswith("._variance"):
            want = 0.9 * case["params"][k].astype(np.float64) + 0.1 * gold["bn/" + k]      # SURVEY App. B.2 (biased batch variance)
            e_bn = max(e_bn, float(np.abs(sd[k] - want).max() / (1.0 + np.abs(want).max())))
    worst = max(l2s, key=l2s.get)
    wb = max(bias, key=lambda k_: abs(bias[k_]))
    med = float(np.median(list(l2s.values())))
    print("\n%s full size: logits %.2e | class weights %.2e | CE %.2e dice loss %.2e per-class dice %.2e | gradients of %d tensors "
          "(+%d identically zero): rel-L2 median %.2e worst %.2e (%s) | worst scale bias %.2e (%s) | BN running stats %.2e"
          % (name, e_lg, e_w, e_ce, e_dl, e_per, len(l2s), zero, med, l2s[worst], worst, bias[wb], wb, e_bn))
    return dict(grads=grads, tags=tags, e_lg=e_lg, e_w=e_w, e_ce=e_ce, e_dl=e_dl, e_per=e_per, l2s=l2s, bias=bias, med=med, worst=worst,
                wb=wb, e_bn=e_bn)


def _fp32_spread(name, r):
    """The step once more with ANOTHER summation order of the forward BatchNorm statistics (debug option "reduce_vpl_site":
    64 instead of 8 voxels per lane in bn_stats_partial -- activations move by 3e-7 ... 3e-6, i.e. by fp32 rounding): how far
    two equally valid fp32 evaluations of each gradient tensor lie apart, relative to the oracle's norm.  Measured (round 4,
    tools/diag_fullsize_ab.py): 1e-3 ... 9e-3 per tensor -- the backward pass amplifies fp32 rounding of the forward pass by
    ~1000x (gradients of 1e-6 behind 24 BatchNorm backward projections), so any single evaluation sits anywhere inside that
    band (sweep of eight orders: worst tensor 1.9e-3 ... 9.7e-3 against the float64 oracle, tools/diag_fullsize_parity.py)."""
    d = dev()
    d.set_option("reduce_vpl_site", 64)
    try:
        other = _run_case(name, grads_only=True)
    finally:
        d.set_option("reduce_vpl_site", 0)
    spread = {}
    for pname, (g, refnorm) in r["grads"].items():
        g2, idx = other[pname]
        g2 = g2 if idx is None else g2[idx]
        spread[pname] = float(np.linalg.norm(g - g2) / (refnorm + 1e-300))
    return spread


def _assert_bounds(r, name):
    assert r["e_lg"] < 2e-5, r["e_lg"]
    assert r["e_w"] < 1e-5 and r["e_ce"] < 2e-5 and r["e_dl"] < 2e-5 and r["e_per"] < 1e-5
    # per tensor: the 32^3-calibrated 8e-3, or -- where fp32 itself is less certain than that -- twice the distance between
    # two fp32 evaluations of this very step (rounding noise has no preferred evaluation; a defect would stand out of it);
    # never beyond 3e-2, and the systematic part (scale bias, below) stays at 1e-3
    spread = _fp32_spread(name, r) if r["l2s"][r["worst"]] >= 4e-3 else {}
    for pname, e in r["l2s"].items():
        lim = min(3e-2, max(8e-3, 2e-3 + 2.0 * spread.get(pname, 0.0)))
        assert e < lim, (pname, e, spread.get(pname))
    if spread:
        ws = max(spread, key=spread.get)
        print("fp32 spread between two summation orders: worst %.2e (%s), at the worst tensor %.2e" % (spread[ws], ws, spread[r["worst"]]))
    assert r["med"] < 4e-3
    assert abs(r["bias"][r["wb"]]) < 1e-3, (r["wb"], r["bias"][r["wb"]])
    assert r["e_bn"] < 2e-5


def _has(tags, prefix, *parts):
    return any(k.startswith(prefix) and all(p in k for p in parts) for k in tags)


def test_vnet_128_batch2_full_step_matches_float64_oracle():
    NAME = "vnet128"
    r = _run_case(NAME)
    t = r["tags"]
    # the variants only this size reaches
    assert _has(t, "wbf_gemm_h2_k", "dhw=128x128x128", "fused"), sorted(t)        # fused matrix + output transform, 32ch @ 128^3
    assert _has(t, "wbf_gemm_h2_k", "dhw=64x64x64", "ck=64", "fused")              # 64ch @ 64^3 class
    assert _has(t, "wbf_gemm_h2_k", "dhw=16x16x16", "ck=256")                      # deep three-stage form
    assert _has(t, "wbf_wgrad_h2_k", "dhw=128x128x128")
    assert _has(t, "wbf_tin_dual_k") or _has(t, "wbf_tin_bn_k")
    assert _has(t, "conv_c1_mfma", "dhw=128x128x128")
    assert _has(t, "wgrad_c1_mfma", "bn-fused")
    assert _has(t, "conv_foldn_h2", "dhw=128x128x128") and _has(t, "conv_tk_h2", "dhw=128x128x128") and _has(t, "wgrad_cbs_h2")
    assert _has(t, "wgrad_ks2_mfma") and (_has(t, "convT_scatter_lds") or _has(t, "convT_scatter_mfma"))
    _assert_bounds(r, NAME)


def test_vnet_mri_512x512x12_20_classes_full_step_matches_float64_oracle():
    NAME = "mri"
    r = _run_case(NAME)
    t = r["tags"]
    assert _has(t, "wbf_gemm_h2_k", "dhw=512x512x12"), sorted(t)                   # padded-plane path of the 12-deep level
    assert _has(t, "wbf_gemm_h2_k", "dhw=256x256x9")
    _assert_bounds(r, NAME)
