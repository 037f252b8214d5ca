"""The BASELINE.json full-size parity cases: inputs, weights and dropout masks as pure functions of fixed seeds, shared
by the generator of the float64 oracle fixtures (tests/golden/make_fullsize_golden.py, run in the build container)
and by the GPU tests that compare the HIP path against them (tests/test_gpu_fullsize_parity.py)."""
import hashlib

import numpy as np

from oracle import vnet_numpy as O

SITES = [("down_tr128", 128), ("down_tr256", 256), ("up_tr256.x", 256), ("up_tr256.skip", 128),
         ("up_tr128.x", 256), ("up_tr128.skip", 64)]
ISO = ((2, 2, 2),) * 4
MRI_K = ((2, 2, 4), (2, 2, 2), (2, 2, 2), (2, 2, 2))      # vnet_mri_spine_seg_512_512_12_15k.yml:9
MRI_S = ((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2))      # :10

CASES = {
    # BASELINE.json configs[1]: VNet 128^3 fp32, batch 2, synthetic CT volumes, 3 classes
    "vnet128": dict(shape=(128, 128, 128), N=2, ncls=3, K=ISO, S=ISO, seed=2, golden="fullsize_vnet128_golden.npz"),
    # BASELINE.json configs[4]: VNet MRISpineSeg 512x512x12, 20 classes, anisotropic kernels
    "mri": dict(shape=(512, 512, 12), N=1, ncls=20, K=MRI_K, S=MRI_S, seed=3, golden="fullsize_mri_golden.npz"),
    # the reference's other model on the MRI slab (configs/mri_spine_seg/vnetdeepsup_mri_spine_seg_512_512_12_15k.yml:12-20;
    # models/vnet_deepsup.py:247-281): four outputs, one MixedLoss[CE, Dice] each, coef 0.25 each (round-5 verdict, Next 7)
    "mri_deepsup": dict(shape=(512, 512, 12), N=1, ncls=20, K=MRI_K, S=MRI_S, seed=5, golden="fullsize_mri_deepsup_golden.npz",
                        model="VNetDeepSup", coef=0.25),
}
SAMPLE = 8192          # entries kept per parameter-gradient tensor larger than that
LOGIT_SAMPLE = 32768   # voxels (all samples, all classes) kept of the logits


def synthetic_ct(idx, shape, seed=1234):
    """The generator of medicalseg_amd.datasets.SyntheticCT followed by Compose's max-normalisation
    (transforms/transform.py:67-69), restated here so that the fixture generator needs no product import."""
    rng = np.random.default_rng(seed + idx)
    D, H, W = shape
    hu = np.clip(rng.standard_normal(shape, dtype=np.float32) * 450.0 - 600.0, -2000, 2000)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    label = np.zeros(shape, dtype=np.int32)
    for c in range(1, 3):
        cen = rng.uniform(0.3, 0.7, 3) * np.array(shape)
        rad = rng.uniform(0.12, 0.25, 3) * np.array(shape)
        m = ((zz - cen[0]) / rad[0]) ** 2 + ((yy - cen[1]) / rad[1]) ** 2 + ((xx - cen[2]) / rad[2]) ** 2 <= 1
        label[m] = c
        hu[m] += 300.0 * c
    im = (hu + 1200.0) / (1800.0 / 255.0)
    np.clip(im, 0, 255, out=im)
    im = im.astype(np.float32)
    return (im / im.max()).astype(np.float32), label


def mri_slab(shape, ncls, seed):
    """SURVEY d1 cfg5 at the network's input: intensities U(0, 1) after normalisation, labels = 20-class blobs
    (a coarse random class map repeated 16 x 16 x 3 voxels per cell, so every class owns connected regions)."""
    rng = np.random.default_rng(seed)
    x = rng.random(shape, dtype=np.float32)
    coarse = rng.integers(0, ncls, (shape[0] // 16, shape[1] // 16, shape[2] // 3)).astype(np.int32)
    y = np.repeat(np.repeat(np.repeat(coarse, 16, 0), 16, 1), 3, 2)
    x += 0.05 * y.astype(np.float32)
    return (x / x.max()).astype(np.float32), y


def build(name):
    """-> dict(x [N,1,D,H,W] f32, y [N,D,H,W] i32, params {name: f32}, masks {site: [N,C] f32 in {0,2}}, cfg)."""
    c = CASES[name]
    if name == "vnet128":
        items = [synthetic_ct(i, c["shape"]) for i in range(c["N"])]
    else:
        items = [mri_slab(c["shape"], c["ncls"], (100 if name == "mri" else 200) + i) for i in range(c["N"])]
    x = np.stack([i[0] for i in items])[:, None].astype(np.float32)
    y = np.stack([i[1] for i in items]).astype(np.int32)
    if c.get("model") == "VNetDeepSup":
        params = O.init_params_deepsup(c["seed"], 1, c["ncls"], c["K"], c["S"])
    else:
        params = O.init_params(c["seed"], 1, c["ncls"], c["K"], c["S"])
    rng = np.random.default_rng(17)
    masks = {s: (rng.random((c["N"], ch)) < 0.5).astype(np.float32) * 2.0 for s, ch in SITES}
    return dict(x=x, y=y, params=params, masks=masks, cfg=c)


def digest(case):
    h = hashlib.sha256()
    h.update(case["x"].tobytes())
    h.update(case["y"].tobytes())
    for k in sorted(case["params"]):
        h.update(np.ascontiguousarray(case["params"][k]).tobytes())
    for k in sorted(case["masks"]):
        h.update(case["masks"][k].tobytes())
    return h.hexdigest()


def sample_indices(name, size):
    """Deterministic subset of a gradient tensor (all of it up to SAMPLE entries)."""
    if size <= SAMPLE:
        return None
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
    return np.sort(np.random.default_rng(seed).choice(size, SAMPLE, replace=False)).astype(np.int64)


def torch_model(case, dtype):
    """The torch-CPU restatement of the case's network (oracle/vnet_torch.py) with the case's parameters loaded -- generator side
    only (the GPU tests never import torch)."""
    import torch
    from oracle import vnet_torch as VT
    c = case["cfg"]
    cls = VT.TorchVNetDeepSup if c.get("model") == "VNetDeepSup" else VT.TorchVNet
    tm = cls(1, c["ncls"], c["K"], c["S"])
    tm = tm.double() if dtype == torch.float64 else tm.float()
    npdt = np.float64 if dtype == torch.float64 else np.float32
    tm.load_oracle_params({k: np.asarray(v, dtype=npdt) for k, v in case["params"].items()})
    tm.train()
    return tm


def torch_losses(case, outs, y):
    """-> (total loss, [(ce, dice loss, per-class dice, class weights)] per output): one MixedLoss[CE, Dice] per output, each with
    its own first-call class weights (losses/loss_utils.py:31-40), outer coefficient cfg['coef'] (1 for the single-output net)."""
    import torch
    from oracle import vnet_torch as VT
    c = case["cfg"]
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    coef = float(c.get("coef", 1.0))
    total, parts = 0.0, []
    for lg in outs:
        with torch.no_grad():
            p = torch.softmax(lg, 1).transpose(0, 1).reshape(c["ncls"], -1)
            w = (1.0 - p).sum(-1) / p.sum(-1)
        ce, dl, per = VT.torch_mixed_loss(lg, y, w)
        total = total + coef * (ce + dl)
        parts.append((ce, dl, per, w))
    return total, parts
