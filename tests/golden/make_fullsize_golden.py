"""Generates tests/golden/fullsize_{vnet128,mri}_golden.npz: ONE full training step of BASELINE.json configs[1]
(VNet 2 x 128^3, 3 classes) and configs[4] (VNet 1 x 512 x 512 x 12, 20 classes, MRI kernels) evaluated by the
torch-CPU restatement of the reference network (oracle/vnet_torch.py) in FLOAT64 -- logits, class weights, both
losses, per-class dice, the gradient of every parameter tensor, BatchNorm batch statistics.

    python tests/golden/make_fullsize_golden.py [vnet128] [mri]        (build container; ~10 / ~25 min on 8 cores)

The full gradients are 182 MB per case, so the fixture keeps: every tensor of <= 8192 entries whole, a fixed random
subset of 8192 entries (tests/fullsize_cases.py::sample_indices) of every larger one plus its full L2 norm, 32768
sampled voxels of the logits, and a SHA-256 of the regenerated inputs.  PARITY UNPINNED by the reference (Paddle is
absent): this is the restatement's arithmetic, cross-checked against the numpy oracle in tests/test_oracle.py.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import fullsize_cases as FC  # noqa: E402
from oracle import vnet_torch as VT  # noqa: E402


def run(name):
    case = FC.build(name)
    c = case["cfg"]
    torch.set_num_threads(os.cpu_count() or 1)
    VT.SLAB_BYTES = 2 << 30
    tm = FC.torch_model(case, torch.float64)
    stats = {}

    def hook(mod_name):
        def f(mod, inp):
            v = inp[0].detach()
            stats[mod_name + "._mean"] = v.mean(dim=(0, 2, 3, 4)).numpy().copy()
            stats[mod_name + "._variance"] = v.var(dim=(0, 2, 3, 4), unbiased=False).numpy().copy()
        return f

    for mod_name, mod in tm.named_modules():
        if isinstance(mod, torch.nn.BatchNorm3d):
            mod.register_forward_pre_hook(hook(mod_name))
    t0 = time.time()
    x = torch.tensor(case["x"], dtype=torch.float64)
    y = torch.tensor(case["y"])
    outs = tm(x, case["masks"])
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    total, parts = FC.torch_losses(case, outs, y)     # losses/loss_utils.py:31-40: first-call class weights, detached, per output
    ce, dl, per, w = parts[0]
    t1 = time.time()
    total.backward()
    t2 = time.time()
    print("%s: forward %.0f s, backward %.0f s; ce %.9f dice %.9f (%d output(s))" % (name, t1 - t0, t2 - t1, float(ce), float(dl), len(outs)), flush=True)
    out = {"digest": np.array(FC.digest(case)), "class_weights": w.numpy(), "ce": np.float64(ce.item()),
           "dice_loss": np.float64(dl.item()), "per_channel_dice": per.detach().numpy()}
    N, C = outs[0].shape[:2]
    vox = int(np.prod(outs[0].shape[2:]))
    li = np.sort(np.random.default_rng(5).choice(N * vox, FC.LOGIT_SAMPLE, replace=False))
    out["logit_idx"] = li
    for oi, (lg, (ce_i, dl_i, per_i, w_i)) in enumerate(zip(outs, parts)):
        lgn = lg.detach().numpy()
        flat = np.moveaxis(lgn.reshape(N, C, vox), 1, 2).reshape(N * vox, C)
        sfx = "" if oi == 0 else "@%d" % oi      # output 0 keeps the single-output key names
        sel = li if oi == 0 else li[::4]         # the extra outputs of a deep-supervision net: a quarter of the sample (fixture size)
        if oi:
            out["logit_idx" + sfx] = sel
        out["logit_val" + sfx] = flat[sel]
        out["logit_absmax" + sfx] = np.float64(np.abs(lgn).max())
        if oi:
            out["class_weights" + sfx], out["ce" + sfx] = w_i.numpy(), np.float64(ce_i.item())
            out["dice_loss" + sfx], out["per_channel_dice" + sfx] = np.float64(dl_i.item()), per_i.detach().numpy()
    out["n_outputs"] = np.int64(len(outs))
    for k, g in tm.named_oracle_grads().items():
        idx = FC.sample_indices(k, g.size)
        out["g/" + k] = g.ravel() if idx is None else g.ravel()[idx]
        out["gn/" + k] = np.float64(np.linalg.norm(g.ravel()))
    for k, v in stats.items():
        out["bn/" + k] = v
    path = os.path.join(HERE, c["golden"])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["vnet128", "mri"]):
        run(n)
