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
    tm = VT.TorchVNet(1, c["ncls"], c["K"], c["S"]).double()
    tm.load_oracle_params({k: np.asarray(v, dtype=np.float64) for k, v in case["params"].items()})
    tm.train()
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
    lg = tm(x, case["masks"])
    with torch.no_grad():   # losses/loss_utils.py:31-40 (first-call class weights, detached)
        p = torch.softmax(lg, 1).transpose(0, 1).reshape(c["ncls"], -1)
        w = (1.0 - p).sum(-1) / p.sum(-1)
    ce, dl, per = VT.torch_mixed_loss(lg, y, w)
    t1 = time.time()
    (ce + dl).backward()
    t2 = time.time()
    print("%s: forward %.0f s, backward %.0f s; ce %.9f dice %.9f" % (name, t1 - t0, t2 - t1, float(ce), float(dl)), flush=True)
    out = {"digest": np.array(FC.digest(case)), "class_weights": w.numpy(), "ce": np.float64(ce.item()),
           "dice_loss": np.float64(dl.item()), "per_channel_dice": per.detach().numpy()}
    lgn = lg.detach().numpy()
    N, C = lgn.shape[:2]
    vox = int(np.prod(lgn.shape[2:]))
    li = np.sort(np.random.default_rng(5).choice(N * vox, FC.LOGIT_SAMPLE, replace=False))
    flat = np.moveaxis(lgn.reshape(N, C, vox), 1, 2).reshape(N * vox, C)
    out["logit_idx"], out["logit_val"] = li, flat[li]
    out["logit_absmax"] = np.float64(np.abs(lgn).max())
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
