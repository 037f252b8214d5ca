"""Generates tests/golden/fullsize_{vnet128,mri}_kink.npz: how far the parameter gradients of the full-size training step
move when the PReLU branch decisions that NO fp32 evaluation can make reliably are made the other way -- measured in
FLOAT64 on the torch-CPU restatement of the reference network (oracle/vnet_torch.py), i.e. without the HIP path.

Why (profiles/r05_fullsize_bimodal_root_cause.txt): the gradient of a PReLU is discontinuous at u = 0.  The product's
convolution layers are specified to 4e-6 of max |y| (tests/test_gpu_wbf.py); a pre-activation with |u| below that band has an
undecidable sign in fp32, and whichever way it falls the element's gradient changes by the factor alpha = 0.25.  The
gradients of this network are heavy-tailed (max = 10 ... 25 x rms at the 16^3 level), so ONE such element under a large
upstream gradient moves whole gradient tensors by ~1e-2 rel-L2: round 4's "bimodal" 9.6e-3 / 2e-3 pattern is exactly one
voxel of down_tr128.ops.2 (|u| = 4.5e-6, gradient 10 x rms) falling on either side.  A float64 oracle decides every sign
exactly, so the distance oracle <-> fp32 has a component no kernel can remove; this fixture measures its size.

    python tests/golden/make_fullsize_kink_sensitivity.py [vnet128] [mri]        (build container, 8 cores: ~12 / ~15 min)

One float64 forward pass; three backward passes over the same graph (the forward VALUE of a PReLU is continuous at the
kink, only the derivative's branch changes): exact; EVERY element with |u| < TAU * max|u| of its layer on the other branch
("all"); a fixed pseudo-random half of those ("half").  Stored per parameter tensor: ||g_variant - g_exact|| / ||g_exact||
("kink_all/<name>", "kink_half/<name>"), and per PReLU layer the number of elements inside the band.
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

TAU = 4e-6          # the per-layer accuracy class of the product's convolutions, relative to max |u| of the layer
MODE = {"v": 0}     # 0 exact, 1 every element inside the band on the other branch, 2 a fixed half of them
COUNTS = {}


class KinkPReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, alpha, name):
        ctx.save_for_backward(u, alpha)
        ctx.name = name
        a = alpha.view(1, -1, 1, 1, 1)
        return torch.where(u > 0, u, a * u)

    @staticmethod
    def backward(ctx, g):
        u, alpha = ctx.saved_tensors
        pos = u > 0
        tau = TAU * float(u.abs().max())
        near = u.abs() < tau
        COUNTS[ctx.name] = (int(near.sum()), u.numel(), tau)
        if MODE["v"] == 2:
            # a fixed pseudo-random half: parity of a hash of the flat index
            idx = torch.arange(u.numel(), dtype=torch.int64).view(u.shape)
            near = near & (((idx * 2654435761) >> 7) % 2 == 0)
        if MODE["v"] != 0:
            pos = pos ^ near
        a = alpha.view(1, -1, 1, 1, 1)
        gu = torch.where(pos, g, a * g)
        ga = torch.where(pos, torch.zeros_like(g), g * u).sum(dim=(0, 2, 3, 4))
        return gu, ga, None


def patch(tm):
    for name, mod in tm.named_modules():
        if isinstance(mod, torch.nn.PReLU):
            mod.forward = (lambda m, n: (lambda x: KinkPReLU.apply(x, m.weight, n)))(mod, name)


def run(name):
    case = FC.build(name)
    c = case["cfg"]
    gold = np.load(os.path.join(HERE, c["golden"]))
    assert str(gold["digest"]) == FC.digest(case)
    torch.set_num_threads(os.cpu_count() or 1)
    VT.SLAB_BYTES = 2 << 30
    tm = FC.torch_model(case, torch.float64)
    patch(tm)
    t0 = time.time()
    x = torch.tensor(case["x"], dtype=torch.float64)
    y = torch.tensor(case["y"])
    outs = tm(x, case["masks"])
    loss, parts = FC.torch_losses(case, outs, y)
    ce, dl, per, w = parts[0]
    print("%s: forward %.0f s; ce %.9f dice %.9f" % (name, time.time() - t0, float(ce.detach()), float(dl.detach())), flush=True)
    assert abs(float(ce.detach()) - float(gold["ce"])) < 1e-12
    grads = {}
    for mode in (0, 1, 2):
        t0 = time.time()
        MODE["v"] = mode
        for prm in tm.parameters():
            prm.grad = None
        loss.backward(retain_graph=mode != 2)
        grads[mode] = {k: g.astype(np.float64) for k, g in tm.named_oracle_grads().items()}
        print("%s: backward mode %d %.0f s" % (name, mode, time.time() - t0), flush=True)
    # mode 0 must BE the committed float64 fixture
    for k, g in grads[0].items():
        idx = FC.sample_indices(k, g.size)
        mine = g.ravel() if idx is None else g.ravel()[idx]
        ref = gold["g/" + k]
        assert np.linalg.norm(mine - ref) <= 1e-9 * (np.linalg.norm(ref) + 1e-30) + 1e-18, k
    out = {"digest": np.array(FC.digest(case)), "tau": np.float64(TAU)}
    rows = []
    for k, g0 in grads[0].items():
        n0 = float(np.linalg.norm(g0.ravel()))
        if n0 < 1e-9 * max(1.0, np.sqrt(g0.size)):
            continue
        ea = float(np.linalg.norm((grads[1][k] - g0).ravel()) / n0)
        eh = float(np.linalg.norm((grads[2][k] - g0).ravel()) / n0)
        out["kink_all/" + k] = np.float64(ea)
        out["kink_half/" + k] = np.float64(eh)
        rows.append((max(ea, eh), k, ea, eh))
    for lname, (cnt, tot, tau) in COUNTS.items():
        out["near/" + lname] = np.array([cnt, tot], dtype=np.int64)
    path = os.path.join(HERE, "fullsize_%s_kink.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")
    print("elements inside the band |u| < %.0e max|u|: %s" % (TAU, ", ".join("%s %d/%d" % (k, v[0], v[1]) for k, v in COUNTS.items())))
    for m, k, ea, eh in sorted(rows, reverse=True)[:16]:
        print("  %-40s all %.2e  half %.2e" % (k, ea, eh))


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["vnet128", "mri"]):
        run(n)
