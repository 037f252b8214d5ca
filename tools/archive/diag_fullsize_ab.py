"""Two runs of the full-size parity case under different options, compared with each other: activations at the block
boundaries, their gradients, every parameter gradient.  Usage: python tools/diag_fullsize_ab.py "optA=val" "optB=val" """
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import fullsize_cases as FC  # noqa: E402
from helpers import dev  # noqa: E402


def l2(a, b):
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def run(case, opts):
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation
    d = dev()
    for k, v in opts:
        d.set_option(k, int(v))
    c = case["cfg"]
    model = VNet(elu=False, in_channels=1, num_classes=c["ncls"], kernel_size=c["K"], stride_size=c["S"])
    model.set_state_dict(case["params"])
    model.train()
    model.set_dropout_masks(case["masks"])
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    logits = model(case["x"])
    out = {}
    names = ("out16", "out32", "out64", "out128", "out256")
    for n, a in zip(names, model._acts):
        out["act/" + n] = a.numpy().copy()
    out["act/feat"] = model._feat.numpy().copy()
    out["act/logits"] = logits[0].numpy().copy()
    loss_list, per = loss_computation(logits, to_tensor(case["y"]), losses)
    model.clear_gradients()
    sum(loss_list).backward()
    d.sync()
    for n, a in zip(names, model._acts):
        g = getattr(a, "grad", None)
        if g is not None:
            out["dact/" + n] = g.numpy().copy()
    for blk in ("up_tr32", "up_tr64", "up_tr128", "up_tr256"):
        x = getattr(getattr(model, blk), "_x", None)
        if x is not None and getattr(x, "grad", None) is not None:
            out["dact/" + blk + "._x"] = x.grad.numpy().copy()
    for pname, p in model.named_parameters():
        out["g/" + pname] = p.grad_numpy().copy()
    return out


case = FC.build("vnet128")
sets = sys.argv[1:3]
parse = lambda s: [p.split("=") for p in s.split(",")] if s != "-" else []
A = run(case, parse(sets[0]))
B = run(case, parse(sets[1]))
for k in A:
    if k in B:
        e = l2(A[k], B[k])
        if not k.startswith("g/") or e > 2e-3:
            extra = ""
            if k.startswith("act/") or k.startswith("dact/"):
                a, b = A[k].astype(np.float64), B[k].astype(np.float64)
                ax = tuple(i for i in range(a.ndim) if i != 1)
                pc = np.sqrt(((a - b) ** 2).sum(axis=ax) / ((b ** 2).sum(axis=ax) + 1e-300))
                top = np.argsort(-pc)[:4]
                extra = " | worst channels " + " ".join("%d:%.1e" % (i, pc[i]) for i in top) + " | max|x| %.6g %.6g" % (np.abs(a).max(), np.abs(b).max())
            print("%-40s %.3e%s" % (k, e, extra), flush=True)
