"""Root-cause probe for the BIMODAL full-size gradient error (round-4 verdict, Weak 1 / Next 3): the step of
tests/fullsize_cases.py::vnet128 lands either at ~9.6e-3 (down_tr128 tensors) or at ~2e-3 (down_tr256 tensors) against the
float64 fixture, depending only on the summation order of the forward BatchNorm statistics (option reduce_vpl_site).

  python tools/diag_fullsize_flip.py sweep          every option set of SWEEP: worst tensors against the fixture
  python tools/diag_fullsize_flip.py pair A B       two option sets ("k=v,k=v" or "-"): every intermediate tensor of the
                                                    down_tr128 / down_tr256 blocks (forward activations, pre-activations,
                                                    BatchNorm coefficients, gradients), A against B, plus PReLU sign flips
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import fullsize_cases as FC  # noqa: E402
from helpers import dev  # noqa: E402

SWEEP = ["-", "reduce_vpl_site=64", "conv_split=3", "conv_split=3,reduce_vpl_site=64", "conv_split=3,reduce_vpl_site=4",
         "conv_split=3,reduce_vpl_site=16", "conv_split=3,reduce_vpl_site=128", "reduce_vpl_site=4", "reduce_vpl_site=12",
         "bwd_fuse=0,reduce_vpl_site=0", "wino_bf3=0"]
RESET = {"reduce_vpl_site": 0, "conv_split": 2, "bwd_fuse": -1, "wino_bf3": 1, "dy_bound_shift": 0}


def l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def parse(s):
    return [p.split("=") for p in s.split(",")] if s != "-" else []


def run(case, opts, deep=False):
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation
    d = dev()
    for k, v in RESET.items():
        d.set_option(k, v)
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
    if deep:
        for bname in ("down_tr128", "down_tr256"):
            blk = getattr(model, bname)
            units = [("down", blk._down)] + [("ops%d" % i, op._unit) for i, op in enumerate(blk.ops)]
            for uname, u in units:
                out["fwd/%s.%s.y" % (bname, uname)] = u.y.numpy().copy()
                sc = u.bn.scratch(d)
                Cn = u.bn.num_features
                for nm in ("mean", "invstd", "scale", "shift"):
                    out["bn/%s.%s.%s" % (bname, uname, nm)] = d.d2h(sc[nm], (Cn,), np.float32)
            out["fwd/%s.out" % bname] = dict(zip(("out16", "out32", "out64", "out128", "out256"), model._acts))[
                "out128" if bname == "down_tr128" else "out256"].numpy().copy()
    loss_list, per = loss_computation(logits, to_tensor(case["y"]), losses)
    model.clear_gradients()
    sum(loss_list).backward()
    d.sync()
    if deep:
        for n, a in zip(("out16", "out32", "out64", "out128", "out256"), model._acts):
            if getattr(a, "grad", None) is not None:
                out["dact/" + n] = a.grad.numpy().copy()
        for bname in ("down_tr128", "down_tr256"):
            blk = getattr(model, bname)
            for i, op in enumerate(blk.ops):
                u = op._unit
                for nm, t in (("out.grad", getattr(u.out, "grad", None)), ("x.grad", getattr(u.x, "grad", None)),
                              ("dy", getattr(u, "dy", None))):
                    if t is not None:
                        out["bwd/%s.ops%d.%s" % (bname, i, nm)] = t.numpy().copy()
            if getattr(blk._t_down, "grad", None) is not None:
                out["bwd/%s.down.grad" % bname] = blk._t_down.grad.numpy().copy()
    for pname, p in model.named_parameters():
        out["g/" + pname] = p.grad_numpy().copy()
    for k, v in RESET.items():
        d.set_option(k, v)
    return out


def against_fixture(case, res):
    gold = np.load(os.path.join(HERE, "..", "tests", "golden", case["cfg"]["golden"]))
    errs = {}
    for k, g in res.items():
        if not k.startswith("g/"):
            continue
        pname = k[2:]
        ref = gold["g/" + pname]
        if float(gold["gn/" + pname]) < 1e-9 * max(1.0, np.sqrt(ref.size)):
            continue
        idx = FC.sample_indices(pname, g.size)
        gg = g.astype(np.float64).ravel()
        errs[pname] = l2(gg if idx is None else gg[idx], ref)
    return errs


def main():
    case = FC.build("vnet128")
    mode = sys.argv[1]
    if mode == "sweep":
        for s in (sys.argv[2:] or SWEEP):
            e = against_fixture(case, run(case, parse(s)))
            top = sorted(e, key=e.get, reverse=True)[:4]
            print("[%s] median %.2e | %s | down_tr128.ops.0.conv1.weight %.2e" % (s, float(np.median(list(e.values()))),
                  " | ".join("%s %.2e" % (k, e[k]) for k in top), e["down_tr128.ops.0.conv1.weight"]), flush=True)
        return
    A = run(case, parse(sys.argv[2]), deep=True)
    B = run(case, parse(sys.argv[3]), deep=True)
    ea, eb = against_fixture(case, A), against_fixture(case, B)
    print("A [%s]: worst %s" % (sys.argv[2], max(ea.items(), key=lambda kv: kv[1])))
    print("B [%s]: worst %s" % (sys.argv[3], max(eb.items(), key=lambda kv: kv[1])))
    for k in A:
        if k not in B or k.startswith("g/"):
            continue
        a, b = A[k].astype(np.float64), B[k].astype(np.float64)
        e = l2(a, b)
        extra = ""
        if a.ndim == 5:
            ax = (0, 2, 3, 4)
            pc = np.sqrt(((a - b) ** 2).sum(axis=ax) / ((b ** 2).sum(axis=ax) + 1e-300))
            top = np.argsort(-pc)[:4]
            extra = " | worst channels " + " ".join("%d:%.1e" % (i, pc[i]) for i in top) + " | max|x| %.6g" % np.abs(b).max()
            if k.startswith("fwd/") and k.endswith(".y"):
                # PReLU kink: sign of the BatchNorm output scale*y + shift
                pre = k[len("fwd/"):-2]
                ua = a * A["bn/%s.scale" % pre][None, :, None, None, None] + A["bn/%s.shift" % pre][None, :, None, None, None]
                ub = b * B["bn/%s.scale" % pre][None, :, None, None, None] + B["bn/%s.shift" % pre][None, :, None, None, None]
                flips = int(((ua > 0) != (ub > 0)).sum())
                extra += " | PReLU sign flips %d of %d (min |u| %.2e)" % (flips, ua.size, float(np.abs(ub).min()))
        elif a.ndim == 1:
            i = int(np.argmax(np.abs(a - b) / (np.abs(b) + 1e-30)))
            extra = " | worst channel %d: %.9g vs %.9g | min %.4g max %.4g" % (i, a[i], b[i], b.min(), b.max())
        print("%-44s %.3e%s" % (k, e, extra), flush=True)
    # PReLU kink: for every LUConv unit, the voxels whose pre-activation u = scale*y + shift changes sign between A and B, with
    # the gradient that arrives there -- du = g * (u > 0 ? 1 : alpha) changes by (1 - alpha) * g at such a voxel
    for bname in ("down_tr128", "down_tr256"):
        for i in range(3):
            pre = "%s.ops%d" % (bname, i)
            if "fwd/%s.y" % pre not in A or "bwd/%s.out.grad" % pre not in A:
                continue
            ya, yb = A["fwd/%s.y" % pre].astype(np.float64), B["fwd/%s.y" % pre].astype(np.float64)
            bc = lambda v: v[None, :, None, None, None].astype(np.float64)
            ua = ya * bc(A["bn/%s.scale" % pre]) + bc(A["bn/%s.shift" % pre])
            ub = yb * bc(B["bn/%s.scale" % pre]) + bc(B["bn/%s.shift" % pre])
            ga, gb = A["bwd/%s.out.grad" % pre].astype(np.float64), B["bwd/%s.out.grad" % pre].astype(np.float64)
            al = 0.25
            dua, dub = ga * np.where(ua > 0, 1.0, al), gb * np.where(ub > 0, 1.0, al)
            flip = (ua > 0) != (ub > 0)
            tot = np.linalg.norm(dua - dub) / np.linalg.norm(dub)
            at = np.linalg.norm((dua - dub)[flip]) / np.linalg.norm(dub)
            print("kink %-18s: %d sign flips; ||du_A - du_B|| / ||du_B|| = %.3e, of which at the flipped voxels %.3e; rms|g| %.3e max|g| %.3e"
                  % (pre, int(flip.sum()), tot, at, float(np.sqrt((gb ** 2).mean())), float(np.abs(gb).max())))
            for idx in zip(*np.nonzero(flip)):
                print("     voxel (n, c, d, h, w) = %s: u_A %+.3e u_B %+.3e | g %.3e = %.0f x rms (share of ||du||^2: %.2e)"
                      % (idx, ua[idx], ub[idx], gb[idx], abs(gb[idx]) / np.sqrt((gb ** 2).mean()), (0.75 * gb[idx]) ** 2 / (dub ** 2).sum()))
    for k in sorted(ea):
        if k.startswith("down_tr128") or k.startswith("down_tr256.down"):
            print("g/%-40s A %.2e  B %.2e  A-vs-B %.2e" % (k, ea[k], eb[k], l2(A["g/" + k], B["g/" + k])))


if __name__ == "__main__":
    main()
