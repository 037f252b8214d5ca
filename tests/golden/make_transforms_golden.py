"""Generate tests/golden/transforms_golden.npz by running the REFERENCE's own
medicalseg/transforms/functional.py (flip_3d, rotate_3d, resized_crop_3d, resize_3d; numpy/scipy)
in the build container.  Only inputs, parameters and expected outputs are stored.

    python tests/golden/make_transforms_golden.py

functional.py imports SimpleITK at module level without using it in these functions; the module
is loaded by path with an empty placeholder for that one absent I/O library.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/medicalseg/transforms/functional.py"


def load_reference():
    sys.dont_write_bytecode = True
    try:
        import SimpleITK  # noqa: F401
    except Exception:
        sys.modules["SimpleITK"] = types.ModuleType("SimpleITK")
    spec = importlib.util.spec_from_file_location("ref_functional", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_classes(F):
    """transform.py (the classes) imports `medicalseg.cvlibs.manager` and the functional module through
    the package, whose __init__ needs paddle: register path-only packages and load the three files by path."""
    root = "/root/reference/medicalseg"
    for name, path in (("medicalseg", root), ("medicalseg.cvlibs", root + "/cvlibs"),
                       ("medicalseg.transforms", root + "/transforms")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("medicalseg.cvlibs.manager", root + "/cvlibs/manager.py")
    mgr = importlib.util.module_from_spec(spec)
    sys.modules["medicalseg.cvlibs.manager"] = mgr
    spec.loader.exec_module(mgr)
    sys.modules["medicalseg.cvlibs"].manager = mgr
    sys.modules["medicalseg.transforms.functional"] = F
    sys.modules["medicalseg.transforms"].functional = F
    spec = importlib.util.spec_from_file_location("medicalseg.transforms.transform", root + "/transforms/transform.py")
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    return T


def class_goldens(T, out):
    """Random-parameter streams and end-to-end outputs of the transform CLASSES under python's
    `random`/numpy seeds: a drop-in must draw the same numbers in the same order."""
    import random
    rng = np.random.default_rng(7)
    img = (rng.random((20, 18, 12)) * 255).astype(np.float32)
    lab = rng.integers(0, 4, (20, 18, 12)).astype(np.int32)
    out["cls_img"], out["cls_lab"] = img, lab
    # RandomResizedCrop3D.get_params (transform.py:246-282), default scale/ratio and a tight one
    for tag, kw in (("rrc_default", {}), ("rrc_small", {"scale": (0.3, 0.6), "ratio": (0.5, 2.0)})):
        t = T.RandomResizedCrop3D(size=(8, 8, 8), **kw)
        rows = []
        for seed in range(24):
            random.seed(seed)
            p = t.get_params(img, t.scale, t.ratio)
            rows.append(list(p))
        out[tag + "_params"] = np.array(rows, dtype=np.int64)
    random.seed(5)
    np.random.seed(5)
    o = T.RandomResizedCrop3D(size=(10, 9, 8), scale=(0.3, 0.6))(img, lab)
    out["rrc_call_img"], out["rrc_call_lab"] = o
    # pre_crop branch (transform.py:284-323): numpy RNG for the pre-crop box, python RNG afterwards
    random.seed(9)
    np.random.seed(9)
    o = T.RandomResizedCrop3D(size=(8, 8, 6), scale=(0.8, 1.2), pre_crop=True)(img, lab)
    out["rrc_precrop_img"], out["rrc_precrop_lab"] = o
    random.seed(10)
    np.random.seed(10)
    lab2 = np.zeros_like(lab)
    lab2[5:15, 4:12, 3:9] = lab[5:15, 4:12, 3:9] + 1
    out["cls_lab2"] = lab2
    o = T.RandomResizedCrop3D(size=(6, 6, 4), scale=(0.8, 1.2), pre_crop=True, nonzero_mask=True)(img, lab2)
    out["rrc_nonzero_img"], out["rrc_nonzero_lab"] = o
    # RandomRotation3D.get_params (transform.py:142-151)
    t = T.RandomRotation3D(degrees=30)
    rows = []
    for seed in range(16):
        random.seed(seed)
        a, pl = t.get_params(t.degrees)
        rows.append([a, pl[0], pl[1]])
    out["rot_params"] = np.array(rows, dtype=np.float64)
    random.seed(3)
    o = T.RandomRotation3D(degrees=(-10, 50), rotate_planes=[[0, 1], [1, 2]])(img, lab)
    out["rot_call_img"], out["rot_call_lab"] = o
    # RandomFlip3D (transform.py:186-204): axis draw THEN the coin
    rows = []
    t = T.RandomFlip3D(prob=0.5)
    for seed in range(16):
        random.seed(seed)
        o_img, o_lab = t(img, lab)
        ax = [a for a in range(3) if np.array_equal(o_img, np.flip(img, a))]
        rows.append([-1 if np.array_equal(o_img, img) else ax[0]])
    out["flip_axes"] = np.array(rows, dtype=np.int64)
    # Resize3D (transform.py:75-110) with an int size (shorter-side rule, functional.py:41-46) and a tuple
    out["resize_int_img"], out["resize_int_lab"] = T.Resize3D(6)(img, lab)
    out["resize_tuple_img"], out["resize_tuple_lab"] = T.Resize3D([9, 10, 11])(img, lab)
    # Compose tail (transform.py:64-69)
    random.seed(1)
    c_img, c_lab = T.Compose([T.RandomFlip3D(prob=1.0, flip_axis=1)])(img.copy(), lab.copy())
    out["compose_img"], out["compose_lab"] = c_img, c_lab


def main():
    F = load_reference()
    rng = np.random.default_rng(20260929)
    out = {}
    shapes = [(12, 10, 9), (16, 16, 6), (7, 13, 5)]
    for si, shp in enumerate(shapes):
        img = (rng.random(shp) * 255).astype(np.float32)
        lab = rng.integers(0, 5, shp).astype(np.int32)
        out[f"s{si}_img"], out[f"s{si}_lab"] = img, lab
        for ax in range(3):
            out[f"s{si}_flip{ax}_img"] = np.ascontiguousarray(F.flip_3d(img, ax))
            out[f"s{si}_flip{ax}_lab"] = np.ascontiguousarray(F.flip_3d(lab, ax))
        # transform.py:150-168: image AND label are rotated with order 1, cval 0
        rots = [([0, 1], 17.3), ([0, 2], -42.0), ([1, 2], 90.0), ([1, 2], 5.5), ([0, 1], -179.0), ([0, 2], 63.7)]
        out[f"s{si}_rot_params"] = np.array([[p[0], p[1], a] for p, a in rots], dtype=np.float64)
        for ri, (plane, ang) in enumerate(rots):
            out[f"s{si}_rot{ri}_img"] = F.rotate_3d(img, plane, ang)
            out[f"s{si}_rot{ri}_lab"] = F.rotate_3d(lab, plane, ang)
        # transform.py:334-339: image with `interpolation` (1), label with order 0
        d, h, w = shp
        crops = [(0, 0, 0, d, h, w, (8, 8, 8)), (2, 1, 0, d - 3, h - 2, w - 1, (16, 12, 10)),
                 (1, 3, 2, 4, 5, 3, (6, 6, 6)), (d - 1, h - 1, w - 1, 1, 1, 1, (3, 2, 2)), (0, 2, 1, d, 3, 3, (5, 7, 1))]
        out[f"s{si}_crop_params"] = np.array([c[:6] + c[6] for c in crops], dtype=np.int64)
        for ci, (i, j, k, cd, ch, cw, size) in enumerate(crops):
            out[f"s{si}_crop{ci}_img"] = F.resized_crop_3d(img, i, j, k, cd, ch, cw, size, 1)
            out[f"s{si}_crop{ci}_lab"] = F.resized_crop_3d(lab, i, j, k, cd, ch, cw, size, 0)
    class_goldens(load_reference_classes(F), out)
    np.savez_compressed(os.path.join(HERE, "transforms_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
