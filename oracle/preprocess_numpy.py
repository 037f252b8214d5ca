"""numpy restatement of the reference's volumetric preprocessing (test-only oracle).

PINNED: tests/golden/preprocess_golden.npz holds outputs of the reference's own
tools/preprocess_utils/{geometry,values}.py (numpy/scipy backend) captured in the
build container by tests/golden/make_preprocess_golden.py.

Restated without scipy so that the arithmetic the HIP kernels must reproduce is
explicit.
"""
from __future__ import annotations

import numpy as np


def zoom_coords(n_in: int, n_out: int) -> np.ndarray:
    """scipy.ndimage.zoom(grid_mode=False) coordinate map used by
    geometry.py:66-67: output index o -> input coordinate o*(n_in-1)/(n_out-1)
    (align-corners); 0 when n_out == 1."""
    if n_out <= 1:
        return np.zeros(n_out, dtype=np.float64)
    return np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1))


def resample(image, new_shape, order=1, spacing=None):
    """tools/preprocess_utils/geometry.py:31-69 with ``new_shape`` given.

    order 0: nearest = floor(c + 0.5); order 1: separable linear between
    floor(c) and min(floor(c)+1, n_in-1).  dtype preserved.  Returns
    (array, new_spacing) like the reference."""
    image = np.asarray(image)
    new_shape = [int(s) for s in new_shape]
    if spacing is not None and len(spacing) == 4:  # geometry.py:59-60
        spacing = spacing[1:]
    new_spacing = (tuple((np.array(image.shape) / np.array(new_shape)) * np.array(spacing))
                   if spacing is not None else None)
    out = image
    for ax in range(3):
        n_in, n_out = out.shape[ax], new_shape[ax]
        c = zoom_coords(n_in, n_out)
        if order == 0:
            idx = np.floor(c + 0.5).astype(np.int64)
            idx = np.clip(idx, 0, n_in - 1)
            out = np.take(out, idx, axis=ax)
        elif order == 1:
            i0 = np.clip(np.floor(c).astype(np.int64), 0, n_in - 1)
            i1 = np.minimum(i0 + 1, n_in - 1)
            t = c - i0
            shp = [1, 1, 1]
            shp[ax] = n_out
            t = t.reshape(shp)
            a = np.take(out, i0, axis=ax).astype(np.float64)
            b = np.take(out, i1, axis=ax).astype(np.float64)
            out = a * (1.0 - t) + b * t
        else:
            raise ValueError("only order 0/1 are used by the reference pipelines")
    if order == 1:
        if np.issubdtype(image.dtype, np.integer):
            out = np.rint(out)
        out = out.astype(image.dtype)
    return out, new_spacing


def HUnorm(image, HU_min=-1200, HU_max=600, HU_nan=-2000):
    """tools/preprocess_utils/values.py:67-87."""
    image = np.array(image, copy=True)
    image = np.nan_to_num(image, copy=False, nan=HU_nan)
    image = (image - HU_min) / ((HU_max - HU_min) / 255)
    np.clip(image, 0, 255, out=image)
    return image


def normalize(image, min_val=None, max_val=None):
    """tools/preprocess_utils/values.py:54-64."""
    image = np.asarray(image)
    if min_val is None and max_val is None:
        image = (image - image.min()) / (image.max() - image.min())
    else:
        image = (image - min_val) / (max_val - min_val)
    np.clip(image, 0, 1, out=image)
    return image


def label_remap(label, map_dict):
    """tools/preprocess_utils/values.py:37-51 (sequential in-place remap)."""
    label = np.array(label, copy=True)
    for key, val in map_dict.items():
        label[label == key] = val
    return label


def max_normalize(im):
    """medicalseg/transforms/transform.py:67-69: im/im.max() when max > 0, then
    expand_dims(axis=0)."""
    im = np.asarray(im)
    if np.max(im) > 0:
        im = im / np.max(im)
    return np.expand_dims(im, axis=0)
