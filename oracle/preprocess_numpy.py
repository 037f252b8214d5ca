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


# --------------------------------------------------------------------------
# loader augmentations (medicalseg/transforms/functional.py) -- SURVEY 8 f3
# PINNED: tests/golden/transforms_golden.npz (outputs of the reference's functional.py,
# captured by tests/golden/make_transforms_golden.py)
# --------------------------------------------------------------------------
def flip3d(img, axis):
    """functional.py:80-88."""
    return np.flip(np.asarray(img), axis)


def resized_crop3d(img, i, j, k, d, h, w, size, order):
    """functional.py:103-110: crop_3d then resize_3d == ndimage.zoom(mode='nearest') of the
    crop to ``size`` (same coordinate map as geometry.py's resample)."""
    crop = np.asarray(img)[i:i + d, j:j + h, k:k + w]
    return resample(crop, size, order=order)[0]


def cos_sin_deg(angle):
    """scipy.special.cosdg/sindg as used by ndimage.rotate: exact at multiples of 90 degrees
    (so right-angle rotations land on grid points), cos/sin of the radian angle otherwise."""
    a = float(angle) % 360.0
    if a % 90.0 == 0.0:
        return [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)][int(a // 90) % 4]
    r = np.deg2rad(float(angle))
    return float(np.cos(r)), float(np.sin(r))


def rotate3d(img, plane, angle, order=1, cval=0.0):
    """functional.py:91-100: scipy.ndimage.rotate(img, angle, axes=plane, order=1, cval=0,
    reshape=False) restated: per plane slice an affine map about the plane centre,
        in = M @ out + (c - M @ c),  M = [[cos, sin], [-sin, cos]],  c = (n - 1) / 2,
    linear interpolation inside the slice; mode='constant' does NOT interpolate beyond the edges:
    a sample whose coordinate is < 0 or > n-1 on either axis is ``cval`` (scipy map_coordinate);
    integer dtypes round half away from zero (scipy NI_GeometricTransform)."""
    img = np.asarray(img)
    a0, a1 = sorted(int(a) for a in plane)
    c, s = cos_sin_deg(angle)
    n0, n1 = img.shape[a0], img.shape[a1]
    m = np.array([[c, s], [-s, c]], dtype=np.float64)
    centre = (np.array([n0, n1], dtype=np.float64) - 1) / 2
    shift = centre - m @ centre
    o0, o1 = np.meshgrid(np.arange(n0, dtype=np.float64), np.arange(n1, dtype=np.float64), indexing="ij")
    c0 = shift[0] + o0 * m[0, 0] + o1 * m[0, 1]
    c1 = shift[1] + o0 * m[1, 0] + o1 * m[1, 1]
    arr = np.moveaxis(img, (a0, a1), (0, 1)).astype(np.float64)
    rest = arr.shape[2:]
    outside = (c0 < 0) | (c0 > n0 - 1) | (c1 < 0) | (c1 > n1 - 1)

    def tap(i0, i1):
        ok = (i0 >= 0) & (i0 < n0) & (i1 >= 0) & (i1 < n1)
        v = arr[np.clip(i0, 0, n0 - 1), np.clip(i1, 0, n1 - 1)]
        return np.where(ok.reshape(ok.shape + (1,) * len(rest)), v, float(cval))

    ex = lambda a: a.reshape(a.shape + (1,) * len(rest))
    if order == 0:
        out = tap(np.floor(c0 + 0.5).astype(np.int64), np.floor(c1 + 0.5).astype(np.int64))
    elif order == 1:
        f0, f1 = np.floor(c0).astype(np.int64), np.floor(c1).astype(np.int64)
        t0, t1 = c0 - f0, c1 - f1
        # scipy accumulates taps in (dim0, dim1) order with the product of the two weights
        out = (tap(f0, f1) * ex((1 - t0) * (1 - t1)) + tap(f0, f1 + 1) * ex((1 - t0) * t1) +
               tap(f0 + 1, f1) * ex(t0 * (1 - t1)) + tap(f0 + 1, f1 + 1) * ex(t0 * t1))
    else:
        raise ValueError("the reference rotates with order 1 only")
    out = np.where(ex(outside), float(cval), out)
    out = np.moveaxis(out, (0, 1), (a0, a1))
    if np.issubdtype(img.dtype, np.integer):
        out = np.where(out > 0, np.floor(out + 0.5), np.ceil(out - 0.5))
    return out.astype(img.dtype)
