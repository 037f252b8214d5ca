"""Shared helpers for the GPU parity tests (call the C ABI through medicalseg_amd)."""
import ctypes as C

import numpy as np


def dev():
    from medicalseg_amd.device import get_device
    return get_device()


def t_from_ncdhw(a, ld=None):
    """numpy NCDHW -> persistent device Tensor (NDHWC, optional wider voxel stride)."""
    from medicalseg_amd.device import Tensor
    d = dev()
    a = np.ascontiguousarray(a, dtype=np.float32)
    n, c, D, H, W = a.shape
    ld = ld or c
    host = np.zeros((n, D, H, W, ld), dtype=np.float32)
    host[..., :c] = np.moveaxis(a, 1, -1)
    ptr = d.malloc(host.nbytes)
    d.h2d(ptr, host)
    return Tensor(d, ptr, n, D, H, W, c, ld, None)


def t_empty(n, c, D, H, W, ld=None, fill=None):
    from medicalseg_amd.device import Tensor
    d = dev()
    ld = ld or c
    ptr = d.malloc(n * D * H * W * ld * 4)
    if fill is not None:
        d.h2d(ptr, np.full((n, D, H, W, ld), fill, dtype=np.float32))
    return Tensor(d, ptr, n, D, H, W, c, ld, None)


def t_to_ncdhw(t):
    d = dev()
    full = d.d2h(t.ptr - 0, (t.n, t.d, t.h, t.w, t.ld), np.float32) if t.ld == t.c else None
    if full is not None:
        return np.moveaxis(full, -1, 1).copy()
    return t.numpy()


def vec(a):
    """1-D float array -> device pointer"""
    d = dev()
    a = np.ascontiguousarray(a, dtype=np.float32)
    ptr = d.malloc(max(a.nbytes, 16))
    if a.size:
        d.h2d(ptr, a)
    return ptr


def vec_back(ptr, n, dtype=np.float32):
    return dev().d2h(ptr, (n,), dtype)


def vp(ptr):
    return C.c_void_p(ptr) if ptr else None


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
