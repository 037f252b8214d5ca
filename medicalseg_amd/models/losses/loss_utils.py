"""Device counterparts of medicalseg/models/losses/loss_utils.py."""
import ctypes as C

import numpy as np

from ...device import Tensor


def class_weights(tensor: Tensor):
    """w_c = sum(1 - softmax_c) / sum(softmax_c) over all voxels (reference
    loss_utils.py:31-40); returns a persistent device pointer (the reference caches it)."""
    dev = tensor.dev
    ptr = dev.small(tensor.c)
    dev.call("msk_class_weights", tensor.msk(), C.c_void_p(ptr))
    return ptr


def flatten(tensor: Tensor):
    """(N, C, D, H, W) -> (C, N*D*H*W) on the host (reference loss_utils.py:18-28); only for
    inspection -- the fused loss kernel never materialises it."""
    a = tensor.numpy()
    return np.moveaxis(a, 1, 0).reshape(a.shape[1], -1)
