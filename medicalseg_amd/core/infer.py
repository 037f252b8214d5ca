"""Inference helpers (reference medicalseg/core/infer.py:62-94): forward + argmax."""
import collections.abc
import ctypes as C

from .. import nn
from ..device import IntTensor


def inference(model, im, ori_shape=None, transforms=None):
    """Returns (pred int32 [N,1,D,H,W] on device, logits Tensor).  Reverse-resize of the
    reference (infer.py:43-59,88-90) is only reachable with Resize3D in the val transforms,
    which no shipped config uses (and which fails upstream): it raises here.

    An eval-mode model runs its conv -> BN -> PReLU units as single folded convolutions here
    (nn.fused_inference); a model left in training mode runs the ordinary kernels."""
    with nn.fused_inference():
        logits = model(im)
    if not isinstance(logits, collections.abc.Sequence):
        raise TypeError("The type of logits must be one of collections.abc.Sequence, e.g. list, tuple. "
                        "But received {}".format(type(logits)))
    logit = logits[0]
    if ori_shape is not None and tuple(ori_shape) != tuple(logit.shape[2:]):
        raise NotImplementedError("reverse_transform (Resize3D in val transforms) is not built")
    dev = logit.dev
    ptr = dev.arena.alloc(logit.voxels * 4)
    dev.call("msk_argmax_c", logit.msk(), C.c_void_p(ptr))
    pred = IntTensor(dev, ptr, (logit.n, 1, logit.d, logit.h, logit.w), dev.arena.gen)
    return pred, logit
