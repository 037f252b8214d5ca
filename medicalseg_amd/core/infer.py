"""Inference helpers (reference medicalseg/core/infer.py:62-94): forward + argmax."""
import collections.abc
import ctypes as C

from .. import nn
from ..device import IntTensor, Tensor


def get_reverse_list(ori_shape, transforms):
    """infer.py:21-40: [('resize', shape before that Resize3D), ...] in application order."""
    reverse_list = []
    d, h, w = ori_shape[0], ori_shape[1], ori_shape[2]
    for op in (transforms or []):
        if op.__class__.__name__ in ['Resize3D']:
            reverse_list.append(('resize', (d, h, w)))
            d, h, w = op.size[0], op.size[1], op.size[2]
    return reverse_list


def reverse_transform(pred, ori_shape, transforms, mode='trilinear'):
    """infer.py:43-59: undo the Resize3D ops of the val transforms on the logits, last one first, with
    F.interpolate(mode='trilinear', align_corners=False) -- the resize kernel of the deep-supervision heads
    (msk_interp_trilinear_fwd).  The reference's own call site passes mode='bilinear', which Paddle rejects for
    5-D input; the function's default mode is what is built, and 'bilinear' is read as it."""
    if mode not in ('trilinear', 'bilinear'):
        raise ValueError("reverse_transform supports mode='trilinear' only, got %r" % (mode,))
    for kind, (d, h, w) in get_reverse_list(ori_shape, transforms)[::-1]:
        if kind != 'resize':
            raise Exception("Unexpected info '{}' in im_info".format(kind))
        if (d, h, w) == (pred.d, pred.h, pred.w):
            continue
        out = Tensor.empty(pred.dev, pred.n, int(d), int(h), int(w), pred.c)
        pred.dev.call("msk_interp_trilinear_fwd", pred.msk(), out.msk())
        pred = out
    return pred


def inference(model, im, ori_shape=None, transforms=None):
    """Returns (pred int32 [N,1,D,H,W] on device, logits Tensor); with `ori_shape` different from the logits' and
    Resize3D ops in `transforms`, the logits are resized back first (infer.py:88-90).

    An eval-mode model runs its conv -> BN -> PReLU units as single folded convolutions here
    (nn.fused_inference); a model left in training mode runs the ordinary kernels."""
    with nn.fused_inference():
        logits = model(im)
    if not isinstance(logits, collections.abc.Sequence):
        raise TypeError("The type of logits must be one of collections.abc.Sequence, e.g. list, tuple. "
                        "But received {}".format(type(logits)))
    logit = logits[0]
    if ori_shape is not None and tuple(ori_shape) != tuple(logit.shape[2:]):
        logit = reverse_transform(logit, ori_shape, transforms, mode='bilinear')
    dev = logit.dev
    ptr = dev.arena.alloc(logit.voxels * 4)
    dev.call("msk_argmax_c", logit.msk(), C.c_void_p(ptr))
    pred = IntTensor(dev, ptr, (logit.n, 1, logit.d, logit.h, logit.w), dev.arena.gen)
    return pred, logit
