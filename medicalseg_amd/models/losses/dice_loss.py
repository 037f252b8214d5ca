import numpy as np

from ... import nn
from ...cvlibs import manager
from .fused import Scalar, node_for, per_channel_dice


@manager.LOSSES.add_component
class DiceLoss(nn.Layer):
    """V-Net dice with squared denominator (reference losses/dice_loss.py:24-102):
    per_channel = 2*w_c*sum(p*t)/clip(sum(p^2)+sum(t^2), 1e-6); loss = 1 - mean_c; p = sigmoid(logits)
    (sigmoid_norm=True, the shipped configs) or softmax over the classes (dice_loss.py:40-43); `weight` is the
    optional per-class factor on the intersection (dice_loss.py:68-69).  Returns (loss, per_channel_dice) like
    the reference (the second value is a lazily fetched host array)."""

    def __init__(self, sigmoid_norm=True, weight=None):
        super(DiceLoss, self).__init__()
        self.sigmoid_norm = bool(sigmoid_norm)
        self.weight = None if weight is None else np.asarray(weight, dtype=np.float32).reshape(-1)
        self._weight_dev = None
        self.eps = 1e-5

    def forward(self, logits, labels):
        node = node_for(logits, labels)
        node.dice_softmax = not self.sigmoid_norm
        if self.weight is not None:
            if self.weight.size != logits.c:
                raise ValueError("DiceLoss weight has %d entries for %d classes" % (self.weight.size, logits.c))
            if self._weight_dev is None or self._weight_dev[0] is not logits.dev:
                ptr = logits.dev.malloc(max(self.weight.size, 4) * 4)
                logits.dev.h2d(ptr, self.weight)
                self._weight_dev = (logits.dev, ptr)
            node.dice_weight_ptr = self._weight_dev[1]
        loss = Scalar([(1.0, node, "dice")])
        return loss, per_channel_dice(node)
