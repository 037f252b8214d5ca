import numpy as np

from ... import nn
from ...cvlibs import manager
from .fused import Scalar, node_for
from .loss_utils import class_weights


@manager.LOSSES.add_component
class CrossEntropyLoss(nn.Layer):
    """Class-weighted cross entropy (reference losses/cross_entropy_loss.py:24-87).

    weight None -> computed ONCE from the first batch's logits and cached
    (cross_entropy_loss.py:68-69); ignore_index voxels are excluded from numerator and
    denominator; logits get +1e-8 like the reference (:79)."""

    def __init__(self, weight=None, ignore_index=255, data_format='NCDHW'):
        super(CrossEntropyLoss, self).__init__()
        self.ignore_index = ignore_index
        self.EPS = 1e-8
        self.data_format = data_format
        self.weight = None if weight is None else np.asarray(weight, dtype=np.float32)
        self._weight_ptr = None

    def forward(self, logit, label):
        if self.data_format not in ('NCDHW', ):
            raise ValueError("only data_format='NCDHW' is supported at the boundary")
        node = node_for(logit, label)
        if self._weight_ptr is None:
            if self.weight is None:
                self._weight_ptr = class_weights(logit)
                self.weight = node.dev.d2h(self._weight_ptr, (logit.c,), np.float32)
            else:
                self._weight_ptr = node.dev.small(len(self.weight))
                node.dev.h2d(self._weight_ptr, self.weight)
        if self.weight is not None and logit.c != len(self.weight):
            raise ValueError('The number of weights = {} must be the same as the number of classes = {}.'
                             .format(len(self.weight), logit.c))
        node.weights_ptr = self._weight_ptr
        node.ignore_index = self.ignore_index
        return Scalar([(1.0, node, "ce")])
