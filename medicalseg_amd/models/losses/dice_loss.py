from ... import nn
from ...cvlibs import manager
from .fused import Scalar, node_for, per_channel_dice


@manager.LOSSES.add_component
class DiceLoss(nn.Layer):
    """Sigmoid-normalised V-Net dice with squared denominator (reference
    losses/dice_loss.py:24-102): per_channel = 2*sum(p*t)/clip(sum(p^2)+sum(t^2), 1e-6);
    loss = 1 - mean_c.  Returns (loss, per_channel_dice) like the reference (the second
    value is a lazily fetched host array)."""

    def __init__(self, sigmoid_norm=True, weight=None):
        super(DiceLoss, self).__init__()
        if not sigmoid_norm:
            raise NotImplementedError("softmax-normalised dice is not built: no shipped config sets "
                                      "sigmoid_norm=False (dice_loss.py:36-43)")
        if weight is not None:
            raise NotImplementedError("per-class dice weights are not built (unused by the shipped configs)")
        self.weight = weight
        self.eps = 1e-5

    def forward(self, logits, labels):
        node = node_for(logits, labels)
        loss = Scalar([(1.0, node, "dice")])
        return loss, per_channel_dice(node)
