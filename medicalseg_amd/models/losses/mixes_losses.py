from ... import nn
from ...cvlibs import manager


@manager.LOSSES.add_component
class MixedLoss(nn.Layer):
    """Weighted list of losses (reference losses/mixes_losses.py:23-60); dispatch on the
    class NAME 'DiceLoss' to pick up its side output, like the reference (:57)."""

    def __init__(self, losses, coef):
        super(MixedLoss, self).__init__()
        if not isinstance(losses, list):
            raise TypeError('`losses` must be a list!')
        if not isinstance(coef, list):
            raise TypeError('`coef` must be a list!')
        len_losses = len(losses)
        len_coef = len(coef)
        if len_losses != len_coef:
            raise ValueError('The length of `losses` should equal to `coef`, but they are {} and {}.'
                             .format(len_losses, len_coef))
        self.losses = losses
        self.coef = coef

    def forward(self, logits, labels):
        loss_list = []
        per_channel_dice = None
        for i, loss in enumerate(self.losses):
            output = loss(logits, labels)
            if type(loss).__name__ == "DiceLoss":
                output, per_channel_dice = output
            loss_list.append(output * self.coef[i])
        return loss_list, per_channel_dice
