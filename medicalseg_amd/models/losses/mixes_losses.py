from ... import nn
from ...cvlibs import manager


@manager.LOSSES.add_component
class MixedLoss(nn.Layer):
    """sum_i coef_i * loss_i over ONE logits tensor (reference losses/mixes_losses.py:23-60).

    forward returns ``(loss_list, per_channel_dice)``: the weighted terms stay separate so that
    the train loop can log them individually; a member whose class is NAMED ``DiceLoss`` also
    returns its per-class dice, which is passed through (the reference dispatches on the class
    name too, :57).  Here every term is a device-side ``Scalar`` of the fused loss node."""

    def __init__(self, losses, coef):
        super().__init__()
        for what, value in (("losses", losses), ("coef", coef)):
            if not isinstance(value, list):
                raise TypeError('`{}` must be a list!'.format(what))
        if len(losses) != len(coef):
            raise ValueError('The length of `losses` should equal to `coef`, but they are {} and {}.'
                             .format(len(losses), len(coef)))
        self.losses, self.coef = losses, coef

    def forward(self, logits, labels):
        terms, dice_per_class = [], None
        for weight, member in zip(self.coef, self.losses):
            value = member(logits, labels)
            if type(member).__name__ == "DiceLoss":
                value, dice_per_class = value
            terms.append(value * weight)
        return terms, dice_per_class
