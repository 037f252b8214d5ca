"""logits list x loss list -> (loss_list, per_channel_dice): reference
medicalseg/utils/loss_utils.py:16-52.  Dispatch is on the loss CLASS NAME."""
from ..device import Tensor


def check_logits_losses(logits_list, losses):
    len_logits = len(logits_list)
    len_losses = len(losses['types'])
    if len_logits != len_losses:
        raise RuntimeError('The length of logits_list should equal to the types of loss config: {} != {}.'.format(
            len_logits, len_losses))


def loss_computation(logits_list, labels, losses, edges=None):
    # evaluate() passes the bare batch-1 logits tensor (core/val.py:118); the reference then
    # indexes away the batch axis and the losses unsqueeze it back (dice_loss.py:83-84).
    if isinstance(logits_list, Tensor):
        logits_list = [logits_list]
    check_logits_losses(logits_list, losses)
    loss_list = []
    per_channel_dice = None
    for i in range(len(logits_list)):
        logits = logits_list[i]
        loss_i = losses['types'][i]
        coef_i = losses['coef'][i]
        name = loss_i.__class__.__name__
        if name in ('BCELoss', 'FocalLoss') and getattr(loss_i, 'edge_label', False):
            loss_list.append(coef_i * loss_i(logits, edges))
        elif name == 'MixedLoss':
            mixed_loss_list, per_channel_dice = loss_i(logits, labels)
            for mixed_loss in mixed_loss_list:
                loss_list.append(coef_i * mixed_loss)
        elif name in ("KLLoss", ):
            loss_list.append(coef_i * loss_i(logits_list[0], logits_list[1].detach()))
        elif name == "DiceLoss":
            loss, per_channel_dice = loss_i(logits, labels)
            loss_list.append(coef_i * loss)
        else:
            loss_list.append(coef_i * loss_i(logits, labels))
    return loss_list, per_channel_dice
