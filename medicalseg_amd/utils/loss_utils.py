"""Pairs the model's output list with the configured loss list (reference
medicalseg/utils/loss_utils.py:16-52): output i is scored by ``losses['types'][i]`` and scaled by
``losses['coef'][i]``; returns the flat list of weighted terms plus the per-class dice of the
LAST dice-bearing loss.  As in the reference the dispatch is on the loss CLASS NAME, so
user-defined losses with those names take the same branches."""
from ..device import Tensor

_EDGE_LOSSES = ('BCELoss', 'FocalLoss')


def check_logits_losses(logits_list, losses):
    n_out, n_loss = len(logits_list), len(losses['types'])
    if n_out != n_loss:
        raise RuntimeError('The length of logits_list should equal to the types of loss config: {} != {}.'.format(
            n_out, n_loss))


def _score(kind, loss_fn, logits_list, i, labels, edges):
    """-> (list of unweighted terms, per-class dice or None) for output i."""
    logits = logits_list[i]
    if kind == 'MixedLoss':
        return loss_fn(logits, labels)
    if kind == 'DiceLoss':
        value, dice = loss_fn(logits, labels)
        return [value], dice
    if kind in _EDGE_LOSSES and getattr(loss_fn, 'edge_label', False):
        return [loss_fn(logits, edges)], None
    if kind == 'KLLoss':  # distillation: student vs detached teacher (outputs 0 and 1)
        return [loss_fn(logits_list[0], logits_list[1].detach())], None
    return [loss_fn(logits, labels)], None


def loss_computation(logits_list, labels, losses, edges=None):
    # evaluate() hands over the bare batch-1 logits tensor (core/val.py:118)
    if isinstance(logits_list, Tensor):
        logits_list = [logits_list]
    check_logits_losses(logits_list, losses)
    loss_list, per_channel_dice = [], None
    for i, (loss_fn, weight) in enumerate(zip(losses['types'], losses['coef'])):
        terms, dice = _score(type(loss_fn).__name__, loss_fn, logits_list, i, labels, edges)
        if dice is not None:
            per_channel_dice = dice
        loss_list.extend(weight * t for t in terms)
    return loss_list, per_channel_dice
