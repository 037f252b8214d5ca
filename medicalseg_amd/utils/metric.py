"""Host-side metrics of the evaluation loop.

auc_roc: area under the ROC curve of the softmax scores collected by ``core.val.evaluate(auc_roc=True)`` -- the quantity the
reference takes from ``sklearn.metrics.roc_auc_score`` (medicalseg/utils/metric.py:64-107): binary = AUC of the class-1
score, more classes = one-vs-rest, macro average.  Computed here from the rank statistic (Mann-Whitney U with average
ranks for ties = the trapezoidal area sklearn integrates), numpy only.

Difference to the reference, on purpose: its function insists on 4-D (N, C, H, W) arrays (it was written for 2-D
segmentation) and therefore raises on the 5-D logits of every 3-D model of this package; here any (N, C, *spatial) layout
is accepted and flattened the same way."""
import numpy as np


def _average_ranks(x):
    """1-based ranks of x with ties sharing their average rank (scipy.stats.rankdata(method='average'))"""
    order = np.argsort(x, kind="mergesort")
    xs = x[order]
    n = xs.size
    start = np.flatnonzero(np.concatenate(([True], xs[1:] != xs[:-1])))       # first index of every run of equal values
    end = np.concatenate((start[1:], [n]))
    avg = (start + end + 1) / 2.0                                             # mean of the 1-based ranks start+1 .. end
    ranks = np.empty(n, dtype=np.float64)
    ranks[order] = np.repeat(avg, end - start)
    return ranks


def binary_auc(score, positive):
    """P(score of a positive > score of a negative) + 0.5 P(equal)"""
    positive = np.asarray(positive, dtype=bool).ravel()
    score = np.asarray(score, dtype=np.float64).ravel()
    n_pos = int(positive.sum())
    n_neg = positive.size - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    r = _average_ranks(score)
    return float((r[positive].sum() - n_pos * (n_pos + 1) / 2.0) / (float(n_pos) * n_neg))


def auc_roc(logits, label, num_classes, ignore_index=None):
    """logits: softmax scores (N, C, *spatial); label: (N, 1, *spatial) or (N, *spatial) class indices."""
    logits = np.asarray(logits)
    label = np.asarray(label)
    if ignore_index or len(np.unique(label)) > num_classes:
        raise RuntimeError('labels with ignore_index is not supported yet.')
    if logits.ndim < 3:
        raise ValueError('The shape of logits is not (N, C, *spatial), it is {}'.format(logits.shape))
    C = logits.shape[1]
    scores = np.moveaxis(logits, 1, -1).reshape(-1, C)
    lab = label.reshape(-1)
    if scores.shape[0] != lab.shape[0]:
        raise ValueError('length of `logit` and `label` should be equal, but they are {} and {}.'.format(scores.shape[0],
                                                                                                    lab.shape[0]))
    if num_classes == 2:
        return binary_auc(scores[:, 1], lab == 1)
    present = np.unique(lab)
    if len(present) != C:
        raise ValueError("Number of classes in y_true not equal to the number of columns in 'y_score'")
    return float(np.mean([binary_auc(scores[:, c], lab == c) for c in range(C)]))
