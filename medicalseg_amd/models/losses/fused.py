"""Device side of the loss: one fused statistics kernel + one backward kernel shared by
CrossEntropyLoss and DiceLoss, and the scalar objects the training loop manipulates
(``sum(loss_list)``, ``coef * loss``, ``loss.backward()``, ``loss.numpy()[0]``;
reference core/train.py:135-139,158)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ...device import IntTensor, LazyArray, Tensor, to_tensor


class LossNode:
    """Fused CE + Dice evaluation for one (logits, labels) pair."""

    def __init__(self, logits: Tensor, labels: IntTensor):
        if tuple(labels.shape) != (logits.n, logits.d, logits.h, logits.w):
            raise ValueError(f"label shape {labels.shape} does not match logits {logits.shape}")
        self.logits, self.labels = logits, labels
        self.dev = logits.dev
        self.C = logits.c
        self.weights_ptr = None
        self.ignore_index = 255
        self.dice_softmax = False      # DiceLoss(sigmoid_norm=False)
        self.dice_weight_ptr = None    # DiceLoss(weight=[...]) on the device
        self.out_ptr = None
        self.stats_ptr = None
        self.evaluated_with = None

    def evaluate(self):
        key = (self.weights_ptr, self.dice_softmax, self.dice_weight_ptr)
        if self.evaluated_with == (key, self.ignore_index) and self.out_ptr is not None:
            return
        dev, Cn = self.dev, self.C
        if self.out_ptr is None:
            self.out_ptr = dev.arena.alloc((2 + Cn) * 4)
            self.stats_ptr = dev.arena.alloc((3 * Cn + 2) * 8)
        w = self.weights_ptr
        if w is None:  # Dice alone: CE term unused, any weights do
            w = dev.arena.alloc(Cn * 4)
            dev.h2d(w, np.ones(Cn, dtype=np.float32))
            self._dummy_w = w
        dev.call("msk_loss_fwd_ex", self.logits.msk(), C.c_void_p(self.labels.ptr), C.c_void_p(w),
                 int(self.ignore_index), int(self.dice_softmax), C.c_void_p(self.dice_weight_ptr),
                 C.c_void_p(self.out_ptr), C.c_void_p(self.stats_ptr))
        self.evaluated_with = (key, self.ignore_index)

    def backward(self, coef_ce: float, coef_dice: float):
        self.evaluate()
        dev = self.dev
        w = self.weights_ptr if self.weights_ptr is not None else self._dummy_w
        dz = self.logits.empty_like()
        dev.call("msk_loss_bwd_ex", self.logits.msk(), C.c_void_p(self.labels.ptr), C.c_void_p(w),
                 int(self.ignore_index), int(self.dice_softmax), C.c_void_p(self.dice_weight_ptr),
                 C.c_void_p(self.stats_ptr), C.c_float(coef_ce), C.c_float(coef_dice), dz.msk())
        self.logits.grad = dz
        self.logits.grad_written = True
        return dz


def node_for(logits: Tensor, labels) -> LossNode:
    labels = to_tensor(labels, logits.dev)
    key = (labels.ptr, tuple(labels.shape))
    node = _NODE_CACHE.get((id(logits), logits.ptr, logits.gen) + key)
    if node is None or node.logits is not logits:
        if len(_NODE_CACHE) > 8:
            _NODE_CACHE.clear()
        node = LossNode(logits, labels)
        _NODE_CACHE[(id(logits), logits.ptr, logits.gen) + key] = node
    return node


_NODE_CACHE = {}


class Scalar:
    """A scalar loss = sum_i coef_i * term_i, term = (LossNode, 'ce'|'dice').  Values stay on
    the device until ``numpy()``/``float()`` (one sync), so the train loop can defer host
    syncs to log boundaries."""

    def __init__(self, terms):
        self.terms = list(terms)  # [(coef, node, which)]

    # arithmetic used by MixedLoss / loss_computation / sum()
    def __mul__(self, k):
        k = float(k)
        return Scalar([(c * k, n, w) for c, n, w in self.terms])

    __rmul__ = __mul__

    def __add__(self, other):
        if isinstance(other, (int, float)):
            if other != 0:
                raise TypeError("only sum()'s zero start value can be added to a loss")
            return self
        return Scalar(self.terms + other.terms)

    __radd__ = __add__

    def __truediv__(self, k):
        return self * (1.0 / float(k))

    def value(self) -> float:
        tot = 0.0
        for c, node, which in self.terms:
            node.evaluate()
            v = node.dev.d2h(node.out_ptr, (2,), np.float32)
            tot += c * float(v[0 if which == "ce" else 1])
        return tot

    def numpy(self):
        return np.array([self.value()], dtype=np.float32)

    def item(self):
        return self.value()

    __float__ = value

    def backward(self):
        groups = {}
        for c, node, which in self.terms:
            g = groups.setdefault(id(node), [node, 0.0, 0.0])
            if which == "ce":
                g[1] += c
            else:
                g[2] += c
        # dL/dlogits per evaluated output, then ONE backward per producing model: a
        # multi-output model (VNetDeepSup.num_outputs = 4) receives the list in forward order
        producers = {}
        for node, cce, cdice in groups.values():
            dz = node.backward(cce, cdice)
            prod = node.logits.producer
            if prod is not None:
                producers.setdefault(id(prod), (prod, {}))[1][getattr(node.logits, "out_index", 0)] = dz
        for prod, grads in producers.values():
            n_out = getattr(prod, "num_outputs", 1)
            if n_out == 1:
                prod.backward(grads[0])
            else:
                prod.backward([grads.get(i) for i in range(n_out)])

    def detach(self):
        return self

    def __repr__(self):
        return f"Scalar({self.value():.6f})"


def per_channel_dice(node: LossNode) -> LazyArray:
    node.evaluate()
    return LazyArray(node.dev, node.out_ptr + 8, node.C)
