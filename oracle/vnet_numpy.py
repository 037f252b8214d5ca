"""numpy restatement of the reference VNet training step (CPU oracle, test-only).

PARITY UNPINNED BY THE REFERENCE (no Paddle, no reference tests; see
oracle/__init__.py).  Cross-checked against torch-CPU in tests/test_oracle.py.

Layouts are the reference's: activations NCDHW, Conv3D weight
[Cout, Cin, kD, kH, kW], Conv3DTranspose weight [Cin, Cout, kD, kH, kW],
labels N x D x H x W integers.  Every function cites the reference lines it
restates.  Arithmetic runs in ``dtype`` (float64 by default so the oracle is a
tight reference for the fp32 HIP path).

A tiny reverse-mode tape records one closure per op so that the backward pass
is the exact adjoint of what the forward executed.
"""
from __future__ import annotations

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

BN_EPS = 1e-5  # paddle.nn.BatchNorm3D default epsilon
BN_MOMENTUM = 0.9  # paddle momentum: running = 0.9*running + 0.1*batch


# --------------------------------------------------------------------------
# primitive ops (forward + explicit gradients)
# --------------------------------------------------------------------------
def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return tuple(int(i) for i in v)
    return (int(v),) * 3


def conv3d(x, w, b=None, stride=1, padding=0):
    """paddle.nn.Conv3D: cross-correlation, zero padding (vnet.py:36,67,98,165)."""
    s, p = _triple(stride), _triple(padding)
    xp = np.pad(x, ((0, 0), (0, 0), (p[0], p[0]), (p[1], p[1]), (p[2], p[2])))
    kd, kh, kw = w.shape[2:]
    win = sliding_window_view(xp, (kd, kh, kw), axis=(2, 3, 4))
    win = win[:, :, ::s[0], ::s[1], ::s[2]]
    # win: [n, ci, od, oh, ow, kd, kh, kw]
    y = np.tensordot(win, w, axes=([1, 5, 6, 7], [1, 2, 3, 4]))  # n,od,oh,ow,co
    y = np.moveaxis(y, -1, 1)
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1, 1)
    return np.ascontiguousarray(y)


def conv3d_dgrad(dy, w, x_shape, stride=1, padding=0):
    s, p = _triple(stride), _triple(padding)
    n, ci, D, H, W = x_shape
    kd, kh, kw = w.shape[2:]
    od, oh, ow = dy.shape[2:]
    dxp = np.zeros((n, ci, D + 2 * p[0], H + 2 * p[1], W + 2 * p[2]), dtype=dy.dtype)
    dyl = np.moveaxis(dy, 1, -1)  # n,od,oh,ow,co
    for a in range(kd):
        for bb in range(kh):
            for c in range(kw):
                t = np.tensordot(dyl, w[:, :, a, bb, c], axes=([4], [0]))  # n,od,oh,ow,ci
                dxp[:, :, a:a + od * s[0]:s[0], bb:bb + oh * s[1]:s[1],
                    c:c + ow * s[2]:s[2]] += np.moveaxis(t, -1, 1)
    return np.ascontiguousarray(
        dxp[:, :, p[0]:p[0] + D, p[1]:p[1] + H, p[2]:p[2] + W])


def conv3d_wgrad(dy, x, k, stride=1, padding=0):
    s, p = _triple(stride), _triple(padding)
    kd, kh, kw = k
    xp = np.pad(x, ((0, 0), (0, 0), (p[0], p[0]), (p[1], p[1]), (p[2], p[2])))
    win = sliding_window_view(xp, (kd, kh, kw), axis=(2, 3, 4))
    od, oh, ow = dy.shape[2:]
    win = win[:, :, ::s[0], ::s[1], ::s[2]][:, :, :od, :oh, :ow]
    dw = np.tensordot(dy, win, axes=([0, 2, 3, 4], [0, 2, 3, 4]))  # co,ci,kd,kh,kw
    db = dy.sum(axis=(0, 2, 3, 4))
    return dw, db


def conv_transpose3d(x, w, b=None, stride=1):
    """paddle.nn.Conv3DTranspose, padding 0, output_padding 0 (vnet.py:133-137).

    out = (in-1)*s + k;  weight [Cin, Cout, kD, kH, kW] (App. B.1).
    Implemented as the adjoint of a strided Conv3D whose weight is
    [out=Cin, in=Cout]."""
    s = _triple(stride)
    n, ci, D, H, W = x.shape
    co = w.shape[1]
    kd, kh, kw = w.shape[2:]
    out_shape = (n, co, (D - 1) * s[0] + kd, (H - 1) * s[1] + kh, (W - 1) * s[2] + kw)
    y = conv3d_dgrad(x, w, out_shape, stride=s, padding=0)
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1, 1)
    return y


def conv_transpose3d_dgrad(dy, w, stride=1):
    return conv3d(dy, w, None, stride=stride, padding=0)


def conv_transpose3d_wgrad(dy, x, k, stride=1):
    dw, _ = conv3d_wgrad(x, dy, k, stride=stride, padding=0)  # [ci, co, k]
    db = dy.sum(axis=(0, 2, 3, 4))
    return dw, db


def bn_train(x, gamma, beta, eps=BN_EPS):
    """BatchNorm3D / SyncBatchNorm training forward (App. B.2): biased batch
    variance over (N, D, H, W)."""
    ax = (0, 2, 3, 4)
    mean = x.mean(axis=ax)
    var = x.var(axis=ax)  # biased
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean.reshape(1, -1, 1, 1, 1)) * invstd.reshape(1, -1, 1, 1, 1)
    y = xhat * gamma.reshape(1, -1, 1, 1, 1) + beta.reshape(1, -1, 1, 1, 1)
    return y, xhat, mean, var, invstd


def bn_eval(x, gamma, beta, rmean, rvar, eps=BN_EPS):
    invstd = 1.0 / np.sqrt(rvar + eps)
    return ((x - rmean.reshape(1, -1, 1, 1, 1)) * (invstd * gamma).reshape(1, -1, 1, 1, 1)
            + beta.reshape(1, -1, 1, 1, 1))


def bn_train_bwd(dy, xhat, gamma, invstd):
    ax = (0, 2, 3, 4)
    m = dy.size // dy.shape[1]
    dgamma = (dy * xhat).sum(axis=ax)
    dbeta = dy.sum(axis=ax)
    dx = (gamma * invstd / m).reshape(1, -1, 1, 1, 1) * (
        m * dy - dbeta.reshape(1, -1, 1, 1, 1) - xhat * dgamma.reshape(1, -1, 1, 1, 1))
    return dx, dgamma, dbeta


def prelu(x, alpha):
    """paddle.nn.PReLU(num_parameters=C): x>0 ? x : alpha_c*x (App. B.3)."""
    a = alpha.reshape(1, -1, 1, 1, 1)
    return np.where(x > 0, x, a * x)


def prelu_bwd(dy, x, alpha):
    a = alpha.reshape(1, -1, 1, 1, 1)
    neg = ~(x > 0)
    dx = np.where(neg, a * dy, dy)
    dalpha = (dy * x * neg).sum(axis=(0, 2, 3, 4))
    return dx, dalpha


# --------------------------------------------------------------------------
# tape
# --------------------------------------------------------------------------
class Tape:
    def __init__(self):
        self.ops = []

    def record(self, fn):
        self.ops.append(fn)

    def backward(self):
        for fn in reversed(self.ops):
            fn()


class Var:
    """Activation with an accumulating gradient slot."""
    __slots__ = ("v", "g")

    def __init__(self, v):
        self.v = v
        self.g = None

    def acc(self, g):
        self.g = g if self.g is None else self.g + g


# --------------------------------------------------------------------------
# VNet (models/vnet.py:32-267)
# --------------------------------------------------------------------------
DEFAULT_K = ((2, 2, 2),) * 4
DEFAULT_S = ((2, 2, 2),) * 4


def param_specs(in_channels=1, num_classes=4, kernel_size=DEFAULT_K, stride_size=DEFAULT_S):
    """Ordered list of (state_dict name, shape, kind) mirroring the reference's
    attribute tree (App. B.7).  kind in {conv_w, convT_w, bias, bn_w, bn_b,
    bn_mean, bn_var, prelu}."""
    specs = []

    def conv(name, co, ci, k):
        k = _triple(k)
        specs.append((name + ".weight", (co, ci) + k, "conv_w"))
        specs.append((name + ".bias", (co,), "bias"))

    def convT(name, ci, co, k):
        k = _triple(k)
        specs.append((name + ".weight", (ci, co) + k, "convT_w"))
        specs.append((name + ".bias", (co,), "bias"))

    def bn(name, c):
        specs.append((name + ".weight", (c,), "bn_w"))
        specs.append((name + ".bias", (c,), "bn_b"))
        specs.append((name + "._mean", (c,), "bn_mean"))
        specs.append((name + "._variance", (c,), "bn_var"))

    def pr(name, c):
        specs.append((name + "._weight", (c,), "prelu"))

    def luconv(name, c):  # LUConv vnet.py:32-43 (attribute order relu1, conv1, bn1)
        pr(name + ".relu1", c)
        conv(name + ".conv1", c, c, 5)
        bn(name + ".bn1", c)

    # InputTransition vnet.py:57-79
    conv("in_tr.conv1", 16, in_channels, 5)
    bn("in_tr.bn1", 16)
    pr("in_tr.relu1", 16)

    def down(name, cin, nconv, k):  # DownTransition vnet.py:82-113
        co = 2 * cin
        conv(name + ".down_conv", co, cin, k)
        bn(name + ".bn1", co)
        pr(name + ".relu1", co)
        pr(name + ".relu2", co)
        for i in range(nconv):
            luconv(f"{name}.ops.{i}", co)

    def up(name, cin, cout, nconv, k):  # UpTransition vnet.py:116-156
        convT(name + ".up_conv", cin, cout // 2, k)
        bn(name + ".bn1", cout // 2)
        pr(name + ".relu1", cout // 2)
        pr(name + ".relu2", cout)
        for i in range(nconv):
            luconv(f"{name}.ops.{i}", cout)

    down("down_tr32", 16, 1, kernel_size[0])
    down("down_tr64", 32, 2, kernel_size[1])
    down("down_tr128", 64, 3, kernel_size[2])
    down("down_tr256", 128, 2, kernel_size[3])
    up("up_tr256", 256, 256, 2, kernel_size[3])
    up("up_tr128", 256, 128, 2, kernel_size[2])
    up("up_tr64", 128, 64, 1, kernel_size[1])
    up("up_tr32", 64, 32, 1, kernel_size[0])
    # OutputTransition vnet.py:159-175
    conv("out_tr.conv1", num_classes, 32, 5)
    bn("out_tr.bn1", num_classes)
    conv("out_tr.conv2", num_classes, num_classes, 1)
    pr("out_tr.relu1", num_classes)
    return specs


def init_params(seed=0, in_channels=1, num_classes=4, kernel_size=DEFAULT_K,
                stride_size=DEFAULT_S, dtype=np.float32, perturb=True):
    """Deterministic initialisation (App. B.6 distributions).  With ``perturb``
    BN gamma/beta, PReLU alpha, biases and running stats get non-trivial values so
    that parity tests exercise every term."""
    return _init_from_specs(param_specs(in_channels, num_classes, kernel_size, stride_size), seed, dtype, perturb)


def _init_from_specs(specs, seed, dtype, perturb):
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, kind in specs:
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif kind == "convT_w":
            # Xavier-uniform: fan_in = Cin*k, fan_out = Cout*k (paddle default for Conv3DTranspose)
            rec = shape[2] * shape[3] * shape[4]
            lim = np.sqrt(6.0 / (shape[0] * rec + shape[1] * rec))
            v = rng.uniform(-lim, lim, shape)
        elif kind == "bias":
            v = rng.uniform(-0.1, 0.1, shape) if perturb else np.zeros(shape)
        elif kind == "bn_w":
            v = rng.uniform(0.8, 1.2, shape) if perturb else np.ones(shape)
        elif kind == "bn_b":
            v = rng.uniform(-0.1, 0.1, shape) if perturb else np.zeros(shape)
        elif kind == "bn_mean":
            v = rng.uniform(-0.05, 0.05, shape) if perturb else np.zeros(shape)
        elif kind == "bn_var":
            v = rng.uniform(0.8, 1.2, shape) if perturb else np.ones(shape)
        elif kind == "prelu":
            v = rng.uniform(0.15, 0.35, shape) if perturb else np.full(shape, 0.25)
        out[name] = v.astype(dtype)
    return out


TRAINABLE_KINDS = ("conv_w", "convT_w", "bias", "bn_w", "bn_b", "prelu")


class VNetOracle:
    """Forward/backward restatement of VNet (models/vnet.py:178-267).

    ``dropout_masks``: dict site -> array [N, C] of multipliers (0 or 1/(1-p));
    sites are 'down_tr128', 'down_tr256', 'up_tr256.x', 'up_tr256.skip',
    'up_tr128.x', 'up_tr128.skip' (vnet.py:205,212,220-221,229-230).  Missing
    sites mean identity (eval mode or dropout disabled)."""

    def __init__(self, params, in_channels=1, num_classes=4, kernel_size=DEFAULT_K,
                 stride_size=DEFAULT_S, dtype=np.float64):
        self.dtype = dtype
        self.p = {k: np.asarray(v, dtype=dtype).copy() for k, v in params.items()}
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.kernel_size = [_triple(k) for k in kernel_size]
        self.stride_size = [_triple(s) for s in stride_size]
        self.specs = param_specs(in_channels, num_classes, kernel_size, stride_size)
        self.trainable = [n for n, _, k in self.specs if k in TRAINABLE_KINDS]
        self.grads = {}
        self.training = True
        self.tape = None

    # ---- helpers -------------------------------------------------------
    def _gacc(self, name, g):
        self.grads[name] = g if name not in self.grads else self.grads[name] + g

    def _conv(self, name, x, stride=1, padding=0):
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        y = Var(conv3d(x.v, w, b, stride, padding))
        if self.tape is not None:
            def bw():
                dw, db = conv3d_wgrad(y.g, x.v, w.shape[2:], stride, padding)
                self._gacc(name + ".weight", dw)
                self._gacc(name + ".bias", db)
                x.acc(conv3d_dgrad(y.g, w, x.v.shape, stride, padding))
            self.tape.record(bw)
        return y

    def _convT(self, name, x, stride):
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        y = Var(conv_transpose3d(x.v, w, b, stride))
        if self.tape is not None:
            def bw():
                dw, db = conv_transpose3d_wgrad(y.g, x.v, w.shape[2:], stride)
                self._gacc(name + ".weight", dw)
                self._gacc(name + ".bias", db)
                x.acc(conv_transpose3d_dgrad(y.g, w, stride))
            self.tape.record(bw)
        return y

    def _bn(self, name, x):
        g, b = self.p[name + ".weight"], self.p[name + ".bias"]
        if self.training:
            yv, xhat, mean, var, invstd = bn_train(x.v, g, b)
            # running statistics update (App. B.2; biased batch variance)
            self.p[name + "._mean"] = BN_MOMENTUM * self.p[name + "._mean"] + (1 - BN_MOMENTUM) * mean
            self.p[name + "._variance"] = BN_MOMENTUM * self.p[name + "._variance"] + (1 - BN_MOMENTUM) * var
            y = Var(yv)
            if self.tape is not None:
                def bw():
                    dx, dg, db = bn_train_bwd(y.g, xhat, g, invstd)
                    self._gacc(name + ".weight", dg)
                    self._gacc(name + ".bias", db)
                    x.acc(dx)
                self.tape.record(bw)
            return y
        y = Var(bn_eval(x.v, g, b, self.p[name + "._mean"], self.p[name + "._variance"]))
        if self.tape is not None:
            def bw():
                invstd = 1.0 / np.sqrt(self.p[name + "._variance"] + BN_EPS)
                xhat = (x.v - self.p[name + "._mean"].reshape(1, -1, 1, 1, 1)) * invstd.reshape(1, -1, 1, 1, 1)
                self._gacc(name + ".weight", (y.g * xhat).sum(axis=(0, 2, 3, 4)))
                self._gacc(name + ".bias", y.g.sum(axis=(0, 2, 3, 4)))
                x.acc(y.g * (g * invstd).reshape(1, -1, 1, 1, 1))
            self.tape.record(bw)
        return y

    def _prelu(self, name, x):
        if getattr(self, "elu", False):  # ELUCons(elu=True) (vnet.py:25-29): nn.ELU(), alpha = 1, no parameter
            y = Var(np.where(x.v > 0, x.v, np.expm1(np.minimum(x.v, 0))))
            if self.tape is not None:
                self.tape.record(lambda: x.acc(y.g * np.where(x.v > 0, 1.0, np.exp(np.minimum(x.v, 0)))))
            return y
        a = self.p[name + "._weight"]
        y = Var(prelu(x.v, a))
        if self.tape is not None:
            def bw():
                dx, da = prelu_bwd(y.g, x.v, a)
                self._gacc(name + "._weight", da)
                x.acc(dx)
            self.tape.record(bw)
        return y

    def _add(self, a, b):
        y = Var(a.v + b.v)
        if self.tape is not None:
            def bw():
                a.acc(y.g)
                b.acc(y.g)
            self.tape.record(bw)
        return y

    def _drop(self, site, x):
        m = self.masks.get(site) if self.training else None
        if m is None:
            return x
        mm = np.asarray(m, dtype=self.dtype).reshape(m.shape[0], m.shape[1], 1, 1, 1)
        y = Var(x.v * mm)
        if self.tape is not None:
            self.tape.record(lambda: x.acc(y.g * mm))
        return y

    def _cat(self, a, b):
        ca = a.v.shape[1]
        y = Var(np.concatenate((a.v, b.v), axis=1))
        if self.tape is not None:
            def bw():
                a.acc(y.g[:, :ca])
                b.acc(y.g[:, ca:])
            self.tape.record(bw)
        return y

    def _luconv(self, name, x):  # vnet.py:40-43
        return self._prelu(name + ".relu1", self._bn(name + ".bn1", self._conv(name + ".conv1", x, 1, 2)))

    def _down(self, name, x, nconv, k, s, dropout):  # vnet.py:106-113
        down = self._prelu(name + ".relu1", self._bn(name + ".bn1", self._conv(name + ".down_conv", x, s, 0)))
        out = self._drop(name, down) if dropout else down
        for i in range(nconv):
            out = self._luconv(f"{name}.ops.{i}", out)
        return self._prelu(name + ".relu2", self._add(out, down))

    def _up(self, name, x, skip, nconv, k, s, dropout):  # vnet.py:147-156
        out = self._drop(name + ".x", x) if dropout else x
        skipd = self._drop(name + ".skip", skip) if dropout else skip
        out = self._prelu(name + ".relu1", self._bn(name + ".bn1", self._convT(name + ".up_conv", out, s)))
        xcat = self._cat(out, skipd)
        out = xcat
        for i in range(nconv):
            out = self._luconv(f"{name}.ops.{i}", out)
        return self._prelu(name + ".relu2", self._add(out, xcat))

    # ---- public --------------------------------------------------------
    def _begin(self, x, train, dropout_masks, record):
        self.training = train
        self.masks = dropout_masks or {}
        self.tape = Tape() if record else None
        self.grads = {}
        xin = Var(np.asarray(x, dtype=self.dtype))
        self._xin = xin
        return xin

    def _trunk(self, xin):
        """in_tr .. up_tr32 (vnet.py:255-265); fills self.acts."""
        K, S = self.kernel_size, self.stride_size
        # InputTransition vnet.py:74-79
        c = self._bn("in_tr.bn1", self._conv("in_tr.conv1", xin, 1, 2))
        rep = 16 // self.in_channels
        tile = Var(np.tile(xin.v, (1, rep, 1, 1, 1)))
        out16 = self._prelu("in_tr.relu1", self._add(c, tile))
        out32 = self._down("down_tr32", out16, 1, K[0], S[0], False)
        out64 = self._down("down_tr64", out32, 2, K[1], S[1], False)
        out128 = self._down("down_tr128", out64, 3, K[2], S[2], True)
        out256 = self._down("down_tr256", out128, 2, K[3], S[3], True)
        up256 = self._up("up_tr256", out256, out128, 2, K[3], S[3], True)
        up128 = self._up("up_tr128", up256, out64, 2, K[2], S[2], True)
        up64 = self._up("up_tr64", up128, out32, 1, K[1], S[1], False)
        out = self._up("up_tr32", up64, out16, 1, K[0], S[0], False)
        # named activations (value .v, and gradient .g after backward) for debugging/tests
        self.acts = {"out16": out16, "out32": out32, "out64": out64, "out128": out128, "out256": out256,
                     "up256": up256, "up128": up128, "up64": up64, "up32": out}
        return out

    def _out_transition(self, name, x):  # OutputTransition vnet.py:172-175
        o = self._prelu(name + ".relu1", self._bn(name + ".bn1", self._conv(name + ".conv1", x, 1, 2)))
        return self._conv(name + ".conv2", o, 1, 0)

    def forward(self, x, train=True, dropout_masks=None, record=True):
        xin = self._begin(x, train, dropout_masks, record)
        logits = self._out_transition("out_tr", self._trunk(xin))
        self._logits = logits
        return logits.v

    def backward(self, dlogits):
        self._logits.g = np.asarray(dlogits, dtype=self.dtype)
        self.tape.backward()
        return self.grads


# --------------------------------------------------------------------------
# VNetDeepSup (models/vnet_deepsup.py:178-281)
# --------------------------------------------------------------------------
def trilinear_matrix(n_in, n_out, dtype=np.float64):
    """[n_out, n_in] interpolation matrix of one axis of paddle F.interpolate(mode=
    'trilinear', align_corners=False, align_mode=0) with an explicit output size
    (vnet_deepsup.py:268-277; [PADDLE] interpolate kernel): ratio = n_in/n_out,
    src = max(ratio*(o+0.5)-0.5, 0), i0 = floor(src), i1 = min(i0+1, n_in-1),
    out = (1-lam)*x[i0] + lam*x[i1] with lam = src - i0."""
    m = np.zeros((n_out, n_in), dtype=dtype)
    ratio = n_in / n_out
    for o in range(n_out):
        src = max(ratio * (o + 0.5) - 0.5, 0.0)
        i0 = min(int(np.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        lam = src - i0
        m[o, i0] += 1.0 - lam
        m[o, i1] += lam
    return m


def trilinear_resize(x, size):
    """x [N,C,D,H,W] -> [N,C,*size] (separable form of the 8-corner sum)."""
    md, mh, mw = (trilinear_matrix(i, o, x.dtype) for i, o in zip(x.shape[2:], size))
    y = np.einsum("od,ncdhw->ncohw", md, x)
    y = np.einsum("ph,ncohw->ncopw", mh, y)
    return np.einsum("qw,ncopw->ncopq", mw, y)


def trilinear_resize_bwd(g, in_size):
    """Adjoint of trilinear_resize: g [N,C,*out] -> [N,C,*in_size]."""
    md, mh, mw = (trilinear_matrix(i, o, g.dtype) for i, o in zip(in_size, g.shape[2:]))
    y = np.einsum("qw,ncopq->ncopw", mw, g)
    y = np.einsum("ph,ncopw->ncohw", mh, y)
    return np.einsum("od,ncohw->ncdhw", md, y)


def param_specs_deepsup(in_channels=1, num_classes=4, kernel_size=DEFAULT_K, stride_size=DEFAULT_S):
    """State-dict names of VNetDeepSup in attribute order (vnet_deepsup.py:210-251): the
    VNet trunk, out_tr32 (= VNet's out_tr), three conv3^3 heads and the never-called
    out_tr_all (:251; its parameters exist, receive no gradient and are skipped by the
    optimizer [PADDLE])."""
    specs = [(n.replace("out_tr.", "out_tr32."), s, k)
             for n, s, k in param_specs(in_channels, num_classes, kernel_size, stride_size)]
    for name, cin in (("out_tr64", 64), ("out_tr128", 128), ("out_tr256", 256)):
        specs.append((name + ".weight", (num_classes, cin, 3, 3, 3), "conv_w"))
        specs.append((name + ".bias", (num_classes,), "bias"))
    n = "out_tr_all"
    specs += [(n + ".conv1.weight", (num_classes, 4 * num_classes, 5, 5, 5), "conv_w"),
              (n + ".conv1.bias", (num_classes,), "bias"),
              (n + ".bn1.weight", (num_classes,), "bn_w"), (n + ".bn1.bias", (num_classes,), "bn_b"),
              (n + ".bn1._mean", (num_classes,), "bn_mean"), (n + ".bn1._variance", (num_classes,), "bn_var"),
              (n + ".conv2.weight", (num_classes, num_classes, 1, 1, 1), "conv_w"),
              (n + ".conv2.bias", (num_classes,), "bias"),
              (n + ".relu1._weight", (num_classes,), "prelu")]
    return specs


def init_params_deepsup(seed=0, in_channels=1, num_classes=4, kernel_size=DEFAULT_K, stride_size=DEFAULT_S,
                        dtype=np.float32, perturb=True):
    return _init_from_specs(param_specs_deepsup(in_channels, num_classes, kernel_size, stride_size), seed, dtype,
                            perturb)


class VNetDeepSupOracle(VNetOracle):
    """forward -> [out, d1, d2, d3] (vnet_deepsup.py:257-281): d1/d2/d3 are conv3^3(p=1)
    heads on the up_tr256/up_tr128/up_tr64 outputs, trilinearly resized to the input
    size.  backward takes the list of the four logit gradients."""

    def __init__(self, params, in_channels=1, num_classes=4, kernel_size=DEFAULT_K, stride_size=DEFAULT_S,
                 dtype=np.float64):
        super().__init__(params, in_channels, num_classes, kernel_size, stride_size, dtype)
        self.specs = param_specs_deepsup(in_channels, num_classes, kernel_size, stride_size)
        self.unused = [n for n, _, _ in self.specs if n.startswith("out_tr_all.")]
        self.trainable = [n for n, _, k in self.specs if k in TRAINABLE_KINDS and n not in self.unused]

    def _resize(self, x, size):
        y = Var(trilinear_resize(x.v, size))
        if self.tape is not None:
            self.tape.record(lambda: x.acc(trilinear_resize_bwd(y.g, x.v.shape[2:])))
        return y

    def forward(self, x, train=True, dropout_masks=None, record=True):
        xin = self._begin(x, train, dropout_masks, record)
        size = xin.v.shape[2:]
        feat = self._trunk(xin)
        outs = [self._out_transition("out_tr32", feat)]
        for head, act in (("out_tr256", "up256"), ("out_tr128", "up128"), ("out_tr64", "up64")):
            d = self._conv(head, self.acts[act], 1, 1)
            self.acts[head] = d
            outs.append(self._resize(d, size))
        self._outs = outs
        return [o.v for o in outs]

    def backward(self, dlogits_list):
        for o, g in zip(self._outs, dlogits_list):
            o.g = None if g is None else np.asarray(g, dtype=self.dtype)
        # a head without a gradient contributes nothing
        for o in self._outs:
            if o.g is None:
                o.g = np.zeros_like(o.v)
        self.tape.backward()
        return self.grads


# --------------------------------------------------------------------------
# losses (models/losses/*.py)
# --------------------------------------------------------------------------
def softmax(z, axis=1):
    m = z.max(axis=axis, keepdims=True)
    e = np.exp(z - m)
    return e / e.sum(axis=axis, keepdims=True)


def class_weights(logit):
    """losses/loss_utils.py:31-40: w_c = sum(1-p_c)/sum(p_c), p = softmax over C."""
    p = softmax(logit, 1)
    flat = np.moveaxis(p, 1, 0).reshape(p.shape[1], -1)
    return (1.0 - flat).sum(-1) / flat.sum(-1)


def cross_entropy(logit, label, weight, ignore_index=255, eps=1e-8):
    """losses/cross_entropy_loss.py:47-87 + paddle F.cross_entropy(weight,
    ignore_index, reduction='mean') (App. B.5): sum_i w[y_i]*nll_i / sum_i w[y_i]
    over y_i != ignore_index.  Returns (loss, dL/dlogit)."""
    z = logit + eps
    p = softmax(z, 1)
    C = z.shape[1]
    valid = label != ignore_index
    ysafe = np.where(valid, label, 0).astype(np.int64)
    onehot = np.moveaxis(np.eye(C, dtype=z.dtype)[ysafe], -1, 1)
    wy = weight[ysafe] * valid
    logp = np.log(np.take_along_axis(p, ysafe[:, None], axis=1)[:, 0])
    den = wy.sum()
    den_safe = den if den != 0 else 1.0
    loss = -(wy * logp).sum() / den_safe
    dz = (p - onehot) * (wy / den_safe)[:, None]
    return loss, dz


def dice(logit, label, epsilon=1e-6, sigmoid_norm=True, weight=None):
    """losses/dice_loss.py:45-102: V-Net dice with squared denominator; probabilities from the sigmoid
    (dice_loss.py:40-41) or the softmax over classes (:42-43), optional per-class weight on the intersection
    (:68-69).  Returns (loss, per_channel_dice, dL/dlogit)."""
    if not sigmoid_norm or weight is not None:
        return _dice_general(logit, label, epsilon, sigmoid_norm, weight)
    C = logit.shape[1]
    s = 1.0 / (1.0 + np.exp(-logit))
    t = np.moveaxis(np.eye(C, dtype=logit.dtype)[label.astype(np.int64)], -1, 1)
    ax = (0, 2, 3, 4)
    inter = (s * t).sum(axis=ax)
    den = (s * s).sum(axis=ax) + (t * t).sum(axis=ax)
    denc = np.maximum(den, epsilon)
    per = 2.0 * inter / denc
    loss = 1.0 - per.mean()
    r = lambda v: v.reshape(1, -1, 1, 1, 1)
    dden = np.where(den > epsilon, 1.0, 0.0)  # clip passes gradient only above min
    dper_ds = 2.0 * t / r(denc) - r(2.0 * inter / denc ** 2 * dden) * 2.0 * s
    dz = -(1.0 / C) * dper_ds * s * (1.0 - s)
    return loss, per, dz


def _dice_general(logit, label, epsilon, sigmoid_norm, weight):
    C = logit.shape[1]
    p = 1.0 / (1.0 + np.exp(-logit)) if sigmoid_norm else softmax(logit, 1)
    t = np.moveaxis(np.eye(C, dtype=logit.dtype)[label.astype(np.int64)], -1, 1)
    ax = (0, 2, 3, 4)
    w = np.ones(C, dtype=logit.dtype) if weight is None else np.asarray(weight, dtype=logit.dtype)
    inter = w * (p * t).sum(axis=ax)
    den = (p * p).sum(axis=ax) + (t * t).sum(axis=ax)
    denc = np.maximum(den, epsilon)
    per = 2.0 * inter / denc
    loss = 1.0 - per.mean()
    r = lambda v: v.reshape(1, -1, 1, 1, 1)
    dden = np.where(den > epsilon, 1.0, 0.0)
    dper_dp = 2.0 * r(w) * t / r(denc) - r(2.0 * inter / denc ** 2 * dden) * 2.0 * p
    g = -(1.0 / C) * dper_dp  # dL/dp
    if sigmoid_norm:
        dz = g * p * (1.0 - p)
    else:
        dz = p * (g - (g * p).sum(axis=1, keepdims=True))
    return loss, per, dz


def adam_step(params, grads, m1, m2, t, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.0, names=None):
    """paddle.optimizer.Adam with a float weight_decay (= L2Decay added to the gradient), step t = 1, 2, ...:
    the bias-corrected update in the form Paddle's kernel evaluates it (epsilon scaled by sqrt(1 - beta2^t));
    algebraically torch.optim.Adam's update, against which tests/test_oracle.py checks it."""
    c2 = np.sqrt(1.0 - beta2 ** t)
    lr_t = lr * c2 / (1.0 - beta1 ** t)
    for n in (names if names is not None else grads.keys()):
        g = grads[n] + weight_decay * params[n]
        m1[n] = beta1 * m1.get(n, 0.0) + (1.0 - beta1) * g
        m2[n] = beta2 * m2.get(n, 0.0) + (1.0 - beta2) * g * g
        params[n] = params[n] - lr_t * (m1[n] / (np.sqrt(m2[n]) + epsilon * c2))


class MixedLossOracle:
    """MixedLoss([CrossEntropyLoss, DiceLoss], coef) (mixes_losses.py:23-60) wrapped
    by loss_computation's outer coef (utils/loss_utils.py:43-46).  Caches the CE
    class weights from the first call (cross_entropy_loss.py:68-69)."""

    def __init__(self, coef=(1.0, 1.0), outer_coef=1.0, ignore_index=255, dtype=np.float64):
        self.coef = coef
        self.outer = outer_coef
        self.ignore_index = ignore_index
        self.weight = None
        self.dtype = dtype

    def __call__(self, logits, labels):
        z = np.asarray(logits, dtype=self.dtype)
        if self.weight is None:
            self.weight = class_weights(z)
        ce, dce = cross_entropy(z, labels, self.weight, self.ignore_index)
        dl, per, ddl = dice(z, labels)
        loss_list = [self.outer * self.coef[0] * ce, self.outer * self.coef[1] * dl]
        dz = self.outer * (self.coef[0] * dce + self.coef[1] * ddl)
        return loss_list, per, dz


# --------------------------------------------------------------------------
# optimizer (cvlibs/config.py:156-224; App. B.8 v, vi)
# --------------------------------------------------------------------------
def poly_lr(step, lr0=1e-3, decay_steps=15000, end_lr=0.0, power=0.9):
    t = min(step, decay_steps)
    return (lr0 - end_lr) * (1.0 - t / decay_steps) ** power + end_lr


def sgd_momentum_step(params, grads, vel, lr, momentum=0.9, weight_decay=1e-4, names=None):
    """paddle Momentum + L2Decay(float): g += wd*p; v = mu*v + g; p -= lr*v."""
    for n in (names if names is not None else grads.keys()):
        g = grads[n] + weight_decay * params[n]
        vel[n] = momentum * vel.get(n, 0.0) + g
        params[n] = params[n] - lr * vel[n]


def train_step(model: VNetOracle, loss: MixedLossOracle, vel, x, y, step, lr0=1e-3,
               decay_steps=15000, power=0.9, momentum=0.9, weight_decay=1e-4,
               train=True, dropout_masks=None):
    """One iteration of core/train.py:120-155."""
    logits = model.forward(x, train=train, dropout_masks=dropout_masks)
    loss_list, per, dz = loss(logits, y)
    grads = model.backward(dz)
    lr = poly_lr(step, lr0, decay_steps, 0.0, power)
    sgd_momentum_step(model.p, grads, vel, lr, momentum, weight_decay, names=model.trainable)
    return float(sum(loss_list)), [float(l) for l in loss_list], per, logits
