"""Minimal paddle.nn-shaped layer system over libmsegk.

Only what medicalseg's VNet path uses (models/vnet.py:24-26): Layer, Sequential,
Conv3D, Conv3DTranspose, BatchNorm3D (SyncBatchNorm semantics when world > 1,
cvlibs/config.py:322), PReLU, Dropout3D.  Parameter names and state-dict keys follow
Paddle so reference checkpoints map one to one.

Execution is fused: ``ConvBNAct`` runs conv -> BN statistics -> one elementwise kernel
(normalise + affine + residual + PReLU); the individual BatchNorm3D/PReLU layers are
parameter holders.  Backward is explicit (no tape): every fused unit keeps what its
adjoint needs.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import NULL_TENSOR, MskBnFin, MskConvDesc, MskError
from .device import Tensor, get_device

BN_EPS = 1e-5
BN_MOMENTUM = 0.9


def _triple(v):
    if isinstance(v, (list, tuple)):
        if len(v) != 3:
            raise ValueError(f"expected 3 values, got {v}")
        return tuple(int(i) for i in v)
    return (int(v),) * 3


_init_rng = np.random.default_rng(0)

# Host-side accounting of ALGORITHMIC convolution FLOPs (2 k^3 Cin Cout per output voxel and pass) for the
# workload benchmarks: {"on": True} makes every convolution call add its count under its kernel class.
FLOPS = {"on": False, "same_k5": 0.0, "same_k3": 0.0, "other": 0.0}


def _count_flops(conv, n, out_voxels, passes):
    if not FLOPS["on"]:
        return
    k = conv.k[0] * conv.k[1] * conv.k[2]
    f = 2.0 * k * conv.cin * conv.cout * n * out_voxels * passes
    same = conv.s == (1, 1, 1) and not conv.transposed and conv.cin >= 16 and conv.cout >= 16
    key = "same_k5" if (same and conv.k == (5, 5, 5)) else ("same_k3" if (same and conv.k == (3, 3, 3)) else "other")
    FLOPS[key] += f


def seed(s: int):
    """Seed parameter initialisation (paddle.seed, train.py:120-123)."""
    global _init_rng
    _init_rng = np.random.default_rng(s)


class Parameter:
    """A trainable tensor or a buffer; lives in a ParamArena once the model is built."""

    def __init__(self, value: np.ndarray, trainable=True):
        self.init_value = np.ascontiguousarray(value, dtype=np.float32)
        self.shape = tuple(self.init_value.shape)
        self.size = int(self.init_value.size)
        self.trainable = trainable
        self.arena = None
        self.offset = 0  # in floats
        self.name = None

    @property
    def ptr(self):
        return self.arena.value_ptr + 4 * self.offset

    @property
    def grad_ptr(self):
        return self.arena.grad_ptr + 4 * self.offset

    def numpy(self):
        if self.arena is None:
            return self.init_value.copy()
        return self.arena.dev.d2h(self.ptr, self.shape, np.float32)

    def grad_numpy(self):
        return self.arena.dev.d2h(self.grad_ptr, self.shape, np.float32)

    def set_value(self, v):
        v = np.ascontiguousarray(v, dtype=np.float32)
        if tuple(v.shape) != self.shape:
            raise ValueError(f"shape mismatch for {self.name}: {v.shape} vs {self.shape}")
        if self.arena is None:
            self.init_value = v
        else:
            self.arena.dev.h2d(self.ptr, v)


class ParamArena:
    """All trainable parameters of a model in ONE flat fp32 buffer (+ a parallel gradient
    buffer), each tensor 16-byte aligned: the optimizer is a single kernel launch and the
    data-parallel exchange a single RCCL all-reduce."""

    def __init__(self, dev, params, with_grad=True):
        self.dev = dev
        off = 0
        for p in params:
            p.offset = off
            off += (p.size + 3) & ~3
        self.count = off
        self.value_ptr = dev.malloc(max(off, 4) * 4)
        dev.memset(self.value_ptr, 0, max(off, 4) * 4)
        self.grad_ptr = None
        if with_grad:
            self.grad_ptr = dev.malloc(max(off, 4) * 4)
            dev.memset(self.grad_ptr, 0, max(off, 4) * 4)
        self.grad_scale = 1.0
        for p in params:
            p.arena = self
            dev.h2d(p.ptr, p.init_value)
        self.params = list(params)

    def zero_grad(self):
        self.dev.memset(self.grad_ptr, 0, self.count * 4)


class Layer:
    def __init__(self):
        object.__setattr__(self, "_sub_layers", OrderedDict())
        object.__setattr__(self, "_parameters", OrderedDict())
        object.__setattr__(self, "_buffers", OrderedDict())
        object.__setattr__(self, "training", True)

    def __setattr__(self, name, value):
        if isinstance(value, Layer):
            self._sub_layers[name] = value
        elif isinstance(value, Parameter):
            (self._parameters if value.trainable else self._buffers)[name] = value
        object.__setattr__(self, name, value)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    # -- tree walks -------------------------------------------------------------------
    def named_sublayers(self, prefix=""):
        for name, layer in self._sub_layers.items():
            full = f"{prefix}.{name}" if prefix else name
            yield full, layer
            yield from layer.named_sublayers(full)

    def sublayers(self):
        return [l for _, l in self.named_sublayers()]

    def named_parameters(self, prefix=""):
        for name, p in self._parameters.items():
            yield (f"{prefix}.{name}" if prefix else name), p
        for name, layer in self._sub_layers.items():
            yield from layer.named_parameters(f"{prefix}.{name}" if prefix else name)

    def named_buffers(self, prefix=""):
        for name, p in self._buffers.items():
            yield (f"{prefix}.{name}" if prefix else name), p
        for name, layer in self._sub_layers.items():
            yield from layer.named_buffers(f"{prefix}.{name}" if prefix else name)

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def train(self):
        self.training = True
        for l in self.sublayers():
            l.training = True

    def eval(self):
        self.training = False
        for l in self.sublayers():
            l.training = False

    # -- state dict (paddle: parameters and persistable buffers, in attribute order) ----
    def _named_state(self, prefix=""):
        for name, p in self._parameters.items():
            yield (f"{prefix}.{name}" if prefix else name), p
        for name, p in self._buffers.items():
            yield (f"{prefix}.{name}" if prefix else name), p
        for name, layer in self._sub_layers.items():
            yield from layer._named_state(f"{prefix}.{name}" if prefix else name)

    def state_dict(self):
        return OrderedDict((k, p.numpy()) for k, p in self._named_state())

    def set_state_dict(self, state):
        own = dict(self._named_state())
        missing = [k for k in own if k not in state]
        for k, v in state.items():
            if k in own:
                own[k].set_value(np.asarray(v))
        return missing, [k for k in state if k not in own]

    set_dict = set_state_dict
    load_dict = set_state_dict

    def clear_gradients(self):
        arenas = {id(p.arena): p.arena for p in self.parameters()
                  if p.arena is not None and p.arena.grad_ptr is not None}
        for a in arenas.values():
            a.zero_grad()

    def forward(self, *a, **k):
        raise NotImplementedError


class Sequential(Layer):
    def __init__(self, *layers):
        super().__init__()
        for i, l in enumerate(layers):
            setattr(self, str(i), l)

    def __iter__(self):
        return iter(self._sub_layers.values())

    def __len__(self):
        return len(self._sub_layers)

    def __getitem__(self, i):
        return list(self._sub_layers.values())[i]

    def forward(self, x):
        for l in self._sub_layers.values():
            x = l(x)
        return x


class Conv3D(Layer):
    """paddle.nn.Conv3D: weight [Cout, Cin, kD, kH, kW], Normal(0, sqrt(2/(k^3 Cin)))
    init, zero bias (App. B.6)."""

    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.cin, self.cout = int(in_channels), int(out_channels)
        self.k, self.s, self.p = _triple(kernel_size), _triple(stride), _triple(padding)
        fan_in = self.cin * self.k[0] * self.k[1] * self.k[2]
        self.weight = Parameter(_init_rng.standard_normal((self.cout, self.cin) + self.k) * np.sqrt(2.0 / fan_in))
        self.bias = Parameter(np.zeros(self.cout))

    def desc(self):
        return MskConvDesc(*self.k, *self.s, *self.p)

    def out_dims(self, x: Tensor):
        return tuple((i + 2 * p - k) // s + 1 for i, p, k, s in zip((x.d, x.h, x.w), self.p, self.k, self.s))

    def run_forward(self, x: Tensor, y: Tensor | None = None, stats_ptr=None, keep_xform=False, fin=None) -> Tensor:
        """stats_ptr: device buffer [2*Cout] that receives the BatchNorm statistics record of y (taken in the
        convolution's output stage when the kernel can, msk_conv3d_fwd_ex).  keep_xform: keep the transformed input the
        convolution computes anyway in the activation arena for this layer's weight gradient.  fin: MskBnFin -- the
        BatchNorm finalisation runs in the launch that merges the statistics (msk_conv3d_fwd_ex3)."""
        if x.c != self.cin:
            raise ValueError(f"Conv3D expects {self.cin} input channels, got {x.c}")
        od, oh, ow = self.out_dims(x)
        if y is None:
            y = Tensor.empty(x.dev, x.n, od, oh, ow, self.cout)
        _count_flops(self, x.n, od * oh * ow, 1)
        dev = x.dev
        self._xform = None
        if stats_ptr is None and not keep_xform:
            dev.call("msk_conv3d_fwd", self.desc(), x.msk(), C.c_void_p(self.weight.ptr),
                     C.c_void_p(self.bias.ptr), y.msk())
            return y
        xf = None
        if keep_xform:
            nbytes = int(dev.lib.msk_conv3d_xform_bytes(dev.ctx, self.desc(), x.msk(), self.cout))
            if nbytes > 0:
                xf = dev.arena.alloc(nbytes)
        dev.call("msk_conv3d_fwd_ex3", self.desc(), x.msk(), C.c_void_p(self.weight.ptr), C.c_void_p(self.bias.ptr),
                 y.msk(), C.c_void_p(stats_ptr) if stats_ptr else None, C.c_void_p(xf) if xf else None,
                 C.c_void_p(x.amax) if (x.amax and PRODUCER_AMAX) else None, C.byref(fin) if fin is not None else None)
        self._xform = (xf, x.ptr, dev.arena.gen) if xf else None
        return y

    def run_backward(self, x: Tensor, dy: Tensor, need_dx=True, bias_grad=True):
        dev = x.dev
        _count_flops(self, x.n, dy.d * dy.h * dy.w, 2 if need_dx else 1)
        xf = getattr(self, "_xform", None)
        xfp = C.c_void_p(xf[0]) if (xf is not None and xf[1] == x.ptr and xf[2] == dev.arena.gen) else None   # same tensor, same arena generation
        # max |dy| folded in by the pass that wrote dy (msk_affine_act_bwd_apply_amax): no absmax pass of their own below
        dya = C.c_void_p(dy.amax) if (dy.amax and PRODUCER_AMAX) else None
        xa = C.c_void_p(x.amax) if (x.amax and PRODUCER_AMAX and xfp is None) else None   # (a kept transform carries its own scale)
        if xfp is not None or dya is not None or xa is not None:
            dev.call("msk_conv3d_wgrad_ex3", self.desc(), x.msk(), dy.msk(), C.c_void_p(self.weight.grad_ptr),
                     C.c_void_p(self.bias.grad_ptr) if bias_grad else None, 1, xfp, dya, xa)
        else:
            dev.call("msk_conv3d_wgrad", self.desc(), x.msk(), dy.msk(), C.c_void_p(self.weight.grad_ptr),
                     C.c_void_p(self.bias.grad_ptr) if bias_grad else None, 1)
        self._xform = None
        if need_dx:
            dx = x.ensure_grad()
            dev.call("msk_conv3d_dgrad_ex", self.desc(), dy.msk(), C.c_void_p(self.weight.ptr), dx.msk(),
                     1 if x.grad_written else 0, dya)
            x.grad_written = True

    forward = run_forward


class Conv3DTranspose(Layer):
    """paddle.nn.Conv3DTranspose: weight [Cin, Cout, kD, kH, kW], Xavier-uniform init,
    out = (in-1)*s + k (App. B.1, B.6)."""

    transposed = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1):
        super().__init__()
        self.cin, self.cout = int(in_channels), int(out_channels)
        self.k, self.s, self.p = _triple(kernel_size), _triple(stride), (0, 0, 0)
        rec = self.k[0] * self.k[1] * self.k[2]
        lim = np.sqrt(6.0 / (self.cin * rec + self.cout * rec))
        self.weight = Parameter(_init_rng.uniform(-lim, lim, (self.cin, self.cout) + self.k))
        self.bias = Parameter(np.zeros(self.cout))

    def desc(self):
        return MskConvDesc(*self.k, *self.s, 0, 0, 0)

    def out_dims(self, x: Tensor):
        return tuple((i - 1) * s + k for i, k, s in zip((x.d, x.h, x.w), self.k, self.s))

    def run_forward(self, x: Tensor, y: Tensor | None = None) -> Tensor:
        if x.c != self.cin:
            raise ValueError(f"Conv3DTranspose expects {self.cin} input channels, got {x.c}")
        od, oh, ow = self.out_dims(x)
        if y is None:
            y = Tensor.empty(x.dev, x.n, od, oh, ow, self.cout)
        _count_flops(self, x.n, x.d * x.h * x.w, 1)      # transposed conv: per INPUT voxel
        x.dev.call("msk_convT3d_fwd", self.desc(), x.msk(), C.c_void_p(self.weight.ptr),
                   C.c_void_p(self.bias.ptr), y.msk())
        return y

    def run_backward(self, x: Tensor, dy: Tensor, need_dx=True, bias_grad=True):
        dev = x.dev
        _count_flops(self, x.n, x.d * x.h * x.w, 2 if need_dx else 1)
        dev.call("msk_convT3d_wgrad", self.desc(), x.msk(), dy.msk(), C.c_void_p(self.weight.grad_ptr),
                 C.c_void_p(self.bias.grad_ptr) if bias_grad else None, 1)
        if need_dx:
            dx = x.ensure_grad()
            dev.call("msk_convT3d_dgrad", self.desc(), dy.msk(), C.c_void_p(self.weight.ptr), dx.msk(),
                     1 if x.grad_written else 0)
            x.grad_written = True

    forward = run_forward


class BatchNorm3D(Layer):
    """paddle.nn.BatchNorm3D(momentum=0.9, epsilon=1e-5); buffers `_mean`, `_variance`.
    With world > 1 the statistics are global-batch (SyncBatchNorm), as
    cvlibs/config.py:322 converts every BatchNorm unconditionally.

    ``BatchNorm3D.sync = False`` (train.py/bench.py ``--no_sync_bn``) is a DOCUMENTED DEVIATION from
    the reference: rank-local statistics, which removes the 24 + 24 latency-bound
    collectives per step and leaves the single gradient all-reduce."""

    sync = True
    # tests: run the SyncBatchNorm collectives (all-gather of the statistics, all-reduce of the backward sums) also on a
    # 1-rank communicator, where they are identities -- exercises the communication-stream hand-over inside a real step
    force_collectives = False
    # bench.py's compute-only replay (parallel.set_dry_run): no statistics collective is issued, the statistics stay rank-local
    dry_run = False

    def __init__(self, num_features, momentum=BN_MOMENTUM, epsilon=BN_EPS):
        super().__init__()
        self.num_features = int(num_features)
        self.momentum, self.epsilon = float(momentum), float(epsilon)
        self.weight = Parameter(np.ones(num_features))
        self.bias = Parameter(np.zeros(num_features))
        self._mean = Parameter(np.zeros(num_features), trainable=False)
        self._variance = Parameter(np.ones(num_features), trainable=False)
        self._scratch = None

    def scratch(self, dev):
        """Persistent device scratch: stats_local[2C] gathered[W*2C] scale shift mean invstd
        sums[3C] sums_total[3C]."""
        if self._scratch is None or self._scratch["world"] != dev.world:
            Cn, W = self.num_features, dev.world
            total = 2 * Cn + W * 2 * Cn + 4 * Cn + 7 * Cn + 128
            base = dev.small(total)
            o = 0
            s = {"world": W}
            for name, cnt in (("stats", 2 * Cn), ("gathered", W * 2 * Cn), ("scale", Cn), ("shift", Cn),
                              ("mean", Cn), ("invstd", Cn), ("sums", 4 * Cn), ("sums_total", 3 * Cn), ("maxes", 128)):   # maxes: two amax arrays of 64 floats (msk_affine_act_bwd_reduce_ex)
                s[name] = base + 4 * o
                o += cnt
            self._scratch = s
        return self._scratch


class SyncBatchNorm(BatchNorm3D):
    @staticmethod
    def convert_sync_batchnorm(layer):
        """No-op kept for API parity (cvlibs/config.py:322): BatchNorm3D here already uses
        cross-rank statistics whenever a communicator is initialised."""
        return layer


class PReLU(Layer):
    """paddle.nn.PReLU(num_parameters=C, init=0.25); parameter name `_weight`."""

    def __init__(self, num_parameters=1, init=0.25):
        super().__init__()
        self.num_parameters = int(num_parameters)
        self._weight = Parameter(np.full(num_parameters, init))


class ELU(Layer):
    """paddle.nn.ELU(alpha=1.0) -- ELUCons(elu=True) of the reference (vnet.py:25-29).  No parameter.  The units below run
    their activation-less kernels and msk_elu_fwd / msk_elu_bwd as separate passes (the shipped configs use PReLU, which is
    the fused path; the reference notes NaN gradients with ELU, core/train.py:139)."""

    def __init__(self, alpha=1.0):
        super().__init__()
        self.alpha = float(alpha)


def _act_alpha(act):
    """device pointer of a PReLU's slopes; None for no activation and for ELU (applied as its own pass)"""
    return act._weight.ptr if isinstance(act, PReLU) else None


def _act_alpha_grad(act):
    return act._weight.grad_ptr if isinstance(act, PReLU) else None


class Dropout3D(Layer):
    """paddle.nn.Dropout3D(p=0.5): drops whole (n, c) channels, kept ones scaled 1/(1-p)
    (upscale_in_train); identity in eval (App. B.4).  Masks come from a counter-based RNG
    keyed by (seed, step, site) or are injected for parity tests."""

    _site_counter = 0
    seed = 0
    step = 0

    def __init__(self, p=0.5):
        super().__init__()
        self.p = float(p)
        Dropout3D._site_counter += 1
        self.site = Dropout3D._site_counter
        self.injected = None  # numpy [N, C] multipliers
        self.enabled = True

    def make_mask(self, x: Tensor):
        """Returns a device pointer to mask[N*C] or None for identity."""
        if not self.training or not self.enabled or self.p == 0.0:
            return None
        dev = x.dev
        cnt = x.n * x.c
        ptr = dev.arena.alloc(cnt * 4)
        if self.injected is not None:
            m = np.ascontiguousarray(self.injected, dtype=np.float32)
            if m.shape != (x.n, x.c):
                raise ValueError(f"injected dropout mask must be [{x.n},{x.c}], got {m.shape}")
            dev.h2d(ptr, m)
        else:
            # every data-parallel rank draws its own masks: the rank rides in the upper bits of the site id
            dev.call("msk_dropout_mask", C.c_uint64(Dropout3D.seed), C.c_uint64(Dropout3D.step),
                     C.c_uint32(self.site | (dev.rank << 16)), cnt, C.c_float(self.p), C.c_void_p(ptr))
        return ptr


# ----------------------------------------------------------------------------------------
# fused execution units
# ----------------------------------------------------------------------------------------
_FUSED_INFERENCE = {"on": False, "epoch": 0}


class fused_inference:
    """Context manager: inside it, eval-mode conv -> BN -> PReLU units run as one folded convolution
    (`ConvBNAct._forward_folded`).  `core.val.evaluate` and `core.infer.inference` use it; nothing that calls
    backward may.  The folded weights are rebuilt on first use in every scope, so parameter or running-statistics
    changes between scopes are always picked up."""

    def __enter__(self):
        self._prev = _FUSED_INFERENCE["on"]
        _FUSED_INFERENCE["on"] = True
        if not self._prev:               # a nested scope keeps the outer scope's folded weights
            _FUSED_INFERENCE["epoch"] += 1
        return self

    def __exit__(self, *exc):
        _FUSED_INFERENCE["on"] = self._prev
        return False


def _fp(ptr):
    return C.c_void_p(ptr) if ptr else None


# A/B switch (env MSEGK_PRODUCER_AMAX=0): layer inputs are measured by an absmax pass of their own instead of in the pass
# that writes them
PRODUCER_AMAX = os.environ.get("MSEGK_PRODUCER_AMAX", "1") != "0"


def _amax_for(out: Tensor):
    """the amax array the pass writing `out` folds into: the tensor's own (a concat buffer's, shared by its slices), else a
    fresh one"""
    if not PRODUCER_AMAX:
        return None
    if out.amax is None:
        out.amax = out.dev.amax_new()
    return C.c_void_p(out.amax)


# A/B switch (env MSEGK_FUSE_SMALL=0): the per-channel kernels of a BatchNorm layer (finalize, parameter gradients) as
# launches of their own instead of inside the merge kernels (msk_conv3d_fwd_ex3, msk_*_pg)
FUSE_SMALL = os.environ.get("MSEGK_FUSE_SMALL", "1") != "0"


# A/B switch (env MSEGK_KS_STATS=0): up-convolution units take their BatchNorm statistics in a pass of their own
KS_STATS = os.environ.get("MSEGK_KS_STATS", "1") != "0"
# A/B switch (env MSEGK_BWD_FUSE=0): conv -> BN -> PReLU units run their backward as three calls (apply, dgrad, wgrad)
FUSE_BN_BACKWARD = os.environ.get("MSEGK_BWD_FUSE", "1") != "0"
# A/B switch (env MSEGK_BWD_FUSE_C1=0): in_tr.conv1's BatchNorm backward as a pass of its own in front of its weight gradient
FUSE_BN_BACKWARD_C1 = os.environ.get("MSEGK_BWD_FUSE_C1", "1") != "0"
# Switch (env MSEGK_BWD_FUSE_T=1, default OFF): up-convolution units evaluate dy inside the data gradient's loads and move the pass
# that writes dy to the weight-gradient stream (msk_convT3d_bwd_bnact).  Measured in the step (round 5, one box, two
# repetitions): 18.71 ms with it against 18.51 without -- the compute stream loses 0.14 ms of kernel time, but (y, dout) are
# read twice and the step is bound by the sum of its HBM and matrix work, not by the compute stream's chain alone
FUSE_BN_BACKWARD_T = os.environ.get("MSEGK_BWD_FUSE_T", "0") == "1"


# A/B switch (env MSEGK_SHARE_JOIN_GRAD=0): a residual join writes the gradient of BOTH operands (round 5) instead of one tensor that
# the layer behind the join reads as its old values (msk_conv3d_bwd_bnact_acc)
SHARE_JOIN_GRAD = os.environ.get("MSEGK_SHARE_JOIN_GRAD", "1") != "0"


def materialize_grad(t: Tensor):
    """t.grad_from (see device.Tensor) -> t.grad: the copy the shared-gradient shortcut avoids, for consumers that accumulate in place"""
    src = getattr(t, "grad_from", None)
    if src is None:
        return
    t.grad_from = None
    g = t.ensure_grad()
    src.dev.call("msk_copy_scale", src.msk(), None, g.msk(), 1 if t.grad_written else 0)
    g.amax = None
    t.grad_written = True


class ConvBNAct:
    """conv (or convT) -> BatchNorm -> (+ residual) -> PReLU, forward and adjoint.

    Reference chains replaced: vnet.py:41 (LUConv), :74-79 (InputTransition),
    :107 (down_conv/bn1/relu1), :150 (up_conv/bn1/relu1), :173 (OutputTransition)."""

    def __init__(self, conv, bn: BatchNorm3D, act: PReLU | None):
        self.conv, self.bn, self.act = conv, bn, act

    def _forward_folded(self, x: Tensor, out: Tensor | None) -> Tensor:
        """Inference: eval-mode BatchNorm folded into the convolution, PReLU in the conv epilogue --
        one kernel, no pre-activation tensor.  The folded weights are rebuilt once per `fused_inference()` scope."""
        dev, conv, bn = x.dev, self.conv, self.bn
        if x.c != conv.cin:
            raise ValueError(f"Conv3D expects {conv.cin} input channels, got {x.c}")
        if getattr(self, "_fold_epoch", None) != _FUSED_INFERENCE["epoch"]:
            if not hasattr(self, "_fold_w"):
                self._fold_w = dev.malloc(conv.weight.size * 4)
                self._fold_b = dev.malloc(conv.cout * 4)
            sc, Cn = bn.scratch(dev), bn.num_features
            dev.call("msk_bn_eval_coeffs", Cn, _fp(bn.weight.ptr), _fp(bn.bias.ptr), _fp(bn._mean.ptr),
                     _fp(bn._variance.ptr), C.c_float(bn.epsilon), _fp(sc["mean"]), _fp(sc["invstd"]),
                     _fp(sc["scale"]), _fp(sc["shift"]))
            dev.call("msk_conv_fold_bn", _fp(conv.weight.ptr), _fp(conv.bias.ptr), _fp(sc["scale"]), _fp(sc["shift"]),
                     conv.cout, C.c_long(conv.weight.size // conv.cout), _fp(self._fold_w), _fp(self._fold_b))
            self._fold_epoch = _FUSED_INFERENCE["epoch"]
        if out is None:
            od, oh, ow = conv.out_dims(x)
            out = Tensor.empty(dev, x.n, od, oh, ow, conv.cout)
        alpha = _act_alpha(self.act)
        dev.call("msk_conv3d_fwd_act", conv.desc(), x.msk(), _fp(self._fold_w), _fp(self._fold_b), _fp(alpha), out.msk())
        out.amax = None     # written by a kernel that does not measure it
        self.x, self.res, self.y, self.out, self.bn_mode = x, None, None, out, 3   # 3: no backward through this
        return out

    def forward(self, x: Tensor, res: Tensor | None = None, out: Tensor | None = None, defer_act=False, out2: Tensor | None = None) -> Tensor:
        """defer_act: stop after the BatchNorm coefficients; the caller (AddAct.forward with unit=self) applies BN + PReLU
        inside the residual join that follows (msk_affine_act_join_fwd) -- the returned tensor is allocated (it carries the
        gradient in backward) but never written."""
        dev = x.dev
        self.deferred = False
        if (_FUSED_INFERENCE["on"] and not self.bn.training and res is None and type(self.conv) is Conv3D
                and (self.act is None or isinstance(self.act, PReLU))):
            return self._forward_folded(x, out)
        self.x, self.res = x, res
        bn, sc = self.bn, self.bn.scratch(dev)
        Cn = bn.num_features
        # one rank (or rank-local statistics): the finalisation rides in the launch that merges the statistics
        fin = None
        multi = (dev.world > 1 or BatchNorm3D.force_collectives) and BatchNorm3D.sync and not BatchNorm3D.dry_run
        if bn.training and FUSE_SMALL and not multi:
            od, oh, ow = self.conv.out_dims(x)
            fin = MskBnFin(bn.weight.ptr, bn.bias.ptr, bn.epsilon, bn.momentum, float(x.n * od * oh * ow), bn._mean.ptr,
                           bn._variance.ptr, sc["mean"], sc["invstd"], sc["scale"], sc["shift"])
        if bn.training and type(self.conv) is Conv3D:
            # statistics from the convolution's output stage, transformed input kept for the weight gradient
            y = self.conv.run_forward(x, stats_ptr=sc["stats"], keep_xform=True, fin=fin)
        elif bn.training and type(self.conv) is Conv3DTranspose and KS_STATS:
            # up-convolution (vnet.py:133-150): the statistics ride in the convolution's store pass where its kernel can
            conv = self.conv
            if x.c != conv.cin:
                raise ValueError(f"Conv3DTranspose expects {conv.cin} input channels, got {x.c}")
            od, oh, ow = conv.out_dims(x)
            y = Tensor.empty(dev, x.n, od, oh, ow, conv.cout)
            _count_flops(conv, x.n, x.d * x.h * x.w, 1)
            dev.call("msk_convT3d_fwd_ex", conv.desc(), x.msk(), _fp(conv.weight.ptr), _fp(conv.bias.ptr), y.msk(), _fp(sc["stats"]),
                     C.byref(fin) if fin is not None else None)
        else:
            y = self.conv.run_forward(x)
            if bn.training:
                if fin is not None:
                    dev.call("msk_bn_stats_fin", y.msk(), _fp(sc["stats"]), C.byref(fin))
                else:
                    dev.call("msk_bn_stats", y.msk(), _fp(sc["stats"]))
        self.y = y
        if bn.training:
            if fin is None:
                gathered, nstat = sc["stats"], 1
                if multi:
                    dev.call("msk_dp_allgather", _fp(sc["stats"]), _fp(sc["gathered"]), C.c_size_t(2 * Cn))
                    gathered, nstat = sc["gathered"], dev.world
                dev.call("msk_bn_finalize", _fp(gathered), nstat, C.c_double(y.voxels), Cn, _fp(bn.weight.ptr),
                         _fp(bn.bias.ptr), C.c_float(bn.epsilon), C.c_float(bn.momentum), _fp(bn._mean.ptr),
                         _fp(bn._variance.ptr), _fp(sc["mean"]), _fp(sc["invstd"]), _fp(sc["scale"]), _fp(sc["shift"]))
            self.bn_mode = 1
        else:
            dev.call("msk_bn_eval_coeffs", Cn, _fp(bn.weight.ptr), _fp(bn.bias.ptr), _fp(bn._mean.ptr),
                     _fp(bn._variance.ptr), C.c_float(bn.epsilon), _fp(sc["mean"]), _fp(sc["invstd"]),
                     _fp(sc["scale"]), _fp(sc["shift"]))
            self.bn_mode = 2
        if out is None:
            out = y.empty_like()
        alpha = _act_alpha(self.act)
        if defer_act and res is None and alpha is not None:
            self.deferred = True
        else:
            if isinstance(self.act, ELU):
                dev.call("msk_affine_act_fwd", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]),
                         res.msk() if res is not None else NULL_TENSOR, _fp(alpha), out.msk())
                dev.call("msk_elu_fwd", out.msk(), C.c_float(self.act.alpha), out.msk())
                out.amax = None
            elif out2 is not None:
                # a second, dense copy of the output (InputTransition: `out` is the skip half of a concat buffer, msk_affine_act_fwd_amax2)
                dev.call("msk_affine_act_fwd_amax2", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]),
                         res.msk() if res is not None else NULL_TENSOR, _fp(alpha), out.msk(), _amax_for(out), out2.msk())
                out2.amax = out.amax
            else:
                dev.call("msk_affine_act_fwd_amax", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]),
                         res.msk() if res is not None else NULL_TENSOR, _fp(alpha), out.msk(), _amax_for(out))
        self.out = out
        return out

    def backward(self, dout: Tensor, need_dx=True, res_needs_grad=True):
        """dout: gradient w.r.t. the unit's output (may be a channel slice)."""
        if self.bn_mode == 3:
            raise RuntimeError("backward through a forward pass run under nn.fused_inference() (BN folded into the "
                               "convolution: no pre-activation tensor was kept)")
        dev = dout.dev
        bn, sc = self.bn, self.bn.scratch(dev)
        Cn = bn.num_features
        y, res = self.y, self.res
        alpha = _act_alpha(self.act)
        if isinstance(self.act, ELU):   # dout <- dout * elu'(u), taken from the unit's output; the rest is the BN-only backward
            g = Tensor.empty(dev, dout.n, dout.d, dout.h, dout.w, dout.c)
            dev.call("msk_elu_bwd", self.out.msk(), dout.msk(), C.c_float(self.act.alpha), g.msk(), 0)
            dout = g
        resm = res.msk() if res is not None else NULL_TENSOR
        fuse = (FUSE_BN_BACKWARD and type(self.conv) is Conv3D and res is None and self.bn_mode == 1 and need_dx
                and self.conv.cin == self.conv.cout)
        if not fuse:
            materialize_grad(self.x)      # (only the fused data gradient below reads old values from another tensor)
        # the maxima of |du| and |xhat| ride along when the fused backward may need to scale dy into fp16 range
        want_maxes = fuse and Cn % 4 == 0 and y.ld % 4 == 0 and dout.ld % 4 == 0 and y.ptr % 16 == 0 and dout.ptr % 16 == 0
        pg_done = False
        maxes = sc["maxes"]
        if getattr(self, "presummed", False):     # the join behind this unit already reduced (msk_add_act_join_bwd_ex / _pg)
            self.presummed = False
            want_maxes = fuse
            pg_done = getattr(self, "presummed_pg", False)
            maxes = getattr(self, "presummed_maxes", None) or sc["maxes"]
        elif want_maxes and FUSE_SMALL:
            maxes = dev.amax_new(2)               # zeroed ring arrays: no memset
            dev.call("msk_affine_act_bwd_reduce_pg", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), resm, _fp(alpha),
                     _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(), _fp(sc["sums"]), _fp(maxes), 0,
                     _fp(bn.weight.grad_ptr), _fp(bn.bias.grad_ptr), _fp(_act_alpha_grad(self.act)))
            pg_done = True
        elif want_maxes:
            dev.call("msk_affine_act_bwd_reduce_ex", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), resm, _fp(alpha),
                     _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(), _fp(sc["sums"]), _fp(sc["maxes"]))
        elif FUSE_SMALL:
            dev.call("msk_affine_act_bwd_reduce_pg", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), resm, _fp(alpha),
                     _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(), _fp(sc["sums"]), None, 0,
                     _fp(bn.weight.grad_ptr), _fp(bn.bias.grad_ptr), _fp(_act_alpha_grad(self.act)))
            pg_done = True
        else:
            dev.call("msk_affine_act_bwd_reduce", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), resm, _fp(alpha),
                     _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(), _fp(sc["sums"]))
        sums_total, m_total = sc["sums"], float(y.voxels)
        if (self.bn_mode == 1 and (dev.world > 1 or BatchNorm3D.force_collectives) and BatchNorm3D.sync
                and not BatchNorm3D.dry_run):
            dev.d2d(sc["sums_total"], sc["sums"], 2 * Cn * 4)
            dev.call("msk_dp_allreduce_stats", _fp(sc["sums_total"]), C.c_size_t(2 * Cn))
            sums_total, m_total = sc["sums_total"], float(y.voxels) * dev.world
        if not pg_done:
            dev.call("msk_affine_act_param_grads", Cn, _fp(sc["sums"]), _fp(bn.weight.grad_ptr), _fp(bn.bias.grad_ptr),
                     _fp(_act_alpha_grad(self.act)), 1)
        dy = y.empty_like()
        if (FUSE_BN_BACKWARD and FUSE_BN_BACKWARD_C1 and type(self.conv) is Conv3D and self.bn_mode == 1 and not need_dx
                and self.conv.cin == 1 and self.conv.k == (5, 5, 5) and self.conv.s == (1, 1, 1)
                and (res is None or (res.ptr == self.x.ptr and res.c == 1 and not res_needs_grad))):
            # in_tr (vnet.py:57-79; one input channel, no data gradient, the residual is the tiled input itself): dy is evaluated
            # inside the weight-gradient kernel -- the last weight gradient of the backward pass starts one full-resolution pass
            # earlier.  1 = declined (nothing launched): the separate passes below
            conv, x = self.conv, self.x
            rc = dev.lib.msk_conv3d_bwd_bnact_c1(dev.ctx, conv.desc(), x.msk(), y.msk(), _fp(sc["scale"]), _fp(sc["shift"]),
                                                 _fp(alpha), _fp(sc["mean"]), _fp(sc["invstd"]), resm, dout.msk(),
                                                 _fp(sums_total), C.c_double(m_total), _fp(conv.weight.grad_ptr), 1)
            if rc < 0:
                raise MskError(f"msk_conv3d_bwd_bnact_c1 failed: {_lib.last_error(dev.ctx)}")
            if rc == 0:
                _count_flops(conv, x.n, y.d * y.h * y.w, 1)
                conv._xform = None
                self.dy = None
                return
        if fuse:
            # LUConv class (vnet.py:36-41): BatchNorm/PReLU backward evaluated inside the kernel that writes both transforms
            # of dy (msk_conv3d_bwd_bnact); dy itself reaches HBM only when the shape is not eligible
            conv, x = self.conv, self.x
            xf = getattr(conv, "_xform", None)
            xfp = xf[0] if (xf is not None and xf[1] == x.ptr and xf[2] == dev.arena.gen) else None
            ybuf = None
            if xfp is not None:
                nbytes = int(dev.lib.msk_conv3d_bwd_bnact_bytes(dev.ctx, conv.desc(), x.msk(), y.msk()))
                if nbytes > 0:
                    ybuf = dev.arena.alloc(nbytes)
            dx = x.ensure_grad()
            _count_flops(conv, x.n, y.d * y.h * y.w, 2)
            split = getattr(self, "dx_split", None)     # (lo, hi): dense halves for the gradient of a zero-copy concat (UpTransition.backward)
            self.dx_split, self.dx_split_done = None, False
            old = getattr(x, "grad_from", None)         # the join behind this layer wrote ITS gradient once: the old values of dx live there
            if old is not None and (old.ld != dx.ld or old.c != dx.c or x.grad_written):
                materialize_grad(x)
                old = None
            if old is not None:
                x.grad_from = None
                done = C.c_int(0)
                dev.call("msk_conv3d_bwd_bnact_acc", conv.desc(), x.msk(), _fp(conv.weight.ptr), y.msk(), _fp(sc["scale"]),
                         _fp(sc["shift"]), _fp(alpha), _fp(sc["mean"]), _fp(sc["invstd"]), _fp(bn.weight.ptr), dout.msk(),
                         _fp(sums_total), C.c_double(m_total), dy.msk(), dx.msk(), 1,
                         _fp(conv.weight.grad_ptr), 1, _fp(xfp), _fp(ybuf), _fp(maxes) if want_maxes else None,
                         old.msk(), split[0].msk() if split is not None else NULL_TENSOR,
                         split[1].msk() if split is not None else NULL_TENSOR, C.byref(done))
                self.dx_split_done = bool(done.value)
            elif split is not None and x.grad_written:
                done = C.c_int(0)
                dev.call("msk_conv3d_bwd_bnact_split", conv.desc(), x.msk(), _fp(conv.weight.ptr), y.msk(), _fp(sc["scale"]),
                         _fp(sc["shift"]), _fp(alpha), _fp(sc["mean"]), _fp(sc["invstd"]), _fp(bn.weight.ptr), dout.msk(),
                         _fp(sums_total), C.c_double(m_total), dy.msk(), dx.msk(), 1,
                         _fp(conv.weight.grad_ptr), 1, _fp(xfp), _fp(ybuf), _fp(maxes) if want_maxes else None,
                         split[0].msk(), split[1].msk(), C.byref(done))
                self.dx_split_done = bool(done.value)
            else:
                dev.call("msk_conv3d_bwd_bnact", conv.desc(), x.msk(), _fp(conv.weight.ptr), y.msk(), _fp(sc["scale"]),
                         _fp(sc["shift"]), _fp(alpha), _fp(sc["mean"]), _fp(sc["invstd"]), _fp(bn.weight.ptr), dout.msk(),
                         _fp(sums_total), C.c_double(m_total), dy.msk(), dx.msk(), 1 if x.grad_written else 0,
                         _fp(conv.weight.grad_ptr), 1, _fp(xfp), _fp(ybuf), _fp(maxes) if want_maxes else None)
            x.grad_written = True
            conv._xform = None
            self.dy = dy if ybuf is None else None
            return
        if (FUSE_BN_BACKWARD and FUSE_BN_BACKWARD_T and type(self.conv) is Conv3DTranspose and self.bn_mode == 1 and need_dx
                and res is None and not isinstance(self.act, ELU) and self.conv.cout <= 16):
            # up-convolution unit (vnet.py:133-150): the data gradient evaluates dy from (y, dout) in its own loads; the pass that
            # writes dy runs on the weight-gradient stream in front of the weight gradient.  1 = declined (nothing launched)
            conv, x = self.conv, self.x
            dx = x.ensure_grad()
            rc = dev.lib.msk_convT3d_bwd_bnact(dev.ctx, conv.desc(), x.msk(), _fp(conv.weight.ptr), y.msk(), _fp(sc["scale"]),
                                               _fp(sc["shift"]), _fp(alpha), _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(),
                                               _fp(sums_total), C.c_double(m_total), dy.msk(), dx.msk(),
                                               1 if x.grad_written else 0, _fp(conv.weight.grad_ptr), 1)
            if rc < 0:
                raise MskError(f"msk_convT3d_bwd_bnact failed: {_lib.last_error(dev.ctx)}")
            if rc == 0:
                _count_flops(conv, x.n, x.d * x.h * x.w, 2)
                x.grad_written = True
                self.dy = dy
                return
        dres = NULL_TENSOR
        dres_acc = 0
        if res is not None and res_needs_grad and res.c == y.c:
            g = res.ensure_grad()
            dres, dres_acc = g.msk(), 1 if res.grad_written else 0
            res.grad_written = True
        # max |dy| rides in this pass when a 'same' convolution (the 16-bit pipeline scales its operands) consumes dy
        dya = _amax_for(dy) if type(self.conv) is Conv3D and self.conv.s == (1, 1, 1) else None
        dev.call("msk_affine_act_bwd_apply_amax", y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), resm, _fp(alpha),
                 _fp(sc["mean"]), _fp(sc["invstd"]), _fp(bn.weight.ptr), dout.msk(), _fp(sums_total),
                 C.c_double(m_total), self.bn_mode, dy.msk(), dres, dres_acc, dya)
        self.dy = dy  # kept for introspection (tests); freed with the arena
        # conv bias gradient = sum_v dy[v][c]: identically zero behind a batch-statistics BN (its backward
        # removes the mean), scale * sum(dout * prelu') behind a running-statistics BN -- either way no
        # extra pass over dy (was 0.8 ms of channel-sum reads per step)
        if self.bn_mode == 2:
            dev.call("msk_bn_bias_grad", Cn, _fp(sc["sums"]), _fp(sc["scale"]), _fp(self.conv.bias.grad_ptr), 1)
        self.conv.run_backward(self.x, dy, need_dx=need_dx, bias_grad=False)


class AddAct:
    """out = PReLU(a + b): the residual joins of DownTransition/UpTransition
    (vnet.py:110-111, 154)."""

    def __init__(self, act: PReLU):
        self.act = act
        self._sums = None

    def forward(self, a: Tensor, b: Tensor, unit: "ConvBNAct | None" = None, out: Tensor | None = None) -> Tensor:
        """unit: the conv -> BN -> PReLU unit that produced `a` with defer_act=True: its BatchNorm apply + PReLU run inside
        this join, on the convolution output (one pass instead of two, `a` is never written).  out: destination (may be a
        channel slice of a wider buffer: zero-copy skip connections, UpTransition.reserve_concat)."""
        self.a, self.b = a, b
        self.unit = unit if (unit is not None and getattr(unit, "deferred", False)) else None
        if out is None:
            out = a.empty_like()
        if isinstance(self.act, ELU):
            a.dev.call("msk_affine_act_fwd", a.msk(), None, None, b.msk(), None, out.msk())   # a + b
            a.dev.call("msk_elu_fwd", out.msk(), C.c_float(self.act.alpha), out.msk())
            out.amax = None
            self.out = out
            return out
        if self.unit is not None:
            u, sc = self.unit, self.unit.bn.scratch(a.dev)
            a.dev.call("msk_affine_act_join_fwd_amax", u.y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), _fp(u.act._weight.ptr),
                       b.msk(), _fp(self.act._weight.ptr), out.msk(), _amax_for(out))
        else:
            a.dev.call("msk_affine_act_fwd_amax", a.msk(), None, None, b.msk(), _fp(self.act._weight.ptr), out.msk(),
                       _amax_for(out))
        return out

    def backward(self, dout: Tensor, share_b=False):
        """share_b: d(a + b) goes to both operands unchanged -- write it ONCE (a.grad) and mark b's gradient as living there
        (b.grad_from) when nothing has been written to b.grad yet; the layer that accumulates onto it next reads a.grad as its old
        values (ConvBNAct.backward -> msk_conv3d_bwd_bnact_acc).  The caller guarantees that this layer is b's next consumer and
        calls nn.materialize_grad(b) afterwards in case it was not."""
        dev = dout.dev
        a, b = self.a, self.b
        Cn = a.c
        if self._sums is None:
            self._sums = dev.small(3 * Cn)
        ga, gb = a.ensure_grad(), b.ensure_grad()
        if a.grad_written:
            raise MskError("AddAct.backward expects to be the first writer of its first operand's gradient")
        share_b = bool(share_b and SHARE_JOIN_GRAD and not b.grad_written and not isinstance(self.act, ELU) and ga.ld == gb.ld
                       and Cn % 4 == 0 and all(t.ld % 4 == 0 and t.ptr % 16 == 0 for t in (a, b, dout, ga)))
        gbm = NULL_TENSOR if share_b else gb.msk()
        if isinstance(self.act, ELU):
            g = Tensor.empty(dev, dout.n, dout.d, dout.h, dout.w, dout.c)
            dev.call("msk_elu_bwd", self.out.msk(), dout.msk(), C.c_float(self.act.alpha), g.msk(), 0)
            # d(a + b): the gradient goes to both operands unchanged
            dev.call("msk_affine_act_bwd_apply", a.msk(), None, None, b.msk(), None, None, None, None, g.msk(),
                     None, C.c_double(1.0), 0, ga.msk(), gb.msk(), 1 if b.grad_written else 0)
            a.grad_written = True
            b.grad_written = True
            return
        alpha = _fp(self.act._weight.ptr)
        if getattr(self, "unit", None) is not None:
            # one pass: the join's data gradients and slope gradient AND the sums (and maxima) the unit's own backward starts
            # with -- its reduce pass is skipped (ConvBNAct.backward, presummed)
            u, sc = self.unit, self.unit.bn.scratch(dev)
            if FUSE_SMALL:
                mx = dev.amax_new(2)
                dev.call("msk_add_act_join_bwd_pg", u.y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), _fp(u.act._weight.ptr),
                         b.msk(), alpha, _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(), ga.msk(), gbm,
                         1 if b.grad_written else 0, _fp(self.act._weight.grad_ptr), _fp(sc["sums"]), _fp(mx), 0,
                         _fp(u.bn.weight.grad_ptr), _fp(u.bn.bias.grad_ptr), _fp(u.act._weight.grad_ptr))
                u.presummed_pg, u.presummed_maxes = True, mx
            else:
                dev.call("msk_add_act_join_bwd_ex", u.y.msk(), _fp(sc["scale"]), _fp(sc["shift"]), _fp(u.act._weight.ptr),
                         b.msk(), alpha, _fp(sc["mean"]), _fp(sc["invstd"]), dout.msk(), ga.msk(), gbm,
                         1 if b.grad_written else 0, _fp(self.act._weight.grad_ptr), _fp(sc["sums"]), _fp(sc["maxes"]))
                u.presummed_pg, u.presummed_maxes = False, None
            u.presummed = True
        elif Cn % 4 == 0 and all(t.ld % 4 == 0 and t.ptr % 16 == 0 for t in (a, b, dout, ga, gb)):
            # one pass: both data gradients and the alpha-gradient sum (a join has no BatchNorm)
            dev.call("msk_add_act_bwd", a.msk(), b.msk(), alpha, dout.msk(), ga.msk(), gbm,
                     1 if b.grad_written else 0, _fp(self.act._weight.grad_ptr))
        else:
            dev.call("msk_affine_act_bwd_reduce", a.msk(), None, None, b.msk(), alpha, None, None, dout.msk(),
                     _fp(self._sums))
            dev.call("msk_affine_act_param_grads", Cn, _fp(self._sums), None, None, _fp(self.act._weight.grad_ptr), 1)
            dev.call("msk_affine_act_bwd_apply", a.msk(), None, None, b.msk(), alpha, None, None, None, dout.msk(),
                     None, C.c_double(1.0), 0, ga.msk(), gb.msk(), 1 if b.grad_written else 0)
        a.grad_written = True
        if share_b:
            b.grad_from = ga          # b.grad itself is still unwritten
        else:
            b.grad_written = True


def copy_scale(src: Tensor, mask_ptr, dst: Tensor, accumulate=False):
    if accumulate:      # the destination's old contents are not covered by any amax array
        dst.amax = None
        src.dev.call("msk_copy_scale", src.msk(), _fp(mask_ptr), dst.msk(), 1)
    else:
        src.dev.call("msk_copy_scale_amax", src.msk(), _fp(mask_ptr), dst.msk(), 0, _amax_for(dst))
