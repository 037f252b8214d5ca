"""V-Net on MI355X -- same classes, constructor arguments, attribute names (hence
state-dict keys) and forward contract as the reference's medicalseg/models/vnet.py
(:32-267), executed by fused HIP kernels in NDHWC.

forward(x[N,C,D,H,W]) -> [logits]; the backward pass is explicit (VNet.backward) and is
entered from the loss (``loss.backward()``, core/train.py:139)."""
from __future__ import annotations

import os

import numpy as np

from .. import nn
from ..cvlibs import manager
from ..device import Tensor, get_device, to_tensor
from ..nn import AddAct, ConvBNAct, copy_scale

# A/B switches (round 5): env MSEGK_DENSE_TWIN=0 -- in_tr writes only the skip half of up_tr32's concat buffer (down_tr32 reads
# 64 of every 128-byte line); env MSEGK_SPLIT_CONCAT_GRAD=0 -- the concat gradient of a 16 + 16-channel up-transition stays one
# interleaved buffer
DENSE_TWIN = os.environ.get("MSEGK_DENSE_TWIN", "1") != "0"
SPLIT_CONCAT_GRAD = os.environ.get("MSEGK_SPLIT_CONCAT_GRAD", "1") != "0"


class LUConv(nn.Layer):
    """conv5^3(p=2) -> BN -> PReLU (reference vnet.py:32-43)."""

    def __init__(self, nchan, elu):
        super(LUConv, self).__init__()
        self.relu1 = nn.ELU() if elu else nn.PReLU(nchan)
        self.conv1 = nn.Conv3D(nchan, nchan, kernel_size=5, padding=2)
        self.bn1 = nn.BatchNorm3D(nchan)
        self._unit = ConvBNAct(self.conv1, self.bn1, self.relu1)

    def forward(self, x):
        return self._unit.forward(x)

    def backward(self, dout):
        self._unit.backward(dout)


def _join_fusable(t, res):
    """msk_affine_act_join_fwd / msk_add_act_join_bwd take float4-aligned tensors of equal channel count"""
    return (t.c % 4 == 0 and res.c == t.c and all(x.ld % 4 == 0 and x.ptr % 16 == 0 for x in (t, res))
            and os.environ.get("MSEGK_JOIN_FUSE", "1") != "0")


def _run_ops(ops, x, res):
    """The LUConv chain of a stage; the LAST unit leaves its BatchNorm apply + PReLU to the residual join that follows
    (vnet.py:107-111, 150-154): returns (its output tensor -- unwritten in that case --, the unit or None)."""
    ops = list(ops)
    out = x
    for op in ops[:-1]:
        out = op(out)
    if not ops:
        return out, None
    last = ops[-1]
    if isinstance(last, LUConv) and _join_fusable(out, res):
        out = last._unit.forward(out, defer_act=True)
        return out, (last._unit if last._unit.deferred else None)
    return last(out), None


def _make_nConv(nchan, depth, elu):
    return nn.Sequential(*[LUConv(nchan, elu) for _ in range(depth)])


class InputTransition(nn.Layer):
    """conv5^3(in->16) -> BN -> + tile(x) -> PReLU (reference vnet.py:57-79)."""

    def __init__(self, in_channels, elu):
        super(InputTransition, self).__init__()
        self.num_features = 16
        self.in_channels = in_channels
        if self.num_features % in_channels:
            raise ValueError("in_channels must divide 16 (x.tile in the reference, vnet.py:76-78)")
        self.conv1 = nn.Conv3D(self.in_channels, self.num_features, kernel_size=5, padding=2)
        self.bn1 = nn.BatchNorm3D(self.num_features)
        self.relu1 = nn.ELU() if elu else nn.PReLU(self.num_features)
        self._unit = ConvBNAct(self.conv1, self.bn1, self.relu1)

    def forward(self, x, out=None, out2=None):
        """out: where the block's output goes (a channel slice of the concat buffer of the up-transition that takes it as
        its skip connection: UpTransition.reserve_concat); out2: a second, dense copy for the readers of that slice alone"""
        # the tiled input is the residual: channel c of the sum reads x[..., c % in_channels]
        return self._unit.forward(x, res=x, out=out, out2=out2)

    def backward(self, dout):
        self._unit.backward(dout, need_dx=False, res_needs_grad=False)


class DownTransition(nn.Layer):
    """reference vnet.py:82-113."""

    def __init__(self, inChans, nConvs, elu, dropout=False, downsample_stride=(2, 2, 2), kernel=(2, 2, 2)):
        super(DownTransition, self).__init__()
        outChans = 2 * inChans
        self.if_dropout = dropout
        self.down_conv = nn.Conv3D(inChans, outChans, kernel_size=kernel, stride=downsample_stride)
        self.bn1 = nn.BatchNorm3D(outChans)
        self.relu1 = nn.ELU() if elu else nn.PReLU(outChans)
        self.relu2 = nn.ELU() if elu else nn.PReLU(outChans)
        self.dropout = nn.Dropout3D()
        self.ops = _make_nConv(outChans, nConvs, elu)
        self._down = ConvBNAct(self.down_conv, self.bn1, self.relu1)
        self._join = AddAct(self.relu2)

    def forward(self, x, out=None):
        """out: see InputTransition.forward"""
        down = self._down.forward(x)
        self._mask = self.dropout.make_mask(down) if self.if_dropout else None
        if self._mask is not None:
            t = down.empty_like()
            copy_scale(down, self._mask, t)
        else:
            t = down
        self._dropped = t
        self._t_down = down
        t, unit = _run_ops(self.ops, t, down)
        self._t_ops = t
        return self._join.forward(t, down, unit=unit, out=out)

    def backward(self, dout):
        down = self._t_down
        ops = list(self.ops)
        # without a dropout between them the first LUConv reads `down` itself: the join's gradient is written once (ops_out.grad) and
        # that layer's accumulating data gradient takes its old values from there (nn.AddAct.backward, share_b)
        self._join.backward(dout, share_b=bool(ops) and self._mask is None and ops[0]._unit.x is down)
        for op in reversed(ops):
            g = op._unit.out.grad
            op.backward(g)
        nn.materialize_grad(down)      # (no-op when ops[0] consumed the shared gradient)
        if self._mask is not None:
            copy_scale(self._dropped.grad, self._mask, down.grad, accumulate=True)
        self._down.backward(down.grad)


class UpTransition(nn.Layer):
    """reference vnet.py:116-156."""

    def __init__(self, inChans, outChans, nConvs, elu, dropout=False, dropout2=False,
                 upsample_stride_size=(2, 2, 2), kernel=(2, 2, 2)):
        super(UpTransition, self).__init__()
        self.up_conv = nn.Conv3DTranspose(inChans, outChans // 2, kernel_size=kernel, stride=upsample_stride_size)
        self.bn1 = nn.BatchNorm3D(outChans // 2)
        self.relu1 = nn.ELU() if elu else nn.PReLU(outChans // 2)
        self.relu2 = nn.ELU() if elu else nn.PReLU(outChans)
        self.if_dropout = dropout
        self.if_dropout2 = dropout2
        self.dropout1 = nn.Dropout3D()
        self.dropout2 = nn.Dropout3D()
        self.ops = _make_nConv(outChans, nConvs, elu)
        self.outChans = outChans
        self._up = ConvBNAct(self.up_conv, self.bn1, self.relu1)
        self._join = AddAct(self.relu2)

    def reserve_concat(self, dev, n, dims):
        """Zero-copy skip connection: allocate this block's concat buffer (vnet.py:152) AHEAD of the encoder block that
        produces the skip tensor and return the channel slice that block should write its output into -- the skip half of
        the concat is then never copied (forward) and its gradient never copied back (backward).  None when the skip path
        has a dropout (the copy applies the mask) or MSEGK_ZERO_COPY_SKIP=0."""
        self._reserved = None
        if self.if_dropout2 or os.environ.get("MSEGK_ZERO_COPY_SKIP", "1") == "0":
            return None
        half = self.outChans // 2
        xcat = Tensor.empty(dev, n, dims[0], dims[1], dims[2], self.outChans)
        if nn.PRODUCER_AMAX:
            xcat.amax = dev.amax_new()     # shared by both halves' writers (Tensor.amax)
        self._reserved = xcat
        self._reserved_skip = xcat.channel_slice(half, self.outChans)
        return self._reserved_skip

    def forward(self, x, skipx):
        dev = x.dev
        self._x, self._skip = x, skipx
        self._m1 = self.dropout1.make_mask(x) if self.if_dropout else None
        self._m2 = self.dropout2.make_mask(skipx) if self.if_dropout2 else None
        xin = x
        if self._m1 is not None:
            xin = x.empty_like()
            copy_scale(x, self._m1, xin)
        self._xin = xin
        half = self.outChans // 2
        od, oh, ow = self.up_conv.out_dims(xin)
        if (od, oh, ow) != (skipx.d, skipx.h, skipx.w) or skipx.c != self.outChans - half:
            raise ValueError(f"skip connection shape {skipx.shape} does not match the up-sampled "
                             f"({x.n}, {half}, {od}, {oh}, {ow})")
        # concat (vnet.py:152) is two channel-slice writes into one NDHWC buffer -- or one, when the skip tensor already
        # lives in its half (reserve_concat)
        res = getattr(self, "_reserved", None)
        self._reserved = None
        self._skip_in_place = (res is not None and self._m2 is None and skipx.ld == res.ld and
                               skipx.ptr == res.ptr + 4 * half and res.gen == dev.arena.gen)
        if self._skip_in_place:
            xcat, s_skip = res, skipx
        else:
            xcat = Tensor.empty(dev, x.n, od, oh, ow, self.outChans)
            if nn.PRODUCER_AMAX:
                xcat.amax = dev.amax_new()
            s_skip = xcat.channel_slice(half, self.outChans)
        s_up = xcat.channel_slice(0, half)
        self._up.forward(xin, out=s_up)
        if not self._skip_in_place:
            copy_scale(skipx, self._m2, s_skip)
        # max |xcat| is known iff BOTH halves were written by passes that folded into xcat's array (a writer that cannot --
        # ELU, a folded inference conv -- clears its slice's reference)
        if xcat.amax is None or s_up.amax != xcat.amax or s_skip.amax != xcat.amax:
            xcat.amax = None
        self._xcat = xcat
        out, unit = _run_ops(self.ops, xcat, xcat)
        return self._join.forward(out, xcat, unit=unit)

    def backward(self, dout):
        xcat, half = self._xcat, self.outChans // 2
        ops = list(self.ops)
        self._join.backward(dout, share_b=bool(ops) and ops[0]._unit.x is xcat)  # -> ops_out.grad (+ xcat.grad unless shared, DownTransition.backward)
        # halves narrower than a 128-byte line (16 channels): ask the layer that completes xcat.grad -- ops[0], accumulating its
        # data gradient behind the join's -- to store the sums as two DENSE tensors (msk_conv3d_bwd_bnact_split); the readers of
        # one half then do not fetch the other with it (round 5).  Honoured only by the one-kernel matrix stage: else as before
        g_lo = g_hi = None
        skip = self._skip
        first = ops[0]._unit if ops else None       # nConvs = 0 (vnet.py:130-131 allows it): no layer behind the concat, nothing to split
        if (first is not None and SPLIT_CONCAT_GRAD and self._skip_in_place and skip.grad is None and half * 4 < 128 and self._m1 is None):
            dev = xcat.dev
            g_lo = Tensor.empty(dev, xcat.n, xcat.d, xcat.h, xcat.w, half)
            g_hi = Tensor.empty(dev, xcat.n, xcat.d, xcat.h, xcat.w, half)
            first.dx_split = (g_lo, g_hi)
        if first is not None:
            first.dx_split_done = False
        for op in reversed(ops):
            op.backward(op._unit.out.grad)
        nn.materialize_grad(xcat)
        gcat = xcat.grad
        split_done = g_lo is not None and getattr(first, "dx_split_done", False)
        if first is not None:
            first.dx_split = None
        # skip branch
        if split_done:
            skip.grad = g_hi                      # the skip half's gradient as a dense tensor; later writers accumulate into it
        elif self._skip_in_place and skip.grad is None:
            # the skip tensor IS the second half of the concat: so is its gradient (later writers accumulate into the slice)
            skip.grad = gcat.channel_slice(half, self.outChans)
        else:
            sg = skip.ensure_grad()
            copy_scale(gcat.channel_slice(half, self.outChans), self._m2, sg, accumulate=skip.grad_written)
        skip.grad_written = True
        # up-conv branch
        self._up.backward(g_lo if split_done else gcat.channel_slice(0, half))
        if self._m1 is not None:
            x = self._x
            xg = x.ensure_grad()
            copy_scale(self._xin.grad, self._m1, xg, accumulate=x.grad_written)
            x.grad_written = True


class OutputTransition(nn.Layer):
    """conv5^3(32->ncls) -> BN -> PReLU -> conv1^3 (reference vnet.py:159-175)."""

    def __init__(self, in_channels, num_classes, elu):
        super(OutputTransition, self).__init__()
        self.conv1 = nn.Conv3D(in_channels, num_classes, kernel_size=5, padding=2)
        self.bn1 = nn.BatchNorm3D(num_classes)
        self.conv2 = nn.Conv3D(num_classes, num_classes, kernel_size=1)
        self.relu1 = nn.ELU() if elu else nn.PReLU(num_classes)
        self._unit = ConvBNAct(self.conv1, self.bn1, self.relu1)

    def forward(self, x):
        o = self._unit.forward(x)
        self._o = o
        return self.conv2.run_forward(o)

    def backward(self, dlogits):
        self.conv2.run_backward(self._o, dlogits, need_dx=True)
        self._unit.backward(self._o.grad)


@manager.MODELS.add_component
class VNet(nn.Layer):
    """Implementation of https://arxiv.org/abs/1606.04797 with the reference's
    constructor (vnet.py:184-190)."""

    def __init__(self, elu=False, in_channels=1, num_classes=4, pretrained=None,
                 kernel_size=((2, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2)),
                 stride_size=((2, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2))):
        super().__init__()
        self.best_loss = 1000000
        self.num_classes = num_classes
        self.in_channels = in_channels

        self.in_tr = InputTransition(in_channels, elu=elu)
        self.down_tr32 = DownTransition(16, 1, elu, downsample_stride=stride_size[0], kernel=kernel_size[0])
        self.down_tr64 = DownTransition(32, 2, elu, downsample_stride=stride_size[1], kernel=kernel_size[1])
        self.down_tr128 = DownTransition(64, 3, elu, dropout=True, downsample_stride=stride_size[2],
                                         kernel=kernel_size[2])
        self.down_tr256 = DownTransition(128, 2, elu, dropout=True, downsample_stride=stride_size[3],
                                         kernel=kernel_size[3])
        self.up_tr256 = UpTransition(256, 256, 2, elu, dropout=True, dropout2=True,
                                     upsample_stride_size=stride_size[3], kernel=kernel_size[3])
        self.up_tr128 = UpTransition(256, 128, 2, elu, dropout=True, dropout2=True,
                                     upsample_stride_size=stride_size[2], kernel=kernel_size[2])
        self.up_tr64 = UpTransition(128, 64, 1, elu, upsample_stride_size=stride_size[1], kernel=kernel_size[1])
        self.up_tr32 = UpTransition(64, 32, 1, elu, upsample_stride_size=stride_size[0], kernel=kernel_size[0])
        self.out_tr = OutputTransition(32, num_classes, elu)

        self.pretrained = pretrained
        self._post_backward_hooks = []
        self._grad_ready_hooks = []      # called with (model, block) once a block's backward is enqueued (parallel.py)
        self._build()
        self.init_weight()

    # -- device residency -----------------------------------------------------------
    def _build(self):
        dev = get_device()
        params, frozen = [], []
        for name, p in self.named_parameters():
            p.name = name
            (frozen if getattr(p, "frozen", False) else params).append(p)
        bufs = []
        for name, p in self.named_buffers():
            p.name = name
            bufs.append(p)
        self.arena = nn.ParamArena(dev, params, with_grad=True)
        self.buffer_arena = nn.ParamArena(dev, bufs, with_grad=False)
        # parameters no forward path reaches (VNetDeepSup.out_tr_all): stored, never updated
        self.frozen_arena = nn.ParamArena(dev, frozen, with_grad=False) if frozen else None
        self.dev = dev

    def init_weight(self):
        if self.pretrained is not None:
            from ..utils import utils
            utils.load_entire_model(self, self.pretrained)

    def dropout_layers(self):
        """site name -> Dropout3D (sites as in oracle/vnet_numpy.py)."""
        return {"down_tr128": self.down_tr128.dropout, "down_tr256": self.down_tr256.dropout,
                "up_tr256.x": self.up_tr256.dropout1, "up_tr256.skip": self.up_tr256.dropout2,
                "up_tr128.x": self.up_tr128.dropout1, "up_tr128.skip": self.up_tr128.dropout2}

    def set_dropout_masks(self, masks):
        """Inject Dropout3D masks ([N, C] multipliers per site; None -> identity) -- parity
        tests only; training uses the counter-based device RNG."""
        for site, layer in self.dropout_layers().items():
            if masks is None:
                layer.injected, layer.enabled = None, True
            elif site in masks and masks[site] is not None:
                layer.injected, layer.enabled = np.asarray(masks[site], dtype=np.float32), True
            else:
                layer.injected, layer.enabled = None, False

    # -- forward / backward ---------------------------------------------------------------
    def forward(self, x):
        if not isinstance(x, Tensor):
            x = to_tensor(x, self.dev)
        if x.c != self.in_channels:
            raise ValueError(f"VNet expects {self.in_channels} input channel(s), got {x.c}")
        self.dev.arena.reset()
        if self.training:
            nn.Dropout3D.step += 1
        # the two dropout-free skip connections are produced straight into the concat buffers of their up-transitions
        slot = self.up_tr32.reserve_concat(self.dev, x.n, (x.d, x.h, x.w))
        # a 16-channel half of the 32-channel concat voxel is 64 of every 128-byte line: the readers of the skip half alone (the
        # down-convolution and its weight gradient) get a dense twin written by the same pass (DENSE_TWIN, round 5)
        twin = None
        if slot is not None and DENSE_TWIN and self.training and slot.c * 4 < 128 and not isinstance(self.in_tr.relu1, nn.ELU):   # (eval: no backward reads it)
            twin = Tensor.empty(self.dev, x.n, x.d, x.h, x.w, slot.c)
        out16 = self.in_tr(x, out=slot, out2=twin)
        self._out16_twin = twin
        out32 = self.down_tr32(twin if twin is not None else out16,
                               out=self.up_tr64.reserve_concat(self.dev, x.n, self.down_tr32.down_conv.out_dims(out16)))
        out64 = self.down_tr64(out32)
        out128 = self.down_tr128(out64)
        out256 = self.down_tr256(out128)
        out = self.up_tr256(out256, out128)
        out = self.up_tr128(out, out64)
        out = self.up_tr64(out, out32)
        self._feat = self.up_tr32(out, out16)
        logits = self.out_tr(self._feat)
        logits.producer = self
        self._acts = (out16, out32, out64, out128, out256)
        return [logits, ]

    def backward(self, dlogits: Tensor):
        """Adjoint of forward; accumulates parameter gradients into the flat arena."""
        out16, out32, out64, out128, out256 = self._acts
        twin = getattr(self, "_out16_twin", None)

        def share_twin_grad():
            # the dense twin of out16 (forward) is what down_tr32 read: its data gradient accumulates into out16's gradient
            if twin is not None and out16.grad is not None:
                twin.grad, twin.grad_written = out16.grad, True
        for block, dout in ((self.out_tr, lambda: dlogits),
                            (self.up_tr32, lambda: self._feat.grad),
                            (self.up_tr64, lambda: self.up_tr32._x.grad),
                            (self.up_tr128, lambda: self.up_tr64._x.grad),
                            (self.up_tr256, lambda: self.up_tr128._x.grad),
                            (self.down_tr256, lambda: out256.grad),
                            (self.down_tr128, lambda: out128.grad),
                            (self.down_tr64, lambda: out64.grad),
                            (self.down_tr32, lambda: out32.grad),
                            (self.in_tr, lambda: out16.grad)):
            block.backward(dout())
            if block is self.up_tr32:
                share_twin_grad()
            self._grads_ready(block)
        for hook in self._post_backward_hooks:
            hook(self)

    def _grads_ready(self, block):
        """The block's parameter gradients are all enqueued: data parallelism may put them on the wire."""
        for hook in self._grad_ready_hooks:
            hook(self, block)

    def test(self):
        np.random.seed(1)
        a = np.random.rand(1, self.in_channels, 32, 32, 32)
        out = self.forward(a.astype("float32"))[0]
        assert out.shape == (1, self.num_classes, 32, 32, 32)
        print("out", out.numpy().mean(), a.mean())
        print("Vnet test is complete")
