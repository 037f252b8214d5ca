"""UNet3D -- BUILDER-DEFINED: the reference has no UNet3D (its READMEs say "Unet ... To be
continue", configs/lung_coronavirus/README.md:15-21); BASELINE.json's configs[3] names one, so this is the plain
3-D U-Net that config describes -- 3x3x3 convolutions, InstanceNorm + PReLU, kernel = stride 2 down / transposed-up
convolutions, skip concatenation -- assembled from the SAME C-ABI kernels as VNet (`msk_conv3d_*` on the MFMA halo
kernel with KS = 3, `msk_convT3d_*`, the BatchNorm statistics / affine+PReLU passes run per sample for the instance
statistics).  `precision="fp16"` (BASELINE configs[3]) runs the 3x3x3 convolutions -- forward, data gradient and weight
gradient -- with fp16 operands on the fp16 matrix pipe (Winograd F(4,3) domain, fp32 accumulation; activations, normalisation
statistics, losses and the optimizer stay fp32: msk_conv_wbf.hip, option "conv_fp16"); the default "fp32" uses the exact
bf16x3 operands.  With no reference model there is no parity claim beyond the torch-CPU restatement in
oracle/unet3d_torch.py (tests/test_gpu_unet3d.py: fp32 at 3e-4, fp16 at the stated fp16 tolerance).

Registered as `UNet3D` so a YAML `model: {type: UNet3D, ...}` builds it through the unchanged Config path."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import nn
from ..cvlibs import manager
from ..device import Tensor, to_tensor
from .._lib import MskBnFin
from ..nn import NULL_TENSOR, ConvBNAct, Parameter, _amax_for, _fp, copy_scale

FUSED_IN_BACKWARD = True        # False = InstanceNorm backward as reduce + apply passes with dy in HBM everywhere (A/B, tests)
ZERO_COPY_SKIPS = True          # False = the skip tensors are copied into / out of the concat buffers (A/B, tests)
INSTANCE_STATS_IN_CONV = True   # False = statistics + finalisation as separate passes per sample (A/B, tests)
from .vnet import VNet


class InstanceNorm3D(nn.Layer):
    """paddle.nn.InstanceNorm3D(num_features, epsilon=1e-5): statistics per (sample, channel) over D*H*W in training
    AND eval mode, learnable `scale` / `bias`, no running statistics."""

    def __init__(self, num_features, epsilon=1e-5):
        super().__init__()
        self.num_features = int(num_features)
        self.epsilon = float(epsilon)
        self.scale = Parameter(np.ones(num_features))
        self.bias = Parameter(np.zeros(num_features))
        self._scratch = None

    @property
    def weight(self):
        return self.scale

    def scratch(self, dev, n):
        """stats[2C] sums[3C] dummy running mean/var [2C], then per sample: scale shift mean invstd [4C]."""
        Cn = self.num_features
        if self._scratch is None or self._scratch["n"] < n:
            base = dev.small(7 * Cn + n * 4 * Cn)
            s = {"n": n, "stats": base, "sums": base + 4 * 2 * Cn, "rmean": base + 4 * 5 * Cn, "rvar": base + 4 * 6 * Cn,
                 "per": base + 4 * 7 * Cn}
            self._scratch = s
        return self._scratch

    def sample_stats(self, dev, n):
        """[n][2C] statistics records (mean, M2) of msk_conv3d_fwd_in"""
        if getattr(self, "_sstats", None) is None or self._sstats[0] < n:
            self._sstats = (n, dev.small(n * 2 * self.num_features))
        return self._sstats[1]

    def sample_sums(self, dev, n):
        """[n][3C] backward sums (sum du, sum du xhat, alpha gradient) kept per sample for msk_conv3d_bwd_inact"""
        if getattr(self, "_ssums", None) is None or self._ssums[0] < n:
            self._ssums = (n, dev.small(n * 3 * self.num_features))
        return self._ssums[1]

    def sample_coeffs(self, sc, i):
        Cn, p = self.num_features, sc["per"] + 4 * i * 4 * self.num_features
        return {"scale": p, "shift": p + 4 * Cn, "mean": p + 8 * Cn, "invstd": p + 12 * Cn}


def _sample(t: Tensor, i: int) -> Tensor:
    """View of sample i of an NDHWC tensor (channel-slice views keep their voxel stride)."""
    v = Tensor(t.dev, t.ptr + 4 * i * t.d * t.h * t.w * t.ld, 1, t.d, t.h, t.w, t.c, t.ld, t.gen)
    return v


class ConvINAct(ConvBNAct):
    """conv (or convT) -> InstanceNorm -> PReLU: the BatchNorm unit of nn.ConvBNAct with the statistics, the
    normalise+PReLU pass and their adjoints run once per sample (one sample = one instance)."""

    def forward(self, x: Tensor, res: Tensor | None = None, out: Tensor | None = None) -> Tensor:
        if res is not None:
            raise ValueError("ConvINAct has no residual input")
        dev, norm = x.dev, self.bn
        self.x, self.res = x, None
        alpha = self.act._weight.ptr if self.act is not None else None
        if type(self.conv) is nn.Conv3D and INSTANCE_STATS_IN_CONV:
            # round 4: the instance statistics AND their finalisation come out of the convolution call -- from the per-tile
            # records of the one-kernel matrix stage where it runs (<= 64 channels: the two fine levels), else from one statistics
            # pass per sample inside the call (msk_conv3d_fwd_in); the transformed input is kept for the weight gradient
            conv = self.conv
            od, oh, ow = conv.out_dims(x)
            y = Tensor.empty(dev, x.n, od, oh, ow, conv.cout)
            sc, Cn = norm.scratch(dev, y.n), norm.num_features
            nn._count_flops(conv, x.n, od * oh * ow, 1)
            nbytes = int(dev.lib.msk_conv3d_xform_bytes(dev.ctx, conv.desc(), x.msk(), conv.cout))
            xf = dev.arena.alloc(nbytes) if nbytes > 0 else None
            co0 = norm.sample_coeffs(sc, 0)
            fin = MskBnFin(norm.scale.ptr, norm.bias.ptr, norm.epsilon, 1.0, float(od * oh * ow), None, None, co0["mean"],
                           co0["invstd"], co0["scale"], co0["shift"])
            dev.call("msk_conv3d_fwd_in", conv.desc(), x.msk(), _fp(conv.weight.ptr), _fp(conv.bias.ptr), y.msk(),
                     _fp(norm.sample_stats(dev, y.n)), _fp(xf), _fp(x.amax) if (x.amax and nn.PRODUCER_AMAX) else None,
                     C.byref(fin), 4 * Cn)
            conv._xform = (xf, x.ptr, dev.arena.gen) if xf else None
            self.y = y
            if out is None:
                out = y.empty_like()
            am = _amax_for(out)
            for i in range(y.n):
                co = norm.sample_coeffs(sc, i)
                dev.call("msk_affine_act_fwd_amax", _sample(y, i).msk(), _fp(co["scale"]), _fp(co["shift"]), NULL_TENSOR, _fp(alpha),
                         _sample(out, i).msk(), am)
            self.out, self.bn_mode = out, 1
            return out
        # a Conv3D keeps its transformed input for the weight gradient when the bf16x3 / fp16 pipeline takes the shape
        y = self.conv.run_forward(x, keep_xform=True) if type(self.conv) is nn.Conv3D else self.conv.run_forward(x)
        self.y = y
        if out is None:
            out = y.empty_like()
        sc, Cn = norm.scratch(dev, y.n), norm.num_features
        vox = float(y.d * y.h * y.w)
        am = _amax_for(out)   # max |out| over all samples rides in the normalise passes (the next 3^3 conv scales by it)
        for i in range(y.n):
            yv, co = _sample(y, i), norm.sample_coeffs(sc, i)
            dev.call("msk_bn_stats", yv.msk(), _fp(sc["stats"]))
            dev.call("msk_bn_finalize", _fp(sc["stats"]), 1, C.c_double(vox), Cn, _fp(norm.scale.ptr), _fp(norm.bias.ptr),
                     C.c_float(norm.epsilon), C.c_float(1.0), _fp(sc["rmean"]), _fp(sc["rvar"]), _fp(co["mean"]),
                     _fp(co["invstd"]), _fp(co["scale"]), _fp(co["shift"]))
            dev.call("msk_affine_act_fwd_amax", yv.msk(), _fp(co["scale"]), _fp(co["shift"]), NULL_TENSOR, _fp(alpha),
                     _sample(out, i).msk(), am)
        self.out, self.bn_mode = out, 1
        return out

    def backward(self, dout: Tensor, need_dx=True, res_needs_grad=True):
        dev, norm, y = dout.dev, self.bn, self.y
        sc, Cn = norm.scratch(dev, y.n), norm.num_features
        alpha = self.act._weight.ptr if self.act is not None else None
        vox = float(y.d * y.h * y.w)
        conv, x = self.conv, self.x
        xf = getattr(conv, "_xform", None)
        xfp = xf[0] if (xf is not None and xf[1] == x.ptr and xf[2] == dev.arena.gen) else None
        vec = Cn % 4 == 0 and y.ld % 4 == 0 and dout.ld % 4 == 0 and y.ptr % 16 == 0 and dout.ptr % 16 == 0
        if (FUSED_IN_BACKWARD and type(conv) is nn.Conv3D and conv.cin == conv.cout and conv.s == (1, 1, 1) and need_dx
                and xfp is not None and vec):
            # round 4: the InstanceNorm / PReLU backward is evaluated inside the kernel that writes both transforms of dy, with
            # the coefficients and sums of the block's own sample (msk_conv3d_bwd_inact): per sample only the reduce pass is left
            agrad = _fp(self.act._weight.grad_ptr) if self.act is not None else None
            sums, stride = norm.sample_sums(dev, y.n), 3 * Cn
            maxes = dev.amax_new(2)      # zeroed ring arrays: the reduce passes of all samples fold into them
            for i in range(y.n):
                yv, dv, co = _sample(y, i), _sample(dout, i), norm.sample_coeffs(sc, i)
                dev.call("msk_affine_act_bwd_reduce_pg", yv.msk(), _fp(co["scale"]), _fp(co["shift"]), NULL_TENSOR, _fp(alpha),
                         _fp(co["mean"]), _fp(co["invstd"]), dv.msk(), _fp(sums + 4 * i * stride), _fp(maxes), 0,
                         _fp(norm.scale.grad_ptr), _fp(norm.bias.grad_ptr), agrad)
            nbytes = int(dev.lib.msk_conv3d_bwd_bnact_bytes(dev.ctx, conv.desc(), x.msk(), y.msk()))
            rc = 1
            if nbytes > 0:
                ybuf = dev.arena.alloc(nbytes)
                dx = x.ensure_grad()
                co0 = norm.sample_coeffs(sc, 0)
                rc = dev.lib.msk_conv3d_bwd_inact(dev.ctx, conv.desc(), x.msk(), _fp(conv.weight.ptr), y.msk(), _fp(co0["scale"]),
                                                  _fp(co0["shift"]), _fp(alpha), _fp(co0["mean"]), _fp(co0["invstd"]), 4 * Cn,
                                                  dout.msk(), _fp(sums), stride, C.c_double(vox), dx.msk(),
                                                  1 if x.grad_written else 0, _fp(conv.weight.grad_ptr), 1, _fp(xfp), _fp(ybuf),
                                                  _fp(maxes))
                if rc < 0:
                    from .._lib import MskError, last_error
                    raise MskError(f"msk_conv3d_bwd_inact failed: {last_error(dev.ctx)}")
            if rc == 0:
                nn._count_flops(conv, x.n, y.d * y.h * y.w, 2)
                x.grad_written = True
                conv._xform = None
                self.dy = None
                return
            # declined (nothing launched): dy through HBM from the sums already taken
            dy = y.empty_like()
            dya = _amax_for(dy)
            for i in range(y.n):
                yv, dv, co = _sample(y, i), _sample(dout, i), norm.sample_coeffs(sc, i)
                dev.call("msk_affine_act_bwd_apply_amax", yv.msk(), _fp(co["scale"]), _fp(co["shift"]), NULL_TENSOR, _fp(alpha),
                         _fp(co["mean"]), _fp(co["invstd"]), _fp(norm.scale.ptr), dv.msk(), _fp(sums + 4 * i * stride),
                         C.c_double(vox), 1, _sample(dy, i).msk(), NULL_TENSOR, 0, dya)
            self.dy = dy
            self.conv.run_backward(self.x, dy, need_dx=need_dx, bias_grad=False)
            return
        dy = y.empty_like()
        dya = _amax_for(dy) if type(self.conv) is nn.Conv3D and self.conv.s == (1, 1, 1) else None
        for i in range(y.n):
            yv, dv, co = _sample(y, i), _sample(dout, i), norm.sample_coeffs(sc, i)
            dev.call("msk_affine_act_bwd_reduce", yv.msk(), _fp(co["scale"]), _fp(co["shift"]), NULL_TENSOR, _fp(alpha),
                     _fp(co["mean"]), _fp(co["invstd"]), dv.msk(), _fp(sc["sums"]))
            dev.call("msk_affine_act_param_grads", Cn, _fp(sc["sums"]), _fp(norm.scale.grad_ptr), _fp(norm.bias.grad_ptr),
                     _fp(self.act._weight.grad_ptr) if self.act is not None else None, 1)
            dev.call("msk_affine_act_bwd_apply_amax", yv.msk(), _fp(co["scale"]), _fp(co["shift"]), NULL_TENSOR, _fp(alpha),
                     _fp(co["mean"]), _fp(co["invstd"]), _fp(norm.scale.ptr), dv.msk(), _fp(sc["sums"]),
                     C.c_double(vox), 1, _sample(dy, i).msk(), NULL_TENSOR, 0, dya)
        self.dy = dy
        # the conv bias gradient is identically zero behind per-sample statistics (the backward removes the mean)
        self.conv.run_backward(self.x, dy, need_dx=need_dx, bias_grad=False)


class DoubleConv(nn.Layer):
    """[conv3^3 (pad 1) -> InstanceNorm -> PReLU] x 2"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3D(cin, cout, kernel_size=3, padding=1)
        self.norm1 = InstanceNorm3D(cout)
        self.relu1 = nn.PReLU(cout)
        self.conv2 = nn.Conv3D(cout, cout, kernel_size=3, padding=1)
        self.norm2 = InstanceNorm3D(cout)
        self.relu2 = nn.PReLU(cout)
        self._u1 = ConvINAct(self.conv1, self.norm1, self.relu1)
        self._u2 = ConvINAct(self.conv2, self.norm2, self.relu2)

    def forward(self, x, out=None):
        return self._u2.forward(self._u1.forward(x), out=out)

    def backward(self, dout, need_dx=True):
        self._u2.backward(dout)
        self._u1.backward(self._u1.out.grad, need_dx=need_dx)


class Down(nn.Layer):
    """conv(k = s = 2) -> InstanceNorm -> PReLU (no pooling kernel exists on the path; a strided conv halves the grid)"""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv3D(c, c, kernel_size=2, stride=2)
        self.norm = InstanceNorm3D(c)
        self.relu = nn.PReLU(c)
        self._u = ConvINAct(self.conv, self.norm, self.relu)

    def forward(self, x):
        return self._u.forward(x)

    def backward(self, dout):
        self._u.backward(dout)


class Up(nn.Layer):
    """convT(k = s = 2, c_hi -> c_lo) -> InstanceNorm -> PReLU, concat with the skip, DoubleConv(2 c_lo -> c_lo)"""

    def __init__(self, c_hi, c_lo):
        super().__init__()
        self.up_conv = nn.Conv3DTranspose(c_hi, c_lo, kernel_size=2, stride=2)
        self.norm = InstanceNorm3D(c_lo)
        self.relu = nn.PReLU(c_lo)
        self.ops = DoubleConv(2 * c_lo, c_lo)
        self.c_lo = c_lo
        self._up = ConvINAct(self.up_conv, self.norm, self.relu)

    def reserve_concat(self, dev, n, d, h, w):
        """Zero-copy skip connection (round 4; the VNet blocks do the same): allocate this block's concat buffer AHEAD of the
        encoder that produces the skip tensor and return the channel slice that encoder writes its output into -- the skip
        half is then never copied (forward) and its gradient never copied back (backward)."""
        xcat = Tensor.empty(dev, n, d, h, w, 2 * self.c_lo)
        _amax_for(xcat)   # one amax array for the buffer: both producers fold into it through their slices
        self._reserved = xcat
        return xcat.channel_slice(self.c_lo, 2 * self.c_lo)

    def forward(self, x, skip):
        od, oh, ow = self.up_conv.out_dims(x)
        if (od, oh, ow) != (skip.d, skip.h, skip.w) or skip.c != self.c_lo:
            raise ValueError(f"skip connection shape {skip.shape} does not match the up-sampled "
                             f"({x.n}, {self.c_lo}, {od}, {oh}, {ow}): every input side must be a multiple of 2^(depth-1)")
        self._x, self._skip = x, skip
        res = getattr(self, "_reserved", None)
        self._reserved = None
        self._in_place = (res is not None and res.gen == x.dev.arena.gen and skip.ld == res.ld and
                          skip.ptr == res.ptr + 4 * self.c_lo and (res.n, res.d, res.h, res.w) == (x.n, od, oh, ow))
        if self._in_place:
            xcat = res
        else:
            xcat = Tensor.empty(x.dev, x.n, od, oh, ow, 2 * self.c_lo)
            _amax_for(xcat)
        self._up.forward(x, out=xcat.channel_slice(0, self.c_lo))
        if not self._in_place:
            copy_scale(skip, None, xcat.channel_slice(self.c_lo, 2 * self.c_lo))
        self._xcat = xcat
        return self.ops.forward(xcat)

    def backward(self, dout):
        self.ops.backward(dout)
        gcat, skip = self._xcat.grad, self._skip
        if self._in_place and skip.grad is None:
            # the skip tensor IS the second half of the concat buffer: so is its gradient (the down path accumulates into the slice)
            skip.grad = gcat.channel_slice(self.c_lo, 2 * self.c_lo)
        else:
            sg = skip.ensure_grad()
            copy_scale(gcat.channel_slice(self.c_lo, 2 * self.c_lo), None, sg, accumulate=skip.grad_written)
        skip.grad_written = True
        self._up.backward(gcat.channel_slice(0, self.c_lo))


@manager.MODELS.add_component
class UNet3D(VNet):
    """UNet3D(in_channels=1, num_classes=3, base_channels=32, depth=4, pretrained=None): `depth` resolution levels
    with base_channels * 2^level channels; returns `[logits]` like every medicalseg model (core/train.py:132)."""

    num_outputs = 1

    def __init__(self, in_channels=1, num_classes=3, base_channels=32, depth=4, pretrained=None, elu=False,
                 precision="fp32"):
        nn.Layer.__init__(self)
        if depth < 2:
            raise ValueError("UNet3D needs depth >= 2")
        if precision not in ("fp32", "fp16"):
            raise ValueError("UNet3D precision must be 'fp32' or 'fp16', got %r" % (precision,))
        self.precision = precision
        if elu:  # the VNet configs this one usually inherits from carry `elu: False`
            raise ValueError("UNet3D has PReLU activations only")
        self.best_loss = 1000000
        self.in_channels, self.num_classes, self.depth = int(in_channels), int(num_classes), int(depth)
        ch = [int(base_channels) * (1 << i) for i in range(depth)]
        self.encoders, self.downs, self.ups = [], [], []
        for i in range(depth):
            enc = DoubleConv(in_channels if i == 0 else ch[i - 1], ch[i])
            setattr(self, f"enc{i}", enc)
            self.encoders.append(enc)
            if i < depth - 1:
                down = Down(ch[i])
                setattr(self, f"down{i}", down)
                self.downs.append(down)
        for i in range(depth - 2, -1, -1):
            up = Up(ch[i + 1], ch[i])
            setattr(self, f"up{i}", up)
            self.ups.append(up)                      # deepest first
        self.head = nn.Conv3D(ch[0], num_classes, kernel_size=1)
        self.pretrained = pretrained
        self._post_backward_hooks = []
        self._grad_ready_hooks = []
        self._build()
        self.init_weight()

    def dropout_layers(self):
        return {}

    def forward(self, x):
        if not isinstance(x, Tensor):
            x = to_tensor(x, self.dev)
        if x.c != self.in_channels:
            raise ValueError(f"UNet3D expects {self.in_channels} input channel(s), got {x.c}")
        self.dev.arena.reset()
        self.dev.set_option("conv_fp16", 1 if self.precision == "fp16" else 0)
        try:
            return self._forward(x)
        finally:
            self.dev.set_option("conv_fp16", 0)

    def _forward(self, x):
        skips, t = [], x
        for i, enc in enumerate(self.encoders):
            # the first encoder's input is the image: it needs no data gradient
            if i < self.depth - 1 and ZERO_COPY_SKIPS:
                # the skip tensor is produced straight into the concat buffer of the decoder block that consumes it
                t = enc.forward(t, out=self.ups[self.depth - 2 - i].reserve_concat(self.dev, t.n, t.d, t.h, t.w))
            else:
                t = enc.forward(t)
            if i < self.depth - 1:
                skips.append(t)
                t = self.downs[i].forward(t)
        self._skips, self._bott = skips, t
        for up, skip in zip(self.ups, reversed(skips)):
            t = up.forward(t, skip)
        self._feat = t
        logits = self.head.run_forward(t)
        logits.producer = self
        return [logits, ]

    def backward(self, dlogits: Tensor):
        self.dev.set_option("conv_fp16", 1 if self.precision == "fp16" else 0)
        try:
            self._backward(dlogits)
        finally:
            self.dev.set_option("conv_fp16", 0)

    def _backward(self, dlogits: Tensor):
        self.head.run_backward(self._feat, dlogits, need_dx=True)
        self._grads_ready(self.head)
        g = self._feat.grad
        for j, up in enumerate(reversed(self.ups)):          # shallowest first
            up.backward(g)
            self._grads_ready(up)
            g = up._x.grad
        # g = gradient of the bottleneck output
        for i in range(self.depth - 1, -1, -1):
            if i < self.depth - 1:
                self.downs[i].backward(g)
                self._grads_ready(self.downs[i])
                g = self._skips[i].grad
            self.encoders[i].backward(g, need_dx=i > 0)
            self._grads_ready(self.encoders[i])
            if i > 0:
                g = self.downs[i - 1]._u.out.grad
        for hook in self._post_backward_hooks:
            hook(self)
