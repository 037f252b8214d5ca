"""VNetDeepSup on MI355X -- reference medicalseg/models/vnet_deepsup.py:178-281: the VNet
trunk plus three conv3^3(p=1) heads on the up_tr256 / up_tr128 / up_tr64 outputs, each
resized to the input size with F.interpolate(mode='trilinear'); forward returns
[out, d1, d2, d3] (loss coef 0.25 each,
configs/mri_spine_seg/vnetdeepsup_mri_spine_seg_512_512_12_15k.yml:12-20).

The building blocks are the ones of models/vnet.py (the reference duplicates them verbatim,
vnet_deepsup.py:32-175).  ``out_tr_all`` (:251) is constructed but never called by the
reference's forward: its parameters are part of the state dict, never receive a gradient and
are therefore skipped by the optimizer [PADDLE]; here they live in a separate frozen arena.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import nn
from ..cvlibs import manager
from ..device import Tensor, get_device, to_tensor
from .vnet import DownTransition, InputTransition, OutputTransition, UpTransition, VNet


def interpolate_trilinear(x: Tensor, size) -> Tensor:
    """F.interpolate(x, size=size, mode='trilinear', data_format='NCDHW') (vnet_deepsup.py:268)."""
    d, h, w = (int(v) for v in size)
    out = Tensor.empty(x.dev, x.n, d, h, w, x.c)
    x.dev.call("msk_interp_trilinear_fwd", x.msk(), out.msk())
    return out


def interpolate_trilinear_backward(dout: Tensor, x: Tensor):
    """Adjoint of interpolate_trilinear: writes (or accumulates into) x.grad."""
    dev = x.dev
    g = x.ensure_grad()
    need = C.c_size_t(0)
    dev.call("msk_interp_scratch_bytes", g.msk(), dout.msk(), C.byref(need))
    scratch = dev.arena.alloc(max(int(need.value), 16))
    dev.call("msk_interp_trilinear_bwd", dout.msk(), g.msk(), 1 if x.grad_written else 0, C.c_void_p(scratch),
             C.c_size_t(int(need.value)))
    x.grad_written = True


@manager.MODELS.add_component
class VNetDeepSup(VNet):
    """Same constructor as the reference (vnet_deepsup.py:184-190)."""

    num_outputs = 4

    def __init__(self, elu=False, in_channels=1, num_classes=4, pretrained=None,
                 kernel_size=((2, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2)),
                 stride_size=((2, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2))):
        nn.Layer.__init__(self)
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
        self.out_tr32 = OutputTransition(32, num_classes, elu)
        self.out_tr64 = nn.Conv3D(64, num_classes, kernel_size=3, padding=1)
        self.out_tr128 = nn.Conv3D(128, num_classes, kernel_size=3, padding=1)
        self.out_tr256 = nn.Conv3D(256, num_classes, kernel_size=3, padding=1)
        self.out_tr_all = OutputTransition(4 * num_classes, num_classes, elu)
        for p in self.out_tr_all.parameters():
            p.frozen = True  # never reached by forward (vnet_deepsup.py:257-281)

        self.pretrained = pretrained
        self._post_backward_hooks = []
        self._grad_ready_hooks = []
        self._build()
        self.init_weight()

    def forward(self, x):
        if not isinstance(x, Tensor):
            x = to_tensor(x, self.dev)
        if x.c != self.in_channels:
            raise ValueError(f"VNetDeepSup expects {self.in_channels} input channel(s), got {x.c}")
        self.dev.arena.reset()
        if self.training:
            nn.Dropout3D.step += 1
        size = (x.d, x.h, x.w)
        # the two dropout-free skip connections are produced straight into the concat buffers of their up-transitions
        out16 = self.in_tr(x, out=self.up_tr32.reserve_concat(self.dev, x.n, (x.d, x.h, x.w)))
        out32 = self.down_tr32(out16, out=self.up_tr64.reserve_concat(self.dev, x.n, self.down_tr32.down_conv.out_dims(out16)))
        out64 = self.down_tr64(out32)
        out128 = self.down_tr128(out64)
        out256 = self.down_tr256(out128)
        u256 = self.up_tr256(out256, out128)
        h1 = self.out_tr256.run_forward(u256)
        d1 = interpolate_trilinear(h1, size)
        u128 = self.up_tr128(u256, out64)
        h2 = self.out_tr128.run_forward(u128)
        d2 = interpolate_trilinear(h2, size)
        u64 = self.up_tr64(u128, out32)
        h3 = self.out_tr64.run_forward(u64)
        d3 = interpolate_trilinear(h3, size)
        self._feat = self.up_tr32(u64, out16)
        out = self.out_tr32(self._feat)
        self._acts = (out16, out32, out64, out128, out256)
        self._ups = (u256, u128, u64)
        self._heads = (h1, h2, h3)
        outs = [out, d1, d2, d3]
        for i, t in enumerate(outs):
            t.producer, t.out_index = self, i
        return outs

    def _head_backward(self, conv, feat: Tensor, head: Tensor, dresized):
        if dresized is None:
            return
        interpolate_trilinear_backward(dresized, head)
        conv.run_backward(feat, head.grad, need_dx=True)
        self._grads_ready(conv)

    def backward(self, dlogits):
        """dlogits: the four logit gradients in forward order (None = output unused)."""
        if isinstance(dlogits, Tensor):
            dlogits = [dlogits, None, None, None]
        dout, dd1, dd2, dd3 = dlogits
        u256, u128, u64 = self._ups
        h1, h2, h3 = self._heads
        def run(block, dy):
            block.backward(dy)
            self._grads_ready(block)

        if dout is not None:
            run(self.out_tr32, dout)
            run(self.up_tr32, self._feat.grad)               # -> u64.grad (first writer)
        self._head_backward(self.out_tr64, u64, h3, dd3)     # accumulates into u64.grad
        run(self.up_tr64, u64.grad)
        self._head_backward(self.out_tr128, u128, h2, dd2)
        run(self.up_tr128, u128.grad)
        self._head_backward(self.out_tr256, u256, h1, dd1)
        run(self.up_tr256, u256.grad)
        out16, out32, out64, out128, out256 = self._acts
        run(self.down_tr256, out256.grad)
        run(self.down_tr128, out128.grad)
        run(self.down_tr64, out64.grad)
        run(self.down_tr32, out32.grad)
        run(self.in_tr, out16.grad)
        for hook in self._post_backward_hooks:
            hook(self)

    def test(self):
        np.random.seed(1)
        a = np.random.rand(1, self.in_channels, 32, 32, 32)
        out = self.forward(a.astype("float32"))[0]
        assert out.shape == (1, self.num_classes, 32, 32, 32)
        print("out", out.numpy().mean(), a.mean())
        print("Vnet test is complete")
