"""torch-CPU restatement of the reference VNet (oracle, test-only; CPU ops only).

Two uses: (1) independent cross-check of oracle/vnet_numpy.py (the reference itself was
ported from a torch implementation and aligned against it, vnet.py:1-3,285-294);
(2) the timed `cpu_baseline` of bench.py -- torch's oneDNN convolutions are the strongest
CPU implementation of this step available in the image (the numpy oracle reaches only
~9 GFLOP/s and would flatter the GPU/CPU ratio).  Never imported by the product.

Differences Paddle<->torch handled here: BN momentum
0.9 <-> 0.1, PReLU parameter name, running-var biased (we overwrite the buffer
by hand), SGD == paddle Momentum + L2Decay.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# float64 convolutions have no oneDNN kernel: torch falls back to im2col + GEMM, whose column buffer is
# Cin * k^3 * V_out elements (64 GB for 32 channels at 128^3).  With a byte budget set, the 'same' convolutions
# are evaluated sample by sample in slabs of output planes along D (same arithmetic per output element: every
# output is still one GEMM row over all Cin * k^3 terms), which keeps the full-size float64 oracle runs of
# tests/golden/make_fullsize_golden.py inside a few GB.  None = plain F.conv3d.
SLAB_BYTES = None


def conv_same(conv, x):
    """conv: an nn.Conv3d with stride 1 and 'same' padding p = k // 2 (vnet.py:36,67,165)."""
    if SLAB_BYTES is None:
        return conv(x)
    w, b = conv.weight, conv.bias
    N, C, D, H, W = x.shape
    k = w.shape[2]
    pad = k // 2
    per_plane = C * k ** 3 * H * W * x.element_size()
    d = max(1, min(D, SLAB_BYTES // per_plane))
    if d >= D and N == 1:
        return F.conv3d(x, w, b, padding=pad)
    xp = F.pad(x, (0, 0, 0, 0, pad, pad))
    rows = []
    for n in range(N):
        row = [F.conv3d(xp[n:n + 1, :, z:min(D, z + d) + 2 * pad], w, b, padding=(0, pad, pad)) for z in range(0, D, d)]
        rows.append(torch.cat(row, 2))
    return torch.cat(rows, 0)


class LUConv(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.relu1 = nn.PReLU(c)
        self.conv1 = nn.Conv3d(c, c, 5, padding=2)
        self.bn1 = nn.BatchNorm3d(c, momentum=0.1, eps=1e-5)

    def forward(self, x):
        return self.relu1(self.bn1(conv_same(self.conv1, x)))


class InTr(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.cin = cin
        self.conv1 = nn.Conv3d(cin, 16, 5, padding=2)
        self.bn1 = nn.BatchNorm3d(16)
        self.relu1 = nn.PReLU(16)

    def forward(self, x):
        out = self.bn1(conv_same(self.conv1, x))
        return self.relu1(out + x.repeat(1, 16 // self.cin, 1, 1, 1))


class Down(nn.Module):
    def __init__(self, cin, n, k, s):
        super().__init__()
        co = 2 * cin
        self.down_conv = nn.Conv3d(cin, co, k, stride=s)
        self.bn1 = nn.BatchNorm3d(co)
        self.relu1 = nn.PReLU(co)
        self.relu2 = nn.PReLU(co)
        self.ops = nn.Sequential(*[LUConv(co) for _ in range(n)])

    def forward(self, x, mask=None):
        down = self.relu1(self.bn1(self.down_conv(x)))
        out = down if mask is None else down * mask
        out = self.ops(out)
        return self.relu2(out + down)


class Up(nn.Module):
    def __init__(self, cin, cout, n, k, s):
        super().__init__()
        self.up_conv = nn.ConvTranspose3d(cin, cout // 2, k, stride=s)
        self.bn1 = nn.BatchNorm3d(cout // 2)
        self.relu1 = nn.PReLU(cout // 2)
        self.relu2 = nn.PReLU(cout)
        self.ops = nn.Sequential(*[LUConv(cout) for _ in range(n)])

    def forward(self, x, skip, mx=None, ms=None):
        out = x if mx is None else x * mx
        skip = skip if ms is None else skip * ms
        out = self.relu1(self.bn1(self.up_conv(out)))
        xcat = torch.cat((out, skip), 1)
        out = self.ops(xcat)
        return self.relu2(out + xcat)


class OutTr(nn.Module):
    def __init__(self, cin, ncls):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, ncls, 5, padding=2)
        self.bn1 = nn.BatchNorm3d(ncls)
        self.conv2 = nn.Conv3d(ncls, ncls, 1)
        self.relu1 = nn.PReLU(ncls)

    def forward(self, x):
        return self.conv2(self.relu1(self.bn1(conv_same(self.conv1, x))))


class TorchVNet(nn.Module):
    def __init__(self, in_channels=1, num_classes=4, kernel_size=((2, 2, 2),) * 4,
                 stride_size=((2, 2, 2),) * 4):
        super().__init__()
        K = [tuple(k) for k in kernel_size]
        S = [tuple(s) for s in stride_size]
        self.in_tr = InTr(in_channels)
        self.down_tr32 = Down(16, 1, K[0], S[0])
        self.down_tr64 = Down(32, 2, K[1], S[1])
        self.down_tr128 = Down(64, 3, K[2], S[2])
        self.down_tr256 = Down(128, 2, K[3], S[3])
        self.up_tr256 = Up(256, 256, 2, K[3], S[3])
        self.up_tr128 = Up(256, 128, 2, K[2], S[2])
        self.up_tr64 = Up(128, 64, 1, K[1], S[1])
        self.up_tr32 = Up(64, 32, 1, K[0], S[0])
        self.out_tr = OutTr(32, num_classes)

    def forward(self, x, masks=None):
        m = masks or {}

        def g(k):
            v = m.get(k)
            return None if v is None else torch.as_tensor(v, dtype=x.dtype)[:, :, None, None, None]

        o16 = self.in_tr(x)
        o32 = self.down_tr32(o16)
        o64 = self.down_tr64(o32)
        o128 = self.down_tr128(o64, g("down_tr128"))
        o256 = self.down_tr256(o128, g("down_tr256"))
        out = self.up_tr256(o256, o128, g("up_tr256.x"), g("up_tr256.skip"))
        out = self.up_tr128(out, o64, g("up_tr128.x"), g("up_tr128.skip"))
        out = self.up_tr64(out, o32)
        out = self.up_tr32(out, o16)
        return self.out_tr(out)

    def trunk(self, x, masks=None):
        """(up_tr256, up_tr128, up_tr64, up_tr32) outputs -- shared with TorchVNetDeepSup."""
        m = masks or {}

        def g(k):
            v = m.get(k)
            return None if v is None else torch.as_tensor(v, dtype=x.dtype)[:, :, None, None, None]

        o16 = self.in_tr(x)
        o32 = self.down_tr32(o16)
        o64 = self.down_tr64(o32)
        o128 = self.down_tr128(o64, g("down_tr128"))
        o256 = self.down_tr256(o128, g("down_tr256"))
        u256 = self.up_tr256(o256, o128, g("up_tr256.x"), g("up_tr256.skip"))
        u128 = self.up_tr128(u256, o64, g("up_tr128.x"), g("up_tr128.skip"))
        u64 = self.up_tr64(u128, o32)
        return u256, u128, u64, self.up_tr32(u64, o16)

    def load_oracle_params(self, params):
        sd = {}
        for k, v in params.items():
            k2 = (k.replace("._weight", ".weight").replace("._mean", ".running_mean")
                  .replace("._variance", ".running_var"))
            sd[k2] = torch.as_tensor(np.asarray(v))
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all("num_batches_tracked" in m for m in missing), missing

    def named_oracle_grads(self):
        out = {}
        for k, p in self.named_parameters():
            parts = k.split(".")
            if parts[-2].startswith("relu"):
                k = ".".join(parts[:-1]) + "._weight"
            out[k] = p.grad.detach().numpy().copy()
        return out


class TorchVNetDeepSup(TorchVNet):
    """vnet_deepsup.py:178-281: VNet trunk, out_tr32, three conv3^3 heads resized with
    F.interpolate(mode='trilinear', align_corners=False) (same half-pixel convention as
    paddle's default align_mode=0), and the never-called out_tr_all."""

    def __init__(self, in_channels=1, num_classes=4, kernel_size=((2, 2, 2),) * 4, stride_size=((2, 2, 2),) * 4):
        super().__init__(in_channels, num_classes, kernel_size, stride_size)
        self.out_tr32 = self.out_tr
        del self.out_tr
        self.out_tr64 = nn.Conv3d(64, num_classes, 3, padding=1)
        self.out_tr128 = nn.Conv3d(128, num_classes, 3, padding=1)
        self.out_tr256 = nn.Conv3d(256, num_classes, 3, padding=1)
        self.out_tr_all = OutTr(4 * num_classes, num_classes)

    def forward(self, x, masks=None):
        u256, u128, u64, feat = self.trunk(x, masks)
        size = x.shape[2:]
        r = lambda t: F.interpolate(t, size=size, mode="trilinear", align_corners=False)
        return [self.out_tr32(feat), r(self.out_tr256(u256)), r(self.out_tr128(u128)), r(self.out_tr64(u64))]

    def named_oracle_grads(self):
        out = {}
        for k, p in self.named_parameters():
            if p.grad is None:
                continue
            parts = k.split(".")
            if parts[-2].startswith("relu"):
                k = ".".join(parts[:-1]) + "._weight"
            out[k] = p.grad.detach().numpy().copy()
        return out


def torch_mixed_loss(logits, labels, weight, ignore_index=255):
    """CE(weighted, ignore_index, mean) + sigmoid V-Net dice, as the reference."""
    C = logits.shape[1]
    ce = F.cross_entropy(logits + 1e-8, labels.long(), weight=weight,
                         ignore_index=ignore_index, reduction="mean")
    t = F.one_hot(labels.long(), C).permute(0, 4, 1, 2, 3).to(logits.dtype)
    s = torch.sigmoid(logits)
    sf = s.transpose(0, 1).reshape(C, -1)
    tf = t.transpose(0, 1).reshape(C, -1)
    inter = (sf * tf).sum(-1)
    den = (sf * sf).sum(-1) + (tf * tf).sum(-1)
    per = 2 * inter / den.clamp(min=1e-6)
    return ce, 1.0 - per.mean(), per
