"""TEST INFRASTRUCTURE ONLY (never imported by the product): torch-CPU restatement of the BUILDER-DEFINED UNet3D of
medicalseg_amd/models/unet3d.py.  The reference has no UNet3D, so there is nothing of the reference to
follow or to pin against here -- parity unpinned by construction; this file only gives the HIP path an independent
implementation of the same arithmetic (torch.nn.Conv3d / InstanceNorm3d(affine) / PReLU / ConvTranspose3d in float64)."""
import numpy as np
import torch
from torch import nn


class _Double(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1, self.norm1, self.relu1 = nn.Conv3d(cin, cout, 3, padding=1), nn.InstanceNorm3d(cout, affine=True), nn.PReLU(cout)
        self.conv2, self.norm2, self.relu2 = nn.Conv3d(cout, cout, 3, padding=1), nn.InstanceNorm3d(cout, affine=True), nn.PReLU(cout)

    def forward(self, x):
        x = self.relu1(self.norm1(self.conv1(x)))
        return self.relu2(self.norm2(self.conv2(x)))


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv, self.norm, self.relu = nn.Conv3d(c, c, 2, stride=2), nn.InstanceNorm3d(c, affine=True), nn.PReLU(c)

    def forward(self, x):
        return self.relu(self.norm(self.conv(x)))


class _Up(nn.Module):
    def __init__(self, c_hi, c_lo):
        super().__init__()
        self.up_conv, self.norm, self.relu = nn.ConvTranspose3d(c_hi, c_lo, 2, stride=2), nn.InstanceNorm3d(c_lo, affine=True), nn.PReLU(c_lo)
        self.ops = _Double(2 * c_lo, c_lo)

    def forward(self, x, skip):
        return self.ops(torch.cat([self.relu(self.norm(self.up_conv(x))), skip], dim=1))


class TorchUNet3D(nn.Module):
    def __init__(self, in_channels=1, num_classes=3, base_channels=32, depth=4):
        super().__init__()
        self.depth = depth
        ch = [base_channels * (1 << i) for i in range(depth)]
        for i in range(depth):
            setattr(self, f"enc{i}", _Double(in_channels if i == 0 else ch[i - 1], ch[i]))
            if i < depth - 1:
                setattr(self, f"down{i}", _Down(ch[i]))
        for i in range(depth - 2, -1, -1):
            setattr(self, f"up{i}", _Up(ch[i + 1], ch[i]))
        self.head = nn.Conv3d(ch[0], num_classes, 1)

    def forward(self, x):
        skips = []
        for i in range(self.depth):
            x = getattr(self, f"enc{i}")(x)
            if i < self.depth - 1:
                skips.append(x)
                x = getattr(self, f"down{i}")(x)
        for i in range(self.depth - 2, -1, -1):
            x = getattr(self, f"up{i}")(x, skips[i])
        return self.head(x)

    def load_msk_state(self, state):
        """state: medicalseg_amd UNet3D.state_dict() (InstanceNorm `scale`, PReLU `_weight`)."""
        own = dict(self.named_parameters())
        for k, v in state.items():
            tk = k.replace("._weight", ".weight")
            if tk.endswith(".scale"):
                tk = tk[:-6] + ".weight"
            own[tk].data.copy_(torch.as_tensor(np.asarray(v, dtype=np.float64)))
        return self

    def grads_as_msk(self, state_keys):
        own = dict(self.named_parameters())
        out = {}
        for k in state_keys:
            tk = k.replace("._weight", ".weight")
            if tk.endswith(".scale"):
                tk = tk[:-6] + ".weight"
            out[k] = own[tk].grad.detach().numpy()
        return out
