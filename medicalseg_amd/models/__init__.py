from .losses import *
from .vnet import VNet
from .vnet_deepsup import VNetDeepSup
from .unet3d import UNet3D
