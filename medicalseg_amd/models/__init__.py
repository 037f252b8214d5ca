from .losses import *
from .vnet import VNet
