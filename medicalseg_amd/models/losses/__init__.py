from .loss_utils import flatten, class_weights
from .cross_entropy_loss import CrossEntropyLoss
from .dice_loss import DiceLoss
from .mixes_losses import MixedLoss
