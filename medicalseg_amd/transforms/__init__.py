from .transform import Compose, RandomFlip3D, RandomResizedCrop3D, RandomRotation3D, Resize3D
