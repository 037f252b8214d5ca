from .transform import (BinaryMaskToConnectComponent, Compose, RandomFlip3D, RandomResizedCrop3D, RandomRotation3D,
                        Resize3D, TopkLargestConnectComponent)
