"""Loader-side augmentation with the reference's class names and arguments
(medicalseg/transforms/transform.py:28-396) so its YAML files build unchanged.  These run
on the host in numpy/scipy exactly like the reference's (they are outside the accelerated
hot path; SURVEY.md section 8 f3 ranks a device version as a follow-up).  ``Compose`` ends with
the per-volume max normalisation and the channel axis (transform.py:67-69)."""
import numbers
import random

import numpy as np
import scipy.ndimage

from ..cvlibs import manager


class Compose:
    def __init__(self, transforms):
        if not isinstance(transforms, list):
            raise TypeError('The transforms must be a list!')
        self.transforms = transforms

    def __call__(self, im, label=None):
        if isinstance(im, str):
            im = np.load(im)
        if isinstance(label, str):
            label = np.load(label)
        if im is None:
            raise ValueError("Can't read the image file")
        for op in self.transforms:
            im, label = op(im, label)
        if np.max(im) > 0:
            im = im / np.max(im)
        im = np.expand_dims(im, axis=0)
        return (im, label)


def _zoom_to(img, size, order):
    factors = [s / float(i) for s, i in zip(size, img.shape)]
    return scipy.ndimage.zoom(img, factors, order=order, mode="nearest")


@manager.TRANSFORMS.add_component
class Resize3D:
    def __init__(self, size, model='constant', order=1):
        self.size = [size] * 3 if isinstance(size, int) else list(size)
        if len(self.size) != 3:
            raise ValueError('`size` should include 3 elements, but it is {}'.format(size))
        self.model, self.order = model, order

    def __call__(self, im, label=None):
        im = _zoom_to(im, self.size, self.order)
        if label is not None:
            label = _zoom_to(label, self.size, 0)
        return im, label


@manager.TRANSFORMS.add_component
class RandomRotation3D:
    def __init__(self, degrees, rotate_planes=[[0, 1], [0, 2], [1, 2]]):
        if isinstance(degrees, numbers.Number):
            if degrees < 0:
                raise ValueError("If degrees is a single number, it must be positive.")
            self.degrees = (-degrees, degrees)
        else:
            if len(degrees) != 2:
                raise ValueError("If degrees is a sequence, it must be of len 2.")
            self.degrees = tuple(degrees)
        self.rotate_planes = rotate_planes

    def __call__(self, im, label=None):
        angle = random.uniform(self.degrees[0], self.degrees[1])
        plane = tuple(self.rotate_planes[random.randint(0, len(self.rotate_planes) - 1)])
        im = scipy.ndimage.rotate(im, angle, axes=plane, reshape=False, order=1, mode="constant", cval=0)
        if label is not None:
            label = scipy.ndimage.rotate(label, angle, axes=plane, reshape=False, order=0, mode="constant", cval=0)
        return im, label


@manager.TRANSFORMS.add_component
class RandomFlip3D:
    def __init__(self, prob=0.5, flip_axis=[0, 1, 2]):
        self.prob = prob
        self.flip_axis = flip_axis

    def __call__(self, im, label=None):
        axes = self.flip_axis if isinstance(self.flip_axis, (list, tuple)) else [self.flip_axis]
        axis = axes[random.randint(0, len(axes) - 1)]
        if random.random() < self.prob:
            im = np.flip(im, axis)
            if label is not None:
                label = np.flip(label, axis)
        return im, label


@manager.TRANSFORMS.add_component
class RandomResizedCrop3D:
    """Random crop of `scale` x volume with aspect jitter `ratio`, resized to `size`."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.), interpolation=1,
                 pre_crop=False, nonzero_mask=False):
        self.size = [size] * 3 if isinstance(size, int) else list(size)
        self.scale, self.ratio, self.interpolation = tuple(scale), tuple(ratio), interpolation
        self.pre_crop, self.nonzero_mask = pre_crop, nonzero_mask

    def __call__(self, im, label=None):
        d, h, w = im.shape
        vol = d * h * w
        cd, ch, cw, i, j, k = d, h, w, 0, 0, 0
        for _ in range(10):
            target = vol * random.uniform(*self.scale)
            ar = np.exp(random.uniform(np.log(self.ratio[0]), np.log(self.ratio[1])))
            side = target ** (1 / 3.)
            td, th, tw = int(round(side)), int(round(side * np.sqrt(ar))), int(round(side / np.sqrt(ar)))
            if 0 < td <= d and 0 < th <= h and 0 < tw <= w:
                cd, ch, cw = td, th, tw
                i, j, k = random.randint(0, d - td), random.randint(0, h - th), random.randint(0, w - tw)
                break
        im = _zoom_to(im[i:i + cd, j:j + ch, k:k + cw], self.size, self.interpolation)
        if label is not None:
            label = _zoom_to(label[i:i + cd, j:j + ch, k:k + cw], self.size, 0)
        return im, label
