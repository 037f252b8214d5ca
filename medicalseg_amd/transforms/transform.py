"""Loader-side augmentation with the reference's class names, constructor arguments and
RANDOM-NUMBER STREAMS (medicalseg/transforms/transform.py:28-396): under the same
``random.seed`` / ``np.random.seed`` every class draws the same crop boxes, angles, planes and
flip axes as the reference's (pinned by tests/golden/transforms_golden.npz), so a YAML file and
a seed reproduce the reference's augmentation sequence.

Two execution paths share the parameter sampling:
  * host arrays (numpy in, numpy out) -- scipy, like the reference;
  * device volumes (``preprocess.DeviceVolume`` in and out) -- the HIP kernels msk_flip3d,
    msk_rotate3d, msk_crop_resample3d, selected with
    ``Compose(..., device=True)`` / ``device_aug: True`` on a dataset.

``Compose`` ends with the per-volume max normalisation and the channel axis (:64-69).
"""
import collections.abc
import numbers
import random

import numpy as np
import scipy.ndimage

from ..cvlibs import manager


def _on_device(x):
    from ..preprocess import DeviceVolume
    return isinstance(x, DeviceVolume)


def _swap(old, new):
    """Release the input buffer of a device op (stream-ordered pool) and pass the result on."""
    if old is not new:
        old.free()
    return new


class Compose:
    def __init__(self, transforms, device=False):
        if not isinstance(transforms, list):
            raise TypeError('The transforms must be a list!')
        self.transforms = transforms
        self.device = bool(device)

    def __call__(self, im, label=None):
        if isinstance(im, str):
            im = np.load(im)
        if isinstance(label, str):
            label = np.load(label)
        if im is None:
            raise ValueError("Can't read the image file")
        if self.device and not _on_device(im):
            from ..preprocess import upload_pooled
            im = upload_pooled(np.asarray(im, dtype=np.float32))
            if label is not None:
                label = upload_pooled(np.asarray(label).astype(np.int32))
        for op in self.transforms:
            outputs = op(im, label)
            im = outputs[0]
            if len(outputs) == 2:
                label = outputs[1]
        if _on_device(im):
            from ..preprocess import max_normalize_device
            return (max_normalize_device(im), label)  # [D,H,W] on the device == [1,D,H,W] (one channel)
        im = np.expand_dims(im, axis=0)
        if im.max() > 0:
            im = im / im.max()
        return (im, label)


def _zoom_to(img, size, order):
    """functional.py:49-58: ndimage.zoom(img, size/shape, mode='nearest', order)."""
    factors = np.array(size) / np.array(img.shape[:3])
    return scipy.ndimage.zoom(img, factors, mode="nearest", order=order)


def _resize(img, size, order):
    """functional.py:25-58 resize_3d: an int size fixes the SHORTEST side and keeps the aspect."""
    d, h, w = img.shape[:3]
    if isinstance(size, int):
        short = min(d, h, w)
        if short == size:
            return img
        size = (int(size * d / short), int(size * h / short), int(size * w / short))
    if _on_device(img):
        from ..preprocess import resized_crop_device
        return resized_crop_device(img, 0, 0, 0, d, h, w, size, order)
    return _zoom_to(img, size, order)


@manager.TRANSFORMS.add_component
class Resize3D:
    def __init__(self, size, order=1):
        if isinstance(size, int):
            self.size = size
        elif isinstance(size, collections.abc.Iterable) and len(size) == 3:
            self.size = tuple(size)
        else:
            raise ValueError('Unknown inputs for size: {}'.format(size))
        self.order = order

    def __call__(self, img, label=None):
        out = _resize(img, self.size, self.order)
        img = _swap(img, out) if _on_device(img) else out
        if label is not None:
            out = _resize(label, self.size, 0)
            label = _swap(label, out) if _on_device(label) else out
        return img, label


@manager.TRANSFORMS.add_component
class RandomRotation3D:
    """One random in-plane rotation; the reference rotates image AND label with order 1,
    cval 0 (transform.py:162-167 -> functional.py:91 defaults), reproduced here."""

    def __init__(self, degrees, rotate_planes=[[0, 1], [0, 2], [1, 2]]):
        if isinstance(degrees, numbers.Number):
            if degrees < 0:
                raise ValueError("If degrees is a single number, it must be positive.")
            self.degrees = (-degrees, degrees)
        else:
            if len(degrees) != 2:
                raise ValueError("If degrees is a sequence, it must be of len 2.")
            self.degrees = degrees
        self.rotate_planes = rotate_planes

    def get_params(self, degrees):
        angle = random.uniform(degrees[0], degrees[1])
        r_plane = self.rotate_planes[random.randint(0, len(self.rotate_planes) - 1)]
        return angle, r_plane

    @staticmethod
    def _rotate(vol, r_plane, angle):
        if _on_device(vol):
            from ..preprocess import rotate_device
            return _swap(vol, rotate_device(vol, r_plane, angle, order=1, cval=0))
        return scipy.ndimage.rotate(vol, angle=angle, axes=r_plane, order=1, cval=0, reshape=False)

    def __call__(self, img, label=None):
        angle, r_plane = self.get_params(self.degrees)
        img = self._rotate(img, r_plane, angle)
        if label is not None:
            label = self._rotate(label, r_plane, angle)
        return img, label


@manager.TRANSFORMS.add_component
class RandomFlip3D:
    def __init__(self, prob=0.5, flip_axis=[0, 1, 2]):
        self.prob = prob
        self.flip_axis = flip_axis

    @staticmethod
    def _flip(vol, axis):
        if _on_device(vol):
            from ..preprocess import flip_device
            return _swap(vol, flip_device(vol, axis))
        return np.flip(vol, axis)

    def __call__(self, img, label=None):
        # the axis is drawn BEFORE the coin (transform.py:193-199)
        if isinstance(self.flip_axis, (tuple, list)):
            flip_axis = self.flip_axis[random.randint(0, len(self.flip_axis) - 1)]
        else:
            flip_axis = self.flip_axis
        if random.random() < self.prob:
            img = self._flip(img, flip_axis)
            if label is not None:
                label = self._flip(label, flip_axis)
        return img, label


CropBox = collections.namedtuple('CropBox', ['i', 'j', 'k', 'd', 'h', 'w'])


@manager.TRANSFORMS.add_component
class RandomResizedCrop3D:
    """Random crop box (volume ``scale`` x the input, aspect jitter ``ratio``) resized to ``size``;
    image with order ``interpolation``, label with order 0.  ``pre_crop`` first cuts a box of about
    ``size`` (optionally inside the label's non-zero bounding box) -- transform.py:207-339."""

    def __init__(self, size, scale=(0.8, 1.2), ratio=(3. / 4., 4. / 3.), interpolation=1, pre_crop=False,
                 nonzero_mask=False):
        if isinstance(size, (tuple, list)):
            assert len(size) == 3, \
                "Size must contain THREE number when it is a tuple or list, got {}.".format(len(size))
            self.size = size
        elif isinstance(size, int):
            self.size = (size, size, size)
        else:
            raise ValueError("Size must be an int, list or tuple, got {}.".format(type(size)))
        self.interpolation = interpolation
        self.scale = scale
        self.ratio = ratio
        self.pre_crop = pre_crop
        self.nonzero_mask = nonzero_mask

    def get_params(self, img, scale, ratio):
        """Up to ten attempts (transform.py:246-276): target volume and aspect -> (d, h), w = full
        width, an optional shuffle of the three sides, accepted when the box fits; else the
        centred cube of the shortest side."""
        D, H, W = img.shape[0], img.shape[1], img.shape[2]
        for _ in range(10):
            target = random.uniform(*scale) * (D * H * W)
            aspect = random.uniform(*ratio)
            d = int(round((target * aspect) ** (1 / 3)))
            h = int(round((target / aspect) ** (1 / 3)))
            w = W
            if random.random() < 0.5:
                d, h, w = random.sample([d, h, w], k=3)
            if w <= W and h <= H and d <= D:
                i = random.randint(0, D - d)
                j = random.randint(0, H - h)
                k = random.randint(0, W - w)
                return CropBox(i, j, k, d, h, w)
        side = min(D, H, W)
        return CropBox((D - side) // 2, (H - side) // 2, (W - side) // 2, side, side, side)

    def _pre_crop_box(self, img, label):
        """transform.py:288-318: numpy's global RNG draws the box (3 uniforms, then 3 randints)."""
        crop = (np.random.uniform(low=self.scale[0], high=self.scale[1], size=3) * self.size).round().astype("int")
        if self.nonzero_mask:
            lab = label.numpy() if _on_device(label) else label
            nz = np.where(lab != 0)
            lo = np.array([int(np.min(c)) for c in nz])
            hi = np.array([int(np.max(c)) + 1 for c in nz])
        else:
            lo = np.zeros(3, dtype=int)
            hi = np.array(img.shape[:3])
        ext = hi - lo
        cz, cy, cx = np.minimum(ext, crop)
        z0 = np.random.randint(ext[0] - cz + 1) + lo[0]
        y0 = np.random.randint(ext[1] - cy + 1) + lo[1]
        x0 = np.random.randint(ext[2] - cx + 1) + lo[2]
        return int(z0), int(y0), int(x0), int(cz), int(cy), int(cx)

    @staticmethod
    def _crop(vol, box):
        z0, y0, x0, cz, cy, cx = box
        if _on_device(vol):
            from ..preprocess import resized_crop_device
            return _swap(vol, resized_crop_device(vol, z0, y0, x0, cz, cy, cx, (cz, cy, cx), 0))  # identity zoom
        return vol[z0:z0 + cz, y0:y0 + cy, x0:x0 + cx]

    def _resized_crop(self, vol, p, order):
        if _on_device(vol):
            from ..preprocess import resized_crop_device
            return _swap(vol, resized_crop_device(vol, p.i, p.j, p.k, p.d, p.h, p.w, self.size, order))
        return _zoom_to(vol[p.i:p.i + p.d, p.j:p.j + p.h, p.k:p.k + p.w], self.size, order)

    def __call__(self, img, label=None):
        if self.pre_crop:
            box = self._pre_crop_box(img, label)
            img = self._crop(img, box)
            if label is not None:
                label = self._crop(label, box)
        p = self.get_params(img, self.scale, self.ratio)
        img = self._resized_crop(img, p, self.interpolation)
        if label is not None:
            label = self._resized_crop(label, p, 0)
        return img, label


def _connected_components(binary_mask, minimum_volume=0):
    """functional.py:117-131 (SimpleITK ConnectedComponent + RelabelComponent): face-connected
    components relabelled 1, 2, ... by decreasing size, components smaller than
    ``minimum_volume`` dropped.  SimpleITK is absent from this image; scipy.ndimage.label uses
    the same 6-connectivity."""
    vals = np.unique(binary_mask)
    assert len(vals) < 3, "Only binary mask is accepted, got mask with {}.".format(vals.tolist())
    lab, n = scipy.ndimage.label(np.asarray(binary_mask) != 0)
    if n == 0:
        return lab.astype(np.uint32)
    sizes = np.bincount(lab.ravel())[1:]
    order = np.argsort(-sizes, kind="stable")
    lut = np.zeros(n + 1, dtype=np.uint32)
    rank = 1
    for comp in order:
        if sizes[comp] >= minimum_volume:
            lut[comp + 1] = rank
            rank += 1
    return lut[lab]


@manager.TRANSFORMS.add_component
class BinaryMaskToConnectComponent:
    def __init__(self, minimum_volume=0):
        self.minimum_volume = minimum_volume

    def __call__(self, pred, label=None):
        pred = _connected_components(pred, self.minimum_volume)
        if label is not None:
            label = _connected_components(label, self.minimum_volume)
        return pred, label


@manager.TRANSFORMS.add_component
class TopkLargestConnectComponent:
    def __init__(self, k=1):
        self.k = k

    def __call__(self, pred, label=None):
        pred = _connected_components(pred)
        pred[pred > self.k] = 0
        return pred, label
