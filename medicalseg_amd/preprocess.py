"""Device versions of tools/preprocess_utils (reference geometry.py:31-69, values.py:37-87):
same function names, arguments and return values; inputs are host arrays (as the
reference's `Prep.load_save` hands them over, tools/prepare.py:200-259), staged through
pinned-host -> device copies, processed by HIP kernels, and returned as host arrays.
``*_device`` variants keep data on the GPU for in-loop use."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .device import get_device


class DeviceVolume:
    """A 3-D float32/int32 volume on the device (persistent allocation)."""

    def __init__(self, dev, ptr, shape, dtype):
        self.dev, self.ptr, self.shape, self.dtype = dev, ptr, tuple(int(s) for s in shape), np.dtype(dtype)
        self.pooled = False

    @property
    def size(self):
        return int(np.prod(self.shape))

    def numpy(self):
        return self.dev.d2h(self.ptr, self.shape, self.dtype)

    def free(self):
        if self.ptr:
            if self.pooled:
                _pool_release(self.dev, self.ptr, self.size * 4)
            else:
                self.dev.free(self.ptr)
            self.ptr = None


# Stream-ordered recycling of augmentation buffers: every op is enqueued on the context's single
# stream, so a buffer released by one op can be handed to a later op without synchronising
# (hipFree would stall the loader on the whole training stream).
_POOL = {}


def _pool_alloc(dev, nbytes):
    free = _POOL.setdefault(id(dev), {}).setdefault(int(nbytes), [])
    return free.pop() if free else dev.malloc(nbytes)


def _pool_release(dev, ptr, nbytes):
    _POOL.setdefault(id(dev), {}).setdefault(int(nbytes), []).append(ptr)


def _pooled_volume(dev, shape, dtype) -> DeviceVolume:
    shape = tuple(int(v) for v in shape)
    v = DeviceVolume(dev, _pool_alloc(dev, int(np.prod(shape)) * 4), shape, dtype)
    v.pooled = True
    return v


def _dt(vol):
    return 0 if vol.dtype == np.float32 else 1


def flip_device(vol: DeviceVolume, axis: int) -> DeviceVolume:
    """functional.py:80-88 flip_3d on the device."""
    out = _pooled_volume(vol.dev, vol.shape, vol.dtype)
    vol.dev.call("msk_flip3d", C.c_void_p(vol.ptr), C.c_void_p(out.ptr), *vol.shape, int(axis), _dt(vol))
    return out


def rotate_device(vol: DeviceVolume, r_plane, angle, order=1, cval=0) -> DeviceVolume:
    """functional.py:91-100 rotate_3d (scipy.ndimage.rotate, reshape=False) on the device."""
    out = _pooled_volume(vol.dev, vol.shape, vol.dtype)
    vol.dev.call("msk_rotate3d", C.c_void_p(vol.ptr), C.c_void_p(out.ptr), *vol.shape, int(r_plane[0]), int(r_plane[1]),
                 C.c_double(float(angle)), int(order), C.c_double(float(cval)), _dt(vol))
    return out


def resized_crop_device(vol: DeviceVolume, i, j, k, d, h, w, size, order) -> DeviceVolume:
    """functional.py:103-110 resized_crop_3d (crop box, then zoom to `size`) on the device."""
    out = _pooled_volume(vol.dev, size, vol.dtype)
    vol.dev.call("msk_crop_resample3d", C.c_void_p(vol.ptr), *vol.shape, int(i), int(j), int(k), int(d), int(h), int(w),
                 C.c_void_p(out.ptr), *out.shape, int(order), _dt(vol))
    return out


def max_normalize_device(vol: DeviceVolume) -> DeviceVolume:
    """transforms/transform.py:67-69 in place: im / im.max() when the maximum is positive."""
    vol.dev.call("msk_max_norm", C.c_void_p(vol.ptr), C.c_void_p(vol.ptr), C.c_size_t(vol.size))
    return vol


def upload_pooled(image, dev=None) -> DeviceVolume:
    """Host volume -> pooled device buffer (loader path of the device augmentations)."""
    dev = dev or get_device()
    a = np.asarray(image)
    if a.ndim != 3:
        raise ValueError("expected a 3-D volume, got shape {}".format(a.shape))
    a = np.ascontiguousarray(a, dtype=np.int32 if np.issubdtype(a.dtype, np.integer) else np.float32)
    v = _pooled_volume(dev, a.shape, a.dtype)
    dev.h2d(v.ptr, a)
    return v


def upload(image, dev=None) -> DeviceVolume:
    dev = dev or get_device()
    a = np.asarray(image)
    if a.ndim != 3:
        raise ValueError("expected a 3-D volume, got shape {}".format(a.shape))
    if np.issubdtype(a.dtype, np.integer):
        a = np.ascontiguousarray(a, dtype=np.int32)
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
    ptr = dev.malloc(a.nbytes)
    dev.h2d(ptr, a)
    return DeviceVolume(dev, ptr, a.shape, a.dtype)


def resample_device(vol: DeviceVolume, new_shape, order=1, pooled=False) -> DeviceVolume:
    dev = vol.dev
    new_shape = tuple(int(s) for s in new_shape)
    if order not in (0, 1):
        raise ValueError("only order 0 and 1 are built (the orders the reference pipelines use)")
    if pooled:
        res = _pooled_volume(dev, new_shape, vol.dtype)
    else:
        res = DeviceVolume(dev, dev.malloc(int(np.prod(new_shape)) * 4), new_shape, vol.dtype)
    dev.call("msk_resample3d", C.c_void_p(vol.ptr), *vol.shape, C.c_void_p(res.ptr), *new_shape, int(order),
             0 if vol.dtype == np.float32 else 1)
    return res


def resample(image, spacing=None, new_spacing=[1.0, 1.0, 1.0], new_shape=None, order=1):
    """reference geometry.py:31-69 -> (array, new_spacing)."""
    image = np.asarray(image)
    if new_shape is None:
        spacing = np.array([spacing[0], spacing[1], spacing[2]])
        new_shape = np.round(image.shape * spacing / new_spacing)
    else:
        new_shape = np.array(new_shape)
        if spacing is not None and len(spacing) == 4:
            spacing = spacing[1:]
        new_spacing = tuple((image.shape / new_shape) * spacing) if spacing is not None else None
    src_dtype = image.dtype
    vol = upload(image)
    out = resample_device(vol, [int(s) for s in new_shape], order)
    res = out.numpy()
    vol.free()
    out.free()
    if res.dtype != src_dtype and (np.issubdtype(src_dtype, np.integer) or src_dtype == np.float64):
        res = res.astype(src_dtype)
    return res, new_spacing


def HUnorm(image, HU_min=-1200, HU_max=600, HU_nan=-2000):
    """reference values.py:67-87."""
    vol = upload(np.asarray(image, dtype=np.float32))
    vol.dev.call("msk_hu_norm", C.c_void_p(vol.ptr), C.c_void_p(vol.ptr), C.c_size_t(vol.size), C.c_float(HU_min),
                 C.c_float(HU_max), C.c_float(HU_nan))
    out = vol.numpy()
    vol.free()
    return out


def normalize(image, min_val=None, max_val=None):
    """reference values.py:54-64."""
    vol = upload(np.asarray(image, dtype=np.float32))
    use = 0 if (min_val is None and max_val is None) else 1
    vol.dev.call("msk_minmax_norm", C.c_void_p(vol.ptr), C.c_void_p(vol.ptr), C.c_size_t(vol.size), use,
                 C.c_float(min_val or 0.0), C.c_float(max_val or 0.0))
    out = vol.numpy()
    vol.free()
    return out


def max_normalize(image):
    """transforms/transform.py:67-69: im/im.max() if max > 0, plus the channel axis."""
    vol = upload(np.asarray(image, dtype=np.float32))
    vol.dev.call("msk_max_norm", C.c_void_p(vol.ptr), C.c_void_p(vol.ptr), C.c_size_t(vol.size))
    out = vol.numpy()
    vol.free()
    return np.expand_dims(out, axis=0)


def label_remap(label, map_dict=None):
    """reference values.py:37-51 (sequential key -> value passes)."""
    vol = upload(np.asarray(label).astype(np.int32))
    dev = vol.dev
    keys = np.array(list(map_dict.keys()), dtype=np.int32)
    vals = np.array(list(map_dict.values()), dtype=np.int32)
    kp, vp = dev.malloc(max(keys.nbytes, 4)), dev.malloc(max(vals.nbytes, 4))
    if len(keys):
        dev.h2d(kp, keys)
        dev.h2d(vp, vals)
    dev.call("msk_label_remap", C.c_void_p(vol.ptr), C.c_size_t(vol.size), C.c_void_p(kp), C.c_void_p(vp), len(keys))
    out = vol.numpy().astype(np.asarray(label).dtype)
    vol.free()
    dev.free(kp)
    dev.free(vp)
    return out


class DevicePipeline:
    """In-loop preprocessing that never returns to the host: raw volume -> pinned staging buffer ->
    asynchronous H2D copy -> HIP kernels -> model input tensor.

    Mirrors the op lists of the reference's prepare scripts
    (tools/prepare_lung_coronavirus.py:81-90: [HUnorm, resample(128^3, order 1)];
    tools/prepare_mri_spine_seg.py:71-80: [normalize(0, 2650), resample([512,512,12], 1)]) followed by
    Compose's max-normalisation (transforms/transform.py:67-69).  Labels take `resample(order=0)`.

        pipe = DevicePipeline()
        x = pipe.image(raw).HUnorm().resample([128, 128, 128], 1).max_normalize().tensor()
        y = pipe.label(raw_label).resample([128, 128, 128], 0).int_tensor()
    """

    def __init__(self, dev=None, pooled=False):
        """pooled: device buffers come from (and intermediate ones return to) the stream-ordered pool instead of
        hipMalloc / hipFree per op -- hipFree synchronises the whole device, which an in-loop pipeline running BESIDE a
        training step (a second Device() = a second stream, tools/bench_workloads.py --inloop-preprocess) cannot afford;
        the caller hands a finished sample's buffers back with release()."""
        self.dev = dev or get_device()
        self._pinned = [None, 0]
        self.pooled = bool(pooled)

    def release(self, *tensors):
        """return the buffers of finished samples (tensor() / int_tensor() results) to the pool: the NEXT op enqueued on this
        pipeline's stream may overwrite them, so order that stream behind their last reader first (Device.wait_for)"""
        for t in tensors:
            if t is not None and getattr(t, "_pool_bytes", 0):
                _pool_release(self.dev, t.ptr, t._pool_bytes)
                t._pool_bytes = 0

    def _stage(self, a: np.ndarray) -> DeviceVolume:
        dev = self.dev
        a = np.ascontiguousarray(a)
        if self._pinned[1] < a.nbytes:
            if self._pinned[0]:
                dev.sync()
                dev.call("msk_pinned_free", C.c_void_p(self._pinned[0]))
            p = C.c_void_p()
            dev.call("msk_pinned_alloc", C.c_size_t(a.nbytes), C.byref(p))
            self._pinned = [p.value, a.nbytes]
        else:
            dev.sync()  # the previous copy out of the staging buffer must have drained
        C.memmove(self._pinned[0], a.ctypes.data, a.nbytes)
        if self.pooled:
            vol = _pooled_volume(dev, a.shape, a.dtype)
        else:
            vol = DeviceVolume(dev, dev.malloc(a.nbytes), a.shape, a.dtype)
        dev.call("msk_h2d_async", C.c_void_p(vol.ptr), C.c_void_p(self._pinned[0]), C.c_size_t(a.nbytes))
        return vol

    def from_pinned(self, pinned_ptr: int, shape, dtype=np.float32) -> "_Chain":
        """a raw volume that already sits in PINNED host memory (msk_pinned_alloc: a loader that reads into pinned buffers):
        one asynchronous copy, no staging memmove and no host synchronisation -- the caller keeps the buffer unchanged until the
        copy has run"""
        dev = self.dev
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * 4
        if self.pooled:
            vol = _pooled_volume(dev, shape, dtype)
        else:
            vol = DeviceVolume(dev, dev.malloc(nbytes), shape, dtype)
        dev.call("msk_h2d_async", C.c_void_p(vol.ptr), C.c_void_p(pinned_ptr), C.c_size_t(nbytes))
        return _Chain(self, vol)

    def image(self, raw) -> "_Chain":
        return _Chain(self, self._stage(np.asarray(raw, dtype=np.float32)))

    def label(self, raw) -> "_Chain":
        return _Chain(self, self._stage(np.asarray(raw).astype(np.int32)))


class _Chain:
    def __init__(self, pipe, vol):
        self.pipe, self.vol = pipe, vol

    def HUnorm(self, HU_min=-1200, HU_max=600, HU_nan=-2000):
        v = self.vol
        v.dev.call("msk_hu_norm", C.c_void_p(v.ptr), C.c_void_p(v.ptr), C.c_size_t(v.size), C.c_float(HU_min),
                   C.c_float(HU_max), C.c_float(HU_nan))
        return self

    def normalize(self, min_val=None, max_val=None):
        v = self.vol
        use = 0 if (min_val is None and max_val is None) else 1
        v.dev.call("msk_minmax_norm", C.c_void_p(v.ptr), C.c_void_p(v.ptr), C.c_size_t(v.size), use,
                   C.c_float(min_val or 0.0), C.c_float(max_val or 0.0))
        return self

    def resample(self, new_shape, order=1):
        out = resample_device(self.vol, new_shape, order, pooled=self.pipe.pooled)
        old, self.vol = self.vol, out
        # the source is still being read by the enqueued kernel: free() synchronises first
        old.free()
        return self

    def max_normalize(self):
        v = self.vol
        v.dev.call("msk_max_norm", C.c_void_p(v.ptr), C.c_void_p(v.ptr), C.c_size_t(v.size))
        return self

    def tensor(self):
        """[1, 1, D, H, W] model input (one channel: NDHWC and NCDHW coincide); keeps the buffer."""
        from .device import Tensor
        d, h, w = self.vol.shape
        t = Tensor(self.vol.dev, self.vol.ptr, 1, d, h, w, 1, 1, None)
        t._pool_bytes = self.vol.size * 4 if self.vol.pooled else 0      # DevicePipeline.release
        return t

    def int_tensor(self):
        from .device import IntTensor
        t = IntTensor(self.vol.dev, self.vol.ptr, (1,) + tuple(self.vol.shape))
        t._pool_bytes = self.vol.size * 4 if self.vol.pooled else 0
        return t

    def numpy(self):
        return self.vol.numpy()
