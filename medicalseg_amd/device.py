"""Device context, memory pools and tensors over libmsegk (no torch, no numpy compute).

Memory model (sized for 288 GB HBM3E per MI355X):
  * persistent allocations (parameters, optimizer state, BN scratch): plain msk_malloc;
  * the ACTIVATION ARENA: a bump allocator over large blocks that is reset at the start of
    every top-level model forward, so a training step performs no hipMalloc/hipFree after
    the first iteration (activations of VNet 128^3 batch 2 are ~10 GB);
  * the INPUT POOL: a few rotating buffers per (shape, dtype) for images/labels handed to
    the model from the host loader.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import MskError, MskTensor


class Device:
    _current = None

    def __init__(self, index: int = 0):
        self.lib = _lib.load()
        ctx = C.c_void_p()
        rc = self.lib.msk_ctx_create(index, C.byref(ctx))
        if rc != 0:
            raise MskError("msk_ctx_create failed: " + _lib.last_error(None))
        self.ctx = ctx
        self.index = index
        self.arena = ActivationArena(self)
        self.inputs = InputPool(self)
        self.rank = 0
        self.world = 1
        self._small = []

    # -- low level -----------------------------------------------------------------
    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.ctx, *args)
        if rc != 0:
            raise MskError(f"{name} failed: {_lib.last_error(self.ctx)}")

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.call("msk_malloc", C.c_size_t(max(int(nbytes), 16)), C.byref(p))
        return p.value

    def wait_for(self, other: "Device"):
        """this device object's compute stream waits, on the GPU, for everything enqueued so far on `other`'s (a second
        Device() on the same GPU is a second stream: in-loop preprocessing beside the training step)"""
        self.call("msk_ctx_wait", other.ctx)

    def amax_new(self, n=1) -> int:
        """n consecutive zeroed device amax arrays (64 floats each, from the context's ring: valid for the current step)"""
        p = self.lib.msk_amax_new(self.ctx, n)
        if not p:
            raise MskError(f"msk_amax_new failed: {_lib.last_error(self.ctx)}")
        return p

    def free(self, ptr: int):
        self.call("msk_free", C.c_void_p(ptr))

    def memset(self, ptr: int, value: int, nbytes: int):
        self.call("msk_memset", C.c_void_p(ptr), int(value), C.c_size_t(int(nbytes)))

    def h2d(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.call("msk_h2d", C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes))

    def d2h(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.call("msk_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes))
        return out

    def d2d(self, dst: int, src: int, nbytes: int):
        self.call("msk_d2d", C.c_void_p(dst), C.c_void_p(src), C.c_size_t(int(nbytes)))

    def sync(self):
        self.call("msk_sync")

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        self.call("msk_device_name", buf, 256)
        return buf.value.decode()

    def pci_bus_id(self) -> str:
        buf = C.create_string_buffer(64)
        self.call("msk_device_pci_bus_id", buf, 64)
        return buf.value.decode()

    def set_option(self, key: str, value: int):
        self.call("msk_set_option", key.encode(), int(value))

    def get_option(self, key: str) -> int:
        v = C.c_int(0)
        self.call("msk_get_option", key.encode(), C.byref(v))
        return v.value

    # -- timing / profiling ----------------------------------------------------------
    def timer_start(self):
        self.call("msk_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.call("msk_timer_stop", C.byref(ms))
        return ms.value

    def prof_enable(self, on: bool):
        self.call("msk_prof_enable", 1 if on else 0)

    def prof_reset(self):
        self.call("msk_prof_reset")

    def prof_report(self):
        n = C.c_int()
        self.call("msk_prof_report", None, 0, C.byref(n))
        buf = C.create_string_buffer(n.value + 16)
        self.call("msk_prof_report", buf, n.value + 16, C.byref(n))
        out = {}
        for line in buf.value.decode().splitlines():
            tag, calls, ms = line.split("\t")
            out[tag] = (int(calls), float(ms))
        return out

    # -- small persistent float buffers ------------------------------------------------
    def small(self, nfloats: int, zero=True) -> int:
        ptr = self.malloc(nfloats * 4)
        if zero:
            self.memset(ptr, 0, nfloats * 4)
        self._small.append(ptr)
        return ptr


_MASK_VARS = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL")


def local_device_index(local_rank=None, count=None, env=None):
    """LOCAL_RANK -> HIP device index of this process (one process per GPU), checked against the devices that are visible:
      * LOCAL_RANK < count: device LOCAL_RANK;
      * exactly one device visible AND a *_VISIBLE_DEVICES mask is set: the launcher gave every rank its own GPU -> device 0;
      * otherwise MskError naming both numbers (before any communicator is created: a rank that silently shared a GPU with
        another would fail much later inside RCCL, or worse, run at half speed).
    Replaces paddle.set_device / ParallelEnv().device_id (reference core/train.py:69-70)."""
    env = os.environ if env is None else env
    lr = int(env.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    if count is None:
        n = C.c_int(0)
        _lib.load().msk_device_count(C.byref(n))
        count = n.value
    if count <= 0 or lr < count:
        return lr            # count 0: Device() raises "no HIP device" (there is no CPU path)
    if count == 1 and any(env.get(v) for v in _MASK_VARS):
        return 0
    raise MskError("LOCAL_RANK=%d but only %d HIP device(s) are visible (masks: %s): one process per GPU needs "
                   "LOCAL_RANK < device count" % (lr, count, {v: env[v] for v in _MASK_VARS if env.get(v)} or "none"))


def get_device() -> Device:
    """The process-wide device (one process per GPU; LOCAL_RANK selects it, `local_device_index` checks it)."""
    if Device._current is None:
        Device._current = Device(local_device_index())
    return Device._current


class ActivationArena:
    BLOCK = 1 << 30

    def __init__(self, dev: Device):
        self.dev = dev
        self.blocks = []  # [ptr, size]
        self.cur = 0
        self.off = 0
        self.gen = 0
        self.peak = 0

    RING_HALF = 512      # msk_scalar_slots: a handed-out amax array is zeroed again this many requests later

    def reset(self):
        self.cur = 0
        self.off = 0
        self.gen += 1
        # amax arrays taken in forward are read at the end of backward (Tensor.amax -> msk_conv3d_wgrad_ex3): the ring must not
        # wrap inside one arena generation (advisor, round 4) -- a zeroed array would give a wrong fp16 scale, silently
        try:
            served = (self.dev.get_option("scalar_ring_served"), self.dev.get_option("scalar_ring_served_side"))
        except Exception:      # a stand-in library without the keys (tests/fake_msegk.c)
            served = None
        last = getattr(self, "_ring_served", None)
        self._ring_served = served
        if served is not None and last is not None:
            used = max((a - b) & 0x3FFFFFFF for a, b in zip(served, last))
            self.ring_used = used
            if used > self.RING_HALF - 64:
                raise MskError("one step took %d amax arrays from a ring that recycles them after %d requests: arrays held from "
                               "forward to backward would be zeroed under their readers (enlarge kRing in msk_scalar_slots)"
                               % (used, self.RING_HALF))

    def alloc(self, nbytes: int) -> int:
        nbytes = (int(nbytes) + 255) & ~255
        while True:
            if self.cur < len(self.blocks):
                ptr, size = self.blocks[self.cur]
                if self.off + nbytes <= size:
                    p = ptr + self.off
                    self.off += nbytes
                    return p
                self.cur += 1
                self.off = 0
                continue
            size = max(self.BLOCK, nbytes)
            self.blocks.append([self.dev.malloc(size), size])
            self.peak += size


class InputPool:
    DEPTH = 3

    def __init__(self, dev: Device):
        self.dev = dev
        self.slots = {}

    def get(self, key, nbytes: int) -> int:
        ring = self.slots.setdefault(key, [[], 0])
        if len(ring[0]) < self.DEPTH:
            ring[0].append(self.dev.malloc(nbytes))
            return ring[0][-1]
        ring[1] = (ring[1] + 1) % self.DEPTH
        return ring[0][ring[1]]


class Tensor:
    """fp32 device tensor, physical layout NDHWC with voxel stride ``ld``.

    ``shape`` reports the reference's logical NCDHW order so code written against the
    reference (``_, c, d, h, w = images.shape``; core/train.py:266) keeps working."""

    __slots__ = ("dev", "ptr", "n", "d", "h", "w", "c", "ld", "gen", "grad", "grad_written", "grad_from", "producer", "out_index", "_amax", "_amax_gen", "_pool_bytes")

    def __init__(self, dev, ptr, n, d, h, w, c, ld=None, gen=None):
        self.dev, self.ptr = dev, ptr
        self.n, self.d, self.h, self.w, self.c = int(n), int(d), int(h), int(w), int(c)
        self.ld = int(ld if ld is not None else c)
        self.gen = gen
        self.grad = None
        self.grad_written = False
        # a Tensor whose VALUES are this tensor's gradient so far although `grad` itself has not been written (a residual join hands
        # the same gradient to both operands and writes it once, nn.AddAct.backward(share_b=True)): the next accumulating writer reads
        # its old values from there (msk_conv3d_bwd_bnact_acc), anyone else calls nn.materialize_grad first
        self.grad_from = None
        self.producer = None
        self.out_index = 0
        # device "amax array" (msk_amax_new) that the passes WRITING this tensor fold max |value| into, or None.  A channel
        # slice shares its parent's array (the parent's maximum is the maximum over its slices' writers); it is only handed
        # to a consumer by code that knows every channel was written by such a pass (nn.ConvBNAct / AddAct / copy_scale).
        # The array is a slot of a ring that is recycled after ~500 requests (a few steps): it is stamped with the arena
        # generation it was taken in and reads back as None afterwards, so a Tensor object that outlives its step (a
        # persistent buffer, a user-held slice) gets a fresh slot / an absmax pass instead of a recycled one.
        self._amax = None
        self._amax_gen = -1

    # -- construction ------------------------------------------------------------------
    @staticmethod
    def empty(dev, n, d, h, w, c, arena=True):
        nbytes = int(n) * d * h * w * c * 4
        if arena:
            return Tensor(dev, dev.arena.alloc(nbytes), n, d, h, w, c, c, dev.arena.gen)
        return Tensor(dev, dev.malloc(nbytes), n, d, h, w, c, c, None)

    def empty_like(self):
        return Tensor.empty(self.dev, self.n, self.d, self.h, self.w, self.c)

    @property
    def amax(self):
        return self._amax if self._amax_gen == self.dev.arena.gen else None

    @amax.setter
    def amax(self, v):
        self._amax = v
        self._amax_gen = self.dev.arena.gen if v is not None else -1

    # -- views -----------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.n, self.c, self.d, self.h, self.w)

    @property
    def voxels(self):
        return self.n * self.d * self.h * self.w

    def channel_slice(self, c0, c1):
        t = Tensor(self.dev, self.ptr + 4 * c0, self.n, self.d, self.h, self.w, c1 - c0, self.ld, self.gen)
        t.amax = self.amax
        return t

    def check_live(self):
        if self.gen is not None and self.gen != self.dev.arena.gen:
            raise MskError("stale activation tensor: the activation arena was reset by a newer model forward")

    def msk(self) -> MskTensor:
        self.check_live()
        return MskTensor(self.ptr, self.n, self.d, self.h, self.w, self.c, self.ld)

    def ensure_grad(self):
        if self.grad is None:
            self.grad = self.empty_like()
            self.grad_written = False
        return self.grad

    # -- host transfer -----------------------------------------------------------------------
    def numpy(self) -> np.ndarray:
        """NCDHW float32 copy on the host (synchronises)."""
        self.check_live()
        V = self.d * self.h * self.w
        if self.c == 1 and self.ld == 1:
            return self.dev.d2h(self.ptr, (self.n, 1, self.d, self.h, self.w), np.float32)
        tmp = self.dev.malloc(self.n * self.c * V * 4)
        try:
            self.dev.call("msk_ndhwc_to_ncdhw", self.msk(), C.c_void_p(tmp))
            return self.dev.d2h(tmp, (self.n, self.c, self.d, self.h, self.w), np.float32)
        finally:
            self.dev.free(tmp)

    def numpy_ndhwc(self) -> np.ndarray:
        self.check_live()
        if self.ld == self.c:
            return self.dev.d2h(self.ptr, (self.n, self.d, self.h, self.w, self.c), np.float32)
        # strided channel slice: go through the NCDHW path
        return np.ascontiguousarray(np.moveaxis(self.numpy(), 1, -1))

    def astype(self, dtype):
        if str(dtype) in ("float32", "<class 'numpy.float32'>"):
            return self
        raise TypeError("device float tensors are float32 only")

    def __repr__(self):
        return f"Tensor(shape={self.shape}, ld={self.ld}, device=gfx950:{self.dev.index})"


class IntTensor:
    """int32 device array (labels N x D x H x W, argmax predictions)."""

    __slots__ = ("dev", "ptr", "shape_", "gen", "_pool_bytes")

    def __init__(self, dev, ptr, shape, gen=None):
        self.dev, self.ptr, self.shape_, self.gen = dev, ptr, tuple(int(s) for s in shape), gen

    @property
    def shape(self):
        return self.shape_

    def astype(self, dtype):
        if "int" in str(dtype):
            return self
        raise TypeError("label tensors are int32 on the device")

    def numpy(self):
        return self.dev.d2h(self.ptr, self.shape_, np.int32)

    def __len__(self):
        return self.shape_[0]


def to_tensor(arr, dev: Device | None = None):
    """Host array -> device.  float 5-D NCDHW -> Tensor; integer arrays -> IntTensor."""
    if isinstance(arr, (Tensor, IntTensor)):
        return arr
    dev = dev or get_device()
    a = np.asarray(arr)
    if np.issubdtype(a.dtype, np.integer) or a.dtype == np.bool_:
        a = np.ascontiguousarray(a, dtype=np.int32)
        ptr = dev.inputs.get(("i", a.shape), a.nbytes)
        dev.h2d(ptr, a)
        return IntTensor(dev, ptr, a.shape)
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 4:
        a = a[None]
    if a.ndim != 5:
        raise ValueError(f"expected a [N,C,D,H,W] image batch, got shape {a.shape}")
    n, c, d, h, w = a.shape
    ptr = dev.inputs.get(("f", a.shape), a.nbytes)
    t = Tensor(dev, ptr, n, d, h, w, c, c, None)
    if c == 1:
        dev.h2d(ptr, a)  # NCDHW == NDHWC for one channel
    else:
        stage = dev.inputs.get(("fs", a.shape), a.nbytes)
        dev.h2d(stage, a)
        dev.call("msk_ncdhw_to_ndhwc", C.c_void_p(stage), t.msk())
    return t


class LazyArray:
    """Host view of a small device float vector, fetched on first use (one sync)."""

    def __init__(self, dev, ptr, count, scale=1.0):
        self.dev, self.ptr, self.count, self.scale = dev, ptr, count, scale
        self._v = None

    def value(self):
        if self._v is None:
            self._v = self.dev.d2h(self.ptr, (self.count,), np.float32) * np.float32(self.scale)
        return self._v

    def __array__(self, dtype=None, copy=None):
        v = self.value()
        return v.astype(dtype) if dtype is not None else v

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        return self.value()[i]

    def __iadd__(self, other):
        return self.value() + np.asarray(other)

    def __add__(self, other):
        return self.value() + np.asarray(other)

    __radd__ = __add__

    def __truediv__(self, other):
        return self.value() / other

    def __repr__(self):
        return repr(self.value())
