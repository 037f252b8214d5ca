"""The C-ABI library loads without a GPU and exports exactly what include/msegk.h declares;
the ctypes table mirrors the header one to one.  No compute entry point is called."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "msegk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msk_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    assert len(syms) >= 50
    for must in ("msk_conv3d_fwd", "msk_conv3d_dgrad", "msk_conv3d_wgrad", "msk_convT3d_fwd", "msk_bn_stats",
                 "msk_affine_act_fwd", "msk_loss_fwd", "msk_loss_bwd", "msk_sgd_momentum", "msk_resample3d",
                 "msk_hu_norm", "msk_dp_allreduce_sum", "msk_ctx_create"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from medicalseg_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run ./build.sh (or __graft_entry__.build()) first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), f"libmsegk.so does not export {s}"
    assert lib.msk_version() >= 100


def test_ctypes_table_matches_header():
    from medicalseg_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        assert fn.restype is res and list(fn.argtypes) == list(args)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a visible GPU creating the device must raise (there is no CPU path)."""
    from medicalseg_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int(0)
    lib.msk_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible here")
    ctx = ctypes.c_void_p()
    assert lib.msk_ctx_create(0, ctypes.byref(ctx)) != 0
    assert b"no HIP device" in lib.msk_last_error(None)
    from medicalseg_amd.device import Device
    with pytest.raises(_lib.MskError):
        Device(0)


def test_struct_layouts():
    from medicalseg_amd._lib import MskConvDesc, MskTensor
    assert ctypes.sizeof(MskTensor) == 32 and MskTensor.ld.offset == 28   # void* + 6 x int32
    assert ctypes.sizeof(MskConvDesc) == 36
