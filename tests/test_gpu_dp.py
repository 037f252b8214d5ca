"""RCCL plumbing on one GPU (world size 1): unique id, communicator init, all-reduce,
all-gather, broadcast, barrier through the C ABI, and a DataParallel-wrapped training step.
The real multi-GPU run is the driver's; this pins linking/symbol resolution and call order."""
import ctypes as C

import numpy as np
import pytest

from helpers import dev, vec, vec_back, vp

pytestmark = pytest.mark.gpu


def test_rccl_world1_collectives():
    from medicalseg_amd import _lib
    d = dev()
    lib = _lib.load()
    buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    assert lib.msk_dp_unique_id(buf) == 0, _lib.last_error(None)
    d.call("msk_dp_init", buf.raw, 0, 1)
    try:
        x = np.arange(1000, dtype=np.float32)
        p = vec(x)
        d.call("msk_dp_allreduce_sum", vp(p), C.c_size_t(1000))
        assert np.array_equal(vec_back(p, 1000), x)
        d.call("msk_dp_allreduce_stats", vp(p), C.c_size_t(512))      # SyncBN sums: no side-stream join
        assert np.array_equal(vec_back(p, 1000), x)
        q = vec(np.zeros(1000))
        d.call("msk_dp_allgather", vp(p), vp(q), C.c_size_t(1000))
        assert np.array_equal(vec_back(q, 1000), x)
        # gradient buckets: second communicator on the communication stream, then the compute stream waits
        g = vec(x * 3)
        d.call("msk_dp_allreduce_async", vp(g), C.c_size_t(600))
        d.call("msk_dp_allreduce_async", C.c_void_p(g + 4 * 600), C.c_size_t(400))
        d.call("msk_dp_wait")
        assert np.array_equal(vec_back(g, 1000), x * 3)
        d.call("msk_dp_broadcast", vp(p), C.c_size_t(1000), 0)
        d.call("msk_dp_barrier")
        assert np.array_equal(vec_back(p, 1000), x)
    finally:
        d.call("msk_dp_destroy")


def test_dataparallel_wrapper_step_world1():
    """DataParallel hooks (broadcast at wrap, all-reduce after backward, 1/nranks in the
    optimizer) are no-ops at world 1 and must not change the step."""
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd import parallel
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation
    from oracle import vnet_numpy as O
    params = O.init_params(9, 1, 3)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 1, 16, 16, 16)).astype(np.float32)
    y = rng.integers(0, 3, (1, 16, 16, 16)).astype(np.int32)
    outs = []
    for wrap in (False, True):
        model = VNet(num_classes=3)
        model.set_state_dict(params)
        model.eval()
        net = parallel.DataParallel(model) if wrap else model
        opt = optim.Momentum(1e-2, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
        losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
        ll, _ = loss_computation(net(x), to_tensor(y), losses)
        sum(ll).backward()
        opt.step()
        outs.append(model.state_dict())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("model_name", ["VNet", "VNetDeepSup"])
def test_overlapped_gradient_buckets_world1(model_name):
    """The bucketed, overlapped gradient exchange (communication stream + second communicator + weight-gradient side
    stream) on a 1-rank communicator: a sum over one rank is the identity, so the updated parameters must be
    bit-identical to the unwrapped step -- any missing stream dependency shows up as a difference."""
    from medicalseg_amd import _lib, models
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd import parallel
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss
    from medicalseg_amd.utils import loss_computation
    d = dev()
    lib = _lib.load()
    buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    assert lib.msk_dp_unique_id(buf) == 0, _lib.last_error(None)
    d.call("msk_dp_init", buf.raw, 0, 1)
    try:
        nout = 4 if model_name == "VNetDeepSup" else 1
        rng = np.random.default_rng(0)
        x = rng.standard_normal((2, 1, 32, 32, 32)).astype(np.float32)
        y = rng.integers(0, 3, (2, 32, 32, 32)).astype(np.int32)
        outs, init = [], None
        for mode in ("plain", "single", "buckets"):
            model = getattr(models, model_name)(num_classes=3)
            if init is None:
                init = model.state_dict()
            model.set_state_dict(init)
            model.eval()
            net = model if mode == "plain" else parallel.DataParallel(model, force=True, overlap=mode == "buckets",
                                                                      bucket_bytes=4 << 20)
            opt = optim.Momentum(1e-2, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
            losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])] * nout,
                      "coef": [1.0 / nout] * nout}
            for _ in range(2):
                ll, _ = loss_computation(net(x), to_tensor(y), losses)
                sum(ll).backward()
                opt.step()
                model.clear_gradients()
            if mode == "buckets":
                assert len(net.buckets_last_step) >= 3
            outs.append(model.state_dict())
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), k
            assert np.array_equal(outs[0][k], outs[2][k]), k
    finally:
        d.call("msk_dp_destroy")
