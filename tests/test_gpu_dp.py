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


@pytest.mark.parametrize("dp_mode", [0, 1, 2, 3])
def test_syncbn_collectives_inside_a_real_step_world1(dp_mode):
    """TRAIN-mode steps on a 1-rank RCCL communicator with the SyncBatchNorm collectives forced on (24 all-gathers + 24
    all-reduces per step, identities on one rank) next to the overlapped gradient buckets: every stream hand-over of the
    multi-GPU step (compute <-> communication stream, weight-gradient side stream -> buckets) is exercised inside the real
    training flow, and the parameters after two steps must equal the plain single-GPU step BIT FOR BIT -- a missing
    dependency or a collective on the wrong stream shows as a difference (or a hang, caught by the test timeout)."""
    from medicalseg_amd import _lib, models, nn
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd import parallel
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss
    from medicalseg_amd.utils import loss_computation
    d = dev()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 1, 32, 32, 32)).astype(np.float32)
    y = rng.integers(0, 3, (2, 32, 32, 32)).astype(np.int32)
    outs, init = [], None
    for mode in ("plain", "dp"):
        if mode == "dp":
            buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
            assert _lib.load().msk_dp_unique_id(buf) == 0, _lib.last_error(None)
            d.set_option("dp_mode", dp_mode)
            d.call("msk_dp_init", buf.raw, 0, 1)
            nn.BatchNorm3D.force_collectives = True
        try:
            model = models.VNet(num_classes=3)
            if init is None:
                init = model.state_dict()
            model.set_state_dict(init)
            model.train()
            model.set_dropout_masks({})     # (the mask stream is keyed by the step counter and the site ids: not comparable across models)
            net = model if mode == "plain" else parallel.DataParallel(model, force=True, overlap=dp_mode != 0, bucket_bytes=4 << 20)
            opt = optim.Momentum(1e-2, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
            losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
            if mode == "dp":
                d.prof_reset()
                d.prof_enable(True)
            for _ in range(2):
                ll, _ = loss_computation(net(x), to_tensor(y), losses)
                sum(ll).backward()
                opt.step()
                model.clear_gradients()
            outs.append(model.state_dict())
            if mode == "dp":
                d.sync()
                d.prof_enable(False)
                rep = d.prof_report()
                assert rep["rccl_allgather"][0] == 48 and rep["rccl_allreduce_stats"][0] == 48, rep
        finally:
            if mode == "dp":
                nn.BatchNorm3D.force_collectives = False
                d.call("msk_dp_destroy")
                d.set_option("dp_mode", 0)
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("dp_mode", [1, 2])
@pytest.mark.parametrize("model_name", ["VNet", "VNetDeepSup"])
def test_overlapped_gradient_buckets_world1(model_name, dp_mode):
    """The bucketed, overlapped gradient exchange (communication stream + weight-gradient side stream; dp_mode 1: the one
    communicator that also carries the SyncBatchNorm exchanges -- the default --, dp_mode 2: a second communicator) on a
    1-rank communicator: a sum over one rank is the identity, so the updated parameters must be bit-identical to the
    unwrapped step -- any missing stream dependency shows up as a difference.  Statistics-type collectives are issued
    between the buckets (a 1-rank model does not exchange statistics by itself) and must return their input."""
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
    d.set_option("dp_mode", dp_mode)
    d.call("msk_dp_init", buf.raw, 0, 1)
    probe = np.arange(64, dtype=np.float32)
    pa, pb = vec(probe), vec(np.zeros(64, np.float32))

    def stats_probe(model, block):        # runs between the gradient buckets of backward
        d.call("msk_dp_allreduce_stats", C.c_void_p(pa), C.c_size_t(64))
        d.call("msk_dp_allgather", C.c_void_p(pa), C.c_void_p(pb), C.c_size_t(64))
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
            if mode == "buckets":
                model._grad_ready_hooks.append(stats_probe)
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
        assert np.array_equal(d.d2h(pa, (64,), np.float32), probe) and np.array_equal(d.d2h(pb, (64,), np.float32), probe)
    finally:
        d.call("msk_dp_destroy")
        d.set_option("dp_mode", 0)


@pytest.mark.parametrize("C_,shape", [(32, (2, 8, 12, 12)), (256, (2, 4, 4, 4)), (3, (4, 5, 6, 7))])
def test_syncbn_two_rank_arithmetic_on_one_gpu(C_, shape):
    """The N-rank arithmetic of SyncBatchNorm (cvlibs/config.py:322; nn.ConvBNAct with world > 1), executed on ONE GPU:
    the batch is cut in two "rank" halves; each half gets its own msk_bn_stats record, both records go through
    msk_bn_finalize(world = 2) (the cross-rank Chan merge of bn_finalize_k), and the backward sums of the halves are ADDED
    (what msk_dp_allreduce_stats does) before msk_affine_act_bwd_apply runs per half with the global count.  Reference
    values: the float64 oracle on the CONCATENATED batch.  (Round-1 verdict: the world > 1 branches had never run.)"""
    import ctypes as C
    from helpers import rel_err, t_empty, t_from_ncdhw, t_to_ncdhw, vec, vec_back, vp
    from oracle import vnet_numpy as O
    d = dev()
    N, D, H, W = shape
    rng = np.random.default_rng(C_ + N)
    x = (rng.standard_normal((N, C_, D, H, W)) * 2 + 10.0).astype(np.float32)
    x[N // 2:] += 3.0                                       # the two ranks see different means
    gamma, beta = rng.uniform(0.5, 1.5, C_).astype(np.float32), rng.standard_normal(C_).astype(np.float32)
    alpha = rng.uniform(0.1, 0.4, C_).astype(np.float32)
    dout = rng.standard_normal(x.shape).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref, xhat, mean, var, invstd = O.bn_train(f8(x), f8(gamma), f8(beta))
    out_ref = O.prelu(y_ref, f8(alpha))
    du, dalpha = O.prelu_bwd(f8(dout), y_ref, f8(alpha))
    dx_ref, dg_ref, db_ref = O.bn_train_bwd(du, xhat, f8(gamma), invstd)

    halves = [slice(0, N // 2), slice(N // 2, N)]
    xs = [t_from_ncdhw(x[h]) for h in halves]
    ds_ = [t_from_ncdhw(dout[h]) for h in halves]
    Mh = (N // 2) * D * H * W
    gathered = vec(np.zeros(2 * 2 * C_))
    for r, xt in enumerate(xs):
        d.call("msk_bn_stats", xt.msk(), vp(gathered + r * 2 * C_ * 4))
    g, b_ = vec(gamma), vec(beta)
    rm, rv = vec(np.zeros(C_)), vec(np.ones(C_))
    sm, si, sc, sh = (vec(np.zeros(C_)) for _ in range(4))
    d.call("msk_bn_finalize", vp(gathered), 2, C.c_double(Mh), C_, vp(g), vp(b_), C.c_float(1e-5), C.c_float(0.9),
           vp(rm), vp(rv), vp(sm), vp(si), vp(sc), vp(sh))
    assert np.abs(vec_back(sm, C_) - mean).max() < 2e-6 * np.abs(mean).max()
    assert rel_err(vec_back(si, C_), invstd) < 2e-5
    assert rel_err(vec_back(rv, C_), 0.9 * 1.0 + 0.1 * var) < 2e-5
    al = vec(alpha)
    sums = [vec(np.zeros(3 * C_)) for _ in range(2)]
    for r in range(2):
        ot = t_empty(N // 2, C_, D, H, W)
        d.call("msk_affine_act_fwd", xs[r].msk(), vp(sc), vp(sh), d_null(), vp(al), ot.msk())
        assert np.abs(t_to_ncdhw(ot) - out_ref[halves[r]]).max() < 5e-5 * np.abs(out_ref).max()
        d.call("msk_affine_act_bwd_reduce", xs[r].msk(), vp(sc), vp(sh), d_null(), vp(al), vp(sm), vp(si), ds_[r].msk(),
               vp(sums[r]))
    total = vec_back(sums[0], 3 * C_) + vec_back(sums[1], 3 * C_)       # the all-reduce
    assert rel_err(total[:C_], db_ref) < 1e-4 and rel_err(total[C_:2 * C_], dg_ref) < 1e-4
    assert rel_err(total[2 * C_:], dalpha) < 1e-4
    tot = vec(total)
    for r in range(2):
        dxt = t_empty(N // 2, C_, D, H, W)
        d.call("msk_affine_act_bwd_apply", xs[r].msk(), vp(sc), vp(sh), d_null(), vp(al), vp(sm), vp(si), vp(g),
               ds_[r].msk(), vp(tot), C.c_double(2 * Mh), 1, dxt.msk(), d_null(), 0)
        assert rel_err(t_to_ncdhw(dxt), dx_ref[halves[r]]) < 2e-4
    # and the rank-local alternative really differs (the test can tell the two apart)
    loc = vec(np.zeros(2 * C_))
    d.call("msk_bn_stats", xs[0].msk(), vp(loc))
    assert np.abs(vec_back(loc, 2 * C_)[:C_] - mean).max() > 1.0


def d_null():
    from medicalseg_amd._lib import NULL_TENSOR
    return NULL_TENSOR
