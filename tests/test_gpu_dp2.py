"""Two data-parallel ranks as two PROCESSES on one GPU (host transport, MSEGK_DP_TRANSPORT=host) against the float64
oracle's emulation of the same job: global-batch BatchNorm statistics (SyncBatchNorm, cvlibs/config.py:322), rank-local
losses and class weights, gradients averaged over ranks (core/train.py:81-85), identical parameters everywhere.

What this catches that world-1 tests cannot (round-1 verdict, "What's weak" 10): a wrong 1/nranks, a missing broadcast, a
gradient bucket sent before its producers finished, ranks disagreeing on the order or size of the SyncBatchNorm
exchanges (the transport tags every collective with kind and element count and fails on a mismatch)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(tmp_path, overlap, steps, world=2):
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MSEGK_DP_TRANSPORT="host", MSEGK_DP_OVERLAP="1")
        out = str(tmp_path / ("rank%d_%d.npz" % (r, overlap)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), out, str(overlap), str(steps)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    return [np.load(o) for o in outs]


def _oracle_two_ranks(steps):
    """float64 emulation: ONE forward over the global batch (batch statistics of BatchNorm = SyncBatchNorm), one loss
    object per rank on its half (own first-batch class weights, own dice), dlogits/world, backward, SGD."""
    sys.path.insert(0, HERE)
    from dp_worker import make_data
    from oracle import vnet_numpy as O
    x, y = make_data()
    params = O.init_params(3, 1, 3)
    om = O.VNetOracle(params, 1, 3)
    Ls = [O.MixedLossOracle(), O.MixedLossOracle()]
    vel, rank_losses = {}, []
    for step in range(steps):
        logits = om.forward(x.astype(np.float64), train=True, dropout_masks={})
        dz = np.zeros_like(logits)
        ls = []
        for r in range(2):
            sl = slice(2 * r, 2 * r + 2)
            ll, _, d = Ls[r](logits[sl], y[sl])
            dz[sl] = d / 2.0
            ls.append(float(sum(ll)))
        rank_losses.append(ls)
        grads = om.backward(dz)
        O.sgd_momentum_step(om.p, grads, vel, 1e-2, 0.9, 1e-4, names=om.trainable)
    return om, np.array(rank_losses), params


@pytest.mark.parametrize("overlap", [0, 1])
def test_two_ranks_on_one_gpu_match_the_oracle(tmp_path, overlap):
    steps = 2
    r0, r1 = _run_ranks(tmp_path, overlap, steps)
    keys = [k for k in r0.files if k.startswith("p:")]
    assert len(keys) > 100
    for k in keys:                                     # same parameters and BN buffers on both ranks, bit for bit
        assert np.array_equal(r0[k], r1[k]), k
    if overlap:
        assert len(r0["buckets"]) >= 2                 # the arena really went out in several buckets
        assert 45607944 <= int(r0["buckets"][:, 1].sum()) < 45607944 + 4 * 130   # the arena (16-byte aligned tensors)
    om, ref_losses, p0 = _oracle_two_ranks(steps)
    for r, rr in enumerate((r0, r1)):                  # each rank's own loss trajectory
        assert np.abs(rr["losses"] - ref_losses[:, r]).max() < 2e-4 * np.abs(ref_losses).max(), (rr["losses"], ref_losses)
    # parameter updates: compare the CHANGE of every tensor (two SGD steps) with the oracle's; a wrong 1/nranks is a
    # factor 2, a missed gradient is O(1).  Tolerance: the calibrated whole-net gradient tolerance of test_gpu_model.py
    # (16^3 inputs leave 1-8 voxels per channel in the deepest BatchNorm layers).
    worst = 0.0
    for k in keys:
        name = k[2:]
        if name not in om.p or name.endswith(("._mean", "._variance")):
            continue
        d_got, d_ref = r0[k].astype(np.float64) - p0[name], om.p[name] - p0[name]
        den = np.linalg.norm(d_ref)
        if den < 1e-12:
            continue
        worst = max(worst, np.linalg.norm(d_got - d_ref) / den)
    print("worst relative update error over all tensors: %.2e" % worst)
    assert worst < 5e-2
    for k in keys:                                     # BatchNorm running statistics of the GLOBAL batch
        name = k[2:]
        if name.endswith(("._mean", "._variance")):
            assert np.abs(r0[k] - om.p[name]).max() < 1e-4 * (np.abs(om.p[name]).max() + 1), name


def test_bench_py_two_ranks_on_one_gpu(tmp_path):
    """bench.py's multi-rank path (env contract of `python -m torch.distributed.run`, barrier, max over ranks, the `dp`
    object with per-rank collective times) executed with two processes on GPU 0 over the host transport."""
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MSEGK_DP_TRANSPORT="host")
        # (tests/run_bench_testlib.py: bench.py on the test build of the library -- the release build has no host transport)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "run_bench_testlib.py"), "--gpus", "2",
                                       "--steps", "2", "--warmup", "1", "--size", "32", "--batch", "1", "--no-cpu-baseline",
                                       "--skip-serialized"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e.decode(errors="replace")[-2000:]
        outs.append(o.decode())
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1].splitlines() if l.startswith("{")]     # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and len(j["dp"]["per_rank_step_ms"]) == 2
    assert j["dp"]["calls_per_step"]["rccl_allgather"] == 24 and j["dp"]["calls_per_step"]["rccl_allreduce_stats"] == 24
    # default (--dp-mode auto, more than one rank): gradient buckets on the communication stream, overlapped with backward;
    # a fall-back to mode 0 (second communicator refused) shows as ONE all-reduce of the arena after backward
    calls = j["dp"]["calls_per_step"]
    if j["dp"]["dp_mode"] == 2:
        assert j["dp"]["dp_mode_requested"] == "auto" and j["dp"]["overlap_buckets"]
        # (the host transport of this test runs a bucket through msk_dp_allreduce_sum on the compute stream: tag rccl_allreduce;
        # on RCCL the same buckets are tagged rccl_allreduce_bucket)
        assert calls["rccl_allreduce"] + calls["rccl_allreduce_bucket"] == j["dp"]["buckets_last_step"] >= 2
    else:
        assert j["dp"]["dp_mode"] == 0 and calls["rccl_allreduce"] == 1
    assert j["dp"]["exposed_comm_ms_per_step"] is not None and j["dp"]["compute_only_ms_per_step"] > 0
