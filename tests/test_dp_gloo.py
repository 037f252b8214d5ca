"""Multi-process (world_size 2) checks of the data-parallel path on CPU (`gloo`):
rendezvous protocol, SyncBatchNorm statistics merge, gradient-averaging semantics and
sampler sharding.  Device collectives are RCCL inside libmsegk; here the same host logic and
the same merge arithmetic run against torch.distributed's gloo backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def chan_merge(stats, count):
    """Restatement of bn_finalize_k's cross-rank merge: stats[r] = (mean[C], M2[C])."""
    n, mean, m2 = 0.0, 0.0, 0.0
    for bm, bm2 in stats:
        d = bm - mean
        tot = n + count
        f = count / tot
        m2 = m2 + bm2 + d * d * n * f
        mean = mean + d * f
        n = tot
    return mean, m2 / n


def _worker(rank, world, port, rdzv_port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(rdzv_port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from medicalseg_amd import parallel
    from oracle import vnet_numpy as O
    out = {}
    # 1. unique-id style rendezvous over plain TCP (what init_parallel_env does before msk_dp_init)
    payload = bytes(range(128)) if rank == 0 else None
    got = parallel.exchange_bytes(payload, rank, world, timeout=60)
    out["rdzv"] = got == bytes(range(128))
    env = parallel.ParallelEnv()
    out["env"] = (env.nranks, env.rank, env.local_rank) == (world, rank, rank)

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # 2. SyncBN: all-gather of per-rank (mean, M2), Chan merge == statistics of the global batch
    rng = np.random.default_rng(0)
    full = rng.standard_normal((4, 8, 3, 4, 5)) * 3 + 7                   # global batch N=4
    mine = full[rank * 2:(rank + 1) * 2]
    mean_l = mine.mean(axis=(0, 2, 3, 4))
    m2_l = ((mine - mean_l.reshape(1, -1, 1, 1, 1)) ** 2).sum(axis=(0, 2, 3, 4))
    local = torch.tensor(np.concatenate([mean_l, m2_l]))
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    stats = [(g.numpy()[:8], g.numpy()[8:]) for g in gathered]
    mean_g, var_g = chan_merge(stats, mine.size / 8)
    out["syncbn"] = (np.abs(mean_g - full.mean(axis=(0, 2, 3, 4))).max() < 1e-12 and
                     np.abs(var_g - full.var(axis=(0, 2, 3, 4))).max() < 1e-12)

    # 3. gradient exchange: all-reduce(sum) * 1/world == gradient of the mean of the rank losses
    ncls, shape = 3, (16, 16, 16)
    params = O.init_params(5, 1, ncls)
    xs = rng.standard_normal((2, 1) + shape).astype(np.float32)
    ys = rng.integers(0, ncls, (2,) + shape).astype(np.int32)
    w = np.array([1.0, 2.0, 3.0])

    def grads_of(x, y):
        m = O.VNetOracle(params, 1, ncls)
        lg = m.forward(x, train=False, dropout_masks=None)
        L = O.MixedLossOracle()
        L.weight = w
        _, _, dz = L(lg, y)
        return m.backward(dz)

    g_local = grads_of(xs[rank:rank + 1], ys[rank:rank + 1])
    names = sorted(g_local)
    flat = torch.tensor(np.concatenate([g_local[k].ravel() for k in names]))   # the flat gradient arena
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat *= 1.0 / world                                                       # optimizer's grad_scale
    if rank == 0:
        ga, gb = grads_of(xs[0:1], ys[0:1]), grads_of(xs[1:2], ys[1:2])
        ref = np.concatenate([(0.5 * (ga[k] + gb[k])).ravel() for k in names])
        out["grad_avg"] = float(np.abs(flat.numpy() - ref).max() / np.abs(ref).max()) < 1e-12
    # 4. sampler shards are disjoint across ranks and cover the data
    mine_idx = [i for b in parallel.shard_indices(10, 2, rank, world, True, 0) for i in b]
    t = torch.tensor(mine_idx)
    allidx = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allidx, t)
    out["shards"] = sorted(int(i) for a in allidx for i in a) == list(range(10))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, rdzv = _free_port(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, rdzv, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for k, v in res[r].items():
            assert v, (r, k)
    assert "grad_avg" in res[0]


def test_chan_merge_matches_device_formula_edge_cases():
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((100, 4)) + 1000.0, rng.standard_normal((100, 4)) - 1000.0   # far-apart means
    stats = [(a.mean(0), ((a - a.mean(0)) ** 2).sum(0)), (b.mean(0), ((b - b.mean(0)) ** 2).sum(0))]
    mean, var = chan_merge(stats, 100)
    full = np.concatenate([a, b])
    assert np.allclose(mean, full.mean(0)) and np.allclose(var, full.var(0), rtol=1e-12)
