"""One rank of the two-process data-parallel test (tests/test_gpu_dp2.py): both processes share GPU 0 and exchange through
the host transport (MSEGK_DP_TRANSPORT=host).  Usage: python tests/dp_worker.py <out.npz> <overlap 0|1> <steps>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make_data(N=4, S=16, ncls=3):
    rng = np.random.default_rng(77)
    x = rng.standard_normal((N, 1, S, S, S)).astype(np.float32)
    y = rng.integers(0, ncls, (N, S, S, S)).astype(np.int32)
    return x, y


def main():
    out, overlap, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    # the host transport lives in the TEST build of the library only (build.sh: libmsegk_test.so)
    from medicalseg_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmsegk_test.so")
    _lib._lib = None            # (importing the package loaded the release library; no context exists yet)
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd import parallel
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation
    from oracle import vnet_numpy as O
    env = parallel.init_parallel_env(dp_mode=2 if overlap else None)    # buckets need an arrangement with a communication stream
    rank, world = env.rank, env.nranks
    params = O.init_params(3, 1, 3)
    if rank != 0:   # DataParallel must broadcast rank 0's parameters: start the other ranks from garbage
        params = {k: v + 1.0 for k, v in params.items()}
    model = VNet(num_classes=3)
    model.set_state_dict(params)
    model.train()
    model.set_dropout_masks({})
    net = parallel.DataParallel(model, overlap=bool(overlap), bucket_bytes=8 << 20)
    x, y = make_data()
    per = x.shape[0] // world
    xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
    opt = optim.Momentum(1e-2, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    vals = []
    for _ in range(steps):
        logits = net(xs)
        ll, _ = loss_computation(logits, to_tensor(ys), losses)
        loss = sum(ll)
        loss.backward()
        opt.step()
        model.clear_gradients()
        vals.append(float(loss))
    sd = model.state_dict()
    np.savez(out, losses=np.array(vals), buckets=np.array(net.buckets_last_step, dtype=np.int64).reshape(-1, 2),
             **{"p:" + k: v for k, v in sd.items()})
    parallel.barrier()


if __name__ == "__main__":
    main()
