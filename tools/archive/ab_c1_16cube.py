import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import test_gpu_model as T
from helpers import dev
from oracle import vnet_numpy as O
from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss
from medicalseg_amd.utils import loss_computation
shape, ncls, K, S, N = T.CFGS[0]
for seed in (0, 1, 2):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, 1) + shape).astype(np.float32)
    y = rng.integers(0, ncls, (N,) + shape).astype(np.int32)
    for opt in (1, 0):
        dev().set_option("c1_h2", opt)
        model, params = T._build(ncls, K, S)
        om, lg_ref, ll_ref, per_ref, g_ref = T._oracle_run(params, ncls, K, S, x, y, True, {}, np.float64)
        _, lg32, _, _, g32 = T._oracle_run(params, ncls, K, S, x, y, True, {}, np.float32)
        model.train(); model.set_dropout_masks({})
        logits = model(x)
        losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
        ll, per = loss_computation(logits, T.to_labels(y), losses)
        model.clear_gradients(); sum(ll).backward()
        l2 = [T._l2(p.grad_numpy(), g_ref[n]) for n, p in model.named_parameters() if np.abs(g_ref[n]).max() > 1e-9]
        n2 = [T._l2(g32[n], g_ref[n]) for n in g_ref if np.abs(g_ref[n]).max() > 1e-9]
        print("seed", seed, "c1_h2", opt, "median l2 %.2e worst %.2e | float32-oracle noise median %.2e worst %.2e" % (np.median(l2), max(l2), np.median(n2), max(n2)))
dev().set_option("c1_h2", 1)
