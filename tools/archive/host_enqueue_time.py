import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from medicalseg_amd import models, optimizer as optim
from medicalseg_amd.device import get_device, to_tensor
from medicalseg_amd.utils import loss_computation
dev = get_device()
for B in (1, 2):
    model = models.VNet(num_classes=3); model.train()
    losses = {"types": [models.MixedLoss([models.CrossEntropyLoss(), models.DiceLoss()], [1, 1])], "coef": [1]}
    opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(0)
    x = to_tensor(rng.random((B, 1, 128, 128, 128), dtype=np.float32)); y = to_tensor(rng.integers(0, 3, (B, 128, 128, 128)).astype(np.int32))
    def step():
        ll, _ = loss_computation(model(x), y, losses); sum(ll).backward(); opt.step(); model.clear_gradients()
    for _ in range(3): step()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(5): step()
    t1 = time.perf_counter(); dev.sync(); t2 = time.perf_counter()
    print(f"B={B}: host enqueue {(t1-t0)/5*1e3:.1f} ms/step, total {(t2-t0)/5*1e3:.1f} ms/step")
