"""Thread sweep of bench.py's cpu_baseline leg (torch-CPU/oneDNN restatement of the VNet training step, oracle/vnet_torch.py,
batch 1, 128^3): how the CPU figure beside the GPU line depends on the thread count of the GPU box's host.
    python tools/cpu_baseline_sweep.py [--threads 16,32,64,128,256] [--out profiles/r06_cpu_baseline_threads.json]
Baseline tooling only: nothing here is on the product path."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(threads, size, max_steps, budget_s):
    import numpy as np
    import torch
    from oracle.vnet_torch import TorchVNet, torch_mixed_loss
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = TorchVNet(1, 3)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.random((1, 1, size, size, size)).astype(np.float32))
    y = torch.tensor(rng.integers(0, 3, (1, size, size, size)).astype(np.int64))
    w = torch.ones(3)

    def step(xx, yy):
        opt.zero_grad()
        ce, dl, _ = torch_mixed_loss(m(xx), yy, w)
        (ce + dl).backward()
        opt.step()

    step(x[:, :, :32, :32, :32].contiguous(), y[:, :32, :32, :32].contiguous())
    t0, n = time.time(), 0
    while n < max_steps and (n == 0 or time.time() - t0 < budget_s):
        step(x, y)
        n += 1
    dt = (time.time() - t0) / n
    return {"threads": threads, "steps": n, "s_per_step": round(dt, 2), "voxels_per_s": round(size ** 3 / dt, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="16,32,64,128,256")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    rows = []
    for t in [int(v) for v in a.threads.split(",")]:
        if t > (os.cpu_count() or 1):
            continue
        # each thread count in a fresh process: oneDNN sizes its primitives and thread pool at first use
        import subprocess
        out = subprocess.run([sys.executable, "-c", "import sys, json; sys.path.insert(0, %r); import tools.cpu_baseline_sweep as s; "
                              "print(json.dumps(s.one(%d, %d, 3, 20.0)))" % (ROOT, t, a.size)], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        rows.append(json.loads(line[-1]) if line else {"threads": t, "error": out.stderr[-300:]})
        print(rows[-1], flush=True)
    rec = {"what": "torch-CPU/oneDNN restatement of the VNet training step (oracle/vnet_torch.py), batch 1, %d^3 fp32" % a.size,
           "nproc": os.cpu_count(), "physical_cores": bench.physical_cores(), "rows": rows}
    try:
        with open("/proc/cpuinfo") as f:
            rec["cpu_model"] = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), None)
    except OSError:
        pass
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
