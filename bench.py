#!/usr/bin/env python
"""Headline benchmark: 3D voxels/s of the VNet training step (fwd + loss + bwd + grad
all-reduce + SGD-momentum update), 128^3 fp32, batch 2 per GPU, synthetic CT volumes
already resident in HBM (BASELINE.json configs[1]; configs[2] at --gpus 8).

    python bench.py --gpus N --steps K --warmup W

N > 1: either under an external launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`:
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* are read from the environment; torch is not imported), or PLAIN `python bench.py --gpus N`:
with WORLD_SIZE unset the process spawns the N ranks itself (medicalseg_amd.parallel.spawn_ranks: one process per GPU, own TCP
rendezvous on 127.0.0.1).  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- the dominant kernel (wbf_gemm_h2_k / wbf_gemm_k: the matrix stage of the 5x5x5 convs and their data gradients),
                  HIP-event time over the timed region; achieved/frac = EXECUTED bf16 FLOPs against the 2.5 PFLOP/s
                  dense bf16 MFMA peak, the algorithmic (direct-convolution) rate under its own keys;
  cpu_baseline -- the CPU oracle timed on the host cores on a bounded sample (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def vnet_lu_layers(d, h, w):
    """(cin, cout, voxels_per_sample, W) of every 5x5x5 LUConv-type layer that runs on the MFMA halo
    kernel, forward AND data gradient (in_tr/out_tr use the VALU halo kernel)."""
    v = d * h * w
    layers = []
    lv = [(v // 8, w // 2), (v // 64, w // 4), (v // 512, w // 8), (v // 4096, w // 16)]
    for c, nconv, (vv, ww) in ((32, 1, lv[0]), (64, 2, lv[1]), (128, 3, lv[2]), (256, 2, lv[3])):
        layers += [(c, c, vv, ww)] * nconv               # down_tr*.ops
    for c, nconv, (vv, ww) in ((256, 2, lv[2]), (128, 2, lv[1]), (64, 1, lv[0]), (32, 1, (v, w))):
        layers += [(c, c, vv, ww)] * nconv               # up_tr*.ops
    return layers


PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), no sparsity
WINO_F45_MAC_RATIO = 0.4        # 1-D Winograd F(4,5): 8 multiplications per 4 outputs instead of 20
# 16-bit products per fp32 product: fp16 two-piece operands (option "conv_split" 2, the product default: hi*hi, hi*lo, lo*hi)
# or exact bf16 three-piece operands (conv_split 3: hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi)
SPLIT_PRODUCTS = {2: 3, 3: 6}
GEMM_TAGS = {2: "wbf_gemm_h2_k", 3: "wbf_gemm_k"}
WGRAD_TAGS = {2: "wbf_wgrad_h2_k", 3: "wbf_wgrad_k"}


def lu_conv_work(n, d, h, w, products=6):
    """Per training step, for the 5^3 LUConv layers: algorithmic FLOPs (SURVEY 8 d3's
    direct-convolution count, 2*125*Cin*Cout per output voxel), the FLOPs the bf16 matrix pipe EXECUTES for them in the
    three-stage Winograd F(4,5) x bf16x3 pipeline (0.4 x 6 = 2.4 bf16 MACs per algorithmic MAC), algorithmic HBM bytes
    (input + output + weights once) and launch counts, for
      'wbf_gemm_k'  : forward + data gradient (msk_conv_wbf.hip), 2 launches per layer,
      'wbf_wgrad_k' : weight gradient (msk_wgrad_wbf.hip), 1 launch per layer."""
    work = {"wbf_gemm_k": [0.0, 0.0, 0, 0.0], "wbf_wgrad_k": [0.0, 0.0, 0, 0.0]}
    for ci, co, vv, ww in vnet_lu_layers(d, h, w):
        f = 2.0 * 125 * ci * co * vv * n
        by = 4.0 * (vv * n * (ci + co) + 125 * ci * co)
        for name, passes in (("wbf_gemm_k", 2), ("wbf_wgrad_k", 1)):
            e = work[name]
            e[0] += passes * f
            e[1] += passes * by
            e[2] += passes
            e[3] += passes * f * WINO_F45_MAC_RATIO * products
    return work


# HIP-event tag (msk_launch_scope) -> bucket of tools/summarize_rocprof.py
TAG_BUCKETS = (("lu_gemm", ("wbf_gemm_",)), ("lu_wgrad", ("wbf_wgrad_",)),
               ("lu_transforms", ("wbf_tin_", "wbf_ty_", "wbf_tout_k", "wbf_pack_", "absmax")),
               ("ks_convs", ("gconv_ks_fwd", "gconv_ks_lds", "convT_scatter_mfma", "gconv_gather_mfma", "wgrad_ks_mfma", "wgrad_ks2_mfma", "wgrad_mfma")),
               ("tiny_channel", ("conv_foldn", "conv_tk_", "conv_halo_tightk", "wgrad_cbs", "conv_c1_", "wgrad_c1_", "wgrad_pw_small",
                                 "pointwise_small", "pack_weights_foldn", "pack_weights_tightk")),
               ("loss_optim", ("loss_", "sgd_momentum", "adam", "class_weights")),
               ("bn_prelu_join", ("affine_act", "add_act", "bn_", "sums_merge", "copy_scale", "dropout_mask", "channel_sum")),
               ("weight_packs_reduces", ("pack_weights", "wgrad_reduce")))


def bucket_times(prof, steps):
    out = {}
    for tag, (calls, ms) in prof.items():
        b = next((name for name, keys in TAG_BUCKETS if any(tag.startswith(k) for k in keys)), "other")
        out[b] = out.get(b, 0.0) + ms / steps
    return out


def _kernel_line(prof, name, flops_step, exec_step, steps, note=None):
    ms = sum(v[1] for k, v in prof.items() if k.startswith(name))
    calls = sum(v[0] for k, v in prof.items() if k.startswith(name))
    if ms <= 0:
        return None
    ex = exec_step * steps / (ms * 1e-3) / 1e12
    out = {"achieved": round(ex, 1), "frac": round(ex / PEAK_BF16_MFMA_TFLOPS, 4),
           "algorithmic_tflops": round(flops_step * steps / (ms * 1e-3) / 1e12, 1),
           "launches": calls, "avg_launch_ms": round(ms / max(calls, 1), 4)}
    if note:
        out["note"] = note
    return out


GEMM_TEMPLATES = ("wbf_gemm_k", "wbf_gemm_fused_k")   # <MR, WM, WN, TD, TH, K, NP, ...>: NP (index 6) = 16-bit pieces per operand
GEMM_NP_INDEX = 6


def template_args(kernel_name):
    """('wbf_gemm_k', ['4', '1', '4', '8', '16', '5', '2', '1']) from 'wbf_gemm_k<4, 1, 4, 8, 16, 5, 2, 1>' (rocprofv3 kernel names)"""
    base, _, rest = kernel_name.partition("<")
    return base.strip(), [a.strip() for a in rest.rsplit(">", 1)[0].split(",")] if rest else []


def dominant_kernel_traffic(tj, split):
    """Mean HBM bytes per launch of the dominant kernel (ALL tile variants of both matrix-stage templates whose operand split --
    template argument GEMM_NP_INDEX, selected by POSITION, not by suffix: trailing parameters come and go -- is `split`) from a
    tools/summarize_rocprof.py traffic file; (bytes_per_launch, launches) or (None, 0)."""
    ent = []
    for k, v in tj.items():
        if not isinstance(v, dict) or "hbm_bytes_per_launch" not in v:
            continue
        base, targs = template_args(k)
        if base in GEMM_TEMPLATES and len(targs) > GEMM_NP_INDEX and targs[GEMM_NP_INDEX] == str(split):
            ent.append(v)
    nl = sum(v["launches"] for v in ent)
    return (int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ent) / nl) if nl else None), nl


def physical_cores():
    """physical cores of the host (unique (package, core) pairs of /proc/cpuinfo); None when it cannot be read"""
    try:
        cores, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("physical id"):
                    phys = l.split(":")[1].strip()
                elif l.startswith("core id"):
                    core = l.split(":")[1].strip()
                elif not l.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        return len(cores) or None
    except OSError:
        return None


def step_flops_per_sample():
    return 4431.2e9  # fwd 1479.9 + bwd 2951.3 GFLOP per 128^3 sample, ncls 3


def cpu_baseline(size=128, ncls=3):
    """CPU baseline (kind "port"): the torch-CPU/oneDNN restatement of the SAME step
    (oracle/vnet_torch.py: VNet forward + CE/Dice loss + backward + SGD-momentum-L2), batch-1
    steps of a size^3 volume (as many as fit in ~12 s, at most 8) after one untimed warm-up step."""
    import torch
    from oracle.vnet_torch import TorchVNet, torch_mixed_loss  # baseline only
    # oneDNN's 3D convolutions stop scaling (and regress) beyond a few dozen threads.  Thread sweep on the GPU box's host
    # (profiles/r06_cpu_baseline_threads.json, tools/cpu_baseline_sweep.py; EPYC 9575F x 2, 128 cores / 256 threads), seconds per
    # step: 16 threads 3.58, 32: 4.01, 64: 6.07, 128: 10.26, 256: 91.9 -> the fastest measured setting, reported in the line
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    torch.manual_seed(0)
    m = TorchVNet(1, ncls)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.random((1, 1, size, size, size)).astype(np.float32))
    y = torch.tensor(rng.integers(0, ncls, (1, size, size, size)).astype(np.int64))
    w = torch.ones(ncls)

    def step(xx, yy):
        opt.zero_grad()
        ce, dl, _ = torch_mixed_loss(m(xx), yy, w)
        (ce + dl).backward()
        opt.step()

    step(x[:, :, :32, :32, :32].contiguous(), y[:, :32, :32, :32].contiguous())  # warm-up (thread pool, primitives)
    t0 = time.time()
    nstep = 0
    while nstep < 8 and (nstep == 0 or time.time() - t0 < 12.0):   # a bounded sample: ~10-30 s of CPU work
        step(x, y)
        nstep += 1
    dt = (time.time() - t0) / nstep
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), None)
    except OSError:
        pass
    # nproc, CPU model and the threads really used, so that box-to-box variance of this line can be read off it
    # cores = physical cores the threads can occupy (threads <= physical cores: one thread per core)
    pc = physical_cores()
    return {"value": float(size ** 3 / dt), "unit": "voxels/s", "cores": min(torch.get_num_threads(), pc or torch.get_num_threads()),
            "threads": torch.get_num_threads(), "physical_cores": pc,
            "nproc": os.cpu_count(), "cpu_model": cpu_model, "gflops": round(step_flops_per_sample() * (size / 128.0) ** 3 / dt / 1e9, 1),
            "kind": "port",
            "sample": "torch-CPU/oneDNN restatement (oracle/vnet_torch.py), %d train step(s), batch 1, %d^3 fp32: "
                      "%.1f s per step" % (nstep, size, dt)}


METRIC = "3D-voxels/sec fwd+bwd, VNet 128^3 fp32"


def fallback_plans(args):
    """What `bench.py --gpus N` tries, in order, when an attempt fails or hangs (launch.run_supervised): the arrangement as
    requested; then every collective on the compute stream's ONE communicator (dp_mode 0: a total order by construction,
    nothing to deadlock); then additionally rank-local BatchNorm statistics (no collective besides the gradient all-reduce --
    a documented deviation from the reference's SyncBatchNorm, reported in config.sync_bn)."""
    plans = [{"label": "as requested (--dp-mode %s%s)" % (args.dp_mode, ", --no-sync-bn" if args.no_sync_bn else ""), "extra": []}]
    if args.dp_mode != "0":
        plans.append({"label": "--dp-mode 0", "extra": ["--dp-mode", "0"]})
    if not args.no_sync_bn:
        plans.append({"label": "--dp-mode 0 --no-sync-bn", "extra": ["--dp-mode", "0", "--no-sync-bn"]})
    return plans


def failure_line(args, error, attempts=None):
    """The ONE JSON line of a job that produced no number: same keys as the success line, value null, the reason in `error`."""
    return {"metric": METRIC, "value": None, "unit": "voxels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "error": str(error)[-2000:],
            "config": {"workload": "VNet %dx%dx%d fp32 batch=%d per GPU, synthetic CT volumes" % (args.size, args.size, args.size, args.batch),
                       "global_batch": args.gpus * args.batch, "parallelism": "dp%d" % args.gpus},
            "dp": {"attempts": attempts or []}}


def supervised_main(args):
    """N > 1: this process does not touch a GPU.  It runs the ranks it owns (all N when no launcher set WORLD_SIZE; its own
    RANK under `python -m torch.distributed.run`) as worker processes under a hang watchdog, walks `fallback_plans` when an
    attempt fails or hangs, and the owner of rank 0 prints ONE JSON line on EVERY outcome: the worker's line + dp.attempts, or
    `failure_line` with value null and the error text."""
    from medicalseg_amd import launch
    world, ranks, env = launch.launch_context(args.gpus)
    is_printer = 0 in ranks
    if world != args.gpus:
        msg = ("bench.py --gpus %d under a launcher that set WORLD_SIZE=%d: the two must agree (or unset WORLD_SIZE and let "
               "bench.py spawn its ranks itself)" % (args.gpus, world))
        if is_printer:
            print(json.dumps(failure_line(args, msg)), flush=True)
        sys.stderr.write(msg + "\n")
        return 2
    os.environ.update({k: env[k] for k in ("MASTER_ADDR", "MASTER_PORT") if k in env})   # the supervisors' own rendezvous
    total = float(os.environ.get("MSEGK_BENCH_TOTAL_S", "1500"))
    launcher = "self (bench.py spawned %d ranks)" % world if len(ranks) == world else \
        "external (WORLD_SIZE=%d from the environment; one supervisor + one worker per rank)" % world
    try:
        ok, text, attempts = launch.run_supervised([sys.executable] + sys.argv, fallback_plans(args), world, ranks, env=env,
                                                   total_timeout=total)
    except BaseException as e:   # incl. the supervisors' own rendezvous timing out: still one line
        if is_printer:
            print(json.dumps(failure_line(args, "launcher: %r" % (e,))), flush=True)
        raise
    if not is_printer:
        return 0 if ok else 1
    rec = None
    if ok and text:
        for l in text.splitlines():
            if l.startswith("{"):
                try:
                    rec = json.loads(l)
                except ValueError:
                    pass
    if rec is None:
        err = "no attempt produced a result" if not ok else "rank 0 finished without printing its JSON line"
        last = attempts[-1] if attempts else {}
        if last.get("error"):
            err += "; last attempt (%s): %s at rank %s, phase %r: %s" % (last.get("plan"), last.get("outcome"), last.get("rank"),
                                                                        last.get("phase"), last.get("error"))
        rec = failure_line(args, err, attempts)
        rec["dp"]["launcher"] = launcher
        print(json.dumps(rec), flush=True)
        return 1
    rec.setdefault("dp", {})["attempts"] = attempts
    rec["dp"]["launcher"] = launcher
    print(json.dumps(rec), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # warm-up 5, time >= 20 steps.  100 since round 6: 20 steps are 0.35 s of GPU time -- shorter than the board's clock /
    # power control settles and than an SMI sampler resolves (round-5 review, weak point 9); 100 steps = 1.8 s
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="samples per GPU")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--num-classes", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shapes", action="store_true", help="tag conv kernels with their problem shapes in the profile")
    ap.add_argument("--profile-out", default=None, help="write the per-kernel HIP-event profile here")
    ap.add_argument("--no-sync-bn", action="store_true",
                    help="rank-local BatchNorm statistics: a documented deviation (the reference uses SyncBatchNorm); "
                         "reported in config.sync_bn, never the default")
    ap.add_argument("--skip-serialized", action="store_true",
                    help="skip the untimed extra pass behind roofline.serialized (used under rocprofv3 so that the kernel "
                         "stats contain only the timed region's launch mode)")
    ap.add_argument("--force-syncbn-collectives", action="store_true",
                    help="diagnostic (1 GPU): create a 1-rank RCCL communicator and run the 48 SyncBatchNorm collectives and the "
                         "gradient buckets of the multi-GPU step on it (identities) -- measures their stream hand-over / launch cost")
    ap.add_argument("--dp-mode", default="auto", choices=("auto", "0", "1", "2", "3"),
                    help="N > 1: stream / communicator arrangement of the collectives (msk_dp.hip): auto (default) = 2 when there is "
                         "more than one rank (falls back to 0 with a logged reason when ncclCommSplit fails); 2 = gradient buckets on a "
                         "second communicator + stream, overlapped with backward; 0 = all on the compute stream, ONE gradient "
                         "all-reduce after backward; 1 / 3 = single-communicator variants")
    ap.add_argument("--eager-opt", type=int, default=None, choices=(0, 1),
                    help="optimizer update + weight re-pack of a block on the weight-gradient stream right behind that block's weight "
                         "gradients (Momentum.enable_eager): default 1 at one rank, 0 otherwise")
    ap.add_argument("--roofline-every", type=int, default=4, metavar="N",
                    help="HIP events ride on the dominant kernel's launches in every Nth timed step (1 = every step: the events' "
                         "packets cost ~0.2 ms of a step, DESIGN.md section 6)")
    ap.add_argument("--skip-strict-fp32", action="store_true",
                    help="skip the untimed extra pass behind roofline.strict_fp32 (the step with exact bf16 x 3 operand pieces)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="msk_set_option knob for experiments, e.g. --opt wgrad_async=0 (not for the headline run)")
    args = ap.parse_args()
    if args.gpus > 1 and os.environ.get("MSEGK_SUPERVISED") != "1":
        raise SystemExit(supervised_main(args))
    worker_main(args)


def worker_main(args):
    from medicalseg_amd import launch
    from medicalseg_amd.utils import logger as _logger
    _logger.stream = sys.stderr       # stdout carries exactly ONE JSON line
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd import parallel
    from medicalseg_amd.datasets import SyntheticCT
    from medicalseg_amd.device import get_device, to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import loss_computation

    env = parallel.ParallelEnv()
    world, rank = env.nranks, env.rank
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d under a launcher that set WORLD_SIZE=%d: the two must agree (or unset WORLD_SIZE and "
                         "let bench.py spawn its ranks itself)" % (args.gpus, world))
    dev = get_device()
    launch.heartbeat("import")
    if world > 1:
        parallel.init_parallel_env(dp_mode=args.dp_mode)
        launch.heartbeat("dp_init")

    S, B, ncls = args.size, args.batch, args.num_classes
    ds = SyntheticCT(num_samples=B, shape=(S, S, S), num_classes=ncls, seed=1234 + 1000 * rank)
    items = [ds[i] for i in range(B)]
    images = to_tensor(np.stack([it[0] for it in items]), dev)   # resident in HBM before timing
    labels = to_tensor(np.stack([it[1] for it in items]), dev)

    from medicalseg_amd import nn
    nn.seed(0)
    model = VNet(elu=False, in_channels=1, num_classes=ncls)
    sched = optim.lr.PolynomialDecay(1e-3, decay_steps=15000, end_lr=0, power=0.9)
    opt = optim.Momentum(sched, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    net = parallel.DataParallel(model) if world > 1 else model
    if args.force_syncbn_collectives and world == 1:
        import ctypes as _C
        from medicalseg_amd import _lib as _L
        buf = _C.create_string_buffer(_L.UNIQUE_ID_BYTES)
        assert _L.load().msk_dp_unique_id(buf) == 0
        dev.call("msk_dp_init", buf.raw, 0, 1)
        nn.BatchNorm3D.force_collectives = True
        net = parallel.DataParallel(model, force=True)
    model.train()
    eager = args.eager_opt if args.eager_opt is not None else (1 if (world == 1 and not args.force_syncbn_collectives) else 0)
    eager = bool(eager) and opt.enable_eager(model)

    def step():
        logits_list = net(images)
        loss_list, per = loss_computation(logits_list, labels, losses)
        loss = sum(loss_list)
        loss.backward()
        opt.step()
        sched.step()
        model.clear_gradients()
        return loss

    if args.no_sync_bn:
        from medicalseg_amd import nn as _nn
        _nn.BatchNorm3D.sync = False
    for kv in args.opt:
        k, v = kv.split("=")
        dev.set_option(k, int(v))
    launch.heartbeat("model")
    for i in range(args.warmup):
        last = step()
        if i == 0 and world > 1:
            dev.sync()     # untimed: the FIRST step with collectives has completed on this rank -- the watchdog's main checkpoint
        launch.heartbeat("warmup %d" % (i + 1))
    parallel.barrier()
    dev.sync()
    launch.heartbeat("warm")
    dev.prof_reset()
    if args.shapes:
        dev.set_option("prof_shapes", 1)
    # HIP events on the launch stream.  Headline run: only around the MFMA halo convolutions (the roofline kernel);
    # bracketing EVERY launch (--profile-out / --shapes) costs ~4 % of the step and is for analysis runs only.
    full_profile = bool(args.profile_out or args.shapes)
    dev.set_option("prof_only_halo", 0 if full_profile else 1)
    dev.prof_enable(True)
    use_marks = args.steps < 1000
    t0 = time.perf_counter()
    if use_marks:
        dev.call("msk_mark", 0)
    # the roofline kernel's events: on every launch of every Nth step (a host-side flag, no drain and no synchronisation in between)
    every = 1 if full_profile else max(1, args.roofline_every)
    sampled_steps = len(range(0, args.steps, every))
    for i in range(args.steps):
        if every > 1:
            dev.set_option("prof_paused", 0 if i % every == 0 else 1)
        last = step()
        if use_marks:
            dev.call("msk_mark", i + 1)     # an event on the stream: no host synchronisation inside the timed region
    if every > 1:
        dev.set_option("prof_paused", 0)
    t_enq = time.perf_counter() - t0   # host time to ENQUEUE the steps (no synchronisation inside a step)
    dev.sync()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    launch.heartbeat("timed")
    dev.prof_enable(False)
    step_ms = []
    if use_marks:
        import ctypes as _C
        for i in range(args.steps):
            ms = _C.c_float()
            dev.call("msk_mark_elapsed", i, i + 1, _C.byref(ms))
            step_ms.append(float(ms.value))
    prof = dev.prof_report()
    loss_val = float(last)
    # untimed extra pass with the weight gradients on the MAIN stream: inside the timed region the data-gradient
    # launches of the dominant kernel share the GPU with the weight-gradient stream, which stretches their
    # HIP-event time; this pass gives the kernel's own duration (reported as roofline.serialized, not as value)
    prof_serial = {}
    if not args.skip_serialized:
        dev.set_option("wgrad_async", 0)
        dev.set_option("prof_only_halo", 0)     # untimed: bracket every launch (the weight-gradient kernel's line comes from here)
        dev.prof_reset()
        dev.prof_enable(True)
        for _ in range(2):
            step()
        dev.sync()
        dev.prof_enable(False)
        prof_serial = dev.prof_report()
        dev.set_option("wgrad_async", 1)
        launch.heartbeat("serialized")

    # untimed extra pass with EXACT fp32 operands (conv_split 3: every fp32 operand as three bf16 pieces, six MFMAs per product,
    # no dropped significand bits) -- the reference's arithmetic is plain fp32 (vnet.py:36, no AMP), the headline uses 22-bit
    # operands (dtype_note); this is the same step at full operand precision, reported beside it, never as `value`
    strict = None
    # (skipped whenever the run itself experiments with the operand format; the option is put back to what it was -- advisor, round 4)
    if world == 1 and not args.skip_strict_fp32 and not any(o.split("=")[0] in ("conv_split", "conv_fp16") for o in args.opt):
        import ctypes as _C
        split_before = dev.get_option("conv_split")
        dev.set_option("conv_split", 3)
        for _ in range(2):
            step()
        dev.sync()
        nst = 5
        dev.call("msk_mark", 0)
        for _ in range(nst):
            step()
        dev.call("msk_mark", 1)
        ms = _C.c_float()
        dev.call("msk_mark_elapsed", 0, 1, _C.byref(ms))
        dev.set_option("conv_split", split_before)
        strict = {"ms_per_step": round(float(ms.value) / nst, 3), "value": round(B * S ** 3 / (float(ms.value) / nst * 1e-3), 1),
                  "unit": "voxels/s", "steps": nst,
                  "note": "untimed extra pass (stream marks): the same step with exact fp32 operands (option conv_split 3: bf16 x 3 "
                          "pieces, 6 MFMAs per fp32 product)"}

    # untimed extra pass in the PLAIN optimizer order (loss.backward(); optimizer.step() as one pass over the arena -- what every rank
    # of an N > 1 job runs, because there the gradients are final only after the all-reduce): the like-for-like base of a scaling curve
    plain = None
    if world == 1 and eager and not args.skip_strict_fp32:
        import ctypes as _C
        opt.enable_eager(model, on=False)
        for _ in range(2):
            step()
        dev.sync()
        nst = 10
        dev.call("msk_mark", 0)
        for _ in range(nst):
            step()
        dev.call("msk_mark", 1)
        ms = _C.c_float()
        dev.call("msk_mark_elapsed", 0, 1, _C.byref(ms))
        opt.enable_eager(model)
        plain = {"ms_per_step": round(float(ms.value) / nst, 3), "value": round(B * S ** 3 / (float(ms.value) / nst * 1e-3), 1), "steps": nst}
        launch.heartbeat("plain_order")

    # untimed extra pass: 150 more steps between two stream marks.  The contract's K = 20 steps are 0.35 s of GPU time -- shorter than
    # the board's clock / power control settles (profiles/r06_fused_probes.txt (e): sclk ~2.0 GHz at ~1.17 kW in steady state); this
    # is the same step sustained for ~2.7 s, reported beside `value`, never as it
    sustained = None
    if world == 1 and not args.skip_strict_fp32 and args.steps < 1000:
        import ctypes as _C
        nst = 150
        dev.call("msk_mark", 0)
        for _ in range(nst):
            step()
        dev.call("msk_mark", 1)
        ms = _C.c_float()
        dev.call("msk_mark_elapsed", 0, 1, _C.byref(ms))
        sustained = {"ms_per_step": round(float(ms.value) / nst, 3), "value": round(B * S ** 3 / (float(ms.value) / nst * 1e-3), 1), "steps": nst}
        launch.heartbeat("sustained")

    # max over ranks; per-rank time inside the RCCL collectives (HIP events on the stream each one is enqueued on: it
    # includes waiting for the slowest peer), so that the first multi-GPU run is diagnosable
    dp_info = None
    if world > 1:
        import ctypes as C
        # compute-only replay (untimed extra pass): the same step with NO collective issued (rank-local statistics, no gradient
        # exchange) -- step minus this is the communication that is NOT hidden, whatever the arrangement
        n_buckets = len(getattr(net, "buckets_last_step", []) or [])
        parallel.set_dry_run(True)
        for _ in range(2):
            step()
        parallel.barrier()
        ndry = max(3, min(10, args.steps))
        dev.call("msk_mark", 0)
        for _ in range(ndry):
            step()
        dev.call("msk_mark", 1)
        msd = C.c_float()
        dev.call("msk_mark_elapsed", 0, 1, C.byref(msd))
        parallel.set_dry_run(False)
        compute_only_ms = float(msd.value) / ndry
        launch.heartbeat("compute_only")
        parallel.barrier()
        tags = ("rccl_allreduce", "rccl_allreduce_stats", "rccl_allgather", "rccl_allreduce_bucket")
        mine = [elapsed] + [sum(v[1] for k, v in prof.items() if k == t) / args.steps for t in tags] + [compute_only_ms, float(dev.index)]
        sp, rp = dev.small(len(mine)), dev.small(world * len(mine))
        dev.h2d(sp, np.array(mine, np.float32))
        dev.call("msk_dp_allgather", C.c_void_p(sp), C.c_void_p(rp), C.c_size_t(len(mine)))
        allv = dev.d2h(rp, (world, len(mine)), np.float32)
        elapsed = max(elapsed, float(allv[:, 0].max()))
        arena_bytes = 4.0 * model.arena.count
        ar_ms = [float(a + b) for a, b in zip(allv[:, 1], allv[:, 4])]    # one piece on the compute stream or buckets on the communication stream
        co_ms = float(allv[:, 5].max())
        bind = parallel.binding_info(env, dev)
        dp_info = {"dp_mode": dev.get_option("dp_mode"), "dp_mode_requested": args.dp_mode,
                   # where it ran: rank -> HIP device of every rank, rank 0's device, the IPC mode and the RCCL version in effect
                   "binding": {"device_index_per_rank": [int(v) for v in allv[:, 6]], "rank0_device": bind["device"],
                               "rank0_pci": bind["pci"], "visible_devices": bind["visible_devices"]},
                   "HSA_ENABLE_IPC_MODE_LEGACY": bind["HSA_ENABLE_IPC_MODE_LEGACY"], "rccl_version": bind["rccl_version"],
                   "overlap_buckets": bool(getattr(net, "overlap", False)),
                   # the step (max over ranks) minus the same step with every collective removed (max over ranks): what the
                   # collectives cost on the critical path in THIS arrangement; budget for >= 6.5x at 8 GPUs: 4.45 ms at 19.3 ms/step
                   "exposed_comm_ms_per_step": round(elapsed / args.steps * 1e3 - co_ms, 3),
                   "compute_only_ms_per_step": round(co_ms, 3),
                   "compute_only_ms_per_step_per_rank": [round(float(v), 3) for v in allv[:, 5]],
                   "gradient_arena_MB": round(arena_bytes / 1e6, 1),
                   "buckets_last_step": n_buckets,
                   # ring all-reduce: every GPU sends and receives 2 (N-1)/N x the buffer -> "bus bandwidth" as nccl-tests define it;
                   # the HIP-event time of a collective includes waiting for the slowest peer to arrive
                   "allreduce_busbw_GBps_per_rank": [round(2.0 * (world - 1) / world * arena_bytes / (t * 1e-3) / 1e9, 1) if t > 0 else None
                                                     for t in ar_ms],
                   "syncbn_collective_ms_per_step_per_rank": [round(float(a + b), 3) for a, b in zip(allv[:, 2], allv[:, 3])],
                   "exposed_comm_budget_note": "north_star >= 6.5x at 8 GPUs holds while (rccl_allreduce, when not overlapped) + "
                                               "syncbn collectives stay under ~4.7 ms per 20 ms step; DESIGN 7's model assumes "
                                               "1.3-2.1 ms for the all-reduce and 20-30 us per statistics collective",
                   "per_rank_step_ms": [round(float(v) / args.steps * 1e3, 3) for v in allv[:, 0]],
                   "per_rank_collective_ms_per_step": {t: [round(float(v), 3) for v in allv[:, 1 + i]] for i, t in enumerate(tags)},
                   "calls_per_step": {t: int(sum(v[0] for k, v in prof.items() if k == t) / args.steps) for t in tags},
                   "how_to_ab": "--dp-mode auto (default) = 2 with more than one rank: gradient buckets on a second communicator + stream, "
                                "overlapped with backward; --dp-mode 0 = one all-reduce after backward, everything on the compute stream; "
                                "--no-sync-bn = rank-local BatchNorm statistics (deviation from the reference)",
                   "note": "rccl_allreduce = gradient arena (182 MB per step) in one piece on the compute stream, rccl_allreduce_bucket = "
                           "its buckets on the communication stream (dp_mode 1-3; overlapped with backward, so their time is NOT "
                           "exposed), rccl_allreduce_stats / rccl_allgather = SyncBatchNorm exchanges (2*C floats each)"}

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    voxels_per_step = world * B * S ** 3
    value = voxels_per_step / (elapsed / args.steps)

    # Dominant kernel: wbf_gemm_k (msk_conv_wbf.hip), the matrix stage of every LUConv forward and data gradient
    # (30 launches per step).  HIP events bracket its launches only (option "prof_only_halo").
    #   achieved = bf16 FLOPs the matrix pipe EXECUTES for it per second  (<= peak: a true fraction of the roof)
    #   algorithmic_tflops = SURVEY 8 d3's direct-convolution FLOPs per second (what the work is worth; the pipeline executes
    #   0.4 x 6 = 2.4 bf16 MACs per algorithmic MAC, so this can exceed the fp32 peak but never 2500 / 2.4)
    split = 2 if any(k.startswith("wbf_gemm_h2_k") for k in list(prof) + list(prof_serial)) else 3
    work = lu_conv_work(B, S, S, S, SPLIT_PRODUCTS[split])
    DOM = GEMM_TAGS[split]
    flops_step, bytes_step, launches_step, exec_step = work["wbf_gemm_k"]
    line = _kernel_line(prof, DOM, flops_step, exec_step, sampled_steps) or {"achieved": 0.0, "frac": 0.0,
                                                                          "algorithmic_tflops": 0.0, "launches": 0,
                                                                          "avg_launch_ms": 0.0}
    kms = line["avg_launch_ms"] * line["launches"] * args.steps / max(sampled_steps, 1)   # scaled from the sampled steps to all of them
    total_kernel_ms = sum(v[1] for v in prof.values())
    traffic, traffic_src, hbm = None, None, None
    tpath = next((pth for pth in (os.path.join(ROOT, "profiles", "r%02d_hbm_traffic.json" % r) for r in range(6, 1, -1)) if os.path.exists(pth)), None)
    if tpath and S == 128 and B == 2:   # PMC passes of this exact workload (tools/profile_gpu.sh, tools/summarize_rocprof.py)
        try:
            tj = json.load(open(tpath))
            rel = os.path.relpath(tpath, ROOT)
            traffic, _ = dominant_kernel_traffic(tj, split)
            traffic_src = "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, captured %s, NOT measured in " \
                          "this run" % (rel, tj.get("_captured", "earlier"))
            ws = tj.get("_whole_step")
            if ws:
                # 15.62 GB per sample of which the 0.91 GB optimizer term is per step
                alg = (15.62 - 0.91) * B * (S / 128.0) ** 3 + 0.91
                hbm = {"algorithmic_GB_per_step": round(alg, 2), "counter_GB_per_step": round(ws["hbm_bytes_per_step"] / 1e9, 2),
                       "ratio": round(ws["hbm_bytes_per_step"] / 1e9 / alg, 2), "source": traffic_src,
                       "buckets_counter_GB_per_step": {k: round(v / 1e9, 2) for k, v in ws["buckets_bytes_per_step"].items()}}
        except Exception:
            traffic = None
    if hbm is not None and prof_serial:
        # achieved HBM rate per bucket: counter bytes of the PMC passes / HIP-event time of the bucket's kernels in the untimed
        # serialized pass of THIS run (kernels do not overlap there)
        bt = bucket_times(prof_serial, 2)
        hbm["buckets"] = {b: {"counter_GB": g, "ms": round(bt.get(b, 0.0), 3),
                              "TB_per_s": round(g / bt[b], 2) if bt.get(b, 0.0) > 0 else None}
                          for b, g in hbm.pop("buckets_counter_GB_per_step").items()}
        hbm["serialized_kernel_ms_per_step"] = round(sum(bt.values()), 3)
    lu_flops = sum(v[0] for v in work.values())            # algorithmic FLOPs of the LUConv layers, fwd + dgrad + wgrad
    lu_exec = sum(v[3] for v in work.values())
    total_flops = step_flops_per_sample() * B * (S / 128.0) ** 3
    # time the executed work needs at the peaks: LUConv layers on the bf16 pipe, everything else at the fp32 peak
    t_floor = lu_exec / (PEAK_BF16_MFMA_TFLOPS * 1e12) + max(total_flops - lu_flops, 0.0) / (PEAK_FP32_MFMA_TFLOPS * 1e12)
    roofline = {"bound": "mfma", "kernel": DOM, "achieved": line["achieved"], "peak": PEAK_BF16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": line["frac"],
                # the same launches priced on the ALGORITHMIC (direct-convolution) FLOPs against the same 16-bit peak: the other convention
                "frac_algorithmic": round(line["algorithmic_tflops"] / PEAK_BF16_MFMA_TFLOPS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "launches": line["launches"], "avg_launch_ms": line["avg_launch_ms"],
                "event_steps": sampled_steps,
                "events": "HIP events riding on every launch of this kernel (hipExtLaunchKernelGGL start / stop events, on the launch "
                          "stream) in %d of the %d timed steps (every %s)" % (sampled_steps, args.steps, every),
                "executed_16bit_flop_per_launch": round(exec_step / max(launches_step, 1), 1),
                "algorithmic_flop_per_launch": round(flops_step / max(launches_step, 1), 1),
                "algorithmic_tflops": line["algorithmic_tflops"],
                "algorithmic_speedup_vs_fp32_mfma_peak": round(line["algorithmic_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                "algorithmic_bytes_per_launch": round(bytes_step / max(launches_step, 1), 1),
                "kernel_share_of_step": round(kms / max(elapsed * 1e3, 1e-9), 4),  # of wall time (streams overlap)
                "serialized": _kernel_line(prof_serial, DOM, flops_step, exec_step, 2,
                                           "same launches, weight-gradient stream disabled (untimed extra pass of 2 steps)"),
                "operand_split": "fp16 x 2 pieces, 3 MFMAs per fp32 product" if split == 2 else "bf16 x 3 pieces, 6 MFMAs per fp32 product",
                "wgrad_kernel": _kernel_line(prof_serial, WGRAD_TAGS[split], work["wbf_wgrad_k"][0], work["wbf_wgrad_k"][3], 2,
                                             WGRAD_TAGS[split] + " in the same untimed pass (it runs on the side stream in the timed region)"),
                # fraction of the step the EXECUTED work would take at the hardware peaks (bf16 pipe for the LUConv
                # layers, fp32 MFMA peak for the remaining convolutions); <= 1
                "strict_fp32": strict,
                "hbm": hbm,
                "step_executed_frac": round(t_floor / (ms_per_step * 1e-3), 4),
                "step_algorithmic_speedup_vs_fp32_roofline": round(total_flops / (ms_per_step * 1e-3) / 1e12
                                                                   / PEAK_FP32_MFMA_TFLOPS, 4)}
    out = {"metric": "3D-voxels/sec fwd+bwd, VNet 128^3 fp32", "value": round(value, 1), "unit": "voxels/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "dtype_note": "fp32 tensors end to end; the LUConv matrix products run on the 16-bit matrix pipe with fp32 operands split "
                         "into 16-bit pieces and fp32 accumulation (roofline.operand_split; fp32-class error against the float64 "
                         "oracle, tests/test_gpu_wbf.py)",
           "config": {"workload": "VNet %dx%dx%d fp32 batch=%d per GPU, synthetic CT volumes (BASELINE configs[%d])"
                      % (S, S, S, B, 1 if world == 1 else 2),
                      "global_batch": world * B, "num_classes": ncls, "parallelism": "dp%d" % world,
                      "step": "fwd+loss+bwd+allreduce+sgd_momentum", "sync_bn": not args.no_sync_bn,
                      # the optimizer update of a block runs on the weight-gradient stream right behind that block's weight
                      # gradients (same arithmetic, same results; optimizer.Momentum.enable_eager)
                      "eager_optimizer": bool(eager)},
           # per-step times from stream marks (no synchronisation inside the timed region): SURVEY 8 d1 asks for the median;
           # `value` stays the contract's K-steps-between-two-synchronisations figure
           "ms_per_step_median": round(float(np.median(step_ms)), 3) if step_ms else None,
           "value_median": round(voxels_per_step / (float(np.median(step_ms)) * 1e-3), 1) if step_ms else None,
           "ms_per_step_min_max": [round(min(step_ms), 3), round(max(step_ms), 3)] if step_ms else None,
           # the steps that carry the roofline kernel's events against the ones that do not (the events' packets idle the packet processor)
           "ms_per_step_with_events": round(float(np.mean(step_ms[0::every])), 3) if step_ms and every > 1 else None,
           "ms_per_step_without_events": round(float(np.mean([m for i, m in enumerate(step_ms) if i % every])), 3) if step_ms and every > 1 else None,
           # N = 1 runs the eager optimizer (config.eager_optimizer); N > 1 cannot: this is the same step in the plain order,
           # untimed extra pass of 10 steps between stream marks -- divide N > 1 values by THIS for a like-for-like scaling base
           "value_plain_order": plain["value"] if plain else None,
           "ms_per_step_plain_order": plain["ms_per_step"] if plain else None,
           # the same step sustained over 150 further steps (untimed extra pass between two stream marks): clocks and power settled
           "ms_per_step_sustained": sustained["ms_per_step"] if sustained else None,
           "value_sustained": sustained["value"] if sustained else None,
           "final_loss": round(loss_val, 6),
           # host time to enqueue one step (python + ctypes + HIP launches, no sync inside a step): the step is GPU-bound
           # while this stays below ms_per_step
           "host_enqueue_ms_per_step": round(t_enq * 1e3 / args.steps, 3),
           "roofline": roofline}
    if dp_info is not None:
        out["dp"] = dp_info
    if args.profile_out:
        os.makedirs(os.path.dirname(os.path.abspath(args.profile_out)) or ".", exist_ok=True)
        with open(args.profile_out, "w") as f:
            f.write("# per-kernel HIP-event time over the timed region (%d steps)\n# tag\tcalls\ttotal_ms\tavg_ms\tshare\n"
                    % args.steps)
            for tag, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                f.write("%s\t%d\t%.3f\t%.4f\t%.4f\n" % (tag, c, ms, ms / max(c, 1), ms / max(total_kernel_ms, 1e-9)))
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
