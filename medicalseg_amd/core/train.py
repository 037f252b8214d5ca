"""Training loop with the reference's signature, step order, log line and checkpoint
layout (medicalseg/core/train.py:30-274), restructured for an accelerator that should
never wait for the host:

  * per-iteration host syncs of the reference (`loss.numpy()`, per-loss `.numpy()`,
    dice `.cpu().numpy()`; train.py:158-170, dice_loss.py:99) are deferred to log
    boundaries -- scalars stay on the device in between;
  * gradients live in one flat arena: `clear_gradients` is one memset, the optimizer one
    kernel, the data-parallel exchange one RCCL all-reduce issued right after backward.
"""
import os
import shutil
import time
from collections import deque

import numpy as np

from ..datasets import DataLoader
from ..device import to_tensor
from ..optimizer import lr as lr_mod
from ..parallel import DataParallel, ParallelEnv, init_parallel_env
from ..utils import TimeAverager, calculate_eta, logger, loss_computation, resume, save, train_profiler
from .val import evaluate


def train(model, train_dataset, val_dataset=None, optimizer=None, save_dir='output', iters=10000, batch_size=2,
          resume_model=None, save_interval=1000, log_iters=10, num_workers=0, use_vdl=False, losses=None,
          keep_checkpoint_max=5, profiler_options=None, to_static_training=False, dp_mode='auto'):
    model.train()
    env = ParallelEnv()
    nranks, local_rank = env.nranks, env.local_rank
    start_iter = 0
    if resume_model is not None:
        start_iter = resume(model, optimizer, resume_model)
    if not os.path.isdir(save_dir):
        if os.path.exists(save_dir):
            os.remove(save_dir)
        os.makedirs(save_dir, exist_ok=True)
    ddp_model = model
    if nranks > 1:
        init_parallel_env(dp_mode=dp_mode)    # train.py --dp_mode: auto = 2 (overlapped buckets) when there are several ranks; 0 = one all-reduce after backward
        ddp_model = DataParallel(model)
        logger.info("data parallel: {} ranks, dp_mode {} requested -> {} in effect (MSEGK_DP_MODE {}), gradient buckets overlapped with "
                    "backward: {}".format(nranks, dp_mode, model.dev.get_option("dp_mode"), os.environ.get("MSEGK_DP_MODE", "unset"),
                                          bool(getattr(ddp_model, "overlap", False))))
    elif hasattr(optimizer, "enable_eager") and os.environ.get("MSEGK_EAGER_OPT", "1") != "0":
        # one rank: a block's parameters are updated on the weight-gradient stream as soon as its backward is enqueued;
        # optimizer.step() below joins.  This loop never looks at parameters between backward() and step().
        optimizer.enable_eager(model)
    loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, drop_last=False,
                        num_workers=num_workers)
    if use_vdl:
        logger.warning("VisualDL is not available in this build; --use_vdl is ignored.")
    dev = model.dev

    pending = []  # ring slots of the iterations since the last log boundary (_snapshot)
    carry = None  # running sums of a log window that outgrew the ring
    iters_per_epoch = max(len(loader), 1)
    best_mean_dice = -1.0
    best_model_iter = -1
    reader_cost_averager = TimeAverager()
    batch_cost_averager = TimeAverager()
    save_models = deque()
    batch_start = time.time()
    result_dict = None

    it = start_iter
    while it < iters:
        for data in loader:
            if it >= iters:
                break
            reader_cost_averager.record(time.time() - batch_start)
            images = to_tensor(data[0], dev)
            # device-augmented batches arrive as device tensors already (datasets.DataLoader._stack_device)
            labels = data[1] if hasattr(data[1], 'ptr') else to_tensor(np.asarray(data[1]).astype('int32'), dev)

            logits_list = ddp_model(images)
            loss_list, per_channel_dice = loss_computation(logits_list=logits_list, labels=labels, losses=losses)
            loss = sum(loss_list)
            loss.backward()
            optimizer.step()

            lr = optimizer.get_lr()
            it += 1
            lr_sche = optimizer._learning_rate
            if isinstance(lr_sche, lr_mod.LRScheduler):
                lr_sche.step()
            model.clear_gradients()
            train_profiler.add_profiler_step(profiler_options, dev)   # reference core/train.py:153

            # values of this iteration (tiny device buffers) are fetched at the log boundary
            vals = _snapshot(dev, loss, loss_list, per_channel_dice)
            pending.append(vals)
            if (len(pending) + 1) * max(len(vals[0]), 1) >= RING_DEPTH and it % log_iters != 0:
                # log windows longer than the ring: fold what is pending into running sums before its slots are reused
                carry = _fold_carry(carry, _reduce_pending(dev, pending), len(pending))
                pending = []
            if it % log_iters == 0:
                # the iterations of this window were only enqueued so far: their device time has to land inside the
                # window's batch_cost / ips (recording before the sync under-reported the step by ~14 %)
                dev.sync()
            batch_cost_averager.record(time.time() - batch_start, num_samples=batch_size)

            if it % log_iters == 0:
                avg_loss, avg_loss_list, mdice = _unfold_carry(_fold_carry(carry, _reduce_pending(dev, pending), len(pending)))
                pending, carry = [], None
                if local_rank == 0:
                    remain_iters = iters - it
                    avg_train_batch_cost = batch_cost_averager.get_average()
                    avg_train_reader_cost = reader_cost_averager.get_average()
                    eta = calculate_eta(remain_iters, avg_train_batch_cost)
                    logger.info("[TRAIN] epoch: {}, iter: {}/{}, loss: {:.4f}, DSC: {:.4f}, "
                                "lr: {:.6f}, batch_cost: {:.4f}, reader_cost: {:.5f}, ips: {:.4f} samples/sec | ETA {}"
                                .format(it // iters_per_epoch, it, iters, avg_loss, mdice, lr, avg_train_batch_cost,
                                        avg_train_reader_cost, batch_cost_averager.get_ips_average(), eta))
                reader_cost_averager.reset()
                batch_cost_averager.reset()

            if (it % save_interval == 0 or it == iters) and (val_dataset is not None):
                result_dict = evaluate(model, val_dataset, losses, num_workers=1 if num_workers > 0 else 0,
                                       print_detail=True, auc_roc=False, save_dir=save_dir)
                model.train()

            if (it % save_interval == 0 or it == iters) and local_rank == 0:
                current_save_dir = os.path.join(save_dir, "iter_{}".format(it))
                os.makedirs(current_save_dir, exist_ok=True)
                save(model.state_dict(), os.path.join(current_save_dir, 'model.pdparams'))
                save(optimizer.state_dict(), os.path.join(current_save_dir, 'model.pdopt'))
                save_models.append(current_save_dir)
                if len(save_models) > keep_checkpoint_max > 0:
                    shutil.rmtree(save_models.popleft())
                if val_dataset is not None and result_dict is not None:
                    if result_dict['mdice'] > best_mean_dice:
                        best_mean_dice = result_dict['mdice']
                        best_model_iter = it
                        save(model.state_dict(), os.path.join(save_dir, "best_model", 'model.pdparams'))
                    logger.info('[EVAL] The model with the best validation mDice ({:.4f}) was saved at iter {}.'
                                .format(best_mean_dice, best_model_iter))
            batch_start = time.time()
    dev.sync()
    # reference core/train.py:265-269: after the loop rank 0 reports paddle.flops(model, [1, c, d, h, w]) -- one batch-1
    # forward.  Here: the analytic count of the convolutions of one real eval forward of that shape (nn.FLOPS, the table of
    # SURVEY App. A), printed in paddle's "Total Flops / Total Params" form (paddle.flops counts multiply-accumulates).
    if local_rank == 0 and it > start_iter:
        try:
            _report_flops(model, images.shape)
        except Exception as e:     # a report must never fail a finished training run
            logger.warning("FLOP report skipped: %r" % (e,))
    time.sleep(0.1)


def _report_flops(model, shape):
    from .. import nn as _nn
    _, c, d, h, w = shape
    dev = model.dev
    was_training = getattr(model, "training", True)
    keep = dict(_nn.FLOPS)
    _nn.FLOPS.update(on=True, same_k5=0.0, same_k3=0.0, other=0.0)
    try:
        model.eval()
        model(to_tensor(np.zeros((1, c, d, h, w), np.float32), dev))
        dev.sync()
        flops = _nn.FLOPS["same_k5"] + _nn.FLOPS["same_k3"] + _nn.FLOPS["other"]
    finally:
        _nn.FLOPS.update(keep)
        if was_training:
            model.train()
    params = sum(int(np.prod(p.shape)) for p in model.parameters())
    logger.info("Total Flops: {:d}     Total Params: {:d}".format(int(flops // 2), params))
    logger.info("(convolution multiply-accumulates of one [1, {}, {}, {}, {}] forward, paddle.flops' convention; algorithmic "
                "forward FLOPs {:.1f} G, training step fwd+bwd ~{:.1f} G per sample)".format(c, d, h, w, flops / 1e9, 3 * flops / 1e9))
    return int(flops // 2), params


def _snapshot(dev, loss, loss_list, per_channel_dice):
    """Keep the device scalars of one iteration alive past the next arena reset: the (2 + C)-float record
    {CE, dice loss, per-class dice} of EVERY distinct loss node (one per model output: VNetDeepSup has four) is copied
    into a small persistent ring (async d2d, no sync).  Returns (slots, terms, dsc_slot): terms = [(coef, slot index,
    0 = CE | 1 = dice) per entry of loss_list], dsc_slot = the node per_channel_dice was taken from (the reference
    reports the per-class dice of the LAST dice-bearing loss, utils/loss_utils.py:41-42)."""
    nodes, index = [], {}
    for l in loss_list:
        for _, node, _ in l.terms:
            if id(node) not in index:
                index[id(node)] = len(nodes)
                nodes.append(node)
    slots = []
    for node in nodes:
        node.evaluate()
        n = 2 + node.C
        ptr = _ring_slot(dev, n)
        dev.d2d(ptr, node.out_ptr, n * 4)
        slots.append((ptr, node.C))
    terms = [[(c, index[id(node)], 0 if w == "ce" else 1) for c, node, w in l.terms] for l in loss_list]
    dsc_slot = None
    if per_channel_dice is not None and hasattr(per_channel_dice, "ptr"):
        for i, node in enumerate(nodes):
            if node.out_ptr + 8 == per_channel_dice.ptr:
                dsc_slot = i
    return slots, terms, dsc_slot


_RING = {"ptrs": [], "next": 0, "n": 0, "depth": 0}
RING_DEPTH = 4096


def _ring_slot(dev, n, depth=None):
    depth = depth or RING_DEPTH
    if _RING["n"] < n or not _RING["ptrs"]:
        _RING["ptrs"] = [dev.malloc(depth * n * 4)]
        _RING["n"], _RING["next"], _RING["depth"] = n, 0, depth
    i = _RING["next"]
    _RING["next"] = (i + 1) % _RING["depth"]
    return _RING["ptrs"][0] + i * _RING["n"] * 4


def _reduce_pending(dev, pending):
    """-> (mean loss, [mean of each weighted loss term], mean DSC in percent) over the pending iterations."""
    tot, per_loss, dsc, ndsc = 0.0, None, 0.0, 0
    for slots, terms, dsc_slot in pending:
        vals = [dev.d2h(ptr, (2 + Cn,), np.float32).astype(np.float64) for ptr, Cn in slots]
        li = [sum(c * vals[si][which] for c, si, which in t) for t in terms]
        tot += sum(li)
        per_loss = li if per_loss is None else [a + b for a, b in zip(per_loss, li)]
        if dsc_slot is not None:
            dsc += float(np.mean(vals[dsc_slot][2:])) * 100
            ndsc += 1
    k = max(len(pending), 1)
    return tot / k, [p / k for p in (per_loss or [])], dsc / max(ndsc, 1)


def _fold_carry(carry, red, k):
    """Weighted running sums (count, loss, per-loss list, dsc) of partial window reductions."""
    if k == 0:
        return carry
    avg, per, dsc = red
    if carry is None:
        return [k, avg * k, [p * k for p in per], dsc * k]
    carry[0] += k
    carry[1] += avg * k
    carry[2] = [a + p * k for a, p in zip(carry[2], per)] if carry[2] else [p * k for p in per]
    carry[3] += dsc * k
    return carry


def _unfold_carry(carry):
    if carry is None:
        return 0.0, [], 0.0
    k = max(carry[0], 1)
    return carry[1] / k, [p / k for p in carry[2]], carry[3] / k
