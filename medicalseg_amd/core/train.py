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
from ..utils import TimeAverager, calculate_eta, logger, loss_computation, resume, save
from .val import evaluate


def train(model, train_dataset, val_dataset=None, optimizer=None, save_dir='output', iters=10000, batch_size=2,
          resume_model=None, save_interval=1000, log_iters=10, num_workers=0, use_vdl=False, losses=None,
          keep_checkpoint_max=5, profiler_options=None, to_static_training=False):
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
        init_parallel_env()
        ddp_model = DataParallel(model)
    loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, drop_last=False,
                        num_workers=num_workers)
    if use_vdl:
        logger.warning("VisualDL is not available in this build; --use_vdl is ignored.")
    dev = model.dev

    pending = []  # (loss Scalar, [loss_i Scalars], per_channel_dice lazy) since the last log boundary
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

            # values of this iteration (tiny device buffers) are fetched at the log boundary
            vals = _snapshot(dev, loss, loss_list, per_channel_dice)
            pending.append(vals)
            if it % log_iters == 0:
                # the iterations of this window were only enqueued so far: their device time has to land inside the
                # window's batch_cost / ips (recording before the sync under-reported the step by ~14 %)
                dev.sync()
            batch_cost_averager.record(time.time() - batch_start, num_samples=batch_size)

            if it % log_iters == 0:
                avg_loss, avg_loss_list, mdice = _reduce_pending(dev, pending)
                pending = []
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
    time.sleep(0.1)


def _snapshot(dev, loss, loss_list, per_channel_dice):
    """Keep the device scalars of one iteration alive past the next arena reset by copying
    them into a small persistent ring (async d2d, no sync)."""
    node = loss.terms[0][1]
    node.evaluate()
    n = 2 + node.C
    ptr = _ring_slot(dev, n)
    dev.d2d(ptr, node.out_ptr, n * 4)
    coefs = [(sum(c for c, _, w in l.terms if w == "ce"), sum(c for c, _, w in l.terms if w == "dice"))
             for l in loss_list]
    return ptr, node.C, coefs


_RING = {"ptrs": [], "next": 0, "n": 0}


def _ring_slot(dev, n, depth=4096):
    if _RING["n"] < n or not _RING["ptrs"]:
        _RING["ptrs"] = [dev.malloc(depth * n * 4)]
        _RING["n"], _RING["next"], _RING["depth"] = n, 0, depth
    i = _RING["next"]
    _RING["next"] = (i + 1) % _RING["depth"]
    return _RING["ptrs"][0] + i * _RING["n"] * 4


def _reduce_pending(dev, pending):
    tot, per_loss, dsc = 0.0, None, 0.0
    for ptr, Cn, coefs in pending:
        v = dev.d2h(ptr, (2 + Cn,), np.float32).astype(np.float64)
        li = [cc * v[0] + cd * v[1] for cc, cd in coefs]
        tot += sum(li)
        per_loss = li if per_loss is None else [a + b for a, b in zip(per_loss, li)]
        dsc += float(np.mean(v[2:])) * 100
    k = max(len(pending), 1)
    return tot / k, [p / k for p in (per_loss or [])], dsc / k
