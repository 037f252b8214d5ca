"""Evaluation loop (reference medicalseg/core/val.py:29-187): eval-mode forward per
validation volume (batch 1), mDice = mean over volumes of the mean over classes of the
soft V-Net dice that DiceLoss returns as a side output."""
import os
import time

import numpy as np

from .. import nn
from ..datasets import DataLoader
from ..device import to_tensor
from ..device import Tensor
from ..utils import TimeAverager, logger, loss_computation, metric, save_array
from . import infer

np.set_printoptions(suppress=True)


def _load_dataset_json(eval_dataset):
    path = getattr(eval_dataset, "dataset_json_path", "")
    if path and os.path.exists(path):
        import json
        with open(path, encoding="utf-8") as f:
            return json.load(f)
    return None


_IDENTITY_GEOMETRY = {"spacing": (1.0, 1.0, 1.0), "direction": (1, 0, 0, 0, 1, 0, 0, 0, 1), "origin": (0.0, 0.0, 0.0),
                      "format": "xyz"}
_warned_geometry = [False]


def _image_info(dataset_json, idx):
    """spacing / direction / origin of one validation volume (reference core/val.py:96-97,149-154).  Identity geometry only
    when the dataset HAS no json (synthetic data); with a json, a volume or key that is missing raises KeyError as the
    reference's dictionary lookups do -- a NIfTI file with a silently wrong geometry is worse than no file."""
    if not dataset_json:
        if not _warned_geometry[0]:
            logger.warning("evaluate: the dataset has no dataset.json; predictions are saved with identity spacing / origin / direction.")
            _warned_geometry[0] = True
        return dict(_IDENTITY_GEOMETRY)
    name = str(idx[0]).split("/")[-1].split(".")[0]
    j = dataset_json["training"][name]
    return {"spacing": j["spacing_resample"], "direction": j["direction"], "origin": j["origin"], "format": "xyz"}


def evaluate(model, eval_dataset, losses, num_workers=0, print_detail=True, auc_roc=False, writer=None,
             save_dir=None):
    new_loss = {'types': [losses['types'][0]], 'coef': [losses['coef'][0]]}
    if writer is not None:
        logger.warning("evaluate(writer=...): VisualDL logging is not built; the writer is ignored.")
    model.eval()
    dataset_json = _load_dataset_json(eval_dataset)
    from ..parallel import ParallelEnv
    env = ParallelEnv()
    local_rank = env.local_rank
    loader = DataLoader(eval_dataset, batch_size=1, shuffle=False, drop_last=False, num_workers=num_workers)
    total_iters = len(loader)
    if print_detail:
        logger.info("Start evaluating (total_samples: {}, total_iters: {})...".format(len(eval_dataset), total_iters))
    reader_cost_averager = TimeAverager()
    batch_cost_averager = TimeAverager()
    batch_start = time.time()
    mdice = 0.0
    channel_dice_array = np.array([])
    loss_all = 0.0
    logits_all, label_all = [], []   # auc_roc: softmax scores and labels of the whole set on the host (core/val.py:121-131)
    with nn.fused_inference():       # one scope for the whole set: BN is folded into the conv weights once
        for it, (im, label, idx) in enumerate(loader):
            reader_cost_averager.record(time.time() - batch_start)
            label_t = to_tensor(label.astype('int32'))
            pred, logits = infer.inference(model, to_tensor(im), ori_shape=label.shape[-3:],
                                           transforms=eval_dataset.transforms.transforms)
            loss, per_channel_dice = loss_computation(logits, label_t, new_loss)
            loss = sum(loss)
            if auc_roc:
                lg = logits[0] if isinstance(logits, (list, tuple)) else logits
                probs = Tensor.empty(lg.dev, lg.n, lg.d, lg.h, lg.w, lg.c)
                lg.dev.call("msk_softmax_c", lg.msk(), probs.msk())       # F.softmax(logits, axis=1) on the device
                logits_all.append(probs.numpy())
                label_all.append(np.asarray(label))
            loss_all += loss.numpy()
            pcd = np.asarray(per_channel_dice)
            mdice += np.mean(pcd)
            channel_dice_array = pcd.copy() if channel_dice_array.size == 0 else channel_dice_array + pcd
            if it < 5 and save_dir is not None:
                # reference core/val.py:137-153: npy + nii.gz with the volume's geometry from the dataset json
                info = _image_info(dataset_json, idx)
                save_array(save_path=os.path.join(save_dir, str(it)),
                           save_content={'pred': pred.numpy(), 'label': label, 'img': im},
                           form=('npy', 'nii.gz'), image_infor=info)
            batch_cost_averager.record(time.time() - batch_start, num_samples=len(label))
            reader_cost_averager.reset()
            batch_cost_averager.reset()
            batch_start = time.time()
    total_iters = max(total_iters, 1)
    mdice /= total_iters
    channel_dice_array = channel_dice_array / total_iters
    loss_all = loss_all / total_iters
    result_dict = {"mdice": float(mdice)}
    auc_infor = ""
    if auc_roc:
        auc = metric.auc_roc(np.concatenate(logits_all), np.concatenate(label_all), num_classes=eval_dataset.num_classes)
        auc_infor = ' Auc_roc: {:.4f}'.format(auc)
        result_dict['auc_roc'] = auc
    if print_detail and local_rank == 0:
        logger.info("[EVAL] #Images: {}, Dice: {:.4f}, Loss: {:6f}".format(len(eval_dataset), mdice,
                                                                           float(np.ravel(loss_all)[0])) + auc_infor)
        logger.info("[EVAL] Class dice: \n" + str(np.round(channel_dice_array, 4)))
    return result_dict
