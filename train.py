#!/usr/bin/env python
"""Training entry point with the reference's command line (train.py:26-116): the same flags
drive the MI355X-native path.

    python train.py --config configs/synthetic/vnet_synthetic_ct_128.yml --save_dir out --iters 20 --log_iters 5

Multi-GPU: one process per GPU, e.g.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config ...
"""
import argparse
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    p = argparse.ArgumentParser(description='Model training')
    p.add_argument("--config", dest="cfg", help="The config file.", default=None, type=str)
    p.add_argument('--iters', dest='iters', help='iters for training', type=int, default=None)
    p.add_argument('--batch_size', dest='batch_size', help='Mini batch size of one gpu or cpu', type=int, default=None)
    p.add_argument('--learning_rate', dest='learning_rate', help='Learning rate', type=float, default=None)
    p.add_argument('--save_interval', dest='save_interval', type=int, default=1000,
                   help='How many iters to save a model snapshot once during training.')
    p.add_argument('--resume_model', dest='resume_model', help='The path of resume model', type=str, default=None)
    p.add_argument('--save_dir', dest='save_dir', help='The directory for saving the model snapshot', type=str,
                   default='./output')
    p.add_argument('--keep_checkpoint_max', dest='keep_checkpoint_max', type=int, default=5,
                   help='Maximum number of checkpoints to save')
    p.add_argument('--num_workers', dest='num_workers', help='Num workers for data loader', type=int, default=0)
    p.add_argument('--do_eval', dest='do_eval', help='Eval while training', action='store_true')
    p.add_argument('--log_iters', dest='log_iters', help='Display logging information at every log_iters', default=100,
                   type=int)
    p.add_argument('--use_vdl', dest='use_vdl', help='Whether to record the data to VisualDL during training',
                   action='store_true')
    p.add_argument('--seed', dest='seed', help='Set the random seed during training.', default=None, type=int)
    p.add_argument('--no_sync_bn', dest='no_sync_bn', action='store_true',
                   help='rank-local BatchNorm statistics (documented deviation from the reference, which converts every '
                        'BatchNorm to SyncBatchNorm: cvlibs/config.py:322)')
    p.add_argument('--dp_mode', dest='dp_mode', default='auto', choices=('auto', '0', '1', '2', '3'),
                   help='multi-GPU: arrangement of the collectives (msk_dp.hip): auto (default) = 2 with more than one rank; 2 = '
                        'gradient buckets on a second communicator + stream, overlapped with backward; 0 = everything on the '
                        'compute stream, one gradient all-reduce after backward')
    p.add_argument('--gpus', dest='gpus', type=int, default=None,
                   help='spawn this many ranks (one per GPU) when no launcher set WORLD_SIZE -- replaces '
                        '`python -m paddle.distributed.launch train.py ...`')
    p.add_argument('--data_format', dest='data_format', type=str, default='NCHW',
                   help='Kept for CLI compatibility; the device layout is always NDHWC internally.')
    p.add_argument('--profiler_options', type=str, default=None,
                   help='The option string of the train profiler, e.g. "batch_range=[10,20];profile_path=out.tsv": per-kernel '
                        'HIP-event profile + roctx iteration ranges over the batch range (medicalseg_amd/utils/train_profiler.py)')
    return p.parse_args()


def main(args):
    if args.gpus and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from medicalseg_amd.parallel import spawn_ranks
        raise SystemExit(spawn_ranks(args.gpus))
    if args.seed is not None:
        np.random.seed(args.seed)
        random.seed(args.seed)
    from medicalseg_amd import nn
    from medicalseg_amd.core import train
    from medicalseg_amd.cvlibs import Config
    from medicalseg_amd.device import get_device
    from medicalseg_amd.utils import logger
    if args.seed is not None:
        nn.seed(args.seed)
        nn.Dropout3D.seed = args.seed
    if args.no_sync_bn:
        nn.BatchNorm3D.sync = False
    if args.data_format not in ('NCHW', 'NCDHW'):
        logger.warning("--data_format %s is ignored: tensors cross the API in the reference's NCDHW order, the device "
                       "layout is NDHWC internally." % args.data_format)
    if not args.cfg:
        raise RuntimeError('No configuration file specified.')
    cfg = Config(args.cfg, learning_rate=args.learning_rate, iters=args.iters, batch_size=args.batch_size)
    logger.info("device: " + get_device().name())
    train_dataset = cfg.train_dataset
    if train_dataset is None:
        raise RuntimeError('The training dataset is not specified in the configuration file.')
    elif len(train_dataset) == 0:
        raise ValueError('The length of train_dataset is 0. Please check if your dataset is valid')
    val_dataset = cfg.val_dataset if args.do_eval else None
    losses = cfg.loss
    logger.info('\n------------Config-----------\n' + str(cfg) + '-----------------------------')
    train(cfg.model, train_dataset, val_dataset=val_dataset, optimizer=cfg.optimizer, save_dir=args.save_dir,
          iters=cfg.iters, batch_size=cfg.batch_size, resume_model=args.resume_model, save_interval=args.save_interval,
          log_iters=args.log_iters, num_workers=args.num_workers, use_vdl=args.use_vdl, losses=losses,
          keep_checkpoint_max=args.keep_checkpoint_max, profiler_options=args.profiler_options,
          to_static_training=cfg.to_static_training, dp_mode=args.dp_mode)


if __name__ == '__main__':
    main(parse_args())
