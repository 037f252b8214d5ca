#!/usr/bin/env python
"""Evaluation entry point with the reference's command line (val.py:25-126)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    p = argparse.ArgumentParser(description='Model evaluation')
    p.add_argument("--config", dest="cfg", help="The config file.", default=None, type=str)
    p.add_argument('--model_path', dest='model_path', help='The path of model for evaluation', type=str, default=None)
    p.add_argument('--save_dir', dest='save_dir', help='The path to save result', type=str, default="saved_model/val")
    p.add_argument('--num_workers', dest='num_workers', help='Num workers for data loader', type=int, default=0)
    p.add_argument('--print_detail', dest='print_detail', type=bool, default=True)
    p.add_argument('--auc_roc', dest='auc_roc', help='Whether to use auc_roc metric', type=bool, default=False)
    return p.parse_args()


def main(args):
    from medicalseg_amd.core import evaluate
    from medicalseg_amd.cvlibs import Config
    from medicalseg_amd.utils import load_entire_model, logger
    if not args.cfg:
        raise RuntimeError('No configuration file specified.')
    cfg = Config(args.cfg)
    val_dataset = cfg.val_dataset
    if val_dataset is None:
        raise RuntimeError('The verification dataset is not specified in the configuration file.')
    model = cfg.model
    if args.model_path:
        load_entire_model(model, args.model_path)
        logger.info('Loaded trained params of model successfully')
    print(evaluate(model, val_dataset, cfg.loss, num_workers=args.num_workers, print_detail=args.print_detail,
                   auc_roc=args.auc_roc, save_dir=args.save_dir))


if __name__ == '__main__':
    main(parse_args())
