"""YAML experiment configuration -> objects (drop-in for medicalseg/cvlibs/config.py:29-429).

Same file format and semantics: recursive ``_base_`` inheritance with dict deep-merge and
``_inherited_: False`` to cut a subtree (:94-126); CLI overrides for learning rate, batch
size, iters (:128-143); lazily built ``model`` (cached, so the optimizer holds the live
parameters -- SURVEY Q12), ``optimizer``, ``lr_scheduler``, ``loss``, datasets; component
lookup across the five registries (:371-403)."""
import codecs
import os
import warnings
from typing import Any, Dict

import yaml

from .. import optimizer as optim
from ..nn import SyncBatchNorm
from ..utils import logger
from . import manager


class Config(object):
    def __init__(self, path: str, learning_rate: float = None, batch_size: int = None, iters: int = None):
        if not path:
            raise ValueError('Please specify the configuration file path.')
        if not os.path.exists(path):
            raise FileNotFoundError('File {} does not exist'.format(path))
        self._model = None
        self._losses = None
        if path.endswith('yml') or path.endswith('yaml'):
            self.dic = self._parse_from_yaml(path)
            self.data_root_path_warning()
        else:
            raise RuntimeError('Config file should in yaml format!')
        self.update(learning_rate=learning_rate, batch_size=batch_size, iters=iters)

    # ---- parsing ------------------------------------------------------------------------
    def _update_dic(self, dic, base_dic):
        """Overlay ``dic`` on ``base_dic`` (nested dicts merge unless `_inherited_: False`)."""
        base_dic = base_dic.copy()
        dic = dic.copy()
        if dic.get('_inherited_', True) is False or dic.get('_inherited_', True) == False:  # noqa: E712
            dic.pop('_inherited_')
            return dic
        for key, val in dic.items():
            if isinstance(val, dict) and key in base_dic:
                base_dic[key] = self._update_dic(val, base_dic[key])
            else:
                base_dic[key] = val
        return base_dic

    def _parse_from_yaml(self, path: str):
        with codecs.open(path, 'r', 'utf-8') as file:
            dic = yaml.load(file, Loader=yaml.FullLoader)
        if '_base_' in dic:
            base_path = os.path.join(os.path.dirname(path), dic.pop('_base_'))
            dic = self._update_dic(dic, self._parse_from_yaml(base_path))
        return dic

    def update(self, learning_rate: float = None, batch_size: int = None, iters: int = None):
        if learning_rate:
            if 'lr_scheduler' in self.dic:
                self.dic['lr_scheduler']['learning_rate'] = learning_rate
            else:
                self.dic['learning_rate']['value'] = learning_rate
        if batch_size:
            self.dic['batch_size'] = batch_size
        if iters:
            self.dic['iters'] = iters

    # ---- scalars ----------------------------------------------------------------------------
    @property
    def batch_size(self) -> int:
        return self.dic.get('batch_size', 1)

    @property
    def iters(self) -> int:
        iters = self.dic.get('iters')
        if not iters:
            raise RuntimeError('No iters specified in the configuration file.')
        return iters

    # ---- lr / optimizer ------------------------------------------------------------------------
    @property
    def lr_scheduler(self):
        if 'lr_scheduler' not in self.dic:
            raise RuntimeError('No `lr_scheduler` specified in the configuration file.')
        params = self.dic.get('lr_scheduler').copy()
        lr_type = params.pop('type')
        if lr_type == 'PolynomialDecay':
            params.setdefault('decay_steps', self.iters)
            params.setdefault('end_lr', 0)
            params.setdefault('power', 0.9)
        if not hasattr(optim.lr, lr_type):
            raise RuntimeError('Unknown lr_scheduler type {}.'.format(lr_type))
        return getattr(optim.lr, lr_type)(**params)

    @property
    def learning_rate(self):
        logger.warning('''`learning_rate` in configuration file will be deprecated, please use `lr_scheduler` instead. E.g
            lr_scheduler:
                type: PolynomialDecay
                learning_rate: 0.01''')
        _learning_rate = self.dic.get('learning_rate', {})
        if isinstance(_learning_rate, float):
            return _learning_rate
        _learning_rate = self.dic.get('learning_rate', {}).get('value')
        if not _learning_rate:
            raise RuntimeError('No learning rate specified in the configuration file.')
        args = self.decay_args
        decay_type = args.pop('type')
        if decay_type == 'poly':
            return optim.lr.PolynomialDecay(_learning_rate, **args)
        elif decay_type == 'piecewise':
            return optim.lr.PiecewiseDecay(values=_learning_rate, **args)
        elif decay_type == 'stepdecay':
            return optim.lr.StepDecay(_learning_rate, **args)
        raise RuntimeError('Only poly and piecewise decay support.')

    @property
    def optimizer(self):
        lr = self.lr_scheduler if 'lr_scheduler' in self.dic else self.learning_rate
        args = self.optimizer_args
        optimizer_type = args.pop('type')
        if optimizer_type == 'sgd':
            return optim.Momentum(lr, parameters=self.model.parameters(), **args)
        elif optimizer_type in optim.__all__:
            return getattr(optim, optimizer_type)(lr, parameters=self.model.parameters(), **args)
        raise RuntimeError('Unknown optimizer type {}.'.format(optimizer_type))

    @property
    def optimizer_args(self) -> dict:
        args = self.dic.get('optimizer', {}).copy()
        if args['type'] == 'sgd':
            args.setdefault('momentum', 0.9)
        return args

    @property
    def decay_args(self) -> dict:
        args = self.dic.get('learning_rate', {}).get('decay', {'type': 'poly', 'power': 0.9}).copy()
        if args['type'] == 'poly':
            args.setdefault('decay_steps', self.iters)
            args.setdefault('end_lr', 0)
        return args

    # ---- loss --------------------------------------------------------------------------------
    @property
    def loss(self) -> dict:
        if self._losses is None:
            self._losses = self._prepare_loss('loss')
        return self._losses

    def _prepare_loss(self, loss_name):
        args = self.dic.get(loss_name, {}).copy()
        if 'types' in args and 'coef' in args:
            len_types = len(args['types'])
            len_coef = len(args['coef'])
            if len_types != len_coef:
                if len_types == 1:
                    args['types'] = args['types'] * len_coef
                else:
                    raise ValueError(
                        'The length of types should equal to coef or equal to 1 in loss config, but they are {} and {}.'
                        .format(len_types, len_coef))
        else:
            raise ValueError('Loss config should contain keys of "types" and "coef"')

        losses = dict()
        for key, val in args.items():
            if key == 'types':
                losses['types'] = []
                for item in args['types']:
                    item = dict(item)
                    if item['type'] != 'MixedLoss':
                        if 'ignore_index' in item:
                            assert item['ignore_index'] == self.train_dataset.ignore_index, \
                                'If ignore_index of loss is set, the ignore_index of loss and train_dataset must be ' \
                                'the same. Currently, loss ignore_index = {}, train_dataset ignore_index = {}.'.format(
                                    item['ignore_index'], self.train_dataset.ignore_index)
                        item['ignore_index'] = self.train_dataset.ignore_index
                    losses['types'].append(self._load_object(item))
            else:
                losses[key] = val
        if len(losses['coef']) != len(losses['types']):
            raise RuntimeError('The length of coef should equal to types in loss config: {} != {}.'.format(
                len(losses['coef']), len(losses['types'])))
        return losses

    # ---- model / datasets ------------------------------------------------------------------------
    @property
    def model(self):
        model_cfg = (self.dic.get('model') or {}).copy()
        if not model_cfg:
            raise RuntimeError('No model specified in the configuration file.')
        if 'num_classes' not in model_cfg:
            num_classes = None
            if self.dic.get('train_dataset'):
                if hasattr(self.train_dataset_class, 'NUM_CLASSES'):
                    num_classes = self.train_dataset_class.NUM_CLASSES
                elif hasattr(self.train_dataset, 'num_classes'):
                    num_classes = self.train_dataset.num_classes
            elif self.dic.get('val_dataset'):
                if hasattr(self.val_dataset_class, 'NUM_CLASSES'):
                    num_classes = self.val_dataset_class.NUM_CLASSES
                elif hasattr(self.val_dataset, 'num_classes'):
                    num_classes = self.val_dataset.num_classes
            if num_classes is not None:
                model_cfg['num_classes'] = num_classes
        if not self._model:
            self._model = self._load_object(model_cfg)
        # reference: paddle.nn.SyncBatchNorm.convert_sync_batchnorm (config.py:322)
        self._model = SyncBatchNorm.convert_sync_batchnorm(self._model)
        return self._model

    def _dataset_config(self, key) -> Dict:
        cfg = self.dic.get(key, {}).copy()
        if not cfg:
            return cfg
        cfg['dataset_root'] = os.path.join(self.dic['data_root'], cfg.get('dataset_root') or '')
        cfg['result_dir'] = os.path.join(self.dic['data_root'], cfg.get('result_dir') or '')
        return cfg

    @property
    def train_dataset_config(self) -> Dict:
        return self._dataset_config('train_dataset')

    @property
    def val_dataset_config(self) -> Dict:
        return self._dataset_config('val_dataset')

    @property
    def train_dataset_class(self):
        return self._load_component(self.train_dataset_config['type'])

    @property
    def val_dataset_class(self):
        return self._load_component(self.val_dataset_config['type'])

    @property
    def train_dataset(self):
        cfg = self.train_dataset_config
        return self._load_object(cfg) if cfg else None

    @property
    def val_dataset(self):
        cfg = self.val_dataset_config
        return self._load_object(cfg) if cfg else None

    # ---- component loading ------------------------------------------------------------------------
    def _load_component(self, com_name: str) -> Any:
        for com in (manager.MODELS, manager.BACKBONES, manager.DATASETS, manager.TRANSFORMS, manager.LOSSES):
            if com_name in com.components_dict:
                return com[com_name]
        raise RuntimeError('The specified component was not found {}.'.format(com_name))

    def _load_object(self, cfg: dict) -> Any:
        cfg = cfg.copy()
        if 'type' not in cfg:
            raise RuntimeError('No object information in {}.'.format(cfg))
        component = self._load_component(cfg.pop('type'))
        params = {}
        for key, val in cfg.items():
            if self._is_meta_type(val):
                params[key] = self._load_object(val)
            elif isinstance(val, list):
                params[key] = [self._load_object(item) if self._is_meta_type(item) else item for item in val]
            else:
                params[key] = val
        return component(**params)

    @property
    def export_config(self) -> Dict:
        return self.dic.get('export', {})

    @property
    def to_static_training(self) -> bool:
        return self.dic.get('to_static_training', False)

    def _is_meta_type(self, item: Any) -> bool:
        return isinstance(item, dict) and 'type' in item

    def __str__(self) -> str:
        return yaml.dump(self.dic)

    def data_root_path_warning(self):
        if "data_root" not in self.dic:
            raise RuntimeError('The dataroot need to be set in the config file')
        data_root = self.dic["data_root"]
        if data_root == 'data/':
            warnings.warn("Warning: The data dir now is {}, you should change the data_root in the global.yml if "
                          "this directory didn't have enough space".format(os.path.join(os.getcwd(), data_root)))
