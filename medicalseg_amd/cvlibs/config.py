"""YAML experiment description -> live objects; drop-in for the reference's
medicalseg/cvlibs/config.py (same file format, same attribute surface: ``dic``, ``batch_size``,
``iters``, ``lr_scheduler``, ``optimizer``, ``loss``, ``model``, ``train_dataset``,
``val_dataset``, ``export_config`` ...), organised as three small pieces:

  * ``load_yaml_tree``  -- ``_base_`` inheritance: a child file is overlaid on its base, nested
    mappings merge key by key, ``_inherited_: False`` replaces the subtree (reference :94-126);
  * ``ComponentFactory`` -- ``{type: Name, **kwargs}`` -> ``registry[Name](**kwargs)``,
    recursively through nested mappings and lists, searching the five registries (:371-403);
  * ``Config`` -- scalar getters, CLI overrides (:128-143), the schedule/optimizer/loss
    builders with the reference's defaults, and the ONE cached model instance, so that the
    optimizer built afterwards holds the live parameters.
"""
import io
import os
import warnings
from typing import Any, Dict

import yaml

from .. import optimizer as optim
from ..nn import SyncBatchNorm
from ..utils import logger
from . import manager

_CUT = '_inherited_'
_PARENT = '_base_'


def _overlay(child: dict, parent: dict) -> dict:
    """child wins; mappings present on both sides merge recursively unless the child opts out."""
    if child.get(_CUT, True) in (False, 0):
        return {k: v for k, v in child.items() if k != _CUT}
    merged = dict(parent)
    for key, value in child.items():
        both_maps = isinstance(value, dict) and isinstance(merged.get(key), dict)
        merged[key] = _overlay(value, merged[key]) if both_maps else value
    return merged


def load_yaml_tree(path: str) -> dict:
    with io.open(path, 'r', encoding='utf-8') as fh:
        tree = yaml.load(fh, Loader=yaml.FullLoader)
    parent = tree.pop(_PARENT, None)
    if parent is None:
        return tree
    return _overlay(tree, load_yaml_tree(os.path.join(os.path.dirname(path), parent)))


class ComponentFactory:
    REGISTRIES = ('MODELS', 'BACKBONES', 'DATASETS', 'TRANSFORMS', 'LOSSES')

    @classmethod
    def lookup(cls, name: str) -> Any:
        for reg_name in cls.REGISTRIES:
            registry = getattr(manager, reg_name)
            if name in registry.components_dict:
                return registry[name]
        raise RuntimeError('The specified component was not found {}.'.format(name))

    @staticmethod
    def is_spec(item: Any) -> bool:
        return isinstance(item, dict) and 'type' in item

    @classmethod
    def _value(cls, item: Any) -> Any:
        if cls.is_spec(item):
            return cls.build(item)
        if isinstance(item, list):
            return [cls.build(e) if cls.is_spec(e) else e for e in item]
        return item

    @classmethod
    def build(cls, spec: dict) -> Any:
        if 'type' not in spec:
            raise RuntimeError('No object information in {}.'.format(spec))
        kwargs = {k: cls._value(v) for k, v in spec.items() if k != 'type'}
        return cls.lookup(spec['type'])(**kwargs)


class Config(object):
    def __init__(self, path: str, learning_rate: float = None, batch_size: int = None, iters: int = None):
        if not path:
            raise ValueError('Please specify the configuration file path.')
        if not os.path.exists(path):
            raise FileNotFoundError('File {} does not exist'.format(path))
        if not path.endswith(('yml', 'yaml')):
            raise RuntimeError('Config file should in yaml format!')
        self._model = None
        self._losses = None
        self.dic = load_yaml_tree(path)
        self.data_root_path_warning()
        self.update(learning_rate=learning_rate, batch_size=batch_size, iters=iters)

    # kept for callers of the reference's private helpers
    def _parse_from_yaml(self, path: str):
        return load_yaml_tree(path)

    def _update_dic(self, dic, base_dic):
        return _overlay(dic, base_dic)

    def _load_component(self, com_name: str) -> Any:
        return ComponentFactory.lookup(com_name)

    def _load_object(self, cfg: dict) -> Any:
        return ComponentFactory.build(cfg)

    def _is_meta_type(self, item: Any) -> bool:
        return ComponentFactory.is_spec(item)

    # ---- command-line overrides (train.py --learning_rate/--batch_size/--iters) -------------
    def update(self, learning_rate: float = None, batch_size: int = None, iters: int = None):
        if learning_rate:
            if 'lr_scheduler' in self.dic:
                self.dic['lr_scheduler']['learning_rate'] = learning_rate
            else:
                self.dic['learning_rate']['value'] = learning_rate
        for key, value in (('batch_size', batch_size), ('iters', iters)):
            if value:
                self.dic[key] = value

    # ---- scalars ---------------------------------------------------------------------------------
    @property
    def batch_size(self) -> int:
        return self.dic.get('batch_size', 1)

    @property
    def iters(self) -> int:
        if not self.dic.get('iters'):
            raise RuntimeError('No iters specified in the configuration file.')
        return self.dic['iters']

    @property
    def export_config(self) -> Dict:
        return self.dic.get('export', {})

    @property
    def to_static_training(self) -> bool:
        return self.dic.get('to_static_training', False)

    def __str__(self) -> str:
        return yaml.dump(self.dic)

    def data_root_path_warning(self):
        if "data_root" not in self.dic:
            raise RuntimeError('The dataroot need to be set in the config file')
        if self.dic["data_root"] == 'data/':
            warnings.warn("Warning: The data dir now is {}, you should change the data_root in the global.yml if "
                          "this directory didn't have enough space".format(os.path.join(os.getcwd(), 'data/')))

    # ---- schedule and optimizer ------------------------------------------------------------------
    def _poly_defaults(self, args: dict) -> dict:
        args.setdefault('decay_steps', self.iters)
        args.setdefault('end_lr', 0)
        return args

    @property
    def lr_scheduler(self):
        spec = dict(self.dic.get('lr_scheduler') or {})
        if not spec:
            raise RuntimeError('No `lr_scheduler` specified in the configuration file.')
        kind = spec.pop('type')
        if kind == 'PolynomialDecay':
            self._poly_defaults(spec).setdefault('power', 0.9)
        schedule_cls = getattr(optim.lr, kind, None)
        if schedule_cls is None:
            raise RuntimeError('Unknown lr_scheduler type {}.'.format(kind))
        return schedule_cls(**spec)

    @property
    def decay_args(self) -> dict:
        args = dict(self.dic.get('learning_rate', {}).get('decay', {'type': 'poly', 'power': 0.9}))
        return self._poly_defaults(args) if args['type'] == 'poly' else args

    @property
    def learning_rate(self):
        """Legacy ``learning_rate: {value, decay}`` block (deprecated by the reference too)."""
        logger.warning('`learning_rate` in configuration file will be deprecated, please use `lr_scheduler` '
                       'instead. E.g\n    lr_scheduler:\n        type: PolynomialDecay\n        learning_rate: 0.01')
        block = self.dic.get('learning_rate', {})
        if isinstance(block, float):
            return block
        value = block.get('value')
        if not value:
            raise RuntimeError('No learning rate specified in the configuration file.')
        args = self.decay_args
        legacy = {'poly': lambda: optim.lr.PolynomialDecay(value, **args),
                  'piecewise': lambda: optim.lr.PiecewiseDecay(values=value, **args),
                  'stepdecay': lambda: optim.lr.StepDecay(value, **args)}
        kind = args.pop('type')
        if kind not in legacy:
            raise RuntimeError('Only poly and piecewise decay support.')
        return legacy[kind]()

    @property
    def optimizer_args(self) -> dict:
        args = dict(self.dic.get('optimizer', {}))
        if args['type'] == 'sgd':
            args.setdefault('momentum', 0.9)
        return args

    @property
    def optimizer(self):
        schedule = self.lr_scheduler if 'lr_scheduler' in self.dic else self.learning_rate
        args = self.optimizer_args
        kind = args.pop('type')
        if kind == 'sgd':  # the reference maps sgd to Momentum (config.py:212-214)
            cls = optim.Momentum
        elif kind == 'adam':  # config.py:214-216
            cls = optim.Adam
        elif kind in optim.OPTIMIZERS:      # (not optim.__all__: 'lr' is the scheduler namespace, not an optimizer)
            cls = getattr(optim, kind)
        else:
            raise RuntimeError('Unknown optimizer type {}.'.format(kind))
        return cls(schedule, parameters=self.model.parameters(), **args)

    # ---- loss ------------------------------------------------------------------------------------
    @property
    def loss(self) -> dict:
        if self._losses is None:
            self._losses = self._prepare_loss('loss')
        return self._losses

    def _prepare_loss(self, loss_name):
        """``types`` x ``coef``: a single type is replicated over the coefficients (deep
        supervision); non-mixed losses inherit the training set's ignore_index."""
        block = dict(self.dic.get(loss_name, {}))
        if 'types' not in block or 'coef' not in block:
            raise ValueError('Loss config should contain keys of "types" and "coef"')
        specs, coef = list(block['types']), block['coef']
        if len(specs) != len(coef):
            if len(specs) != 1:
                raise ValueError('The length of types should equal to coef or equal to 1 in loss config, but they '
                                 'are {} and {}.'.format(len(specs), len(coef)))
            specs = specs * len(coef)
        built = []
        for spec in specs:
            spec = dict(spec)
            if spec['type'] != 'MixedLoss':
                ds_ignore = self.train_dataset.ignore_index
                assert spec.get('ignore_index', ds_ignore) == ds_ignore, \
                    'If ignore_index of loss is set, the ignore_index of loss and train_dataset must be the same. ' \
                    'Currently, loss ignore_index = {}, train_dataset ignore_index = {}.'.format(
                        spec.get('ignore_index'), ds_ignore)
                spec['ignore_index'] = ds_ignore
            built.append(ComponentFactory.build(spec))
        losses = {k: v for k, v in block.items() if k != 'types'}
        losses['types'] = built
        if len(losses['coef']) != len(built):
            raise RuntimeError('The length of coef should equal to types in loss config: {} != {}.'.format(
                len(losses['coef']), len(built)))
        return losses

    # ---- datasets --------------------------------------------------------------------------------
    def _dataset_spec(self, key: str) -> Dict:
        spec = dict(self.dic.get(key) or {})
        if spec:
            for sub in ('dataset_root', 'result_dir'):  # both live under data_root (config.py:331-358)
                spec[sub] = os.path.join(self.dic['data_root'], spec.get(sub) or '')
        return spec

    def _dataset(self, key: str):
        spec = self._dataset_spec(key)
        return ComponentFactory.build(spec) if spec else None

    train_dataset_config = property(lambda self: self._dataset_spec('train_dataset'))
    val_dataset_config = property(lambda self: self._dataset_spec('val_dataset'))
    train_dataset_class = property(lambda self: ComponentFactory.lookup(self.train_dataset_config['type']))
    val_dataset_class = property(lambda self: ComponentFactory.lookup(self.val_dataset_config['type']))
    train_dataset = property(lambda self: self._dataset('train_dataset'))
    val_dataset = property(lambda self: self._dataset('val_dataset'))

    # ---- model -----------------------------------------------------------------------------------
    def _infer_num_classes(self):
        """First configured dataset decides: its class constant NUM_CLASSES, else the instance's."""
        for key in ('train_dataset', 'val_dataset'):
            if not self.dic.get(key):
                continue
            cls = ComponentFactory.lookup(self._dataset_spec(key)['type'])
            if hasattr(cls, 'NUM_CLASSES'):
                return cls.NUM_CLASSES
            return getattr(self._dataset(key), 'num_classes', None)
        return None

    @property
    def model(self):
        spec = dict(self.dic.get('model') or {})
        if not spec:
            raise RuntimeError('No model specified in the configuration file.')
        if self._model is None:
            if 'num_classes' not in spec:
                n = self._infer_num_classes()
                if n is not None:
                    spec['num_classes'] = n
            self._model = ComponentFactory.build(spec)
        # the reference converts every BatchNorm to SyncBatchNorm here, unconditionally (config.py:322)
        self._model = SyncBatchNorm.convert_sync_batchnorm(self._model)
        return self._model
