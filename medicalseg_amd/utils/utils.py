"""Checkpoint helpers with the reference's directory layout and resume semantics
(medicalseg/utils/utils.py:76-135, core/train.py:230-254).

File format: ``model.pdparams`` / ``model.pdopt`` are pickles of {name: ndarray} -- the
format paddle.save writes for state dicts.  ``model.pdparams`` is interchangeable in both directions (tensors in
the reference's layouts and structured key names).  ``model.pdopt`` round-trips between runs of THIS
package; a Paddle-written one names its velocities after Paddle's internal parameter names, matches nothing here, and
is reported as such by Momentum.set_state_dict (momentum then restarts from zero)."""
import os
import pickle

import numpy as np

from . import logger


def save(state, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump({k: (np.asarray(v) if not isinstance(v, dict) else v) for k, v in state.items()}, f, protocol=2)


def load(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def load_entire_model(model, pretrained):
    if pretrained is not None:
        load_pretrained_model(model, pretrained)
    else:
        logger.warning('Not all pretrained params of {} are loaded, training from scratch or a '
                       'pretrained backbone.'.format(model.__class__.__name__))


def load_pretrained_model(model, pretrained_model):
    """Shape-checked partial load (reference utils.py:76-112).  URLs cannot be fetched
    (no network): a warning is logged and the model keeps its initialisation."""
    if pretrained_model is None:
        logger.info('No pretrained model to load, {} will be trained from scratch.'.format(model.__class__.__name__))
        return
    logger.info('Loading pretrained model from {}'.format(pretrained_model))
    if str(pretrained_model).startswith(("http://", "https://")):
        logger.warning('{} is a URL and this build has no downloader/network; skipping.'.format(pretrained_model))
        return
    if not os.path.exists(pretrained_model):
        raise ValueError('The pretrained model directory is not Found: {}'.format(pretrained_model))
    para_state_dict = load(pretrained_model)
    model_state_dict = model.state_dict()
    keys = model_state_dict.keys()
    num_params_loaded = 0
    to_set = {}
    for k in keys:
        if k not in para_state_dict:
            logger.warning("{} is not in pretrained model".format(k))
        elif list(np.shape(para_state_dict[k])) != list(model_state_dict[k].shape):
            logger.warning("[SKIP] Shape of pretrained params {} doesn't match.(Pretrained: {}, Actual: {})".format(
                k, np.shape(para_state_dict[k]), model_state_dict[k].shape))
        else:
            to_set[k] = para_state_dict[k]
            num_params_loaded += 1
    model.set_dict(to_set)
    logger.info("There are {}/{} variables loaded into {}.".format(num_params_loaded, len(model_state_dict),
                                                                   model.__class__.__name__))


def resume(model, optimizer, resume_model):
    """Load `<dir>/model.pdparams` + `model.pdopt`; start iter parsed from the `_N` suffix of
    the directory name (reference utils.py:115-135)."""
    if resume_model is None:
        logger.info('No model needed to resume.')
        return 0
    logger.info('Resume model from {}'.format(resume_model))
    if not os.path.exists(resume_model):
        raise ValueError('Directory of the model needed to resume is not Found: {}'.format(resume_model))
    resume_model = os.path.normpath(resume_model)
    missing, unexpected = model.set_state_dict(load(os.path.join(resume_model, 'model.pdparams')))
    if missing or unexpected:
        logger.warning('resume: {} parameters missing from model.pdparams (e.g. {}), {} unexpected keys (e.g. {})'
                       .format(len(missing), missing[:2], len(unexpected), unexpected[:2]))
    optimizer.set_state_dict(load(os.path.join(resume_model, 'model.pdopt')))
    return int(resume_model.split('_')[-1])


def worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)


def save_array(save_path, save_content, form=('npy', ), image_infor=None):
    """Save the first predictions of an evaluation run (reference utils.py:205-256); only
    the .npy form is built (NIfTI needs SimpleITK, absent here)."""
    os.makedirs(os.path.dirname(os.path.abspath(save_path)) or ".", exist_ok=True)
    for key, val in save_content.items():
        np.save('{}_{}.npy'.format(save_path, key), np.asarray(val))
