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
    para = load(os.path.join(resume_model, 'model.pdparams'))
    missing, unexpected = model.set_state_dict(para)
    if missing or unexpected:
        logger.warning('resume: {} parameters missing from model.pdparams (e.g. {}), {} unexpected keys (e.g. {})'
                       .format(len(missing), missing[:2], len(unexpected), unexpected[:2]))
    # a Paddle-written model.pdopt names its accumulators after Paddle's INTERNAL parameter names; the table that maps the
    # structured names onto them travels in model.pdparams
    name_map = para.get("StructuredToParameterName@@") if isinstance(para, dict) else None
    opt_state = load(os.path.join(resume_model, 'model.pdopt'))
    try:
        optimizer.set_state_dict(opt_state, name_map=name_map)
    except TypeError:      # an optimizer without accumulator names (no name_map argument)
        optimizer.set_state_dict(opt_state)
    return int(resume_model.split('_')[-1])


def worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)


_NIFTI_DTYPES = {"uint8": (2, 8), "int16": (4, 16), "int32": (8, 32), "float32": (16, 32), "float64": (64, 64),
                 "int8": (256, 8), "uint16": (512, 16), "uint32": (768, 32), "int64": (1024, 64), "uint64": (1280, 64)}


def write_nifti(path, arr_zyx, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
    """Minimal single-file NIfTI-1 writer (.nii / .nii.gz) for what the reference hands to SimpleITK
    (utils/utils.py:237-249: GetImageFromArray(val) + SetSpacing / SetOrigin / SetDirection + WriteImage): `arr_zyx` is
    indexed [z, y, x] like a SimpleITK array, spacing / origin / direction are ITK's (x, y, z order, LPS frame).  The sform /
    qform hold the same affine in NIfTI's RAS frame (x and y rows negated), as ITK's NIfTI writer stores it."""
    import gzip
    import struct
    a = np.ascontiguousarray(arr_zyx)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype.name not in _NIFTI_DTYPES:
        a = a.astype(np.float32)
    if a.ndim != 3:
        raise ValueError("write_nifti expects a 3-D array, got shape {}".format(a.shape))
    code, bits = _NIFTI_DTYPES[a.dtype.name]
    nz, ny, nx = a.shape
    sp = [float(v) for v in spacing][:3]
    D = np.asarray(direction, dtype=np.float64).reshape(3, 3)
    A = D * np.asarray(sp)[None, :]                       # LPS voxel -> world
    flip = np.diag([-1.0, -1.0, 1.0])
    R = flip @ A                                          # RAS
    t = flip @ np.asarray([float(v) for v in origin][:3])
    # quaternion of the rotation part (qfac = -1 when the matrix is a reflection)
    Rn = R / np.maximum(np.linalg.norm(R, axis=0, keepdims=True), 1e-30)
    qfac = 1.0
    if np.linalg.det(Rn) < 0:
        Rn[:, 2] = -Rn[:, 2]
        qfac = -1.0
    tr = Rn[0, 0] + Rn[1, 1] + Rn[2, 2]
    if tr > 0:
        qa = 0.5 * np.sqrt(1.0 + tr)
        qb, qc, qd = (Rn[2, 1] - Rn[1, 2]) / (4 * qa), (Rn[0, 2] - Rn[2, 0]) / (4 * qa), (Rn[1, 0] - Rn[0, 1]) / (4 * qa)
    else:
        i = int(np.argmax([Rn[0, 0], Rn[1, 1], Rn[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        q = np.zeros(4)
        q[i + 1] = 0.5 * np.sqrt(max(1.0 + Rn[i, i] - Rn[j, j] - Rn[k, k], 0.0))
        q[0] = (Rn[k, j] - Rn[j, k]) / (4 * q[i + 1])
        q[j + 1] = (Rn[j, i] + Rn[i, j]) / (4 * q[i + 1])
        q[k + 1] = (Rn[k, i] + Rn[i, k]) / (4 * q[i + 1])
        if q[0] < 0:
            q = -q
        qa, qb, qc, qd = q
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)                 # dim
    struct.pack_into("<hh", hdr, 70, code, bits)                               # datatype, bitpix
    struct.pack_into("<8f", hdr, 76, qfac, sp[0], sp[1], sp[2], 0.0, 0.0, 0.0, 0.0)   # pixdim
    struct.pack_into("<f", hdr, 108, 352.0)                                    # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)                                      # scl_slope
    hdr[123] = 2                                                               # xyzt_units: mm
    struct.pack_into("<hh", hdr, 252, 1, 1)                                    # qform_code, sform_code (scanner)
    struct.pack_into("<6f", hdr, 256, qb, qc, qd, t[0], t[1], t[2])
    for r in range(3):
        struct.pack_into("<4f", hdr, 280 + 16 * r, R[r, 0], R[r, 1], R[r, 2], t[r])
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + b"\0\0\0\0" + a.astype(a.dtype.newbyteorder("<")).tobytes()   # x fastest = [z][y][x] C order
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


def save_array(save_path, save_content, form=('npy', ), image_infor=None):
    """Save the first predictions of an evaluation run (reference utils.py:205-256): 'npy', and 'nii' / 'nii.gz' through a
    built-in NIfTI-1 writer (the reference uses SimpleITK, absent here) with the reference's axis convention
    (image_infor['format'] 'xyz' arrays are transposed to zyx; spacing / origin / direction as in the dataset json)."""
    if not isinstance(save_content, dict):
        raise TypeError('The save_content need to be dict which the key is the save name and the value is the numpy array '
                        'to be saved, but recieved {}'.format(type(save_content)))
    content = {}
    for key, val in save_content.items():
        val = np.asarray(val)
        content[key] = np.squeeze(val) if val.ndim > 3 else val
    if save_path is None:
        return
    os.makedirs(os.path.dirname(os.path.abspath(save_path)) or ".", exist_ok=True)
    for suffix in form:
        if suffix == 'npy':
            for key, val in content.items():
                np.save('{}_{}.npy'.format(save_path, key), val)
        elif suffix in ('nii', 'nii.gz'):
            info = image_infor or {"format": "zyx"}
            for key, val in content.items():
                if val.ndim != 3:
                    # multi-channel image / batch > 1: the reference's SimpleITK writer takes such arrays, this NIfTI-1
                    # writer is 3-D only -- the .npy above holds the data, the evaluation loop must not die on it
                    logger.warning("save_array: {} has shape {}, not a 3-D volume; no {} written for it".format(
                        key, val.shape, suffix))
                    continue
                if info.get("format", "zyx") == "xyz":
                    val = np.transpose(val, [2, 1, 0])
                elif info.get("format", "zyx") != "zyx":
                    raise RuntimeError("the image format {} is not supported".format(info["format"]))
                write_nifti('{}_{}.{}'.format(save_path, key, suffix), val, info.get("spacing", (1.0, 1.0, 1.0)),
                            info.get("origin", (0.0, 0.0, 0.0)), info.get("direction", (1, 0, 0, 0, 1, 0, 0, 0, 1)))
        else:
            raise RuntimeError('Save format other than npy or nii/nii.gz is not supported yet.')