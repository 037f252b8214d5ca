"""Datasets with the reference's contract (medicalseg/datasets/dataset.py:29-125):
``__getitem__ -> (im float32 [1,D,H,W], label int32 [D,H,W], path)``; attributes
num_classes, ignore_index, transforms.transforms, dataset_json_path, mode.

``SyntheticCT`` generates the benchmark volumes of SURVEY.md section 8 d1 (no dataset can be
downloaded here)."""
import os

import numpy as np

from ..cvlibs import manager
from ..transforms import Compose


def _as_item(im, label, path):
    from ..preprocess import DeviceVolume
    if isinstance(im, DeviceVolume):
        return im, label, path
    return im.astype("float32"), label.astype("int32"), path


@manager.DATASETS.add_component
class MedicalDataset:
    def __init__(self, dataset_root, result_dir, transforms, num_classes, mode='train', ignore_index=255,
                 dataset_json_path="", device_aug=False):
        self.dataset_root = dataset_root
        self.result_dir = result_dir
        # device_aug: run the transform list as HIP kernels on device volumes;
        # __getitem__ then yields device volumes that DataLoader batches without a host round trip
        self.transforms = Compose(transforms, device=device_aug)
        self.file_list = list()
        self.mode = mode.lower()
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.dataset_json_path = dataset_json_path
        if self.dataset_root is None:
            raise ValueError("The dataset is not Found or the folder structure is nonconfoumance.")
        if mode == 'train':
            file_path = os.path.join(self.dataset_root, 'train_list.txt')
        elif mode == 'val':
            file_path = os.path.join(self.dataset_root, 'val_list.txt')
        elif mode == 'test':
            file_path = os.path.join(self.dataset_root, 'test_list.txt')
        else:
            raise ValueError("`mode` should be 'train', 'val' or 'test', but got {}.".format(mode))
        with open(file_path, 'r') as f:
            for line in f:
                items = line.strip().split()
                if len(items) != 2:
                    raise Exception("File list format incorrect! It should be image_name label_name\\n")
                self.file_list.append([os.path.join(self.dataset_root, items[0]),
                                       os.path.join(self.dataset_root, items[1])])
        if mode == 'train':
            self.file_list = self.file_list * 10  # reference dataset.py:110-111

    def __getitem__(self, idx):
        image_path, label_path = self.file_list[idx]
        im, label = self.transforms(im=image_path, label=label_path)
        return _as_item(im, label, self.file_list[idx][0])

    def save_transformed(self):
        pass

    def __len__(self):
        return len(self.file_list)


@manager.DATASETS.add_component
class LungCoronavirus(MedicalDataset):
    """COVID-19 CT scans, lung + infection masks (reference datasets/lung_coronavirus.py)."""


@manager.DATASETS.add_component
class MRISpineSeg(MedicalDataset):
    """MRI spine segmentation, 20 classes (reference datasets/mri_spine_seg.py)."""


@manager.DATASETS.add_component
class SyntheticCT:
    """Deterministic synthetic CT-like volumes: raw HU = clip(N(-600, 450), -2000, 2000) ->
    HUnorm -> /max; label = background 0 plus two ellipsoids (classes 1, 2, ... cycling)."""

    def __init__(self, num_samples=8, shape=(128, 128, 128), num_classes=3, seed=1234, mode='train',
                 ignore_index=255, transforms=None, dataset_root=None, result_dir=None, dataset_json_path="",
                 device_aug=False):
        self.num_samples, self.shape = int(num_samples), tuple(int(s) for s in shape)
        self.num_classes, self.seed, self.mode = num_classes, seed, mode
        self.ignore_index = ignore_index
        self.transforms = Compose(transforms or [], device=device_aug)
        self.dataset_json_path = dataset_json_path
        self.file_list = [["synthetic_{}".format(i), ""] for i in range(self.num_samples)]
        self._cache = {}

    def __len__(self):
        return self.num_samples

    def make(self, idx):
        """Deterministic in (seed, idx): generated once and kept (the numpy generation of a 128^3 volume costs ~70 ms,
        about one training step on an MI355X, and would otherwise sit in reader_cost every epoch)."""
        cached = self._cache.get(idx)
        if cached is None:
            cached = self._cache[idx] = self._generate(idx)
        return cached[0].copy(), cached[1].copy()

    def _generate(self, idx):
        rng = np.random.default_rng(self.seed + idx)
        D, H, W = self.shape
        hu = np.clip(rng.standard_normal(self.shape, dtype=np.float32) * 450.0 - 600.0, -2000, 2000)
        zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
        label = np.zeros(self.shape, dtype=np.int32)
        for c in range(1, max(2, min(self.num_classes, 3))):
            cen = rng.uniform(0.3, 0.7, 3) * np.array(self.shape)
            rad = rng.uniform(0.12, 0.25, 3) * np.array(self.shape)
            m = ((zz - cen[0]) / rad[0]) ** 2 + ((yy - cen[1]) / rad[1]) ** 2 + ((xx - cen[2]) / rad[2]) ** 2 <= 1
            label[m] = c
            hu[m] += 300.0 * c
        im = (hu + 1200.0) / (1800.0 / 255.0)
        np.clip(im, 0, 255, out=im)
        return im.astype(np.float32), label

    def __getitem__(self, idx):
        im, label = self.make(idx)
        im, label = self.transforms(im, label)
        return _as_item(im, label, self.file_list[idx][0])


class DataLoader:
    """Minimal host loader: yields [images NCDHW float32, labels NDHW int32, paths] per batch
    for this rank's shard (replaces paddle.io.DataLoader + DistributedBatchSampler,
    core/train.py:87-95).  A one-batch-ahead prefetch thread hides host transforms."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, num_workers=0, seed=0):
        from ..parallel import ParallelEnv
        env = ParallelEnv()
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.rank, self.world, self.seed = env.rank, env.nranks, seed
        self.epoch = 0
        self.num_workers = num_workers

    def _batches(self):
        from ..parallel import shard_indices
        return shard_indices(len(self.dataset), self.batch_size, self.rank, self.world, self.shuffle, self.epoch,
                             self.seed, self.drop_last)

    def __len__(self):
        return len(self._batches())

    def _load(self, idxs):
        items = [self.dataset[i] for i in idxs]
        from ..preprocess import DeviceVolume
        if isinstance(items[0][0], DeviceVolume):
            return self._stack_device(items)
        return [np.stack([it[0] for it in items]), np.stack([it[1] for it in items]), [it[2] for it in items]]

    @staticmethod
    def _stack_device(items):
        """Device-augmented samples -> one [N,1,D,H,W] Tensor and one [N,D,H,W] IntTensor, device-to-device
        (one channel: NDHWC == NCDHW).  The batch buffers rotate over two slots, so the batch handed out
        one iteration ago stays valid while the next one is assembled."""
        from ..device import IntTensor, Tensor
        ims, labs = [it[0] for it in items], [it[1] for it in items]
        dev, shape = ims[0].dev, ims[0].shape
        if any(v.shape != shape for v in ims + labs):
            raise ValueError("device batching needs equal sample shapes, got {}".format([v.shape for v in ims]))
        n, vox = len(ims), int(np.prod(shape))
        slots = DataLoader._slots.setdefault((id(dev), n, shape), {"bufs": [], "next": 0})
        if len(slots["bufs"]) < 2:
            slots["bufs"].append((dev.malloc(n * vox * 4), dev.malloc(n * vox * 4)))
            xb, yb = slots["bufs"][-1]
        else:
            xb, yb = slots["bufs"][slots["next"]]
            slots["next"] ^= 1
        for i, (x, y) in enumerate(zip(ims, labs)):
            dev.d2d(xb + i * vox * 4, x.ptr, vox * 4)
            dev.d2d(yb + i * vox * 4, y.ptr, vox * 4)
            x.free()
            y.free()
        d, h, w = shape
        return [Tensor(dev, xb, n, d, h, w, 1, 1, None), IntTensor(dev, yb, (n, d, h, w)), [it[2] for it in items]]

    _slots = {}

    def __iter__(self):
        batches = self._batches()
        self.epoch += 1
        if self.num_workers <= 0 or getattr(getattr(self.dataset, "transforms", None), "device", False):
            for b in batches:
                yield self._load(b)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=2)
        stop = threading.Event()

        def put(item):
            # bounded put that gives up when the consumer abandoned the iterator (otherwise the thread blocks forever)
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for b in batches:
                    if not put(self._load(b)):
                        return
                put(None)
            except BaseException as e:  # a bad file / shape mismatch / transform error must reach the training loop
                put(e)

        threading.Thread(target=work, daemon=True).start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
