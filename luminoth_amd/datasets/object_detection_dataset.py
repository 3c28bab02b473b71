"""`ObjectDetectionDataset` — TFRecord-backed detection dataset and the preprocess step of the inference driver
(reference: luminoth/datasets/base_dataset.py:7-76, luminoth/datasets/object_detection_dataset.py:18-234).

MI355X-first layout of the input path, replacing the TF queue runners (20 enqueue threads feeding a 100-element
RandomShuffleQueue, base_dataset.py:48-76):

  .tfrecords  --mmap + C index/CRC32C (libluminoth_io.so)-->  record payloads
              --protobuf wire decode (tfrecord.py)-->          image_raw, boxes
              --PIL decode on a small thread pool-->           uint8 HWC host array (pinned)
              --async H2D-->                                    uint8 on the device
              --lmh_resize_bilinear (flip folded in)-->         float32 (H',W',3) ready for the model

so the host never touches float pixels and the flipped image is never materialised.  Augmentation: `flip` (the Faster
R-CNN default, base_config.yml:94-98) is folded into the resize; `patch`, `resize`, `distortion`, `expand` (the SSD
defaults, utils/image.py:373-620) run on the decoded host image (luminoth_amd/utils/augment.py) for the records that draw
them, in the configured order, before the upload.  Shuffling is a seeded permutation per epoch (the reference's is a
100-record window of a queue: not reproducible, not reproduced).
"""
import io
import logging
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from luminoth_amd.datasets import tfrecord
from luminoth_amd.utils import augment
from luminoth_amd.utils.image import flip_bboxes, resize_image, resize_image_fixed

log = logging.getLogger('luminoth_amd')

KNOWN_AUGMENTATION = ('flip', 'patch', 'resize', 'distortion', 'expand')   # object_detection_dataset.py:9-15


class InvalidDataDirectory(Exception):
    """luminoth/datasets/exceptions.py"""


def decode_image(image_raw):
    """tf.image.decode_image(channels=3) -> uint8 (H,W,3)."""
    from PIL import Image
    return np.array(Image.open(io.BytesIO(image_raw)).convert('RGB'), dtype=np.uint8)   # writable copy


class ObjectDetectionDataset(object):
    def __init__(self, config, name='object_detection_dataset', **kwargs):
        ds, tr = config.dataset, config.get('train', {}) or {}
        ip = ds.image_preprocessing
        self._dataset_dir = ds.get('dir')
        self._split = ds.get('split', 'train')
        self._num_epochs = tr.get('num_epochs', 1)
        self._batch_size = int(tr.get('batch_size', 1) or 1)
        self._random_shuffle = bool(tr.get('random_shuffle', False))
        self._seed = tr.get('seed')
        self._image_min_size = ip.get('min_size')
        self._image_max_size = ip.get('max_size')
        self._fixed_resize = 'fixed_height' in ip and 'fixed_width' in ip
        if self._fixed_resize:
            self._image_fixed_height = ip.fixed_height
            self._image_fixed_width = ip.fixed_width
        self._data_augmentation = ds.get('data_augmentation') or []   # object_detection_dataset.py:65-66
        self._rng = np.random.RandomState(self._seed)
        # data parallel: the record order is permuted identically on every rank and dealt out rank::world; the
        # augmentation draws are per rank (rank 0 / single GPU keep the stream above)
        from luminoth_amd.utils.sharding import rank_world, shared_seed
        self._rank, self._world = kwargs.get('rank'), kwargs.get('world')
        if self._rank is None or self._world is None:
            self._rank, self._world = rank_world()
        self._order_rng = None
        if self._world > 1:
            s = shared_seed(self._seed)
            self._order_rng = np.random.RandomState(s)
            if self._rank > 0:
                self._rng = np.random.RandomState(((s or 0) * 1000003 + self._rank) % (2 ** 32))
        self._decode_threads = int(ds.get('decode_threads', 4))
        self._warned = set()

    # ------------------------------------------------------------ augmentation --
    def _augment_decide(self, default_prob=0.5):
        """One uniform draw per configured strategy (object_detection_dataset.py:163-186), in config order.
        Returns [(aug_type, applied, options)]."""
        plan = []
        for aug_config in self._data_augmentation:
            if len(aug_config.keys()) != 1:
                raise ValueError('Invalid data_augmentation definition: "{}"'.format(aug_config))
            aug_type = list(aug_config.keys())[0]
            if aug_type not in KNOWN_AUGMENTATION:
                if aug_type not in self._warned:
                    log.warning('Invalid data augmentation strategy "%s". Ignoring', aug_type)
                    self._warned.add(aug_type)
                continue
            opts = dict(aug_config[aug_type] or {})
            prob = float(opts.pop('prob', default_prob))
            applied = bool(self._rng.uniform() < prob)
            plan.append((aug_type, applied, opts))
        return plan

    @staticmethod
    def _fold_flips(plan, bboxes, height, width):
        """Net (flip_lr, flip_ud) of the applied flips and the boxes after each of them in order."""
        lr = ud = False
        for aug_type, applied, opts in plan:
            if aug_type == 'flip' and applied:
                l, u = bool(opts.get('left_right', True)), bool(opts.get('up_down', False))
                lr, ud = lr != l, ud != u
                if bboxes is not None:
                    bboxes = flip_bboxes(bboxes, height, width, l, u)
        return lr, ud, bboxes

    def _augment(self, image, bboxes=None, default_prob=0.5):
        """object_detection_dataset.py:141-200: (image, bboxes, [{strategy: applied}, ...]) with the image
        materialised (numpy in -> numpy out, tensor in -> tensor out)."""
        plan = self._augment_decide(default_prob)
        if torch.is_tensor(image):
            host, bboxes = self._augment_host(image.cpu().numpy(), bboxes, plan)
            image = torch.from_numpy(host).to(image.device)
        else:
            image, bboxes = self._augment_host(np.asarray(image), bboxes, plan)
        if bboxes is not None:
            bboxes = np.asarray(bboxes).astype(np.int32)
        return image, bboxes, [{t: a} for t, a, _ in plan]

    # -------------------------------------------------------------- preprocess --
    def _resize_image(self, image, bboxes=None, flip_lr=False, flip_ud=False):
        if self._fixed_resize:                                        # object_detection_dataset.py:223-232
            resized = resize_image_fixed(image, self._image_fixed_height, self._image_fixed_width, bboxes=bboxes,
                                         flip_lr=flip_lr, flip_ud=flip_ud)
        else:
            resized = resize_image(image, bboxes=bboxes, min_size=self._image_min_size,
                                   max_size=self._image_max_size, flip_lr=flip_lr, flip_ud=flip_ud)
        return resized['image'], resized.get('bboxes'), resized['scale_factor']

    def _augment_host(self, image, bboxes, plan):
        """The applied strategies of `plan`, in order, on a HOST image (numpy (H,W,3) uint8 or float32)."""
        for aug_type, applied, opts in plan:
            if not applied:
                continue
            if aug_type == 'flip':
                l, u = bool(opts.get('left_right', True)), bool(opts.get('up_down', False))
                if bboxes is not None:
                    bboxes = flip_bboxes(bboxes, image.shape[0], image.shape[1], l, u)
                image = image[::-1] if u else image
                image = image[:, ::-1] if l else image
                continue
            out = augment.AUGMENTATIONS[aug_type](image, bboxes, rng=self._rng, **opts)
            image, bboxes = out['image'], out.get('bboxes', bboxes)
        return np.ascontiguousarray(image), bboxes

    def preprocess(self, image, bboxes=None):
        """object_detection_dataset.py:71-83: augment, then resize.  Returns (image (H',W',3) float32 on the
        device, bboxes int32 or None, {'scale_factor', 'applied_augmentations'})."""
        plan = self._augment_decide()
        if any(applied and t != 'flip' for t, applied, _ in plan):
            host = image.cpu().numpy() if torch.is_tensor(image) else np.asarray(image)
            host, bboxes = self._augment_host(host, bboxes, plan)
            lr = ud = False
            image = torch.from_numpy(host)
        else:
            lr, ud, bboxes = self._fold_flips(plan, bboxes, image.shape[0], image.shape[1])
        if torch.is_tensor(image) and not image.is_cuda and torch.cuda.is_available():
            image = image.pin_memory().to(torch.device('cuda', torch.cuda.current_device()), non_blocking=True)
        image, bboxes, scale_factor = self._resize_image(image, bboxes, lr, ud)
        return image, bboxes, {'scale_factor': scale_factor,
                               'applied_augmentations': [{t: a} for t, a, _ in plan]}

    # ----------------------------------------------------------------- records --
    def split_path(self):
        return os.path.join(self._dataset_dir or '', '{}.tfrecords'.format(self._split))

    def read_record(self, payload):
        """object_detection_dataset.py:85-139 on one record payload: decoded uint8 image + (G,5) int32 boxes."""
        rec = tfrecord.decode_detection_record(payload)
        image = decode_image(rec['image_raw'])
        if image.shape[0] != rec['height'] or image.shape[1] != rec['width']:
            raise ValueError('record %s: decoded image is %sx%s, header says %sx%s' % (
                rec['filename'], image.shape[0], image.shape[1], rec['height'], rec['width']))
        return {'image': image, 'bboxes': rec['bboxes'], 'filename': rec['filename']}

    def __len__(self):
        path = self.split_path()
        if not os.path.exists(path):
            raise InvalidDataDirectory('"{}" does not exist.'.format(path))
        f = tfrecord.TFRecordFile(path, verify=False)
        n = len(f)
        f.close()
        return ((n // self._world) // self._batch_size) * int(self._num_epochs or 1)

    def __iter__(self):
        """Yields {'image': (B,H',W',3) float32 device tensor, 'bboxes': [ (G,5) float32 ], 'filename': [str],
        'scale_factor': [..]} — the keys of the reference's queue (object_detection_dataset.py:127-137)."""
        path = self.split_path()
        if not os.path.exists(path):                                  # base_dataset.py:36-39
            raise InvalidDataDirectory('"{}" does not exist.'.format(path))
        if not torch.cuda.is_available():
            from luminoth_amd import _lib
            raise _lib.LuminothHipError('the dataset pipeline resizes on a ROCm device (no CPU fallback)')
        device = torch.device('cuda', torch.cuda.current_device())
        records = tfrecord.TFRecordFile(path, verify=True)
        n = len(records)
        epochs = int(self._num_epochs) if self._num_epochs else 1
        pool = ThreadPoolExecutor(max(1, self._decode_threads))
        depth = max(2 * self._decode_threads, self._batch_size)
        try:
            order = []
            for _ in range(epochs):
                if self._world > 1:
                    from luminoth_amd.utils.sharding import shard_order
                    full = self._order_rng.permutation(n).tolist() if self._random_shuffle else range(n)
                    order.extend(shard_order(full, self._rank, self._world))
                else:
                    order.extend(self._rng.permutation(n).tolist() if self._random_shuffle else range(n))
            pending = []
            pos = 0
            batch = []
            while pos < len(order) or pending:
                while pos < len(order) and len(pending) < depth:
                    pending.append(pool.submit(self.read_record, records[order[pos]]))
                    pos += 1
                rec = pending.pop(0).result()
                image, bboxes, meta = self.preprocess(torch.from_numpy(rec['image']), rec['bboxes'])
                batch.append((image, bboxes.astype(np.float32), rec['filename'], meta['scale_factor']))
                if len(batch) == self._batch_size:
                    shapes = set(tuple(b[0].shape) for b in batch)
                    if len(shapes) != 1:
                        raise ValueError('train.batch_size > 1 needs equally sized images (use fixed_height/'
                                         'fixed_width); got %s' % sorted(shapes))
                    yield {'image': torch.stack([b[0] for b in batch]), 'bboxes': [b[1] for b in batch],
                           'filename': [b[2] for b in batch], 'scale_factor': [b[3] for b in batch]}
                    batch = []
        finally:
            pool.shutdown(wait=False)
            records.close()
