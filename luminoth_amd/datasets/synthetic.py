"""Synthetic object-detection dataset: the shapes of SURVEY.md §8d (U[0,255) images, G boxes per image with
sizes U{32..512}, random labels), generated on the host with a seeded generator and batched.

Yields the same record keys as the reference's dataset dict (`luminoth/datasets/object_detection_dataset.py`:
`image`, `bboxes`, `filename`) — `image` (B,H,W,3) fp32, `bboxes` a list of (G,5) arrays."""
import numpy as np
import torch


class SyntheticObjectDetectionDataset(object):
    def __init__(self, config):
        ds = config.dataset
        self.batch_size = int(config.train.get('batch_size', 1) or 1)
        self.num_classes = int(config.model.network.num_classes)
        ip = ds.get('image_preprocessing', {}) or {}
        self.height = int(ip.get('fixed_height') or ds.get('height') or ip.get('max_size') or 1024)
        self.width = int(ip.get('fixed_width') or ds.get('width') or ip.get('max_size') or 1024)
        self.boxes_per_image = int(ds.get('boxes_per_image', 8))
        self.num_images = int(ds.get('num_images', 64))
        self.num_epochs = int(config.train.get('num_epochs', 1) or 1)
        self.seed = int(config.train.get('seed') or 0)
        from luminoth_amd.utils.sharding import rank_world
        self.rank, self.world = rank_world()

    def __len__(self):
        return ((self.num_images // self.batch_size) // self.world) * self.num_epochs

    def __iter__(self):
        for epoch in range(self.num_epochs):
            g = torch.Generator().manual_seed(self.seed)          # same images every epoch
            rs = np.random.RandomState(self.seed)
            nb = self.num_images // self.batch_size
            for i in range(nb):
                mine = i % self.world == self.rank and i // self.world < nb // self.world   # batches dealt rank::world
                H, W, G = self.height, self.width, self.boxes_per_image
                image = torch.rand((self.batch_size, H, W, 3), generator=g) * 255.0
                boxes = []
                for _ in range(self.batch_size):
                    side = min(H, W)                      # box sides between a quarter (<= 32) and half (<= 512) of the short side
                    lo = max(1, min(32, side // 4))
                    wh = rs.randint(lo, max(min(512, side // 2), lo + 1), size=(G, 2))
                    xy = np.stack([rs.randint(0, W - wh[:, 0]), rs.randint(0, H - wh[:, 1])], 1)
                    lab = rs.randint(0, self.num_classes, size=(G, 1))
                    boxes.append(np.concatenate([xy, xy + wh - 1, lab], 1).astype(np.float32))
                if not mine:
                    continue
                yield {'image': image, 'bboxes': boxes,
                       'filename': ['synthetic_%d_%d' % (epoch, i * self.batch_size + b) for b in range(self.batch_size)]}
