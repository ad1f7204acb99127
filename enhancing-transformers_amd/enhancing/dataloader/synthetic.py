"""Synthetic ImageNet-shaped dataset (there is no dataset and no torchvision in this environment): smooth
low-frequency colour fields plus noise in [0,1], deterministic per (seed, rank, index).  Same sample contract
as the reference's ImageNet classes (enhancing/dataloader/imagenet.py:15-23)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


class SyntheticImages(Dataset):
    in_process = True  # generate in the trainer process (no worker fork next to the HIP runtime)

    def __init__(self, resolution: int = 256, length: int = 1024, seed: int = 0, n_classes: int = 1000, smooth: bool = True):
        self.resolution, self.length, self.seed, self.n_classes, self.smooth = resolution, length, seed, n_classes, smooth
        self.rank, self.world = 0, 1

    def set_shard(self, rank: int, world: int) -> None:
        self.rank, self.world = rank, world

    def __len__(self) -> int:
        return self.length // self.world

    def __getitem__(self, i: int):
        rs = np.random.RandomState((self.seed * 1000003 + self.rank * 7919 + i) % (2 ** 31 - 1))
        r = self.resolution
        if self.smooth:
            low = torch.from_numpy(rs.uniform(0, 1, (1, 3, max(r // 16, 2), max(r // 16, 2))).astype(np.float32))
            img = F.interpolate(low, size=(r, r), mode="bilinear", align_corners=False)[0]
            img = (img + 0.05 * torch.from_numpy(rs.standard_normal((3, r, r)).astype(np.float32))).clamp_(0, 1)
        else:
            img = torch.from_numpy(rs.uniform(0, 1, (3, r, r)).astype(np.float32))
        return {"image": img, "class": torch.tensor([rs.randint(0, self.n_classes)])}
