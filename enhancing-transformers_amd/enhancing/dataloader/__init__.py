"""Data module with the reference's contract (reference enhancing/dataloader/__init__.py:14-53):
``DataModuleFromConfig(batch_size, train=None, validation=None, test=None, num_workers=None)`` whose loaders
yield ``{'image': FloatTensor[B,3,H,W] in [0,1], 'class': LongTensor[B,1]}`` (imagenet.py:23,31-36)."""
from __future__ import annotations

from typing import Optional

import torch
from torch.utils.data import DataLoader, Dataset

from ..utils.general import initialize_from_config


def _collate_for(ds):
    if getattr(ds, "device_transform", False):
        from .imagenet import collate_u8
        return collate_u8
    return None


class _Mapped:
    """a DataLoader whose batches go through `fn` (the device-side transform) as they are yielded"""

    def __init__(self, loader, fn) -> None:
        self.loader, self.fn = loader, fn

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self):
        for b in self.loader:
            yield self.fn(b)


class DataModuleFromConfig:
    def __init__(self, batch_size: int, train=None, validation=None, test=None, num_workers: Optional[int] = None):
        self.batch_size = batch_size
        self.dataset_configs = dict()
        self.num_workers = num_workers if num_workers is not None else batch_size * 2
        if train is not None:
            self.dataset_configs["train"] = train
        if validation is not None:
            self.dataset_configs["validation"] = validation
        if test is not None:
            self.dataset_configs["test"] = test
        self.datasets = None
        self.rank, self.world = 0, 1
        self._samplers = {}

    def prepare_data(self):
        for cfg in self.dataset_configs.values():
            initialize_from_config(cfg)

    def setup(self, stage=None, rank: int = 0, world: int = 1):
        self.rank, self.world = rank, world
        self.datasets = {k: initialize_from_config(c) for k, c in self.dataset_configs.items()}
        for d in self.datasets.values():
            if hasattr(d, "set_shard"):
                d.set_shard(rank, world)

    def _loader(self, key: str, shuffle: bool) -> DataLoader:
        if self.datasets is None:
            self.setup()
        ds = self.datasets[key]
        workers = 0 if getattr(ds, "in_process", False) else self.num_workers
        shuffle = shuffle and not getattr(ds, "in_process", False)
        sampler = None
        if self.world > 1 and not hasattr(ds, "set_shard"):
            # data-parallel runs: every rank must see a DISJOINT slice of each epoch (Lightning's strategy="ddp" injects a DistributedSampler,
            # reference main.py:53-56); datasets that shard themselves (set_shard) were handled in setup()
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(ds, num_replicas=self.world, rank=self.rank, shuffle=shuffle)
            self._samplers[key] = sampler
        loader = DataLoader(ds, batch_size=self.batch_size, num_workers=workers, shuffle=shuffle and sampler is None, sampler=sampler,
                            collate_fn=_collate_for(ds), pin_memory=bool(getattr(ds, "device_transform", False)) and torch.cuda.is_available())
        if getattr(ds, "device_transform", False):       # crop + flip + ToTensor run on the GPU (imagenet.DeviceTransform)
            from .imagenet import DeviceTransform
            return _Mapped(loader, DeviceTransform(ds.resolution, resize=ds._resize_arg() if hasattr(ds, "_resize_arg") else None))
        return loader

    def set_epoch(self, epoch: int) -> None:
        for s in self._samplers.values():
            s.set_epoch(epoch)

    def train_dataloader(self):
        return self._loader("train", True)

    def val_dataloader(self):
        return self._loader("validation", False)

    def test_dataloader(self):
        return self._loader("test", False)
