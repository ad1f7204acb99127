"""ImageNet train / validation datasets with the reference's class names, arguments and sample contract
(reference enhancing/dataloader/imagenet.py:15-54): ``{'image': float [3,R,R] in [0,1], 'class': [1]}``,
Resize(R) + RandomCrop(R) + RandomHorizontalFlip for train, Resize(R) + CenterCrop(R) for validation, no
mean/std normalisation.  torchvision is not available here, so the folder walk and the transforms are a small
PIL / numpy implementation; if ``root`` does not exist the constructor raises (use
``enhancing.dataloader.synthetic.SyntheticImages`` for synthetic runs — bench.py and the shipped yaml do)."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

_EXT = (".jpeg", ".jpg", ".png", ".bmp")


def _index(root: str) -> Tuple[List[str], List[int]]:
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    paths, labels = [], []
    for ci, c in enumerate(classes):
        for dp, _, files in sorted(os.walk(os.path.join(root, c))):
            for f in sorted(files):
                if f.lower().endswith(_EXT):
                    paths.append(os.path.join(dp, f)); labels.append(ci)
    return paths, labels


class _ImageNetBase(Dataset):
    split = "train"

    def __init__(self, root: str, resolution: int = 256, resize_ratio: float = 0.75) -> None:
        folder = os.path.join(root, self.split)
        if not os.path.isdir(folder):
            raise FileNotFoundError(f"{folder} not found; for synthetic data use enhancing.dataloader.synthetic.SyntheticImages")
        self.resolution = resolution
        self.paths, self.labels = _index(folder)

    def __len__(self) -> int:
        return len(self.paths)

    def _load(self, path: str, train: bool) -> torch.Tensor:
        from PIL import Image
        r = self.resolution
        im = Image.open(path).convert("RGB")
        w, h = im.size
        s = r / min(w, h)
        im = im.resize((max(r, round(w * s)), max(r, round(h * s))), Image.BILINEAR)
        w, h = im.size
        if train:
            x0, y0 = np.random.randint(0, w - r + 1), np.random.randint(0, h - r + 1)
        else:
            x0, y0 = (w - r) // 2, (h - r) // 2
        a = np.asarray(im.crop((x0, y0, x0 + r, y0 + r)), dtype=np.float32) / 255.0
        if train and np.random.rand() < 0.5:
            a = a[:, ::-1]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    def __getitem__(self, i: int):
        return {"image": self._load(self.paths[i], self.split == "train"), "class": torch.tensor([self.labels[i]])}


class ImageNetTrain(_ImageNetBase):
    split = "train"


class ImageNetValidation(_ImageNetBase):
    split = "val"
