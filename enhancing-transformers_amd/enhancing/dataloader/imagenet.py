"""ImageNet train / validation datasets with the reference's class names, arguments and sample contract
(reference enhancing/dataloader/imagenet.py:15-54): ``{'image': float [3,R,R] in [0,1], 'class': [1]}``,
Resize(R) + RandomCrop(R) + RandomHorizontalFlip for train, Resize(R) + CenterCrop(R) for validation, no
mean/std normalisation.  torchvision is not available here, so the folder walk and the transforms are a small
PIL / numpy implementation; if ``root`` does not exist the constructor raises (use
``enhancing.dataloader.synthetic.SyntheticImages`` for synthetic runs — bench.py and the shipped yaml do).

Device-side tail (``device_transform=True``): the worker processes only decode and resize (PIL) and hand over uint8 pixels plus the crop window
and flip decision they drew; ``DeviceTransform`` then does crop + flip + ToTensor for the whole batch in one HIP kernel (``enh_crop_flip_u8``) and
yields the same ``{'image', 'class'}`` batch — a quarter of the host-to-device bytes (uint8 instead of float32) and no per-sample float work on the
host.  With ``device_resize=True`` as well the workers ONLY DECODE: the antialiased bilinear resize runs on the device too (``enh_resize_u8``,
csrc/resize.hip: Pillow's two integer passes, bit-exact).  The random numbers are drawn in the same order as on the host path, so all three paths
give bit-identical batches for the same seed (tests/test_host_cpu.py, tests/test_ops_gpu.py)."""
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

    def __init__(self, root: str, resolution: int = 256, resize_ratio: float = 0.75, device_transform: bool = False, device_resize: bool = False,
                 host_prereduce: int = 0) -> None:
        folder = os.path.join(root, self.split)
        if not os.path.isdir(folder):
            raise FileNotFoundError(f"{folder} not found; for synthetic data use enhancing.dataloader.synthetic.SyntheticImages")
        self.resolution = resolution
        self.device_transform = device_transform or device_resize
        self.device_resize = device_resize
        # device_resize only: bound the slot a decoded outlier (ImageNet has > 4000 px images) takes in the batch.  0 = off (bit-exact with the host
        # path).  k > 0: an image whose SHORTER side is >= 2k is shrunk in the worker by an integer factor (JPEG DCT scaling via draft(), then
        # Image.reduce's box filter) so that the shorter side lands in [k, 2k) before the device resize — an approximation of the reference's single
        # full-size Pillow resize, so opt-in; k should be a few times `resolution`.
        self.host_prereduce = int(host_prereduce)
        self.paths, self.labels = _index(folder)

    def _resize_arg(self):
        """what the reference hands to T.Resize: the int for training (shorter side -> R, imagenet.py:31), the pair (R, R) for validation (imagenet.py:44-49)"""
        return self.resolution if self.split == "train" else (self.resolution, self.resolution)

    def __len__(self) -> int:
        return len(self.paths)

    def _decode_resize(self, path: str, resize: bool = True):
        """PIL decode (+ T.Resize's PIL.Image.resize(size, BILINEAR) unless the device does it); then the crop window (in RESIZED coordinates) and the
        flip decision, drawn in the order the host path uses them"""
        from PIL import Image
        from .resize import output_size
        r = self.resolution
        train = self.split == "train"
        im = Image.open(path)
        k = self.host_prereduce
        if k > 0 and not resize and min(im.size) >= 2 * k:
            im.draft("RGB", (k, k))                      # JPEG only: the decoder itself scales by 1/2, 1/4, 1/8 keeping both sides >= k
            im = im.convert("RGB")
            if min(im.size) >= 2 * k:
                im = im.reduce(min(im.size) // k)
        else:
            im = im.convert("RGB")
        w, h = output_size(im.size[0], im.size[1], self._resize_arg())      # torchvision's rule: int(size * long / short) for the longer side
        if resize:
            im = im.resize((w, h), Image.BILINEAR)
        if train:
            x0, y0 = np.random.randint(0, w - r + 1), np.random.randint(0, h - r + 1)
        else:
            x0, y0 = (w - r) // 2, (h - r) // 2
        flip = bool(train and np.random.rand() < 0.5)
        return np.array(im, dtype=np.uint8), y0, x0, flip, (h, w)

    def _load(self, path: str) -> torch.Tensor:
        r = self.resolution
        px, y0, x0, flip, _ = self._decode_resize(path)
        a = px[y0:y0 + r, x0:x0 + r].astype(np.float32) / 255.0
        if flip:
            a = a[:, ::-1]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    def __getitem__(self, i: int):
        label = torch.tensor([self.labels[i]])
        if self.device_transform:
            px, y0, x0, flip, (h, w) = self._decode_resize(self.paths[i], resize=not self.device_resize)
            out = {"pixels_u8": torch.from_numpy(px), "window": torch.tensor([y0, x0, int(flip)], dtype=torch.int32), "class": label}
            if self.device_resize:      # the decoded (un-resized) and the target size travel with the pixels: the collate pads to a common slot
                out["in_size"] = torch.tensor([px.shape[0], px.shape[1]], dtype=torch.int32)
                out["out_size"] = torch.tensor([h, w], dtype=torch.int32)
            return out
        return {"image": self._load(self.paths[i]), "class": label}


def collate_u8(samples):
    """device_transform samples have different sizes: each goes into the top-left corner of a common [Hmax, Wmax, 3] slot.  For un-resized samples
    (device_resize) the Pillow coefficient tables of enh_resize_u8 are built HERE, i.e. in the DataLoader worker that collates the batch, and travel as
    three int32 tensors ("resize_meta" / "resize_bounds" / "resize_weights") — the consumer process only copies them to the device."""
    H, W = max(s["pixels_u8"].shape[0] for s in samples), max(s["pixels_u8"].shape[1] for s in samples)
    px = torch.zeros(len(samples), H, W, 3, dtype=torch.uint8)
    for b, s in enumerate(samples):
        h, w, _ = s["pixels_u8"].shape
        px[b, :h, :w] = s["pixels_u8"]
    out = {"pixels_u8": px, "window": torch.stack([s["window"] for s in samples]), "class": torch.stack([s["class"] for s in samples])}
    if "in_size" in samples[0]:
        out["in_size"] = torch.stack([s["in_size"] for s in samples])
        if "out_size" in samples[0]:
            from .resize import build_tables
            out["out_size"] = torch.stack([s["out_size"] for s in samples])
            out["resize_meta"], out["resize_bounds"], out["resize_weights"] = build_tables(
                [tuple(v) for v in out["in_size"].tolist()], [tuple(v) for v in out["out_size"].tolist()])
    return out


class DeviceTransform:
    """(resize +) crop + flip + ToTensor on the device for a collate_u8 batch -> the reference's batch contract {'image': float [B,3,R,R] in [0,1], 'class': [B,1]}.
    `resize`: what the dataset hands to T.Resize (int: shorter side; pair: exact), used when the batch carries un-resized pixels ("in_size")."""

    def __init__(self, resolution: int, device=None, resize=None) -> None:
        self.resolution, self.device = resolution, device
        self.resize = resolution if resize is None else resize

    def __call__(self, batch):
        if "pixels_u8" not in batch:
            return batch
        from .. import _C
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        px = batch["pixels_u8"].to(dev, non_blocking=True)
        if "resize_meta" in batch:           # tables built by the worker (collate_u8); pinned by the DataLoader when pin_memory is on
            from .resize import resize_with_tables_u8
            px = resize_with_tables_u8(px, batch["resize_meta"], batch["resize_bounds"], batch["resize_weights"], batch["out_size"])
        elif "in_size" in batch:             # a hand-made batch without tables: build them here
            from .resize import resize_batch_u8
            px, _ = resize_batch_u8(px, [(int(h), int(w)) for h, w in batch["in_size"].tolist()], self.resize)
        img = _C.crop_flip_u8(px, batch["window"].to(dev, non_blocking=True).contiguous(), self.resolution)
        return {"image": img, "class": batch["class"]}


class ImageNetTrain(_ImageNetBase):
    split = "train"


class ImageNetValidation(_ImageNetBase):
    split = "val"
