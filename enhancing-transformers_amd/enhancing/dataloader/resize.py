"""Device-side ``Resize`` of the input pipeline: ``T.Resize`` of the reference's dataset transforms (reference enhancing/dataloader/imagenet.py:31,49), i.e.
``PIL.Image.resize(size, BILINEAR)`` on 8-bit RGB, done by ``enh_resize_u8`` (csrc/resize.hip) for a whole batch of decoded images of ragged sizes, bit-exact
with Pillow.  The DataLoader workers then only DECODE.

Host part (this file): torchvision's output-size rule and Pillow's coefficient tables — per output index a support window and 22-bit fixed-point triangle
weights with the antialiasing filter scale max(in / out, 1) (Pillow src/libImaging/Resample.c ``precompute_coeffs`` / ``normalize_coeffs_8bpc``) — computed
in double exactly as Pillow does, vectorised with numpy and cached per (in, out) pair.  The device does the two integer passes."""
from __future__ import annotations

import functools
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

_PRECISION_BITS = 32 - 8 - 2


def output_size(w: int, h: int, size) -> Tuple[int, int]:
    """(w_out, h_out) of torchvision.transforms.Resize(size): an int sends the SHORTER side to `size` and the other to int(size * long / short); a pair is
    (h, w) exactly (the reference's validation transform, imagenet.py:44-49)"""
    if isinstance(size, (tuple, list)):
        return int(size[1]), int(size[0])
    if w <= h:
        return int(size), int(size * h / w)
    return int(size * w / h), int(size)


@functools.lru_cache(maxsize=4096)
def coeff_table(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """-> (bounds int32 [out, 2] = (first input index, tap count), weights int32 [out, ksize])"""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)            # C's (int) truncates toward zero; the argument is >= -0.5 -> 0 either way
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    a = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fs))
    w = np.where((a < 1.0) & (x < xmax[:, None]), 1.0 - a, 0.0)
    # Pillow accumulates ww in tap order; numpy's pairwise row sum can differ in the last bit of ww, which moves a 22-bit weight by at most one
    # unit only when k * 2^22 sits within 1e-9 of a half — so the sum is taken sequentially, as Pillow does
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, w * (1 << _PRECISION_BITS) - 0.5, w * (1 << _PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, fixed


def build_tables(in_sizes: Sequence[Tuple[int, int]], out_sizes: Sequence[Tuple[int, int]]):
    """per image (h_in, w_in) -> (h_out, w_out): the metadata rows of enh_resize_u8 (include/enh_hip.h) and the concatenated bounds / weights arrays"""
    meta, bnd, wts = [], [], []
    boff = woff = 0
    for (hin, win), (hout, wout) in zip(in_sizes, out_sizes):
        hb, hk = coeff_table(win, wout)
        vb, vk = coeff_table(hin, hout)
        row = [hin, win, hout, wout, boff, woff, hk.shape[1]]
        bnd.append(hb.reshape(-1)); wts.append(hk.reshape(-1))
        boff += hb.size; woff += hk.size
        row += [boff, woff, vk.shape[1]]
        bnd.append(vb.reshape(-1)); wts.append(vk.reshape(-1))
        boff += vb.size; woff += vk.size
        meta.append(row)
    return (torch.tensor(meta, dtype=torch.int32), torch.from_numpy(np.concatenate(bnd).astype(np.int32)), torch.from_numpy(np.concatenate(wts).astype(np.int32)))


def resize_with_tables_u8(src: torch.Tensor, meta: torch.Tensor, bounds: torch.Tensor, weights: torch.Tensor, out_sizes: torch.Tensor) -> torch.Tensor:
    """the device half of resize_batch_u8 for tables that were built elsewhere (collate_u8 in a DataLoader worker): three asynchronous copies
    (from pinned memory when the loader pins) and the kernel; `out_sizes` int32 [B, 2] = (h_out, w_out) stays on the host"""
    from .. import _C
    dev = src.device
    HD, WD = (int(v) for v in out_sizes.max(dim=0).values.tolist())
    dst = torch.zeros(src.shape[0], HD, WD, 3, dtype=torch.uint8, device=dev)
    _C.resize_u8(src, meta.to(dev, non_blocking=True), bounds.to(dev, non_blocking=True), weights.to(dev, non_blocking=True), dst)
    return dst


def resize_batch_u8(src: torch.Tensor, in_sizes: Sequence[Tuple[int, int]], size) -> Tuple[torch.Tensor, List[Tuple[int, int]]]:
    """src uint8 [B, HS, WS, 3] on the device (image b occupies the top-left h_in x w_in corner of its slot) -> (uint8 [B, HD, WD, 3] with image b resized
    into the top-left corner of its slot, [(h_out, w_out)])"""
    from .. import _C
    outs = []
    for hin, win in in_sizes:
        wo, ho = output_size(win, hin, size)
        outs.append((ho, wo))
    meta, bnd, wts = build_tables(in_sizes, outs)
    HD, WD = max(o[0] for o in outs), max(o[1] for o in outs)
    dev = src.device
    dst = torch.zeros(src.shape[0], HD, WD, 3, dtype=torch.uint8, device=dev)
    _C.resize_u8(src, meta.to(dev), bnd.to(dev), wts.to(dev), dst)
    return dst, outs
