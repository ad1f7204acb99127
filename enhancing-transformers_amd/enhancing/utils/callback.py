"""Trainer callbacks with the reference's names and constructor arguments (reference enhancing/utils/callback.py:21-141, instantiated by
``main.py`` through ``enhancing/utils/general.py:58-74``): ``SetupCallback`` creates the log / checkpoint directories and prints the configs,
``ImageLogger`` calls the module's ``log_images`` every ``batch_frequency`` batches (and at 1, 2, 4, ... while ``increase_log_steps``) and writes one
PNG grid per logged tensor to ``<save_dir>/results/<split>/<key>_gs-<step>_e-<epoch>_b-<batch>.png``.  The wandb / test-tube sinks of the reference
need packages that are not installable here; the local PNG sink is the one every run has.  ``make_grid`` restates torchvision's default layout
(``nrow`` images per row, 2 pixels of zero padding) so that the files have the reference's geometry."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict

import numpy as np
import torch


def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2) -> torch.Tensor:
    """[N,C,H,W] -> [C, rows*(H+padding)+padding, cols*(W+padding)+padding] (torchvision.utils.make_grid defaults: pad_value 0, single-channel
    images repeated to 3 channels)"""
    if images.dim() == 3:
        images = images.unsqueeze(0)
    if images.shape[1] == 1:
        images = images.repeat(1, 3, 1, 1)
    n, c, h, w = images.shape
    if n == 1:
        return images[0]
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = images.new_zeros(c, rows * (h + padding) + padding, cols * (w + padding) + padding)
    for k in range(n):
        r, q = divmod(k, cols)
        y, x = r * (h + padding) + padding, q * (w + padding) + padding
        grid[:, y:y + h, x:x + w] = images[k]
    return grid


class SetupCallback:
    def __init__(self, config, exp_config, basedir: Path, logdir: str = "log", ckptdir: str = "ckpt") -> None:
        self.logdir, self.ckptdir = Path(basedir) / logdir, Path(basedir) / ckptdir
        self.config, self.exp_config = config, exp_config

    def on_pretrain_routine_start(self, trainer, pl_module) -> None:
        if getattr(trainer, "rank", 0) == 0:
            os.makedirs(self.logdir, exist_ok=True)
            os.makedirs(self.ckptdir, exist_ok=True)
            print("Experiment config")
            print(self.exp_config)
            print("Model config")
            print(self.config)


class ImageLogger:
    def __init__(self, batch_frequency: int, max_images: int, clamp: bool = True, increase_log_steps: bool = True) -> None:
        self.batch_freq = batch_frequency
        self.max_images = max_images
        self.log_steps = [2 ** n for n in range(int(np.log2(self.batch_freq)) + 1)] if increase_log_steps else [self.batch_freq]
        self.clamp = clamp

    def log_local(self, save_dir: str, split: str, images: Dict, global_step: int, current_epoch: int, batch_idx: int) -> None:
        from PIL import Image
        root = os.path.join(save_dir, "results", split)
        os.makedirs(root, exist_ok=True)
        for k in images:
            grid = make_grid(images[k], nrow=4)
            grid = (grid.permute(1, 2, 0).numpy() * 255).astype(np.uint8)
            Image.fromarray(grid).save(os.path.join(root, "{}_gs-{:06}_e-{:06}_b-{:06}.png".format(k, global_step, current_epoch, batch_idx)))

    def log_img(self, trainer, pl_module, batch, batch_idx: int, split: str = "train") -> None:
        if getattr(trainer, "rank", 0) != 0:
            return
        if self.check_frequency(batch_idx) and callable(getattr(pl_module, "log_images", None)) and self.max_images > 0:
            is_train = pl_module.training
            if is_train:
                pl_module.eval()
            with torch.no_grad():
                images = pl_module.log_images(batch, split=split, pl_module=pl_module)
            for k in images:
                n = min(images[k].shape[0], self.max_images)
                images[k] = images[k][:n].detach().float().cpu()
                if self.clamp:
                    images[k] = images[k].clamp(0, 1)
            self.log_local(getattr(trainer, "root", "."), split, images, getattr(pl_module, "global_step", 0), getattr(trainer, "current_epoch", 0), batch_idx)
            if is_train:
                pl_module.train()

    def check_frequency(self, batch_idx: int) -> bool:
        if (batch_idx % self.batch_freq) == 0 or (batch_idx in self.log_steps):
            try:
                self.log_steps.pop(0)
            except IndexError:
                pass
            return True
        return False

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx: int) -> None:
        self.log_img(trainer, pl_module, batch, batch_idx, split="train")

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, dataloader_idx: int, batch_idx: int) -> None:
        self.log_img(trainer, pl_module, batch, batch_idx, split="val")
