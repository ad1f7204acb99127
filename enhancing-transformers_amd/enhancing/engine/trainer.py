"""In-repo training loop standing in for ``pl.Trainer.fit`` (reference main.py:51-61; pytorch-lightning is not
installable in this environment).  It drives the Lightning protocol of ``ViTVQ`` the way Lightning 1.5 does for
the stage-1 configs: per batch, ``training_step(batch, batch_idx, optimizer_idx)`` for each optimizer, gradient
accumulation over ``accumulate_grad_batches`` batches, ``strategy="ddp"`` = one process per GPU with gradient
all-reduce (RCCL), a ``{"state_dict": ...}`` checkpoint per epoch (general.py:49-55), scalar logs to a JSONL sink."""
from __future__ import annotations

import json
import os
import time
from typing import Optional

import torch

from .ddp import GradSync, init_process_group_from_env


class Trainer:
    def __init__(self, max_epochs: int = 100, precision: int = 32, gpus: int = 1, num_nodes: int = 1, strategy: Optional[str] = None,
                 accumulate_grad_batches: int = 1, callbacks=None, logger=None, max_steps: Optional[int] = None,
                 default_root_dir: str = "experiments", log_every_n_steps: int = 10, limit_val_batches: Optional[int] = None,
                 val_batches: Optional[int] = None) -> None:
        self.max_epochs, self.max_steps = max_epochs, max_steps
        # Lightning's `precision` (reference main.py:52): 16 = --use_amp = fp16 autocast + GradScaler -> the fp16 engine mode (fp16 MFMA operands, fp32
        # master weights / accumulation, loss-scaled backward with the inf / nan step skip: engine/stage1.py); "bf16" = Lightning's bf16 mixed precision
        # (bf16 operands, no loss scale); 32 = no AMP = the fp32 "exact" engine mode.  The model's engine precision must AGREE with it: fit() checks and raises.
        if precision not in (16, 32, "bf16", "16", "32"):
            raise ValueError(f"precision must be 16 (fp16 mixed precision), 'bf16' or 32, got {precision!r}")
        self.precision = 32 if str(precision) == "32" else ("bf16" if str(precision) == "bf16" else 16)
        self.accum = max(int(accumulate_grad_batches), 1)
        self.strategy = strategy
        self.root = default_root_dir
        self.log_every = log_every_n_steps
        if val_batches is not None:               # deprecated spelling of limit_val_batches (kept so existing callers do not silently validate on everything)
            import warnings
            warnings.warn("Trainer(val_batches=...) is deprecated: use limit_val_batches", DeprecationWarning, stacklevel=2)
            limit_val_batches = val_batches if limit_val_batches is None else limit_val_batches
        self.val_batches = limit_val_batches      # Lightning's limit_val_batches as a batch count; None = the whole validation set (Lightning's default)
        self.rank, self.local_rank, self.world = 0, 0, 1
        self.global_step = 0
        self.current_epoch = 0
        self.callbacks = list(callbacks or [])      # objects with Lightning's hook names (enhancing/utils/callback.py); missing hooks are skipped

    def _hook(self, name: str, *args) -> None:
        for cb in self.callbacks:
            fn = getattr(cb, name, None)
            if callable(fn):
                fn(self, *args)

    def _log(self, rec: dict) -> None:
        if self.rank != 0:
            return
        os.makedirs(self.root, exist_ok=True)
        with open(os.path.join(self.root, "metrics.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")

    def validate(self, model, data) -> dict:
        """One pass over the validation loader.  Lightning's on_epoch=True aggregation (reference vitvqgan.py:137-148): every logged val/* scalar is
        averaged over the batches, weighted by batch size; the two keys logged with sync_dist=True are already cross-rank means per batch (ViTVQ.log),
        so with the DistributedSampler's equal shards their epoch value is the mean over the WHOLE validation set on every rank; the others stay
        rank-local and rank 0's are written, as Lightning does for sync_dist=False."""
        sums, weight = {}, 0
        for vi, vb in enumerate(data.val_dataloader()):
            if self.val_batches is not None and vi >= self.val_batches:
                break
            model.validation_step(vb, vi)
            self._hook("on_validation_batch_end", model, None, vb, 0, vi)
            n = int(vb["image"].shape[0]) if isinstance(vb, dict) and "image" in vb else 1
            weight += n
            for k, v in model.logged.items():
                if k.startswith("val/"):
                    sums[k] = sums.get(k, 0.0) + float(v) * n
        out = {k: v / max(weight, 1) for k, v in sums.items()}
        out["val_images_per_rank"] = weight
        return out

    def fit(self, model, data) -> None:
        self.rank, self.local_rank, self.world = init_process_group_from_env()
        torch.cuda.set_device(self.local_rank)
        want = {16: "fp16", "bf16": "bf16", 32: "fp32"}[self.precision]
        if getattr(model, "_engine", None) is None and hasattr(model, "precision") and model.precision is None:
            model.precision = want          # engine not bound yet: bind it in the requested mode
        eng = model.engine
        if getattr(eng, "precision", want) != want:
            raise RuntimeError(f"Trainer(precision={self.precision}) asks for the {want} engine mode but the model is bound to {eng.precision}")
        if self.world > 1:
            eng.comm = GradSync(eng.store)
            eng.comm.broadcast_parameters(0)
            eng.store.refresh_shadows()
        data.setup(rank=self.rank, world=self.world)
        opts, _scheds = model.configure_optimizers()   # [autoencoder] or [autoencoder, discriminator]
        opt = opts[0]
        for o in opts:
            o.grad_scale = 1.0 / self.accum
        sched = _scheds[0]["scheduler"] if _scheds else None
        base_lr = opt.param_groups[0]["lr"]
        if self.world > 1 and len(opts) > 1:
            import torch.distributed as dist
            dist.broadcast(opts[1].store.p, 0)
            opts[1].attach_sync(model.loss.discriminator)      # bucketed all-reduce behind the discriminator's autograd backward
        t0, seen = time.time(), 0
        self._hook("on_pretrain_routine_start", model)
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            loader = data.train_dataloader()
            if hasattr(data, "set_epoch"):
                data.set_epoch(epoch)          # DistributedSampler reshuffle (what Lightning does for strategy="ddp")
            n_batches = len(loader) if hasattr(loader, "__len__") else None
            for batch_idx, batch in enumerate(loader):
                first = batch_idx % self.accum == 0
                # Lightning also steps on the LAST batch of an epoch when the window is incomplete (a trailing partial window is not dropped)
                last = (batch_idx + 1) % self.accum == 0 or (n_batches is not None and batch_idx + 1 == n_batches)
                # Lightning 1.5 order: per optimizer, training_step -> backward -> step, so the discriminator step sees the updated autoencoder
                eng.sync_grads = last   # accumulate locally, all-reduce the window's sum once (what DDP's no_sync gives Lightning)
                for o in opts[1:]:
                    if getattr(o, "comm", None) is not None:
                        o.comm.enabled = last
                for oi, o in enumerate(opts):
                    model.training_step(batch, batch_idx, oi, zero_grad=first)
                    if last:
                        if sched is not None:
                            # LambdaLR(lr_lambda=scheduler.schedule), vitvqgan.py:172 (a bare callable is taken as the lr_lambda itself)
                            mult = sched.schedule(self.global_step) if hasattr(sched, "schedule") else sched(self.global_step)
                            o.param_groups[0]["lr"] = base_lr * mult
                        o.step()
                seen += batch["image"].shape[0] * self.world
                self._hook("on_train_batch_end", model, None, batch, batch_idx)
                if last:
                    self.global_step += 1
                    model.global_step = self.global_step
                    if self.global_step % self.log_every == 0:
                        rec = {k: float(v) for k, v in model.logged.items() if k.startswith("train/")}
                        rec.update(step=self.global_step, epoch=epoch, images_per_s=seen / (time.time() - t0))
                        self._log(rec)
                        if self.rank == 0:
                            print(json.dumps(rec), flush=True)
                    if self.max_steps is not None and self.global_step >= self.max_steps:
                        break
            if "validation" in data.dataset_configs:
                self._log(self.validate(model, data) | {"epoch": epoch})
            if self.rank == 0:
                ck = os.path.join(self.root, "ckpt")
                os.makedirs(ck, exist_ok=True)
                torch.save({"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "epoch": epoch,
                            "global_step": self.global_step, "optimizer": opt.state_dict(), "optimizer_states": [o.state_dict() for o in opts]}, os.path.join(ck, f"epoch={epoch:02d}.ckpt"))
            if self.max_steps is not None and self.global_step >= self.max_steps:
                break
        for o in opts:      # the autograd hooks of attach_sync must not outlive this fit (a second fit would otherwise reduce every bucket twice: ADVICE r5)
            if hasattr(o, "detach_sync"):
                o.detach_sync()
