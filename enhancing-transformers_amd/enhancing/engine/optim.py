"""Optimizer facade over the engine's flat buffers (what ``ViTVQ.configure_optimizers`` returns in place of
``torch.optim.AdamW``, reference vitvqgan.py:160)."""
from __future__ import annotations


class FusedAdamW:
    def __init__(self, engine, lr: float, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 1e-4) -> None:
        self.engine = engine
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.grad_scale = 1.0  # 1 / accumulate_grad_batches

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.engine.store.zero_grad()

    def step(self) -> None:
        g = self.param_groups[0]
        self.engine.optimizer_step(g["lr"], g["betas"], g["eps"], g["weight_decay"], self.grad_scale)

    def state_dict(self) -> dict:
        s = self.engine.store
        return dict(step=s.step_count, m=s.m.detach().cpu(), v=s.v.detach().cpu(), param_groups=self.param_groups)

    def load_state_dict(self, sd: dict) -> None:
        s = self.engine.store
        s.step_count = int(sd["step"])
        s.m.copy_(sd["m"]); s.v.copy_(sd["v"])
        self.param_groups = sd["param_groups"]


class LossScaler:
    """torch.cuda.amp.GradScaler's state machine with every quantity on the device (the same three launches the fp16 engine uses: enh_nonfinite_flag,
    enh_adamw_step's skip / unscale operands, enh_loss_scale_update), for a ParamStore whose backward runs through plain autograd — the discriminator's.
    `enabled` is decided by the forward that builds the graph (fp16 operands in the loss networks: gradients of ~1e-5 would sit in fp16's subnormal range);
    disabled it is the identity everywhere.  Initial scale ENH_LOSS_NET_SCALE (default 2^12: a gradient of 1 per logit, times the scale, has to fit fp16
    itself — GradScaler's 2^16 would spend its first steps backing off), growth x2 after `interval` clean steps, x0.5 on an inf / nan (that step is skipped)."""

    def __init__(self, device, init_scale: float = None, growth_interval: int = None) -> None:
        import os
        import torch
        init_scale = float(os.environ.get("ENH_LOSS_NET_SCALE", 4096.0)) if init_scale is None else float(init_scale)
        self.growth_interval = int(os.environ.get("ENH_LOSS_SCALE_INTERVAL", 2000)) if growth_interval is None else int(growth_interval)
        self.scale_t = torch.full((1,), init_scale, dtype=torch.float32, device=device)
        self.tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=device)
        self.enabled = False

    def scale(self, t):
        return t * self.scale_t.view(()) if self.enabled else t

    def unscale(self, t):
        return t / self.scale_t.view(()) if self.enabled else t

    @property
    def value(self) -> float:
        return float(self.scale_t.item()) if self.enabled else 1.0


class FlatAdamW:
    """The same fused AdamW launch over any ``ParamStore`` (here: the discriminator's, the second optimizer of reference
    vitvqgan.py:163-164).  Under DDP the gradient buckets are all-reduced behind the discriminator's own backward (``attach_sync``: an
    ``AutogradGradSync`` whose hooks fire as autograd finishes each bucket; ``step`` only waits for what is still in flight) — a process group
    without an attached sync falls back to ONE blocking all-reduce of the flat buffer right before the step."""

    def __init__(self, store, lr: float, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 1e-4) -> None:
        self.store = store
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.grad_scale = 1.0
        self.comm = None

    def attach_sync(self, module, **kw):
        """bucketed asynchronous gradient all-reduce for this store, driven by autograd hooks on `module`'s parameters (engine/ddp.py)"""
        from .ddp import AutogradGradSync
        # one sync per store: a previous attach (Trainer.fit calls configure_optimizers + attach_sync on every invocation) left post-accumulate-grad hooks on
        # the same parameters — still live, they would all-reduce every bucket a second time and the step would divide by the world size only once
        # (ADVICE r5); the old object's hooks are removed and its in-flight handles drained before it is replaced
        old = getattr(self.store, "_grad_sync", None) or self.comm
        if old is not None:
            old.finish()
            old.remove_hooks()
        self.comm = AutogradGradSync(self.store, list(module.named_parameters()), **kw)
        self.store._grad_sync = self.comm
        return self.comm

    def detach_sync(self) -> None:
        """remove the autograd hooks of attach_sync (end of Trainer.fit: the module may be fitted again, or used without a process group)"""
        if self.comm is not None:
            self.comm.finish()
            self.comm.remove_hooks()
            self.comm = None
            self.store._grad_sync = None

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.store.zero_grad()

    def step(self) -> None:
        import torch.distributed as dist
        from .. import _C
        s, g = self.store, self.param_groups[0]
        scale = self.grad_scale
        if self.comm is not None:
            self.comm.finish()
            scale /= self.comm.world
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(s.g)
            scale /= dist.get_world_size()
        s.step_count += 1
        sc = getattr(s, "loss_scaler", None)
        if sc is not None and sc.enabled:        # the backward ran on scale_t x the loss: inf / nan check, unscale inside the update, GradScaler.update — no host sync
            sc.found_inf.zero_()
            _C.nonfinite_flag(s.g, sc.found_inf)
            _C.adamw_step(s.p, s.g, s.m, s.v, None, s.step_count, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], scale,
                          skip_flag=sc.found_inf, loss_scale=sc.scale_t)
            _C.loss_scale_update(sc.scale_t, sc.found_inf, sc.tracker, 2.0, 0.5, sc.growth_interval)
        else:
            _C.adamw_step(s.p, s.g, s.m, s.v, None, s.step_count, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], scale)
        from ..losses.op.conv_nhwc import invalidate_packed_weights
        invalidate_packed_weights()          # the kernel wrote the weights through raw pointers: cached operand images are stale

    def state_dict(self) -> dict:
        s = self.store
        return dict(step=s.step_count, m=s.m.detach().cpu(), v=s.v.detach().cpu(), param_groups=self.param_groups)

    def load_state_dict(self, sd: dict) -> None:
        s = self.store
        s.step_count = int(sd["step"])
        s.m.copy_(sd["m"]); s.v.copy_(sd["v"])
        self.param_groups = sd["param_groups"]
