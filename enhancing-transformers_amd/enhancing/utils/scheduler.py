"""Learning-rate / temperature schedules usable as ``scheduler`` nodes of the stage-1 configs, with the reference's constructor names and
return conventions (reference enhancing/utils/scheduler.py:13-88):

  * ``schedule(n)`` returns a MULTIPLIER of ``start`` — it is what the reference hands to ``LambdaLR(optimizer, lr_lambda=scheduler.schedule)``
    (vitvqgan.py:166-176), and what the in-repo trainer multiplies the base learning rate with;
  * ``scheduler(n)`` (``__call__``) returns the absolute value ``schedule(n) * start`` (used for Gumbel temperatures, scheduler.py:20-23).

Plain Python; no shipped stage-1 config enables one."""
import math


class BaseScheduler:
    start: float

    def schedule(self, n: int) -> float:
        raise NotImplementedError

    def __call__(self, n: int) -> float:
        assert hasattr(self, "start")
        return self.schedule(n) * self.start


class ExponentialDecayScheduler(BaseScheduler):
    """``start * exp(-scale_factor * n)`` floored at ``end``, RE-EVALUATED only at multiples of ``decay_every_step`` and held in between
    (reference scheduler.py:26-41)."""

    def __init__(self, start: float, end: float, decay_every_step: int, scale_factor: float) -> None:
        self.decay_every_step, self.scale_factor = decay_every_step, scale_factor
        self.start, self.end = start, end
        self.current = start

    def schedule(self, n: int) -> float:
        if n % self.decay_every_step == 0:
            self.current = max(self.end, math.exp(-self.scale_factor * n) * self.start)
        return self.current / self.start


class _WarmUp(BaseScheduler):
    def __init__(self, warm_up_steps: int, max_decay_steps: int, min_: float, max_: float, start: float) -> None:
        assert max_decay_steps >= warm_up_steps
        self.warm_up_steps, self.max_decay_steps = warm_up_steps, max_decay_steps
        self.min_, self.max_, self.start = min_, max_, start
        self.last = 0.0

    def _warm(self, n: int) -> float:
        return (self.max_ - self.start) / self.warm_up_steps * n + self.start


class LambdaWarmUpCosineScheduler(_WarmUp):
    """linear warm-up from ``start`` to ``max_`` over ``warm_up_steps``, then half a cosine down to ``min_`` at ``max_decay_steps`` (scheduler.py:44-66)."""

    def schedule(self, n: int) -> float:
        if n < self.warm_up_steps:
            res = self._warm(n)
        else:
            t = min((n - self.warm_up_steps) / max(self.max_decay_steps - self.warm_up_steps, 1), 1.0)
            res = self.min_ + 0.5 * (self.max_ - self.min_) * (1 + math.cos(t * math.pi))
        self.last = res
        return res / self.start


class LambdaWarmUpLinearScheduler(_WarmUp):
    """linear warm-up, then a straight line from ``max_`` at step 0 to ``min_`` at ``max_decay_steps`` (scheduler.py:69-88; the reference's decay
    branch reads an undefined name ``max_decay_steps`` — the attribute is what it means)."""

    def schedule(self, n: int) -> float:
        if n < self.warm_up_steps:
            res = self._warm(n)
        else:
            res = self.min_ + (self.max_ - self.min_) * (self.max_decay_steps - n) / self.max_decay_steps
        self.last = res
        return res / self.start
