"""Learning-rate / temperature schedules usable as ``scheduler`` nodes of the stage-1 configs
(reference enhancing/utils/scheduler.py:13-66).  Plain Python; no shipped stage-1 config enables one."""
import math


class BaseScheduler:
    def schedule(self, n: int) -> float:
        raise NotImplementedError

    def __call__(self, n: int) -> float:
        return self.schedule(n)


class ExponentialDecayScheduler(BaseScheduler):
    """start * exp(-gamma n), floored at `end` (reference scheduler.py:26-41)."""

    def __init__(self, gamma: float, interval: int, start: float, end: float, **_):
        self.gamma, self.interval, self.start, self.end = gamma, interval, start, end

    def schedule(self, n: int) -> float:
        return max(self.start * math.exp(-self.gamma * (n // self.interval)), self.end)


class LambdaWarmUpCosineScheduler(BaseScheduler):
    """linear warm-up to `max` then cosine to `end`, returned as a multiplier of `start`
    (reference scheduler.py:44-66)."""

    def __init__(self, warmup_steps: int, max_decay_steps: int, min_: float, max_: float, start: float, **_):
        self.warmup_steps, self.max_decay_steps = warmup_steps, max_decay_steps
        self.min, self.max, self.start = min_, max_, start

    def schedule(self, n: int) -> float:
        if n < self.warmup_steps:
            lr = (self.max - self.start) / self.warmup_steps * n + self.start
        else:
            t = min((n - self.warmup_steps) / max(self.max_decay_steps - self.warmup_steps, 1), 1.0)
            lr = self.min + 0.5 * (self.max - self.min) * (1 + math.cos(t * math.pi))
        return lr / self.start
