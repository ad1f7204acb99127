"""``FusedLeakyReLU`` / ``fused_leaky_relu(input, bias, negative_slope, scale)`` — the reference's native-op entry point
(reference enhancing/losses/op/fused_act.py:79-126) on ``enh_fused_bias_act``, differentiable to any order; no JIT build, no CPU branch.

Derivation.  y = scale * lrelu(x + b).  Away from the kink the derivative is the diagonal GATE  G(v) = scale * v * (y > 0 ? 1 : slope), read off the
saved OUTPUT (sign(y) = sign(x + b)).  G is linear in v and its own adjoint, and the gate pattern carries no gradient (piecewise constant), so
    dx = G(dy)        db = sum over every axis but the channel axis of dx
and every higher derivative is G again: one self-adjoint node ``_Gate`` serves the first backward and the backward of the backward (the R1 penalty,
vqperceptual.py:157-162).  The bias gradient is a channel reduction (``enh_channel_sum_f32``) whose adjoint is a broadcast.
"""
from __future__ import annotations

import torch
from torch import nn

from ... import _C


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, y, slope: float, scale: float):
        ctx.save_for_backward(y)
        ctx.cfg = (slope, scale)
        return _C.fused_bias_act(v.contiguous(), None, y, 1, slope, scale)

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        return _Gate.apply(g, y, *ctx.cfg), None, None, None


class _ChannelSum(torch.autograd.Function):
    """[B, C, ...] -> [C]"""

    @staticmethod
    def forward(ctx, v):
        ctx.shape = v.shape
        return _C.channel_sum(v.contiguous())

    @staticmethod
    def backward(ctx, g):
        shp = ctx.shape
        return g.view(1, -1, *([1] * (len(shp) - 2))).expand(shp)


class _BiasLeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope: float, scale: float):
        y = _C.fused_bias_act(x.contiguous(), None if bias is None else bias.contiguous(), None, 0, slope, scale)
        ctx.save_for_backward(y)
        ctx.cfg = (slope, scale, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        slope, scale, has_bias = ctx.cfg
        gx = _Gate.apply(gy, y, slope, scale)
        return gx, (_ChannelSum.apply(gx) if has_bias and ctx.needs_input_grad[1] else None), None, None


def fused_leaky_relu(input: torch.Tensor, bias=None, negative_slope: float = 0.2, scale: float = 2 ** 0.5) -> torch.Tensor:
    return _BiasLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel: int, bias: bool = True, negative_slope: float = 0.2, scale: float = 2 ** 0.5) -> None:
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
