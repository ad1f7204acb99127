"""Fused bias + leaky-ReLU with the reference's API (reference enhancing/losses/op/fused_act.py:20-126): ``FusedLeakyReLU``
module, ``fused_leaky_relu(input, bias, negative_slope, scale)``, first AND second derivative (the R1 penalty differentiates
through the discriminator's backward, vqperceptual.py:157-162).  The arithmetic is ``enh_fused_bias_act`` (gfx950 HIP);
unlike the reference there is no JIT build at import and no CPU branch."""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from ... import _C


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = _C.fused_bias_act(grad_output.contiguous(), None, out, 1, negative_slope, scale)
        grad_bias = _C.channel_sum(grad_input) if bias else grad_output.new_empty(0)
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gg_bias = gradgrad_bias.contiguous() if gradgrad_bias is not None and gradgrad_bias.numel() else None
        gradgrad_out = _C.fused_bias_act(gradgrad_input.contiguous(), gg_bias, out, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        ctx.bias = bias is not None
        out = _C.fused_bias_act(input.contiguous(), bias.contiguous() if bias is not None else None, None, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.bias, ctx.negative_slope, ctx.scale)
        return grad_input, (grad_bias if ctx.bias else None), None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
