"""upfirdn2d (upsample - FIR filter - downsample) with the reference's API (reference enhancing/losses/op/upfirdn2d.py:20-165):
``upfirdn2d(input [B,C,H,W], kernel [kh,kw], up=1, down=1, pad=(p0,p1))`` with first and second derivatives (the backward is the same
op with the flipped kernel, swapped up/down and the gradient pads, and is itself differentiable).  Arithmetic: ``enh_upfirdn2d``."""
from __future__ import annotations

from collections import abc

import torch
from torch.autograd import Function

from ... import _C


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1 = g_pad
        go = grad_output.reshape(-1, out_size[0], out_size[1]).contiguous()
        grad_input = _C.upfirdn2d(go, grad_kernel, down_x, down_y, up_x, up_y, g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        (up_x, up_y), (down_x, down_y), (px0, px1, py0, py1), in_size, out_size = ctx.cfg
        gg = gradgrad_input.reshape(-1, in_size[2], in_size[3]).contiguous()
        gradgrad_out = _C.upfirdn2d(gg, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        return gradgrad_out.view(in_size[0], in_size[1], out_size[0], out_size[1]), None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        batch, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        kernel = kernel.contiguous()
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h + down_y) // down_y
        out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w + down_x) // down_x
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
        g_pad_x0 = kernel_w - pad_x0 - 1
        g_pad_y0 = kernel_h - pad_y0 - 1
        g_pad_x1 = in_w * up_x - out_w * down_x + pad_x0 - up_x + 1
        g_pad_y1 = in_h * up_y - out_h * down_y + pad_y0 - up_y + 1
        ctx.g_pad = (g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        out = _C.upfirdn2d(input.reshape(-1, in_h, in_w).contiguous(), kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad, ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if not isinstance(up, abc.Iterable):
        up = (up, up)
    if not isinstance(down, abc.Iterable):
        down = (down, down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    return UpFirDn2d.apply(input, kernel, tuple(up), tuple(down), tuple(pad))
