"""``upfirdn2d(input [B,C,H,W], kernel [kh,kw], up=1, down=1, pad=(p0,p1))`` — the reference's native-op entry point
(reference enhancing/losses/op/upfirdn2d.py:149-165) on ``enh_upfirdn2d``, differentiable to any order.

Derivation.  Per image plane the op is LINEAR in its input: y = D_down( F_k( P_pad( U_up(x) ) ) ) (zero-stuff, pad / crop, correlate with the flipped
kernel, decimate).  Its adjoint is an op of the same family: zero-stuffing and decimation are each other's adjoints, correlation with k becomes
correlation with the flipped k, and the padding that makes the shapes close is
    pad0' = k - 1 - pad0        pad1' = n_in * up - n_out * down + pad0 - up + 1          (per axis; n_out as computed by the forward op),
with up and down exchanged.  So ONE autograd node, parameterised by a geometry record that knows its own adjoint, covers the forward pass, the
gradient and the gradient of the gradient (the R1 penalty differentiates through the discriminator's backward, vqperceptual.py:157-162): the
adjoint of the adjoint is the original geometry again.
"""
from __future__ import annotations

from typing import NamedTuple, Tuple

import torch

from ... import _C


class _Geom(NamedTuple):
    up: Tuple[int, int]            # (x, y)
    down: Tuple[int, int]
    pad: Tuple[int, int, int, int]  # x0, x1, y0, y1
    in_hw: Tuple[int, int]
    k_hw: Tuple[int, int]

    def out_hw(self) -> Tuple[int, int]:
        (ux, uy), (dx, dy), (px0, px1, py0, py1), (h, w), (kh, kw) = self
        return (h * uy + py0 + py1 - kh + dy) // dy, (w * ux + px0 + px1 - kw + dx) // dx

    def adjoint(self) -> "_Geom":
        (ux, uy), (dx, dy), (px0, _, py0, _), (h, w), (kh, kw) = self
        oh, ow = self.out_hw()
        pad = (kw - 1 - px0, w * ux - ow * dx + px0 - ux + 1, kh - 1 - py0, h * uy - oh * dy + py0 - uy + 1)
        return _Geom(self.down, self.up, pad, (oh, ow), self.k_hw)


class _UpFirDn(torch.autograd.Function):
    """x [..., H, W] -> [..., H', W'] under geometry g with kernel k; the backward is the same node under g.adjoint() with the flipped kernel"""

    @staticmethod
    def forward(ctx, x, k, g: _Geom):
        ctx.g = g
        ctx.save_for_backward(k)
        lead = x.shape[:-2]
        (ux, uy), (dx, dy), (px0, px1, py0, py1) = g.up, g.down, g.pad
        y = _C.upfirdn2d(x.reshape(-1, *g.in_hw).contiguous(), k, ux, uy, dx, dy, px0, px1, py0, py1)
        return y.view(*lead, *g.out_hw())

    @staticmethod
    def backward(ctx, gy):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        k, = ctx.saved_tensors
        adj = ctx.g.adjoint()
        assert adj.out_hw() == ctx.g.in_hw
        return _UpFirDn.apply(gy, torch.flip(k, [0, 1]).contiguous(), adj), None, None


def upfirdn2d(input: torch.Tensor, kernel: torch.Tensor, up=1, down=1, pad=(0, 0)) -> torch.Tensor:
    pair = lambda v: (int(v), int(v)) if not isinstance(v, (tuple, list)) else (int(v[0]), int(v[1]))
    pad = tuple(int(p) for p in pad)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    g = _Geom(pair(up), pair(down), pad, (int(input.shape[-2]), int(input.shape[-1])), (int(kernel.shape[0]), int(kernel.shape[1])))
    return _UpFirDn.apply(input, kernel.contiguous(), g)
