"""Arbitrarily differentiable 2-D convolution for the StyleGAN2 discriminator (reference enhancing/losses/op/conv2d_gradfix.py:
``conv2d(input, weight, bias, stride, padding)`` and the ``no_weight_gradients()`` context the R1 penalty runs under,
vqperceptual.py:157-158), lowered onto this library's kernels instead of cuDNN:

    cols = im2col(x)                       enh_im2col       (rows (b,ho,wo) x columns (c,kh,kw); bf16 | fp16 | f32: operand_dtype)
    y    = W[Cout, Kp] . cols^T            enh_gemm_h16     (16-bit MFMA, f32 accumulate, f32 out) | enh_gemm_f32 (the fp32 instrument)
    dW   = dy . cols ;  dcols = dy^T . W   enh_gemm_h16     (same kernel family, other storage flags)
    dx   = col2im(dcols)                   enh_col2im

Every backward is expressed with the same two differentiable primitives (`_Gemm`, `_Im2col` / `_Col2im`), so gradients of gradients
(R1: d/dtheta |d D(x)/dx|^2) come out of autograd with no extra formulas.  The discriminator keeps its activations channel-major
([C, B, H, W]) because that is what the forward GEMM writes; ``conv2d`` (the reference's NCHW signature) wraps ``conv2d_cm``.
There is no CPU branch: the arithmetic only exists as HIP kernels."""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F
from torch.autograd import Function

from ... import _C

enabled = True
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    """skip weight-gradient GEMMs inside the block (the R1 penalty's first-order pass only needs d/d input)"""
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


# Operand format of this lowering's GEMMs and column tensors: "bf16" (default), "fp16" (11 significand bits: ~8x smaller operand rounding at the same MFMA
# rate) or "fp32" — im2col columns in f32 and every product on the exact-f32 GEMM (csrc/exact_f32.hip: fixed ascending-k fp32 accumulation, no 16-bit
# rounding anywhere): the discriminator's PARITY INSTRUMENT, as the fp32 engine mode is for the towers.  StyleDiscriminator(lowering="im2col") under
# `with operand_dtype("fp32"):` reproduces the reference's fp32 forward, R1 double backward and parameter gradients to ~1e-6
# (tests/test_disc_model_gpu.py::test_discriminator_against_reference_golden).
_OPERAND = torch.bfloat16
_OPERANDS = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


@contextlib.contextmanager
def operand_dtype(name: str):
    global _OPERAND
    if name not in _OPERANDS:
        raise ValueError(f"operand_dtype: expected one of {sorted(_OPERANDS)}, got {name!r}")
    old, _OPERAND = _OPERAND, _OPERANDS[name]
    try:
        yield
    finally:
        _OPERAND = old


@contextlib.contextmanager
def _operand(dt):
    """the operand format a node was built with, re-installed while its backward (which builds further nodes: double backward) runs — also when
    .backward() is called outside the operand_dtype block"""
    global _OPERAND
    old, _OPERAND = _OPERAND, dt
    try:
        yield
    finally:
        _OPERAND = old


def _bf16(t: torch.Tensor) -> torch.Tensor:
    """t as a GEMM operand of the current operand format (the name is historical): 16-bit tensors pass through, f32 tensors are rounded to the 16-bit
    format (round-to-nearest-even) — or used as they are in the fp32 instrument"""
    if t.dtype in (torch.bfloat16, torch.float16) or _OPERAND == torch.float32:
        return t
    out = torch.empty(t.shape, dtype=_OPERAND, device=t.device)
    _C.cast_bf16(t.contiguous(), out)
    return out


class _Gemm(Function):
    """C[M,N] = sum_k A(m,k) B(n,k) on enh_gemm_h16.  `a` is stored [M,K] (ta False) or [K,M]; `b` is stored [N,K] (tb False) or
    [K,N]; f32 operands are cast to bf16, bf16 operands are used as they are; the result is f32 unless out_bf16.  wa / wb mark an
    operand as a weight (its gradient is skipped under no_weight_gradients()).  M, N, K must be multiples of 8 so that every
    derivative (which permutes the three roles) satisfies the kernel's alignment rules."""

    @staticmethod
    def forward(ctx, a, b, ta: bool, tb: bool, out_bf16: bool, wa: bool, wb: bool):
        M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
        N, Kb = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
        if K != Kb or M % 8 or N % 8 or K % 8:
            raise RuntimeError(f"_Gemm: need matching K and M, N, K multiples of 8 (a {tuple(a.shape)} ta={ta}, b {tuple(b.shape)} tb={tb})")
        a16, b16 = _bf16(a.contiguous()), _bf16(b.contiguous())
        if a16.dtype == torch.float32 or b16.dtype == torch.float32:      # the fp32 instrument: exact-f32 GEMM, f32 result
            out = torch.empty(M, N, dtype=torch.float32, device=a.device)
            _C.mm(a16.float(), b16.float(), M, N, K, out, trans_a=ta, trans_b=tb)
            ctx.save_for_backward(a, b)
            ctx.cfg = (ta, tb, wa, wb)
            ctx.op = _OPERAND
            return out
        dt = a16.dtype if out_bf16 else torch.float32
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        split = (not out_bf16) and K >= 2048 and tiles < 256      # weight-gradient shape: let the kernel split K (needs a zeroed f32 C)
        out = (torch.zeros if split else torch.empty)(M, N, dtype=dt, device=a.device)
        _C.gemm(a16, b16, M, N, K, trans_a=ta, trans_b=tb, accumulate=split, out_f32=None if out_bf16 else out, out_bf16=out if out_bf16 else None)
        ctx.save_for_backward(a, b)
        ctx.cfg = (ta, tb, wa, wb)
        ctx.op = _OPERAND
        return out

    @staticmethod
    def backward(ctx, g):
        with _operand(ctx.op):
            return _Gemm._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        a, b = ctx.saved_tensors
        ta, tb, wa, wb = ctx.cfg
        g = g.contiguous()
        da = db = None
        if ctx.needs_input_grad[0] and not (wa and weight_gradients_disabled):
            if not ta:   # dA[M,K] = g[M,N] . B(n,k)
                da = _Gemm.apply(g, b, False, not tb, a.dtype in (torch.bfloat16, torch.float16), False, wb)
            else:        # dA^T[K,M] = B(n,k)^T . g^T
                da = _Gemm.apply(b, g, not tb, False, a.dtype in (torch.bfloat16, torch.float16), wb, False)
        if ctx.needs_input_grad[1] and not (wb and weight_gradients_disabled):
            if not tb:   # dB[N,K] = g^T . A(m,k)
                db = _Gemm.apply(g, a, True, not ta, b.dtype in (torch.bfloat16, torch.float16), False, wa)
            else:        # dB^T[K,N] = A(m,k)^T . g
                db = _Gemm.apply(a, g, not ta, True, b.dtype in (torch.bfloat16, torch.float16), wa, False)
        return da, db, None, None, None, None, None


def _strides(layout: str, B: int, C: int, H: int, W: int):
    """(batch stride, channel stride) of a contiguous image tensor: 'nchw' = [B,C,H,W], 'cm' = channel-major [C,B,H,W]"""
    return (C * H * W, H * W) if layout == "nchw" else (H * W, B * H * W)


class _Im2col(Function):
    @staticmethod
    def forward(ctx, x, layout: str, k: int, stride: int, pad: int):
        B, C, H, W = x.shape if layout == "nchw" else (x.shape[1], x.shape[0], x.shape[2], x.shape[3])
        ctx.cfg = (layout, (B, C, H, W), k, stride, pad)
        ctx.op = _OPERAND
        sb, sc = _strides(layout, B, C, H, W)
        return _C.im2col(x.contiguous(), sb, sc, B, C, H, W, k, stride, pad, dtype=_OPERAND)

    @staticmethod
    def backward(ctx, g):
        layout, shape, k, stride, pad = ctx.cfg
        with _operand(ctx.op):
            return _Col2im.apply(g, layout, shape, k, stride, pad), None, None, None, None


class _Col2im(Function):
    @staticmethod
    def forward(ctx, dcols, layout: str, shape, k: int, stride: int, pad: int):
        B, C, H, W = shape
        ctx.cfg = (layout, k, stride, pad)
        ctx.op = _OPERAND
        out = torch.empty((B, C, H, W) if layout == "nchw" else (C, B, H, W), dtype=torch.float32, device=dcols.device)
        sb, sc = _strides(layout, B, C, H, W)
        return _C.col2im(_bf16(dcols.contiguous()), B, C, H, W, k, stride, pad, out, sb, sc)

    @staticmethod
    def backward(ctx, g):
        layout, k, stride, pad = ctx.cfg
        with _operand(ctx.op):
            return _Im2col.apply(g, layout, k, stride, pad), None, None, None, None, None


def conv2d_cm(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, padding: int = 0, layout: str = "cm") -> torch.Tensor:
    """x: f32 image tensor in `layout`; weight [Cout, Cin, k, k] (already scaled); returns channel-major [Cout, B, Ho, Wo]."""
    Cout, Cin, k, k2 = weight.shape
    B, C, H, W = x.shape if layout == "nchw" else (x.shape[1], x.shape[0], x.shape[2], x.shape[3])
    if k != k2 or C != Cin or Cout % 8:
        raise RuntimeError(f"conv2d_cm: weight {tuple(weight.shape)} does not fit input {tuple(x.shape)} ({layout}); Cout must be a multiple of 8")
    Ho, Wo = _C.conv_out_size(H, k, stride, padding), _C.conv_out_size(W, k, stride, padding)
    if (B * Ho * Wo) % 8:
        raise RuntimeError(f"conv2d_cm: B*Ho*Wo = {B * Ho * Wo} must be a multiple of 8")
    cols = _Im2col.apply(x, layout, k, stride, padding)
    w2 = weight.reshape(Cout, Cin * k * k)
    if cols.shape[1] != w2.shape[1]:
        w2 = F.pad(w2, (0, cols.shape[1] - w2.shape[1]))          # the zero columns im2col appends for alignment
    y = _Gemm.apply(w2, cols, False, False, False, True, False)
    return y.view(Cout, B, Ho, Wo)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """the reference's signature: NCHW in, NCHW out (conv2d_gradfix.py:22-42)"""
    if dilation != 1 or groups != 1:
        raise NotImplementedError("conv2d: dilation / groups are not used by the discriminator")
    y = conv2d_cm(input, weight, stride, padding, layout="nchw").permute(1, 0, 2, 3)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y.contiguous()


def linear(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """x [B, in] . weight[out, in]^T on the same GEMM; batch and out are zero-padded to the alignment unit and sliced back"""
    Bn, out_f = x.shape[0], weight.shape[0]
    xp = F.pad(x, (0, 0, 0, -Bn % 8)) if Bn % 8 else x
    wp = F.pad(weight, (0, 0, 0, -out_f % 8)) if out_f % 8 else weight
    y = _Gemm.apply(xp, wp, False, False, False, False, True)
    return y[:Bn, :out_f]
