"""Arbitrarily differentiable channels-last convolution stack of the StyleGAN2 discriminator on implicit-GEMM HIP kernels — no im2col tensor.

Replaces, for activations kept as bf16 ``[B, H, W, C]`` (C padded to a multiple of 8 with zero channels):

    conv2d_gradfix.conv2d(x, weight * scale, stride, padding)      reference enhancing/losses/op/conv2d_gradfix.py:22-42, called from
                                                                   EqualConv2d.forward (enhancing/losses/layers.py:176-185)
    fused_leaky_relu(conv + bias)                                  layers.py:220-264 (ConvLayer = [Blur] -> EqualConv2d -> FusedLeakyReLU)
    upfirdn2d(x, kernel, pad)                                      layers.py:140-160 (Blur)
    (out + skip) / sqrt(2)                                         layers.py:262 (StyleBlock), folded into the skip convolution's epilogue

Three convolution primitives are closed under differentiation, exactly the triangle conv2d_gradfix builds out of cuDNN calls
(conv2d_gradfix.py:81-195): forward ``_Conv``, input gradient ``_Dgrad``, weight gradient ``_Wgrad``; the derivative of each is expressed with
the other two, so the R1 penalty (a gradient of a gradient, vqperceptual.py:157-162) needs no extra formulas.  ``no_weight_gradients()`` of
conv2d_gradfix applies here too.  The element-wise pieces (``_Gate`` = derivative of the leaky-ReLU through its saved output, ``_Blur``,
``_ToNHWC8`` / ``_FromNHWC8``) are linear in their data argument and their own adjoints up to arguments.  No CPU branch exists."""
from __future__ import annotations

import weakref

import torch
from torch.autograd import Function

from ... import _C
from . import conv2d_gradfix


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _no_wgrad() -> bool:
    return conv2d_gradfix.weight_gradients_disabled


def _out_size(n: int, k: int, stride: int, pad: int) -> int:
    return (n + 2 * pad - k) // stride + 1


# ---- kernel-level helpers (not differentiable by themselves) ----------------------------------------------------------------------
# Packed bf16 operand images of the PARAMETERS are cached between optimizer steps: one adversarial step runs the discriminator forward three times
# and its input-gradient twice with unchanged weights.  A cache entry is keyed on the parameter object, its storage, its autograd version (in-place torch
# writes bump it) and the packing arguments; the fused AdamW kernel writes through raw pointers, so FlatAdamW.step() calls
# invalidate_packed_weights().  Anything that is not an nn.Parameter (the weight-shaped gradient of a second-order pass) is packed every time.
_PACKED = {}

# 16-bit operand format of the channels-last path: the container dtype of the activations decides (every kernel entry takes it as `dtype`), and the packed
# weight images follow the activation they meet.  image_to_nhwc8 — the one place where an f32 image enters the path — uses OPERAND_DTYPE: bf16 by default
# (f32's exponent range: gradients of any magnitude survive the R1 double backward unscaled); "fp16" (`operand_dtype("fp16")`, ENH_LOSS_OPERANDS=fp16)
# gives 8x smaller operand rounding — logits 2e-3 instead of 1.5e-2 against the reference's golden — for callers that keep their gradients inside fp16's
# range (loss scale).
import contextlib
import os

OPERAND_DTYPE = torch.float16 if os.environ.get("ENH_LOSS_OPERANDS", "bf16") == "fp16" else torch.bfloat16


@contextlib.contextmanager
def operand_dtype(name: str):
    global OPERAND_DTYPE
    old, OPERAND_DTYPE = OPERAND_DTYPE, {"bf16": torch.bfloat16, "fp16": torch.float16}[name]
    try:
        yield
    finally:
        OPERAND_DTYPE = old


def invalidate_packed_weights() -> None:
    _PACKED.clear()


def _pack(w, scale, transposed, kh0, kw0, kstep, nty, ntx, rows, cols, dtype=torch.bfloat16):
    if not isinstance(w, torch.nn.Parameter):
        return _C.conv_pack_weight(w.contiguous(), scale, transposed, kh0, kw0, kstep, nty, ntx, rows, cols, dtype=dtype)
    key = (id(w), w.data_ptr(), w._version, float(scale), transposed, kh0, kw0, kstep, nty, ntx, rows, cols, dtype)
    hit = _PACKED.get(key)
    if hit is None and len(_PACKED) >= 512:        # an optimizer that never calls invalidate_packed_weights() (in-place torch updates bump the version
        _PACKED.clear()                            # every step) must not grow the cache without bound
    if hit is None or hit[0]() is not w:          # (the weak reference guards against a recycled id / address of a freed parameter)
        hit = _PACKED[key] = (weakref.ref(w), _C.conv_pack_weight(w.detach().contiguous(), scale, transposed, kh0, kw0, kstep, nty, ntx, rows, cols, dtype=dtype))
    return hit[1]


def _fwd(x, w, scale, stride, pad, mode=2, bias=None, add=None, p0=0.0, p1=1.0):
    B, H, W, Cp = x.shape
    Cout, Cin, k, _ = w.shape
    if Cp % 8 or Cp < Cin or Cout % 8:
        raise RuntimeError(f"conv_nhwc: x has {Cp} channels for a weight {tuple(w.shape)}; channels must be padded to a multiple of 8 and Cout % 8 == 0")
    Ho, Wo = _out_size(H, k, stride, pad), _out_size(W, k, stride, pad)
    wt = _pack(w, scale, False, 0, 0, 1, k, k, Cout, Cp, dtype=x.dtype)
    geom = dict(B=B, Hs=H, Ws=W, C=Cp, Hm=Ho, Wm=Wo, gs=stride, oy0=-pad, ox0=-pad, nty=k, ntx=k, sty=1, stx=1, N=Cout, HO=Ho, WO=Wo, os=1, oph=0, opw=0)
    return _C.conv_nhwc(x, wt, geom, mode, bias=bias, add=add, p0=p0, p1=p1)


def _dgrad(dy, w, scale, stride, pad, H, W, Cp):
    B, Ho, Wo, Cout = dy.shape
    k = w.shape[2]
    if stride == 1:
        wt = _pack(w, scale, True, 0, 0, 1, k, k, Cp, Cout, dtype=dy.dtype)
        geom = dict(B=B, Hs=Ho, Ws=Wo, C=Cout, Hm=H, Wm=W, gs=1, oy0=pad, ox0=pad, nty=k, ntx=k, sty=-1, stx=-1, N=Cp, HO=H, WO=W, os=1, oph=0, opw=0)
        return _C.conv_nhwc(dy, wt, geom, 2)
    out = None
    if k < stride:
        # some parity classes are reached by no tap at all (the 1 x 1 / stride-2 skip convolutions: three classes of four): their rows are zero.  One
        # streaming fill of the whole tensor instead of a tile kernel per empty class writing strided 8-byte pieces (195 -> ~100 us at 255^2 x 128 x 16)
        out = torch.zeros(B, H, W, Cp, dtype=dy.dtype, device=dy.device)
    for ph in range(stride):          # rows of one parity class are reached by every stride-th tap only
        kh0 = (ph + pad) % stride
        nty, oy0, Hm = len(range(kh0, k, stride)), (ph + pad - kh0) // stride, (H - ph + stride - 1) // stride
        for pw in range(stride):
            kw0 = (pw + pad) % stride
            ntx, ox0, Wm = len(range(kw0, k, stride)), (pw + pad - kw0) // stride, (W - pw + stride - 1) // stride
            if Hm <= 0 or Wm <= 0 or (out is not None and k < stride and nty * ntx == 0):
                continue
            wt = _pack(w, scale, True, kh0, kw0, stride, nty, ntx, Cp, Cout, dtype=dy.dtype)
            geom = dict(B=B, Hs=Ho, Ws=Wo, C=Cout, Hm=Hm, Wm=Wm, gs=1, oy0=oy0, ox0=ox0, nty=nty, ntx=ntx, sty=-1, stx=-1, N=Cp, HO=H, WO=W,
                        os=stride, oph=ph, opw=pw)
            out = _C.conv_nhwc(dy, wt, geom, 2, out=out)
    return out


def _wgrad(x, dy, scale, stride, pad, k, Cin):
    B, H, W, Cp = x.shape
    _, Ho, Wo, Cout = dy.shape
    geom = dict(B=B, Hs=H, Ws=W, C=Cp, Hm=Ho, Wm=Wo, gs=stride, oy0=-pad, ox0=-pad, nty=k, ntx=k, sty=1, stx=1, N=Cout, HO=Ho, WO=Wo, os=1, oph=0, opw=0)
    dwp = _C.conv_wgrad_nhwc(x, dy, geom)
    return _C.conv_unpack_wgrad(dwp, Cout, Cin, Cp, k, scale)


# ---- the differentiable triangle ---------------------------------------------------------------------------------------------------
class _Conv(Function):
    """y[B,Ho,Wo,Cout] = conv(x[B,H,W,Cp], scale * w[Cout,Cin,k,k])"""

    @staticmethod
    def forward(ctx, x, w, scale, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.cfg = (scale, stride, pad)
        return _fwd(x.contiguous(), w, scale, stride, pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        scale, stride, pad = ctx.cfg
        g = g.contiguous()
        dx = _Dgrad.apply(g, w, scale, stride, pad, x.shape[1], x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = _Wgrad.apply(x, g, scale, stride, pad, w.shape[2], w.shape[1]) if ctx.needs_input_grad[1] and not _no_wgrad() else None
        return dx, dw, None, None, None


class _Dgrad(Function):
    """dx[B,H,W,Cp] = conv_transpose(dy[B,Ho,Wo,Cout], scale * w)"""

    @staticmethod
    def forward(ctx, dy, w, scale, stride, pad, H, W, Cp):
        ctx.save_for_backward(dy, w)
        ctx.cfg = (scale, stride, pad)
        return _dgrad(dy.contiguous(), w, scale, stride, pad, H, W, Cp)

    @staticmethod
    def backward(ctx, G):
        dy, w = ctx.saved_tensors
        scale, stride, pad = ctx.cfg
        G = G.contiguous()
        d_dy = _Conv.apply(G, w, scale, stride, pad) if ctx.needs_input_grad[0] else None
        d_w = _Wgrad.apply(G, dy, scale, stride, pad, w.shape[2], w.shape[1]) if ctx.needs_input_grad[1] and not _no_wgrad() else None
        return d_dy, d_w, None, None, None, None, None, None


class _Wgrad(Function):
    """dw[Cout,Cin,k,k] (f32) = scale * sum over pixels of dy (x) gathered x"""

    @staticmethod
    def forward(ctx, x, dy, scale, stride, pad, k, Cin):
        ctx.save_for_backward(x, dy)
        ctx.cfg = (scale, stride, pad)
        return _wgrad(x.contiguous(), dy.contiguous(), scale, stride, pad, k, Cin)

    @staticmethod
    def backward(ctx, Gw):
        x, dy = ctx.saved_tensors
        scale, stride, pad = ctx.cfg
        Gw = Gw.contiguous()
        d_x = _Dgrad.apply(dy, Gw, scale, stride, pad, x.shape[1], x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        d_dy = _Conv.apply(x, Gw, scale, stride, pad) if ctx.needs_input_grad[1] else None
        return d_x, d_dy, None, None, None, None, None


class _Gate(Function):
    """y = g * (ref > 0 ? 1 : slope) * scale; ref = a saved leaky-ReLU OUTPUT, treated as a constant like the reference's
    FusedLeakyReLUFunctionBackward does (fused_act.py:21-45); ref None: y = g * scale"""

    @staticmethod
    def forward(ctx, g, ref, slope, scale):
        ctx.has_ref, ctx.cfg = ref is not None, (slope, scale)
        if ref is not None:
            ctx.save_for_backward(ref)
        return _C.lrelu_gate(g.contiguous(), ref, slope, scale)

    @staticmethod
    def backward(ctx, G):
        ref = ctx.saved_tensors[0] if ctx.has_ref else None
        return _Gate.apply(G, ref, *ctx.cfg), None, None, None


class _ColSum(Function):
    """bias gradient: sum over (b, h, w) of a channels-last tensor -> f32 [C]"""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (x.shape, x.dtype)
        return _C.colsum_nhwc(x.contiguous())

    @staticmethod
    def backward(ctx, G):
        shape, dtype = ctx.meta
        return G.to(dtype).expand(shape)


class _ConvBiasAct(Function):
    """out = lrelu(conv(x, scale*w) + bias, slope) * act_scale in one kernel (EqualConv2d + FusedLeakyReLU)"""

    @staticmethod
    def forward(ctx, x, w, bias, scale, stride, pad, slope, act_scale):
        out = _fwd(x.contiguous(), w, scale, stride, pad, mode=3, bias=bias.contiguous() if bias is not None else None, p0=slope, p1=act_scale)
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (scale, stride, pad, slope, act_scale, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        scale, stride, pad, slope, act_scale, has_bias = ctx.cfg
        g_pre = _Gate.apply(g.contiguous(), out, slope, act_scale)
        dx = _Dgrad.apply(g_pre, w, scale, stride, pad, x.shape[1], x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        skip_w = _no_wgrad()
        dw = _Wgrad.apply(x, g_pre, scale, stride, pad, w.shape[2], w.shape[1]) if ctx.needs_input_grad[1] and not skip_w else None
        db = _ColSum.apply(g_pre) if has_bias and ctx.needs_input_grad[2] and not skip_w else None
        return dx, dw, db, None, None, None, None, None


class _ConvAdd(Function):
    """out = conv(x, scale*w) + alpha * add (the residual merge of a StyleBlock inside the skip convolution's epilogue)"""

    @staticmethod
    def forward(ctx, x, w, add, scale, stride, pad, alpha):
        ctx.save_for_backward(x, w)
        ctx.cfg = (scale, stride, pad, alpha)
        return _fwd(x.contiguous(), w, scale, stride, pad, mode=4, add=add.contiguous(), p0=alpha)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        scale, stride, pad, alpha = ctx.cfg
        g = g.contiguous()
        dx = _Dgrad.apply(g, w, scale, stride, pad, x.shape[1], x.shape[2], x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = _Wgrad.apply(x, g, scale, stride, pad, w.shape[2], w.shape[1]) if ctx.needs_input_grad[1] and not _no_wgrad() else None
        dadd = (g if alpha == 1.0 else _Gate.apply(g, None, 1.0, alpha)) if ctx.needs_input_grad[2] else None
        return dx, dw, dadd, None, None, None, None


class _Blur(Function):
    """upfirdn2d(x, kernel, pad=(pad0, pad1)) with unit up / down factors on [B,H,W,C]; flip selects the adjoint's tap order"""

    @staticmethod
    def forward(ctx, x, kernel, pad0, pad1, flip):
        ctx.save_for_backward(kernel)
        ctx.cfg = (pad0, pad1, flip)
        return _C.blur_nhwc(x.contiguous(), kernel, pad0, pad1, flip)

    @staticmethod
    def backward(ctx, g):
        kernel, = ctx.saved_tensors
        pad0, pad1, flip = ctx.cfg
        kh = kernel.shape[0]
        return _Blur.apply(g, kernel, kh - 1 - pad0, kh - 1 - pad1, not flip), None, None, None, None


class _ToNHWC8(Function):
    @staticmethod
    def forward(ctx, img, dtype=None):
        ctx.C = img.shape[1]
        return _C.img_to_nhwc8(img.contiguous(), dtype=OPERAND_DTYPE if dtype is None else dtype)

    @staticmethod
    def backward(ctx, g):
        return _FromNHWC8.apply(g, ctx.C), None


class _FromNHWC8(Function):
    @staticmethod
    def forward(ctx, x, C):
        ctx.dtype = x.dtype      # (the double backward re-enters the 16-bit path in the format this graph runs in, whatever OPERAND_DTYPE says by then)
        return _C.nhwc8_to_img(x.contiguous(), C)

    @staticmethod
    def backward(ctx, g):
        return _ToNHWC8.apply(g, ctx.dtype), None


def _stddev_torch(x32, group: int, Cp: int):
    """the same map in differentiable torch ops on f32 (used only for the SECOND-order term of the R1 penalty, on a B x 4 x 4 x 512 tensor)"""
    B, H, W, C = x32.shape
    n = B // group
    sd = torch.sqrt(x32.view(group, n, H, W, C).var(0, unbiased=False) + 1e-8).mean(dim=(1, 2, 3))
    sd_map = sd.repeat(group).view(B, 1, 1, 1).expand(B, H, W, 1)
    return torch.cat([x32, sd_map, x32.new_zeros(B, H, W, Cp - C - 1)], 3)


class _Stddev(Function):
    """minibatch standard deviation + concatenation + channel padding (layers.py:358-367), one kernel"""

    @staticmethod
    def forward(ctx, x, group, Cp):
        ctx.save_for_backward(x)
        ctx.cfg = (group, Cp)
        return _C.minibatch_stddev_nhwc(x.contiguous(), group, Cp)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return _StddevBwd.apply(g, x, *ctx.cfg), None, None


class _StddevBwd(Function):
    @staticmethod
    def forward(ctx, g, x, group, Cp):
        ctx.save_for_backward(g, x)
        ctx.cfg = (group, Cp)
        return _C.minibatch_stddev_nhwc_backward(x.contiguous(), g.contiguous(), group)

    @staticmethod
    def backward(ctx, G):
        g, x = ctx.saved_tensors
        group, Cp = ctx.cfg
        with torch.enable_grad():
            x32, g32 = x.detach().float().requires_grad_(True), g.detach().float().requires_grad_(True)
            dx, = torch.autograd.grad(_stddev_torch(x32, group, Cp), x32, g32, create_graph=True)
            d_g, d_x = torch.autograd.grad(dx, (g32, x32), G.float())
        return d_g.to(g.dtype), d_x.to(x.dtype), None, None


# ---- public functions ---------------------------------------------------------------------------------------------------------------
def conv(x, weight, scale: float = 1.0, stride: int = 1, padding: int = 0):
    return _Conv.apply(x, weight, scale, stride, padding)


def conv_bias_lrelu(x, weight, bias, scale: float, stride: int, padding: int, negative_slope: float = 0.2, act_scale: float = 2 ** 0.5):
    return _ConvBiasAct.apply(x, weight, bias, scale, stride, padding, negative_slope, act_scale)


def conv_add(x, weight, add, scale: float, stride: int, padding: int, alpha: float):
    return _ConvAdd.apply(x, weight, add, scale, stride, padding, alpha)


def blur(x, kernel, pad):
    if kernel.shape[0] != kernel.shape[1]:
        raise RuntimeError("blur: square FIR kernels only")
    return _Blur.apply(x, kernel, int(pad[0]), int(pad[1]), False)


def image_to_nhwc8(img):
    """[B,C<=8,H,W] f32 -> [B,H,W,8] bf16 (zero padding channels)"""
    return _ToNHWC8.apply(img)


def minibatch_stddev(x, group: int):
    """x [B,H,W,C] -> [B,H,W,pad8(C+1)]: x, one standard-deviation channel, zero padding"""
    return _Stddev.apply(x, group, pad8(x.shape[3] + 1))
