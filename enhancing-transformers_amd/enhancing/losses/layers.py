"""StyleGAN2 discriminator and GAN losses with the reference's class names, constructor arguments and state-dict keys
(reference enhancing/losses/layers.py:22-40 d-losses, :140-264 Blur / EqualConv2d / EqualLinear / ConvLayer / StyleBlock, :322-377
StyleDiscriminator), running on this library's HIP kernels:

  convolutions   implicit GEMM on bf16 MFMA (gather in the load stage, no im2col tensor): forward, input gradient and weight gradient,
                 bias + leaky-ReLU and the residual merge fused into the epilogue                    op/conv_nhwc.py  (enh_conv_nhwc_h16, ...)
  blur           4x4 FIR on channels-last bf16                                                        op/conv_nhwc.py  (enh_blur_nhwc_h16)
  linears        enh_gemm_h16                                                                        op/conv2d_gradfix.linear

Inside ``StyleDiscriminator.forward`` the activations are channels-last bf16 ([B, H, W, C], C padded to a multiple of 8 with zero channels:
the 3-channel image enters as 8, the 513-channel input of the final convolution as 520).  Every layer class also keeps the reference's
NCHW f32 ``forward`` (used on its own it runs the older explicit lowering im2col -> GEMM -> col2im of op/conv2d_gradfix.py, which
``StyleDiscriminator(lowering="im2col")`` still selects for A/B comparison).  The three scalar d-losses are torch ops on B logits.
Not built: PatchDiscriminator / ActNorm (layers.py:50-137, 266-319) — no stage-1 loss of the reference constructs them
(VQLPIPSWithDiscriminator always builds StyleDiscriminator, vqperceptual.py:81)."""
from __future__ import annotations

from math import log2, sqrt
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .op import FusedLeakyReLU, conv2d_gradfix, conv_nhwc, fused_leaky_relu, upfirdn2d


def hinge_d_loss(logits_fake, logits_real=None):
    loss_fake = -logits_fake.mean() * 2 if logits_real is None else F.relu(1. + logits_fake).mean()
    loss_real = 0 if logits_real is None else F.relu(1. - logits_real).mean()
    return 0.5 * (loss_real + loss_fake)


def vanilla_d_loss(logits_fake, logits_real=None):
    loss_fake = F.softplus(-logits_fake).mean() * 2 if logits_real is None else F.softplus(logits_fake).mean()
    loss_real = 0 if logits_real is None else F.softplus(-logits_real).mean()
    return 0.5 * (loss_real + loss_fake)


def least_square_d_loss(logits_fake, logits_real=None):
    loss_fake = logits_fake.pow(2).mean() * 2 if logits_real is None else (1 + logits_fake).pow(2).mean()
    loss_real = 0 if logits_real is None else (1 - logits_real).pow(2).mean()
    return 0.5 * (loss_real + loss_fake)


def _to_cm(x: torch.Tensor) -> torch.Tensor:
    return x.permute(1, 0, 2, 3).contiguous()


def _lrelu_cm(x: torch.Tensor, act: FusedLeakyReLU) -> torch.Tensor:
    """bias + leaky-ReLU on a channel-major tensor: viewed as one 'sample' of C channels with B*H rows each"""
    C, B, H, W = x.shape
    return fused_leaky_relu(x.view(1, C, B * H, W), act.bias, act.negative_slope, act.scale).view(C, B, H, W)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor: int = 1) -> None:
        super().__init__()
        kernel = torch.tensor(kernel, dtype=torch.float32)
        if kernel.ndim == 1:
            kernel = kernel[None, :] * kernel[:, None]
        kernel = kernel / kernel.sum()
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input: torch.Tensor) -> torch.Tensor:   # planes are independent: NCHW and channel-major alike
        return upfirdn2d(input, self.kernel, pad=self.pad)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        return conv_nhwc.blur(x, self.kernel, self.pad)


class EqualConv2d(nn.Module):
    def __init__(self, in_channel: int, out_channel: int, kernel_size: int, stride: int = 1, padding: int = 0, bias: bool = True) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None
        self.scale = 1 / sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding

    def forward_cm(self, input: torch.Tensor, layout: str = "cm") -> torch.Tensor:
        out = conv2d_gradfix.conv2d_cm(input, self.weight * self.scale, self.stride, self.padding, layout=layout)
        if self.bias is not None:
            out = out + self.bias.view(-1, 1, 1, 1)
        return out

    def forward_nhwc(self, x: torch.Tensor, act: Optional[FusedLeakyReLU] = None, add: Optional[torch.Tensor] = None, alpha: float = 1.0,
                     gain: float = 1.0, add_scaled: bool = False) -> torch.Tensor:
        """x [B,H,W,Cp] bf16.  act: the FusedLeakyReLU that follows (fused into the kernel's epilogue; `gain` multiplies its output gain); add / alpha:
        returns alpha * (conv(x) + add) — the residual merge of a StyleBlock, with alpha folded into the weights; add_scaled: `add` already carries
        the factor alpha (its producer took it as `gain`), so it enters with weight 1 and its gradient needs no scaling pass"""
        if act is not None:
            if self.bias is not None or add is not None:
                raise RuntimeError("EqualConv2d.forward_nhwc: an activated convolution carries its bias in the activation and takes no residual")
            return conv_nhwc.conv_bias_lrelu(x, self.weight, act.bias, self.scale, self.stride, self.padding, act.negative_slope, act.scale * gain)
        if add is not None:
            if self.bias is not None:
                raise RuntimeError("EqualConv2d.forward_nhwc: the residual merge is only fused into a bias-free convolution")
            return conv_nhwc.conv_add(x, self.weight, add, self.scale * alpha, self.stride, self.padding, 1.0 if add_scaled else alpha)
        out = conv_nhwc.conv(x, self.weight, self.scale * alpha, self.stride, self.padding)
        return out if self.bias is None else out + (alpha * self.bias).to(out.dtype)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return conv2d_gradfix.conv2d(input, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)


class EqualLinear(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, bias: bool = True, bias_init: float = 0, lr_mul: float = 1, activation: Optional[str] = None) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        out = conv2d_gradfix.linear(input, self.weight * self.scale)
        if self.activation:
            return fused_leaky_relu(out.contiguous(), self.bias * self.lr_mul if self.bias is not None else None)
        return out + self.bias * self.lr_mul if self.bias is not None else out


class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> [FusedLeakyReLU], members indexed as in the reference (state-dict keys '0.kernel', '1.weight', ...)."""

    def __init__(self, in_channel: int, out_channel: int, kernel_size: int, downsample: bool = False, blur_kernel=(1, 3, 3, 1),
                 bias: bool = True, activate: bool = True) -> None:
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(list(blur_kernel), pad=((p + 1) // 2, p // 2)))
            stride = 2
            self.padding = 0
        else:
            stride = 1
            self.padding = kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride, bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel, bias=bias))
        super().__init__(*layers)

    def forward_nhwc(self, x: torch.Tensor, add: Optional[torch.Tensor] = None, alpha: float = 1.0, gain: float = 1.0, add_scaled: bool = False) -> torch.Tensor:
        mods = list(self)
        if isinstance(mods[0], Blur):
            x, mods = mods[0].forward_nhwc(x), mods[1:]
        act = mods[1] if len(mods) > 1 else None
        return mods[0].forward_nhwc(x, act=act, add=add, alpha=alpha, gain=gain, add_scaled=add_scaled)

    def forward_cm(self, x: torch.Tensor, layout: str = "cm") -> torch.Tensor:
        for m in self:
            if isinstance(m, EqualConv2d):
                x, layout = m.forward_cm(x, layout), "cm"
            elif isinstance(m, FusedLeakyReLU):
                x = _lrelu_cm(x, m)
            else:
                x = m(x)
        return x


class StyleBlock(nn.Module):
    def __init__(self, in_channel: int, out_channel: int, blur_kernel=(1, 3, 3, 1)) -> None:
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        # (out + skip(x)) / sqrt(2): the factor on `out` rides in conv2's activation gain (sqrt(2) / sqrt(2) = 1: one multiply less per element, and the
        # backward pass has no scaling pass over d(out)), the one on skip(x) in the skip convolution's weights; the sum is taken in that kernel's epilogue
        alpha = 1 / sqrt(2)
        out = self.conv2.forward_nhwc(self.conv1.forward_nhwc(x), gain=alpha)
        return self.skip.forward_nhwc(x, add=out, alpha=alpha, add_scaled=True)

    def forward_cm(self, x: torch.Tensor) -> torch.Tensor:
        out = self.conv2.forward_cm(self.conv1.forward_cm(x))
        return (out + self.skip.forward_cm(x)) / sqrt(2)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self.forward_cm(_to_cm(input)).permute(1, 0, 2, 3).contiguous()


class StyleDiscriminator(nn.Module):
    def __init__(self, size: int = 256, channel_multiplier: int = 2, blur_kernel=(1, 3, 3, 1), lowering: str = "igemm") -> None:
        super().__init__()
        if lowering not in ("igemm", "im2col"):
            raise ValueError(f"StyleDiscriminator: lowering must be 'igemm' or 'im2col', got {lowering!r}")
        self.lowering = lowering
        channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                    256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        log_size = int(log2(size))
        in_channel = channels[size]
        blocks = [ConvLayer(3, channels[size], 1)]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            blocks.append(StyleBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.blocks = nn.Sequential(*blocks)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation="fused_lrelu"), EqualLinear(channels[4], 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,3,size,size] f32 on the ROCm device -> logits [B] (layers.py:354-377)"""
        return self._forward_nhwc(x) if self.lowering == "igemm" else self._forward_cm(x)

    def _forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        out = conv_nhwc.image_to_nhwc8(x)                             # [B,H,W,8] bf16
        for blk in self.blocks:
            out = blk.forward_nhwc(out)
        B, H, W, C = out.shape
        # minibatch standard deviation (layers.py:358-367): sample b belongs to slot b % (B/group); one scalar per slot
        group = min(B, self.stddev_group)
        group = B // (B // group)                                      # the reference's own adjustment for B % group != 0 (layers.py:361-362), kept verbatim in meaning: B = 6 -> one group of 6
        out = conv_nhwc.minibatch_stddev(out, group)                     # 513 channels, zero-padded to 520
        out = self.final_conv.forward_nhwc(out)                          # [B,4,4,512]
        out = out.permute(0, 3, 1, 2).reshape(B, -1).float()             # per-sample (c, h, w) flattening, as the reference's .view
        return self.final_linear(out).squeeze()

    def _forward_cm(self, x: torch.Tensor) -> torch.Tensor:
        out = self.blocks[0].forward_cm(x.contiguous(), layout="nchw")
        for blk in list(self.blocks)[1:]:
            out = blk.forward_cm(out)
        C, B, H, W = out.shape
        # minibatch standard deviation (layers.py:358-367): sample b belongs to slot b % (B/group); one scalar per slot
        group = min(B, self.stddev_group)
        group = B // (B // group)                                      # the reference's own adjustment for B % group != 0 (layers.py:361-362), kept verbatim in meaning: B = 6 -> one group of 6
        n = B // group
        sd = torch.sqrt(out.view(C, group, n, H, W).var(1, unbiased=False) + 1e-8).mean(dim=(0, 2, 3))      # [n]
        sd_map = sd.repeat(group).view(1, B, 1, 1).expand(1, B, H, W)
        out = torch.cat([out, sd_map], 0)
        out = self.final_conv.forward_cm(out)                        # [512, B, 4, 4]
        out = out.permute(1, 0, 2, 3).reshape(B, -1)                 # per-sample (c, h, w) flattening, as the reference's .view
        return self.final_linear(out).squeeze()
