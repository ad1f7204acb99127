"""CPU ORACLE (test infrastructure) for the StyleGAN2 discriminator and the discriminator-side losses: a plain-PyTorch fp32 restatement,
differentiable to any order through torch.autograd, of

  StyleDiscriminator.forward      reference enhancing/losses/layers.py:322-377 (ConvLayer :189-223, StyleBlock :226-245, EqualConv2d :163-185,
                                  EqualLinear :188-214 in the reference's numbering, Blur :140-160)
  vanilla / hinge / lsq d-losses  layers.py:22-40
  generator / discriminator loss  vqperceptual.py:111-172 (pixel terms, g_loss, d_loss, lazy R1 with r1_gamma * do_r1_every / 2)

taking the parameters from a state dict with the reference's key names.  The two native ops come from oracle/disc_ops_oracle.py.
Pinned by oracle/make_golden_disc.py: the reference's own layers.py is imported (with its `.op` package replaced by the pinned
restatements of disc_ops_oracle.py and `kornia` stubbed) and must agree with this file; the outputs are committed as
tests/golden/disc_tiny.npz.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file."""
from math import log2, sqrt

import torch
import torch.nn.functional as F

import disc_ops_oracle as DO


def _blur_kernel(k=(1, 3, 3, 1)):
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


def _conv(x, w, stride, pad):
    """EqualConv2d: weight * 1/sqrt(fan_in), no bias here (the bias lives in the following FusedLeakyReLU)"""
    return F.conv2d(x, w * (1 / sqrt(w.shape[1] * w.shape[2] ** 2)), stride=stride, padding=pad)


def _down_conv(x, w, kernel):
    """ConvLayer(downsample=True): blur with pad ((p+1)//2, p//2), p = (4 - 2) + (k - 1), then stride-2 conv without padding"""
    k = w.shape[2]
    p = (kernel.shape[0] - 2) + (k - 1)
    return _conv(DO.upfirdn2d(x, kernel, pad=((p + 1) // 2, p // 2)), w, 2, 0)


def discriminator(sd, x, size):
    """sd: state dict with the reference's keys; x [B,3,size,size] -> logits [B]"""
    kernel = _blur_kernel()
    out = DO.fused_leaky_relu(_conv(x, sd["blocks.0.0.weight"], 1, 0), sd["blocks.0.1.bias"])
    for i in range(1, int(log2(size)) - 1):
        q = f"blocks.{i}."
        o = DO.fused_leaky_relu(_conv(out, sd[q + "conv1.0.weight"], 1, 1), sd[q + "conv1.1.bias"])
        o = DO.fused_leaky_relu(_down_conv(o, sd[q + "conv2.1.weight"], kernel), sd[q + "conv2.2.bias"])
        out = (o + _down_conv(out, sd[q + "skip.1.weight"], kernel)) / sqrt(2)
    B, C, H, W = out.shape
    group = min(B, 4)
    group = B // (B // group)
    sdv = torch.sqrt(out.view(group, -1, 1, C, H, W).var(0, unbiased=False) + 1e-8)
    sdv = sdv.mean([2, 3, 4], keepdims=True).squeeze(2).repeat(group, 1, H, W)
    out = torch.cat([out, sdv], 1)
    out = DO.fused_leaky_relu(_conv(out, sd["final_conv.0.weight"], 1, 1), sd["final_conv.1.bias"])
    out = out.view(B, -1)
    w0, w1 = sd["final_linear.0.weight"], sd["final_linear.1.weight"]
    out = DO.fused_leaky_relu(F.linear(out, w0 * (1 / sqrt(w0.shape[1]))), sd["final_linear.0.bias"])
    out = F.linear(out, w1 * (1 / sqrt(w1.shape[1])), sd["final_linear.1.bias"])
    return out.squeeze()


def vanilla_d_loss(logits_fake, logits_real=None):
    loss_fake = F.softplus(-logits_fake).mean() * 2 if logits_real is None else F.softplus(logits_fake).mean()
    loss_real = 0 if logits_real is None else F.softplus(-logits_real).mean()
    return 0.5 * (loss_real + loss_fake)


def discriminator_loss(sd, size, inputs, reconstructions, do_r1, r1_gamma=10.0, do_r1_every=16, disc_factor=1):
    """vqperceptual.py:148-172 -> (d_loss, logits_real, logits_fake, r1 or None)"""
    real = inputs.detach().clone().requires_grad_(do_r1)
    logits_real = discriminator(sd, real, size)
    logits_fake = discriminator(sd, reconstructions.detach(), size)
    d_loss = disc_factor * vanilla_d_loss(logits_fake, logits_real)
    r1 = None
    if do_r1:
        gradients, = torch.autograd.grad(outputs=logits_real.sum(), inputs=real, create_graph=True)
        r1 = gradients.square().sum([1, 2, 3]).mean()
        d_loss = d_loss + r1_gamma * do_r1_every * r1 / 2
    return d_loss, logits_real, logits_fake, r1


def generator_loss(sd, size, codebook_loss, inputs, reconstructions, w_l1, w_l2, adversarial_weight, codebook_weight=1.0, disc_factor=1):
    """vqperceptual.py:111-131 without the LPIPS term -> (loss, g_loss)"""
    diff = reconstructions - inputs
    nll = w_l1 * diff.abs().mean() + w_l2 * diff.pow(2).mean()
    g_loss = vanilla_d_loss(discriminator(sd, reconstructions, size))
    return nll + disc_factor * adversarial_weight * g_loss + codebook_weight * codebook_loss, g_loss
