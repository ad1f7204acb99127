"""Stage-1 loss modules with the reference's names, constructor arguments, call signature and log keys
(reference enhancing/losses/vqperceptual.py:17-172).

Pixel terms (L1 / L2) and the codebook term are row a20 of SURVEY.md §8; the adversarial term with the StyleGAN2 discriminator and its lazy R1 penalty
is §8f rank 1; the LPIPS perceptual term (§8f rank 2) is ``enhancing.losses.lpips.LPIPS`` — lpips 0.1.4's ``LPIPS(net="vgg")`` topology on HIP kernels,
constructed only when ``perceptual_weight != 0`` (the reference constructs it unconditionally, vqperceptual.py:29,74; skipping an unused 14.7 M-parameter
VGG16 changes no result).  Without pretrained weights on disk the term runs on a random-init trunk and says so (see lpips.py)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn


class DummyLoss(nn.Module):
    def __init__(self) -> None:
        super().__init__()


class VQLPIPS(nn.Module):
    """loss = w_l1*L1 + w_l2*L2 + w_p*LPIPS + codebook_weight*codebook_loss (vqperceptual.py:22-56)."""

    def __init__(self, codebook_weight: float = 1.0, loglaplace_weight: float = 1.0, loggaussian_weight: float = 1.0,
                 perceptual_weight: float = 1.0) -> None:
        super().__init__()
        if perceptual_weight != 0:
            from .lpips import LPIPS
            self.perceptual_loss = LPIPS(net="vgg", verbose=False)     # vqperceptual.py:29
        self.codebook_weight = codebook_weight
        self.loglaplace_weight = loglaplace_weight
        self.loggaussian_weight = loggaussian_weight
        self.perceptual_weight = perceptual_weight

    # 16-bit operand format of the loss networks (LPIPS trunk, discriminator): None -> ENH_LOSS_OPERANDS if set, else the format of the engine that produced
    # the reconstruction (fp16 engine -> fp16: the reference's --use_amp autocasts the loss networks with everything else, main.py:52), else whatever
    # conv_nhwc.OPERAND_DTYPE currently says; "bf16" / "fp16" pin it for this module.
    operands: Optional[str] = None

    def loss_operands(self, last_layer=None) -> str:
        import os
        from .op import conv_nhwc
        name = self.operands or os.environ.get("ENH_LOSS_OPERANDS")
        if name is None:
            eng = getattr(last_layer, "_enh_engine", None)
            if eng is not None:
                name = "fp16" if getattr(eng, "scaled", False) else "bf16"
            else:
                name = "fp16" if conv_nhwc.OPERAND_DTYPE == torch.float16 else "bf16"
        if name not in ("bf16", "fp16"):
            raise ValueError(f"loss-network operands must be 'bf16' or 'fp16', got {name!r}")
        return name

    def forward(self, codebook_loss: torch.Tensor, inputs: torch.Tensor, reconstructions: torch.Tensor, optimizer_idx: int,
                global_step: int, batch_idx: int, last_layer: Optional[nn.Module] = None, split: Optional[str] = "train") -> Tuple:
        from .op import conv2d_gradfix, conv_nhwc
        name = self.loss_operands(last_layer)
        eng = getattr(last_layer, "_enh_engine", None)
        if name == "fp16" and optimizer_idx == 0 and reconstructions.requires_grad and eng is not None and not getattr(eng, "scaled", False) and \
                (hasattr(self, "discriminator") or hasattr(self, "perceptual_loss")):
            raise ValueError("fp16 loss-network operands need the fp16 engine's loss-scaled backward (generator-side gradients of ~1e-5 per pixel sit in "
                             "fp16's subnormal range unscaled); run the engine in fp16 or set ENH_LOSS_OPERANDS=bf16")
        import contextlib
        # (a caller that pinned conv2d_gradfix to fp32 is running the exact-f32 parity instrument of the im2col lowering: left alone)
        gemm_ctx = contextlib.nullcontext() if conv2d_gradfix._OPERAND == torch.float32 else conv2d_gradfix.operand_dtype(name)
        with conv_nhwc.operand_dtype(name), gemm_ctx:
            return self._forward(codebook_loss, inputs, reconstructions, optimizer_idx, global_step, batch_idx, last_layer, split)

    def _forward(self, codebook_loss: torch.Tensor, inputs: torch.Tensor, reconstructions: torch.Tensor, optimizer_idx: int,
                 global_step: int, batch_idx: int, last_layer: Optional[nn.Module] = None, split: Optional[str] = "train") -> Tuple:
        inputs = inputs.contiguous()
        reconstructions = reconstructions.contiguous()
        diff = reconstructions - inputs
        loglaplace_loss = diff.abs().mean()
        loggaussian_loss = diff.pow(2).mean()
        perceptual_loss = self._perceptual(inputs, reconstructions)
        nll_loss = self.loglaplace_weight * loglaplace_loss + self.loggaussian_weight * loggaussian_loss + self.perceptual_weight * perceptual_loss
        loss = nll_loss + self.codebook_weight * codebook_loss
        log = {"{}/total_loss".format(split): loss.clone().detach(),
               "{}/quant_loss".format(split): codebook_loss.detach(),
               "{}/rec_loss".format(split): nll_loss.detach(),
               "{}/loglaplace_loss".format(split): loglaplace_loss.detach(),
               "{}/loggaussian_loss".format(split): loggaussian_loss.detach(),
               "{}/perceptual_loss".format(split): perceptual_loss.detach()}
        return loss, log


    def _perceptual(self, inputs: torch.Tensor, reconstructions: torch.Tensor) -> torch.Tensor:
        """self.perceptual_loss(inputs*2-1, reconstructions*2-1).mean() (vqperceptual.py:43,115); the 2x-1 map is lpips' own normalize=True, fused
        into the first convolution kernel"""
        if self.perceptual_weight == 0 or not hasattr(self, "perceptual_loss"):
            return torch.zeros((), device=reconstructions.device)
        return self.perceptual_loss(inputs, reconstructions, normalize=True).mean()


class VQLPIPSWithDiscriminator(VQLPIPS):
    """Reference vqperceptual.py:59-172: generator-side loss (optimizer_idx 0: pixel + [LPIPS] + disc_factor * adversarial_weight *
    g_loss + codebook) and discriminator-side loss (optimizer_idx 1: d_loss on real / detached fake + lazy R1 every `do_r1_every`
    batches, differentiated through the discriminator's backward).  The StyleGAN2 discriminator runs on this library's HIP kernels
    (losses/layers.py), the LPIPS term on those of losses/lpips.py.  Deviation: with
    adversarial_weight == 0 no discriminator is built (the reference would build and train one whose output never reaches the
    autoencoder), so such configs keep the single-optimizer fused step."""

    def __init__(self, disc_start: int = 0, disc_loss: str = "vanilla", disc_params=None, codebook_weight: float = 1.0,
                 loglaplace_weight: float = 1.0, loggaussian_weight: float = 1.0, perceptual_weight: float = 1.0,
                 adversarial_weight: float = 1.0, use_adaptive_adv: bool = False, r1_gamma: float = 10, do_r1_every: int = 16) -> None:
        super().__init__(codebook_weight, loglaplace_weight, loggaussian_weight, perceptual_weight)
        from .layers import StyleDiscriminator, hinge_d_loss, least_square_d_loss, vanilla_d_loss
        assert disc_loss in ["hinge", "vanilla", "least_square"], f"Unknown GAN loss '{disc_loss}'."
        if adversarial_weight != 0:
            self.discriminator = StyleDiscriminator(**dict(disc_params or {}))
        self.disc_loss = {"hinge": hinge_d_loss, "vanilla": vanilla_d_loss, "least_square": least_square_d_loss}[disc_loss]
        self.discriminator_iter_start = disc_start
        self.adversarial_weight = adversarial_weight
        self.use_adaptive_adv = use_adaptive_adv
        self.r1_gamma = r1_gamma
        self.do_r1_every = do_r1_every
        self._disc_store = None

    def calculate_adaptive_factor(self, nll_loss: torch.Tensor, g_loss: torch.Tensor, last_layer, reconstructions: torch.Tensor) -> torch.Tensor:
        """reference vqperceptual.py:95-103: ||d nll / d last_layer|| / (||d g_loss / d last_layer|| + 1e-4), clamped to [0, 1e4], detached.  The decoder's
        last layer is not an autograd leaf of the fused engine, so each norm is obtained from the gradient at the reconstruction (autograd) times the
        saved last-layer input (one small GEMM in the engine, Stage1Engine.last_layer_grad_norm)."""
        eng = getattr(last_layer, "_enh_engine", None)
        if eng is None:
            raise RuntimeError("use_adaptive_adv: last_layer must be ViTDecoder.get_last_layer() of a model bound to the HIP engine")
        sc = self.loss_scaler(reconstructions.device) if hasattr(self, "discriminator") else None      # (fp16 loss networks: both probes run scaled)
        if sc is not None and sc.enabled:
            nll_g = sc.unscale(torch.autograd.grad(sc.scale(nll_loss), reconstructions, retain_graph=True)[0])
            g_g = sc.unscale(torch.autograd.grad(sc.scale(g_loss), reconstructions, retain_graph=True)[0])
        else:
            nll_g, = torch.autograd.grad(nll_loss, reconstructions, retain_graph=True)
            g_g, = torch.autograd.grad(g_loss, reconstructions, retain_graph=True)
        adapt = eng.last_layer_grad_norm(nll_g) / (eng.last_layer_grad_norm(g_g) + 1e-4)
        return adapt.clamp(0.0, 1e4).detach()

    # the discriminator's parameters live in one flat fp32 buffer (fused AdamW, one all-reduce), created on first use
    def disc_store(self, device: torch.device):
        if self._disc_store is None:
            from ..engine.stage1 import ParamStore
            self.discriminator.to(device)
            self._disc_store = ParamStore(self.discriminator, device, precision="fp32")
            from ..engine.optim import LossScaler
            self._disc_store.loss_scaler = LossScaler(device)
        return self._disc_store

    def loss_scaler(self, device: torch.device):
        """the loss networks' own scale (engine/optim.py LossScaler): enabled while their 16-bit operands are fp16 (ENH_LOSS_OPERANDS=fp16 /
        conv_nhwc.operand_dtype / conv2d_gradfix.operand_dtype), the identity under bf16"""
        from .op import conv2d_gradfix, conv_nhwc
        sc = self.disc_store(device).loss_scaler
        sc.enabled = bool(conv_nhwc.OPERAND_DTYPE == torch.float16 or conv2d_gradfix._OPERAND == torch.float16)
        return sc

    def scale_disc_loss(self, d_loss: torch.Tensor) -> torch.Tensor:
        """what the caller runs .backward() on for optimizer 1 (torch.cuda.amp's `scaler.scale(loss).backward()`); FlatAdamW.step unscales"""
        return self.disc_store(d_loss.device).loss_scaler.scale(d_loss)      # (`enabled` as the forward that built d_loss left it)

    def _forward(self, codebook_loss: torch.Tensor, inputs: torch.Tensor, reconstructions: torch.Tensor, optimizer_idx: int,
                 global_step: int, batch_idx: int, last_layer: Optional[nn.Module] = None, split: Optional[str] = "train") -> Tuple:
        if not hasattr(self, "discriminator"):
            if optimizer_idx == 0:
                return super()._forward(codebook_loss, inputs, reconstructions, optimizer_idx, global_step, batch_idx, last_layer, split)
            return None, {}
        from .op import conv2d_gradfix, conv_nhwc
        inputs = inputs.contiguous()
        reconstructions = reconstructions.contiguous()
        self.disc_store(reconstructions.device)
        sc = self.loss_scaler(reconstructions.device)      # (sets `enabled` for THIS forward's operand format: scale_disc_loss / FlatAdamW.step read it)
        if optimizer_idx == 0:
            # packed operand images of the discriminator's weights are reused for every pass of ONE training step (the weights only change in the
            # discriminator's optimizer step, which invalidates them itself); dropping them here as well covers writers that bypass both the
            # autograd version counter and that optimizer (a broadcast into the flat buffer, `param.data` assignments)
            conv_nhwc.invalidate_packed_weights()
        disc_factor = 1 if global_step >= self.discriminator_iter_start else 0

        if optimizer_idx == 0:   # generator update (vqperceptual.py:111-146)
            diff = reconstructions - inputs
            loglaplace_loss = diff.abs().mean()
            loggaussian_loss = diff.pow(2).mean()
            perceptual_loss = self._perceptual(inputs, reconstructions)
            nll_loss = self.loglaplace_weight * loglaplace_loss + self.loggaussian_weight * loggaussian_loss + self.perceptual_weight * perceptual_loss
            logits_fake = self.discriminator(reconstructions)
            g_loss = self.disc_loss(logits_fake)
            d_weight = self.adversarial_weight
            if self.use_adaptive_adv:
                if reconstructions.requires_grad:
                    d_weight = d_weight * self.calculate_adaptive_factor(nll_loss, g_loss, last_layer, reconstructions)
                else:                                  # the reference's `except RuntimeError: assert not self.training; d_weight = 0` (vqperceptual.py:127-129)
                    assert not self.training
                    d_weight = torch.tensor(0.0, device=reconstructions.device)
            loss = nll_loss + disc_factor * d_weight * g_loss + self.codebook_weight * codebook_loss
            log = {"{}/total_loss".format(split): loss.clone().detach(),
                   "{}/quant_loss".format(split): codebook_loss.detach(),
                   "{}/rec_loss".format(split): nll_loss.detach(),
                   "{}/loglaplace_loss".format(split): loglaplace_loss.detach(),
                   "{}/loggaussian_loss".format(split): loggaussian_loss.detach(),
                   "{}/perceptual_loss".format(split): perceptual_loss.detach(),
                   "{}/g_loss".format(split): g_loss.detach()}
            if self.use_adaptive_adv:
                log["{}/d_weight".format(split)] = d_weight.detach() if torch.is_tensor(d_weight) else torch.tensor(float(d_weight))
            return loss, log

        if optimizer_idx == 1:   # discriminator update (vqperceptual.py:148-172)
            do_r1 = self.training and bool(disc_factor) and batch_idx % self.do_r1_every == 0 and torch.is_grad_enabled()
            real = inputs.detach().clone().requires_grad_(do_r1)
            logits_real = self.discriminator(real)
            logits_fake = self.discriminator(reconstructions.detach())
            d_loss = disc_factor * self.disc_loss(logits_fake, logits_real)
            if do_r1:
                with conv2d_gradfix.no_weight_gradients():       # (fp16 operands: the first-order pass of R1 runs on scale x sum(logits) and is divided back in f32)
                    gradients, = torch.autograd.grad(outputs=sc.scale(logits_real.sum()), inputs=real, create_graph=True)
                gradients = sc.unscale(gradients)
                gradients_norm = gradients.square().sum([1, 2, 3]).mean()
                d_loss = d_loss + self.r1_gamma * self.do_r1_every * gradients_norm / 2
            log = {"{}/disc_loss".format(split): d_loss.detach() if torch.is_tensor(d_loss) else torch.tensor(float(d_loss)),
                   "{}/logits_real".format(split): logits_real.detach().mean(),
                   "{}/logits_fake".format(split): logits_fake.detach().mean()}
            if do_r1:
                log["{}/r1_reg".format(split)] = gradients_norm.detach()
            return d_loss, log
