"""Stage-1 loss modules with the reference's names, constructor arguments, call signature and log keys
(reference enhancing/losses/vqperceptual.py:17-172).

In scope this round: the pixel terms (L1 / L2) and the codebook term — rows a20 of SURVEY.md §8.  The LPIPS
perceptual term (third-party ``lpips`` + un-obtainable pretrained VGG16 weights) and the StyleGAN
discriminator branch are "next" rows: constructing a loss with a non-zero weight for either
raises, so a config can never silently train with a term missing.  ``ENH_ALLOW_MISSING_TERMS=1`` turns the
error into a warning and treats the missing terms as zero (used to load the reference's yaml files as they are)."""
from __future__ import annotations

import os
import warnings
from typing import Optional, Tuple

import torch
import torch.nn as nn


def _missing(term: str, weight: float) -> None:
    if weight == 0:
        return
    msg = (f"{term} (weight {weight}) is not implemented in this round (SURVEY.md §8f); it would be treated as 0. "
           f"Set the weight to 0 or export ENH_ALLOW_MISSING_TERMS=1 to proceed without it.")
    if os.environ.get("ENH_ALLOW_MISSING_TERMS", "0") == "1":
        warnings.warn(msg)
    else:
        raise NotImplementedError(msg)


class DummyLoss(nn.Module):
    def __init__(self) -> None:
        super().__init__()


class VQLPIPS(nn.Module):
    """loss = w_l1*L1 + w_l2*L2 + w_p*LPIPS + codebook_weight*codebook_loss (vqperceptual.py:22-56)."""

    def __init__(self, codebook_weight: float = 1.0, loglaplace_weight: float = 1.0, loggaussian_weight: float = 1.0,
                 perceptual_weight: float = 1.0) -> None:
        super().__init__()
        _missing("LPIPS perceptual loss", perceptual_weight)
        self.codebook_weight = codebook_weight
        self.loglaplace_weight = loglaplace_weight
        self.loggaussian_weight = loggaussian_weight
        self.perceptual_weight = perceptual_weight

    def forward(self, codebook_loss: torch.Tensor, inputs: torch.Tensor, reconstructions: torch.Tensor, optimizer_idx: int,
                global_step: int, batch_idx: int, last_layer: Optional[nn.Module] = None, split: Optional[str] = "train") -> Tuple:
        inputs = inputs.contiguous()
        reconstructions = reconstructions.contiguous()
        diff = reconstructions - inputs
        loglaplace_loss = diff.abs().mean()
        loggaussian_loss = diff.pow(2).mean()
        perceptual_loss = torch.zeros((), device=diff.device)
        nll_loss = self.loglaplace_weight * loglaplace_loss + self.loggaussian_weight * loggaussian_loss
        loss = nll_loss + self.codebook_weight * codebook_loss
        log = {"{}/total_loss".format(split): loss.clone().detach(),
               "{}/quant_loss".format(split): codebook_loss.detach(),
               "{}/rec_loss".format(split): nll_loss.detach(),
               "{}/loglaplace_loss".format(split): loglaplace_loss.detach(),
               "{}/loggaussian_loss".format(split): loggaussian_loss.detach(),
               "{}/perceptual_loss".format(split): perceptual_loss.detach()}
        return loss, log


class VQLPIPSWithDiscriminator(VQLPIPS):
    """Generator-side loss of vqperceptual.py:59-146 with the adversarial term gated the same way; the
    discriminator itself (StyleGAN2 D + R1) is a 'next' row, so adversarial_weight must be 0 this round."""

    def __init__(self, disc_start: int = 0, codebook_weight: float = 1.0, loglaplace_weight: float = 1.0,
                 loggaussian_weight: float = 1.0, perceptual_weight: float = 1.0, adversarial_weight: float = 1.0,
                 use_adaptive_adv: bool = False, r1_gamma: float = 10, do_r1_every: int = 16) -> None:
        super().__init__(codebook_weight, loglaplace_weight, loggaussian_weight, perceptual_weight)
        _missing("StyleGAN discriminator / adversarial loss", adversarial_weight)
        self.discriminator_iter_start = disc_start
        self.adversarial_weight = adversarial_weight
        self.use_adaptive_adv = use_adaptive_adv
        self.r1_gamma = r1_gamma
        self.do_r1_every = do_r1_every
