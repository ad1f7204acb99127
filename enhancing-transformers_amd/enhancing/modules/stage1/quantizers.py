"""Vector / residual quantizer of the stage-1 tokenizer with the reference's constructor and
``forward(z) -> (z_q, loss, indices)`` contract (reference enhancing/modules/stage1/quantizers.py:19-92).

The arithmetic — l2-normalise, pairwise distance against the whole codebook, argmin, gather, commit/codebook
loss, the shared-codebook residual loop and the straight-through estimator — is ONE fused gfx950 kernel
(``enh_vq_forward``) plus its hand-derived backward (``enh_vq_backward``), wrapped in a
``torch.autograd.Function`` so the module composes with ordinary autograd.  "RQ-VAE" is
``use_residual=True, num_quantizers=D`` on this same class, exactly as in the reference."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from ... import _C


class _QuantizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, codebook, beta, depth, use_residual, use_norm):
        shp = z.shape
        z2 = z.detach().reshape(-1, shp[-1]).contiguous()
        cb = codebook.detach().contiguous()
        zq, _, idx, loss = _C.vq_forward(z2, cb, beta, depth, use_norm, want_bf16=False)
        ctx.save_for_backward(z2, cb, idx)
        ctx.cfg = (beta, depth, use_residual, use_norm, shp)
        idx_out = idx.view(*shp[:-1], depth) if use_residual else idx.view(*shp[:-1])
        ctx.mark_non_differentiable(idx_out)
        return zq.view(shp), loss.view(()), idx_out

    @staticmethod
    def backward(ctx, g_zq, g_loss, _g_idx):
        z2, cb, idx = ctx.saved_tensors
        beta, depth, use_residual, use_norm, shp = ctx.cfg
        if g_zq is None:
            g_zq = torch.zeros(shp, dtype=torch.float32, device=z2.device)
        g_out = g_zq.reshape(-1, shp[-1]).contiguous().float()
        d_cb = torch.zeros_like(cb)
        g_loss_dev = None if g_loss is None else g_loss.reshape(1).float().contiguous()
        dz, _ = _C.vq_backward(z2, cb, idx, g_out, 1.0 if g_loss is not None else 0.0, g_loss_dev, beta, depth, use_residual,
                               use_norm, d_cb, want_bf16=False)
        return dz.view(shp), d_cb, None, None, None, None


class _LookupFn(torch.autograd.Function):
    """z_qnorm = norm(E[idx]) through enh_vq_lookup; backward = the l2-normalise Jacobian scattered into the codebook gradient."""

    @staticmethod
    def forward(ctx, codebook, idx, use_norm):
        cb = codebook.detach().contiguous()
        flat = idx.reshape(-1, 1).contiguous()
        out, _ = _C.vq_lookup(cb, flat, use_norm, want_bf16=False)
        ctx.save_for_backward(cb, flat)
        ctx.use_norm = use_norm
        ctx.mark_non_differentiable(idx)
        return out.view(*idx.shape, cb.shape[1])

    @staticmethod
    def backward(ctx, g):
        cb, flat = ctx.saved_tensors
        e = cb[flat.view(-1)]
        g = g.reshape(-1, cb.shape[1]).float()
        if ctx.use_norm:
            nrm = e.norm(dim=-1, keepdim=True).clamp_min(1e-12)
            en = e / nrm
            g = (g - en * (en * g).sum(-1, keepdim=True)) / nrm
        d_cb = torch.zeros_like(cb).index_add_(0, flat.view(-1), g)
        return d_cb, None, None


class BaseQuantizer(nn.Module):
    def __init__(self, embed_dim: int, n_embed: int, straight_through: bool = True, use_norm: bool = True,
                 use_residual: bool = False, num_quantizers: Optional[int] = None) -> None:
        super().__init__()
        self.straight_through = straight_through
        self.use_norm = use_norm
        self.norm = (lambda x: torch.nn.functional.normalize(x, dim=-1)) if use_norm else (lambda x: x)
        self.use_residual = use_residual
        self.num_quantizers = num_quantizers
        self.embed_dim = embed_dim
        self.n_embed = n_embed
        self.embedding = nn.Embedding(self.n_embed, self.embed_dim)
        self.embedding.weight.data.normal_()  # quantizers.py:33


class VectorQuantizer(BaseQuantizer):
    """reference quantizers.py:66-92.  embed_dim must be 32 (the fused kernel's MFMA tiling; every reference
    config uses 32)."""

    def __init__(self, embed_dim: int, n_embed: int, beta: float = 0.25, use_norm: bool = True,
                 use_residual: bool = False, num_quantizers: Optional[int] = None, **kwargs) -> None:
        super().__init__(embed_dim, n_embed, True, use_norm, use_residual, num_quantizers)
        if embed_dim != 32:
            raise ValueError("the fused gfx950 quantizer kernel requires embed_dim == 32")
        if use_residual and not num_quantizers:
            raise ValueError("use_residual=True needs num_quantizers")
        if use_residual and int(num_quantizers) > 8:
            raise ValueError("the fused gfx950 residual quantizer keeps at most 8 depths in registers (enh_vq_backward): num_quantizers <= 8")
        self.beta = beta

    @property
    def depth(self) -> int:
        return int(self.num_quantizers) if self.use_residual else 1

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return _QuantizeFn.apply(z, self.embedding.weight, float(self.beta), self.depth, bool(self.use_residual), bool(self.use_norm))

    def quantize(self, z: torch.Tensor):
        """single-level quantize (reference quantizers.py:74-92): (z_qnorm, loss, indices) where z_qnorm = norm(embedding(indices)) — the
        NORMALISED CODE itself, not the straight-through value forward() returns (quantizers.py:85-92).  Indices and loss come from the fused
        kernel; z_qnorm from the lookup kernel, differentiable with respect to the codebook like the reference's (normalise Jacobian +
        scatter-add, evaluated with torch ops in backward: this entry point is API surface, not the training hot path)."""
        _, loss, idx = _QuantizeFn.apply(z, self.embedding.weight, float(self.beta), 1, False, bool(self.use_norm))
        zqn = _LookupFn.apply(self.embedding.weight, idx, bool(self.use_norm))
        return zqn.view(z.shape), loss, idx

    def lookup(self, code: torch.Tensor) -> torch.Tensor:
        """decode_codes front half (reference vitvqgan.py:82-87): n(E[code]) summed over the depth axis."""
        depth = self.depth
        shp = code.shape[:-1] if self.use_residual else code.shape
        out, _ = _C.vq_lookup(self.embedding.weight.detach().contiguous(), code.reshape(-1, depth).contiguous(), self.use_norm,
                              want_bf16=False)
        return out.view(*shp, self.embed_dim)


class GumbelQuantizer(BaseQuantizer):
    """Gumbel-softmax relaxation of the codebook lookup (reference quantizers.py:95-126; used by ViTVQGumbel).  Outside the MI355X hot path (no shipped
    stage-1 config selects it): plain PyTorch on the device tensors, differentiated by torch autograd between the two
    halves of the HIP schedule (Stage1Engine.differentiable_encode / _decode).  Semantics restated from the reference:
      logits_k = -(|zn|^2 + |en_k|^2 - 2 zn.en_k)      soft = gumbel_softmax(logits, tau, hard = not training)      z_q = soft @ en
      loss = mean_tokens sum_k p_k (log p_k + log K), p = softmax(logits)  (KL to the uniform prior)            indices = argmax soft
    and BaseQuantizer.forward's residual loop without the straight-through estimator (quantizers.py:38-63, straight_through = False)."""

    def __init__(self, embed_dim: int, n_embed: int, temp_init: float = 1.0, use_norm: bool = True, use_residual: bool = False,
                 num_quantizers: Optional[int] = None, **kwargs) -> None:
        super().__init__(embed_dim, n_embed, False, use_norm, use_residual, num_quantizers)
        if use_residual and not num_quantizers:
            raise ValueError("use_residual=True needs num_quantizers")
        self.temperature = temp_init

    @property
    def depth(self) -> int:
        return int(self.num_quantizers) if self.use_residual else 1

    def quantize(self, z: torch.Tensor, temp: Optional[float] = None):
        import math
        tau = self.temperature if temp is None else temp
        zn = self.norm(z.reshape(-1, self.embed_dim))
        en = self.norm(self.embedding.weight)
        logits = (2.0 * zn @ en.t() - zn.pow(2).sum(1, keepdim=True) - en.pow(2).sum(1)).view(*z.shape[:-1], self.n_embed)
        soft = torch.nn.functional.gumbel_softmax(logits, tau=tau, dim=-1, hard=not self.training)
        z_q = soft @ en
        logp = torch.log_softmax(logits, dim=-1)
        loss = (logp.exp() * (logp + math.log(self.n_embed))).sum(-1).mean()
        return z_q, loss, soft.argmax(dim=-1)

    def forward(self, z: torch.Tensor):
        if not self.use_residual:
            return self.quantize(z)
        z_q, residual = torch.zeros_like(z), z.detach().clone()
        losses, idxs = [], []
        for _ in range(int(self.num_quantizers)):
            z_qi, l_i, i_i = self.quantize(residual.clone())
            residual = residual - z_qi
            z_q = z_q + z_qi
            losses.append(l_i)
            idxs.append(i_i)
        return z_q, torch.stack(losses, dim=-1).mean(), torch.stack(idxs, dim=-1)

    def lookup(self, code: torch.Tensor) -> torch.Tensor:
        """decode_codes front half (reference vitvqgan.py:82-87): n(E[code]), summed over the depth axis when residual"""
        q = self.norm(torch.nn.functional.embedding(code, self.embedding.weight))
        return q.sum(-2) if self.use_residual else q
