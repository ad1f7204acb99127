"""CPU ORACLE — test infrastructure, NOT product code.

A plain-PyTorch, fp32, CPU restatement of the stage-1 ViT-VQGAN / RQ-VAE hot path of
thuanz123/enhancing-transformers.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the shipped package
(``enhancing-transformers_amd/``) never does.

Every function cites the reference lines it restates (paths relative to the reference
repository root).  The restatement is *functional* (weights are passed in as a flat
``dict`` that uses the reference's state-dict key names) so the same tensors can be
fed to the HIP path and to this oracle.

Pinning: the reference ships no tests or golden vectors, so the pin
is the reference's own modules imported from /root/reference in the build container:
``oracle/make_golden.py`` asserts this restatement == the reference modules and writes
the golden vectors under ``tests/golden/``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------
# position embedding  — enhancing/modules/stage1/layers.py:21-68
# ----------------------------------------------------------------------------------------
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """layers.py:50-68 (with the np.float -> float64 fix)."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_h: int, grid_w: int) -> np.ndarray:
    """layers.py:21-47.  NB: meshgrid(grid_w, grid_h) -> grid[0] is the *w* coordinate and it
    fills the FIRST half of the channels (layers.py:30,43-44)."""
    gh = np.arange(grid_h, dtype=np.float32)
    gw = np.arange(grid_w, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_h, grid_w)
    emb_a = sincos_1d(embed_dim // 2, grid[0])
    emb_b = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_a, emb_b], axis=1)  # [grid_h*grid_w, embed_dim] float64


def pos_embedding(dim: int, grid_h: int, grid_w: int) -> Tensor:
    """layers.py:172,201: torch.from_numpy(...).float().unsqueeze(0) -> [1, N, dim] fp32."""
    return torch.from_numpy(sincos_2d(dim, grid_h, grid_w)).float().unsqueeze(0)


# ----------------------------------------------------------------------------------------
# transformer blocks — layers.py:85-150
# ----------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.LayerNorm(dim) defaults: eps 1e-5, biased variance (layers.py:88,143)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def attention(x: Tensor, w_qkv: Tensor, w_out: Tensor, b_out: Tensor, heads: int, dim_head: int = 64) -> Tensor:
    """layers.py:122-132: bias-free qkv, chunk(3) = q,k,v, '(h d)' head-major, softmax(q k^T * d^-0.5) v."""
    B, N, _ = x.shape
    qkv = x @ w_qkv.t()
    q, k, v = qkv.chunk(3, dim=-1)
    sp = lambda t: t.reshape(B, N, heads, dim_head).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    att = torch.softmax((q @ k.transpose(-1, -2)) * dim_head ** -0.5, dim=-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(B, N, heads * dim_head)
    return out @ w_out.t() + b_out


def feed_forward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """layers.py:95-105: Linear -> Tanh -> Linear."""
    return torch.tanh(x @ w1.t() + b1) @ w2.t() + b2


def transformer(x: Tensor, P: Dict[str, Tensor], prefix: str, depth: int, heads: int, trace: Optional[list] = None) -> Tensor:
    """layers.py:145-150: depth x { x = attn(LN(x)) + x ; x = ff(LN(x)) + x } ; final LN.
    ``trace`` (test aid): receives the residual stream entering layer 0 and leaving every layer, for per-layer parity tables."""
    if trace is not None:
        trace.append(x.detach())
    for i in range(depth):
        p = f"{prefix}layers.{i}."
        h = layer_norm(x, P[p + "0.norm.weight"], P[p + "0.norm.bias"])
        x = attention(h, P[p + "0.fn.to_qkv.weight"], P[p + "0.fn.to_out.weight"], P[p + "0.fn.to_out.bias"], heads) + x
        h = layer_norm(x, P[p + "1.norm.weight"], P[p + "1.norm.bias"])
        x = feed_forward(h, P[p + "1.fn.net.0.weight"], P[p + "1.fn.net.0.bias"],
                         P[p + "1.fn.net.2.weight"], P[p + "1.fn.net.2.bias"]) + x
        if trace is not None:
            trace.append(x.detach())
    return layer_norm(x, P[prefix + "norm.weight"], P[prefix + "norm.bias"])


# ----------------------------------------------------------------------------------------
# encoder / decoder — layers.py:153-217
# ----------------------------------------------------------------------------------------
def patchify(img: Tensor, patch: int) -> Tensor:
    """Conv2d(k=s=patch) + 'b c h w -> b (h w) c' as a GEMM operand: [B, N, C*patch*patch] with the
    per-patch element order (c, ph, pw) of the conv weight [dim, C, patch, patch] (layers.py:168-171,178)."""
    B, C, H, W = img.shape
    gh, gw = H // patch, W // patch
    x = img.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


def unpatchify(p: Tensor, patch: int, C: int, H: int, W: int) -> Tensor:
    """Inverse of patchify: the scatter that ConvTranspose2d(k=s=patch) performs (layers.py:202-205,212)."""
    B = p.shape[0]
    gh, gw = H // patch, W // patch
    x = p.reshape(B, gh, gw, C, patch, patch).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, C, H, W)


def encoder(img: Tensor, P: Dict[str, Tensor], cfg: dict, prefix: str = "encoder.", trace: Optional[list] = None) -> Tensor:
    """ViTEncoder.forward layers.py:177-182."""
    patch = cfg["patch_size"]
    w = P[prefix + "to_patch_embedding.0.weight"]  # [dim, C, p, p]
    x = patchify(img, patch) @ w.reshape(w.shape[0], -1).t() + P[prefix + "to_patch_embedding.0.bias"]
    x = x + P[prefix + "en_pos_embedding"]
    return transformer(x, P, prefix + "transformer.", cfg["encoder"]["depth"], cfg["encoder"]["heads"], trace)


def decoder(tok: Tensor, P: Dict[str, Tensor], cfg: dict, prefix: str = "decoder.", trace: Optional[list] = None) -> Tensor:
    """ViTDecoder.forward layers.py:209-214.  ConvTranspose2d weight is [dim, C, p, p] = [K, N] (x @ W)."""
    patch, size = cfg["patch_size"], cfg["image_size"]
    x = tok + P[prefix + "de_pos_embedding"]
    x = transformer(x, P, prefix + "transformer.", cfg["decoder"]["depth"], cfg["decoder"]["heads"], trace)
    w = P[prefix + "to_pixel.1.weight"]
    C = w.shape[1]
    pix = x @ w.reshape(w.shape[0], -1) + P[prefix + "to_pixel.1.bias"].repeat_interleave(patch * patch)
    return unpatchify(pix, patch, C, size, size)


# ----------------------------------------------------------------------------------------
# quantizer — enhancing/modules/stage1/quantizers.py:38-92
# ----------------------------------------------------------------------------------------
def l2norm(x: Tensor) -> Tensor:
    """quantizers.py:24: F.normalize(x, dim=-1) = x / max(||x||_2, 1e-12)."""
    return F.normalize(x, dim=-1)


def vq_distances(zn: Tensor, en: Tensor) -> Tensor:
    """quantizers.py:78-80, the three-term form, evaluated exactly as written."""
    return torch.sum(zn ** 2, dim=1, keepdim=True) + torch.sum(en ** 2, dim=1) - 2 * torch.einsum("bd,nd->bn", zn, en)


def vq_quantize(z: Tensor, E: Tensor, beta: float = 0.25, use_norm: bool = True, force_idx: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """VectorQuantizer.quantize quantizers.py:74-92 -> (z_qnorm, loss, indices[z.shape[:-1]] int64).
    ``force_idx`` (test aid): skip the argmin and use the given indices, so that a mixed-precision path and this fp32 restatement
    can be compared downstream of the SAME discrete decisions (arithmetic error separated from near-tie index flips)."""
    norm = l2norm if use_norm else (lambda t: t)
    d_e = E.shape[1]
    zn = norm(z.reshape(-1, d_e))
    en = norm(E)
    idx = torch.argmin(vq_distances(zn, en), dim=1).view(*z.shape[:-1]) if force_idx is None else force_idx.view(*z.shape[:-1])
    zq = F.embedding(idx, E).view(z.shape)
    zqn, zn2 = norm(zq), norm(z)
    loss = beta * torch.mean((zqn.detach() - zn2) ** 2) + torch.mean((zqn - zn2.detach()) ** 2)
    return zqn, loss, idx


def quantizer_forward(z: Tensor, E: Tensor, beta: float = 0.25, use_norm: bool = True,
                      use_residual: bool = False, num_quantizers: Optional[int] = None, force_idx: Optional[Tensor] = None):
    """BaseQuantizer.forward quantizers.py:38-63 (residual loop + straight-through).  force_idx: see vq_quantize ([..., depth] if residual)."""
    if not use_residual:
        z_q, loss, idx = vq_quantize(z, E, beta, use_norm, force_idx)
    else:
        z_q = torch.zeros_like(z)
        residual = z.detach().clone()
        losses, idxs = [], []
        for d_ in range(num_quantizers):
            z_qi, l_i, i_i = vq_quantize(residual.clone(), E, beta, use_norm, None if force_idx is None else force_idx[..., d_])
            residual.sub_(z_qi)          # in place, as quantizers.py:50 (creates the cross-depth grad path)
            z_q.add_(z_qi)
            idxs.append(i_i)
            losses.append(l_i)
        loss = torch.stack(losses, dim=-1).mean()
        idx = torch.stack(idxs, dim=-1)
    z_q = z + (z_q - z).detach()          # quantizers.py:61
    return z_q, loss, idx


# ----------------------------------------------------------------------------------------
# ViTVQ composition — enhancing/modules/stage1/vitvqgan.py:44-90
# ----------------------------------------------------------------------------------------
def qparams(cfg: dict) -> dict:
    q = cfg["quantizer"]
    return dict(beta=q.get("beta", 0.25), use_norm=q.get("use_norm", True),
                use_residual=q.get("use_residual", False), num_quantizers=q.get("num_quantizers", None))


def encode(img: Tensor, P: Dict[str, Tensor], cfg: dict):
    """vitvqgan.py:61-66 (returns h as well, for op-boundary parity)."""
    h = encoder(img, P, cfg) @ P["pre_quant.weight"].t() + P["pre_quant.bias"]
    quant, emb_loss, idx = quantizer_forward(h, P["quantizer.embedding.weight"], **qparams(cfg))
    return quant, emb_loss, idx, h


def decode(quant: Tensor, P: Dict[str, Tensor], cfg: dict) -> Tensor:
    """vitvqgan.py:68-72."""
    return decoder(quant @ P["post_quant.weight"].t() + P["post_quant.bias"], P, cfg)


def forward(img: Tensor, P: Dict[str, Tensor], cfg: dict):
    """vitvqgan.py:44-48 -> (dec, diff)."""
    quant, diff, idx, h = encode(img, P, cfg)
    return decode(quant, P, cfg), diff


def encode_codes(img: Tensor, P: Dict[str, Tensor], cfg: dict) -> Tensor:
    """vitvqgan.py:74-79."""
    return encode(img, P, cfg)[2]


def decode_codes(code: Tensor, P: Dict[str, Tensor], cfg: dict) -> Tensor:
    """vitvqgan.py:81-90."""
    q = qparams(cfg)
    quant = F.embedding(code, P["quantizer.embedding.weight"])
    if q["use_norm"]:
        quant = l2norm(quant)
    if q["use_residual"]:
        quant = quant.sum(-2)
    return decode(quant, P, cfg)


# ----------------------------------------------------------------------------------------
# loss (pixel + codebook terms) — enhancing/losses/vqperceptual.py:41-56,113-144
# ----------------------------------------------------------------------------------------
def pixel_codebook_loss(codebook_loss: Tensor, inputs: Tensor, recon: Tensor, loglaplace_weight: float = 0.0,
                        loggaussian_weight: float = 1.0, codebook_weight: float = 1.0, split: str = "train"):
    """VQLPIPS / VQLPIPSWithDiscriminator generator branch with perceptual and adversarial weights 0
    (LPIPS needs un-obtainable pretrained weights: SURVEY.md §8c).  vqperceptual.py:113-114,117,131-141."""
    l1 = (recon - inputs).abs().mean()
    l2 = (recon - inputs).pow(2).mean()
    nll = loglaplace_weight * l1 + loggaussian_weight * l2
    loss = nll + codebook_weight * codebook_loss
    log = {f"{split}/total_loss": loss.detach().clone(), f"{split}/quant_loss": codebook_loss.detach(),
           f"{split}/rec_loss": nll.detach(), f"{split}/loglaplace_loss": l1.detach(),
           f"{split}/loggaussian_loss": l2.detach()}
    return loss, log


# ----------------------------------------------------------------------------------------
# parameters — init (layers.py:71-82,175,207; quantizers.py:32-33; vitvqgan.py:38-39) and AdamW
# ----------------------------------------------------------------------------------------
def param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """State-dict contract of SURVEY.md §8(b) (keys and shapes of the reference's ViTVQ minus loss.*)."""
    p, size = cfg["patch_size"], cfg["image_size"]
    n_tok = (size // p) ** 2
    C = cfg.get("channels", 3)
    out: Dict[str, Tuple[int, ...]] = {}

    def tower(prefix: str, c: dict):
        dim, inner, mlp = c["dim"], 64 * c["heads"], c["mlp_dim"]
        for i in range(c["depth"]):
            q = f"{prefix}transformer.layers.{i}."
            out[q + "0.norm.weight"] = (dim,); out[q + "0.norm.bias"] = (dim,)
            out[q + "0.fn.to_qkv.weight"] = (3 * inner, dim)
            out[q + "0.fn.to_out.weight"] = (dim, inner); out[q + "0.fn.to_out.bias"] = (dim,)
            out[q + "1.norm.weight"] = (dim,); out[q + "1.norm.bias"] = (dim,)
            out[q + "1.fn.net.0.weight"] = (mlp, dim); out[q + "1.fn.net.0.bias"] = (mlp,)
            out[q + "1.fn.net.2.weight"] = (dim, mlp); out[q + "1.fn.net.2.bias"] = (dim,)
        out[prefix + "transformer.norm.weight"] = (dim,); out[prefix + "transformer.norm.bias"] = (dim,)

    e, d, q = cfg["encoder"], cfg["decoder"], cfg["quantizer"]
    out["encoder.en_pos_embedding"] = (1, n_tok, e["dim"])
    out["encoder.to_patch_embedding.0.weight"] = (e["dim"], C, p, p)
    out["encoder.to_patch_embedding.0.bias"] = (e["dim"],)
    tower("encoder.", e)
    tower("decoder.", d)
    out["decoder.de_pos_embedding"] = (1, n_tok, d["dim"])
    out["decoder.to_pixel.1.weight"] = (d["dim"], C, p, p)
    out["decoder.to_pixel.1.bias"] = (C,)
    out["pre_quant.weight"] = (q["embed_dim"], e["dim"]); out["pre_quant.bias"] = (q["embed_dim"],)
    out["post_quant.weight"] = (d["dim"], q["embed_dim"]); out["post_quant.bias"] = (d["dim"],)
    out["quantizer.embedding.weight"] = (q["n_embed"], q["embed_dim"])
    return out


def make_params(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic, version-stable parameters (numpy MT19937, NOT torch's RNG) following the reference's
    init *distributions*: xavier-uniform Linear/conv weights, zero biases, LN (1, 0) (layers.py:71-82),
    N(0,1) codebook (quantizers.py:33), torch-default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) pre/post_quant
    (vitvqgan.py:38-39), fixed sin-cos position tables.  LN weights / biases get a small perturbation so
    that gradient tests see non-trivial values."""
    rs = np.random.RandomState(seed)
    p, size = cfg["patch_size"], cfg["image_size"]
    g = size // p
    P: Dict[str, Tensor] = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("pos_embedding"):
            P[k] = pos_embedding(shp[2], g, g)
        elif k == "quantizer.embedding.weight":
            P[k] = torch.from_numpy(rs.standard_normal(shp).astype(np.float32))
        elif k.startswith("pre_quant") or k.startswith("post_quant"):
            fan_in = cfg["encoder"]["dim"] if k.startswith("pre_quant") else cfg["quantizer"]["embed_dim"]
            b = 1.0 / math.sqrt(fan_in)
            P[k] = torch.from_numpy(rs.uniform(-b, b, shp).astype(np.float32))
        elif k.endswith("norm.weight"):
            P[k] = torch.from_numpy((1.0 + 0.05 * rs.standard_normal(shp)).astype(np.float32))
        elif k.endswith("bias"):
            P[k] = torch.from_numpy((0.02 * rs.standard_normal(shp)).astype(np.float32))
        else:  # xavier-uniform on weight viewed [shape[0], -1]
            fan_out, fan_in = shp[0], int(np.prod(shp[1:]))
            b = math.sqrt(6.0 / (fan_in + fan_out))
            P[k] = torch.from_numpy(rs.uniform(-b, b, shp).astype(np.float32))
    return P


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.99, eps: float = 1e-8, wd: float = 1e-4) -> None:
    """torch.optim.AdamW as configured at vitvqgan.py:160 (decoupled decay, bias correction), in place."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def train_step_grads(img: Tensor, P: Dict[str, Tensor], cfg: dict, loss_kw: Optional[dict] = None):
    """One AE training step's loss and gradients (ViTVQ.training_step optimizer_idx 0, vitvqgan.py:101-115)
    with the pixel+codebook loss; gradients come from torch.autograd over this restatement, exactly as the
    reference obtains them."""
    leaves = {k: v.detach().clone().requires_grad_(not k.endswith("pos_embedding")) for k, v in P.items()}
    xrec, qloss = forward(img, leaves, cfg)
    loss, log = pixel_codebook_loss(qloss, img, xrec, **(loss_kw or {}))
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return loss.detach(), log, grads, xrec.detach()


# ----------------------------------------------------------------------------------------
# seeded synthetic inputs shared by the golden generator, the tests and bench.py
# (numpy MT19937: stable across numpy / torch versions, so fixtures only store OUTPUTS)
# ----------------------------------------------------------------------------------------
def make_vq_inputs(seed: int, M: int, K: int, d: int = 32):
    rs = np.random.RandomState(seed)
    z = torch.from_numpy(rs.standard_normal((M, d)).astype(np.float32))
    E = torch.from_numpy(rs.standard_normal((K, d)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((M, d)).astype(np.float32))  # upstream grad wrt returned z_q
    return z, E, g


def make_images(seed: int, B: int, size: int, smooth: bool = True, C: int = 3) -> Tensor:
    """Synthetic ImageNet-shaped batch in [0,1] (dataloader/imagenet.py:31-36 contract).  ``smooth`` =
    bilinear-upsampled low-frequency noise (uniform noise collapses code usage: SURVEY.md §8d config 1)."""
    rs = np.random.RandomState(seed)
    if not smooth:
        return torch.from_numpy(rs.uniform(0, 1, (B, C, size, size)).astype(np.float32))
    low = torch.from_numpy(rs.uniform(0, 1, (B, C, max(size // 16, 2), max(size // 16, 2))).astype(np.float32))
    up = F.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)
    return (up + 0.05 * torch.from_numpy(rs.standard_normal((B, C, size, size)).astype(np.float32))).clamp_(0, 1)


TINY_CFG = dict(image_size=64, patch_size=8,
                encoder=dict(dim=128, depth=2, heads=2, mlp_dim=256),
                decoder=dict(dim=128, depth=2, heads=2, mlp_dim=256),
                quantizer=dict(embed_dim=32, n_embed=512))


def train_step_traced(img: Tensor, P: Dict[str, Tensor], cfg: dict, loss_kw: Optional[dict] = None, force_idx: Optional[Tensor] = None) -> dict:
    """train_step_grads plus every intermediate a per-layer parity table needs (test aid): the residual stream entering / leaving
    each encoder and decoder layer, h (quantizer input), z_q, indices, xrec, loss, gradients.  Same arithmetic as forward()."""
    leaves = {k: v.detach().clone().requires_grad_(not k.endswith("pos_embedding")) for k, v in P.items()}
    enc_tr, dec_tr = [], []
    h = encoder(img, leaves, cfg, trace=enc_tr) @ leaves["pre_quant.weight"].t() + leaves["pre_quant.bias"]
    quant, qloss, idx = quantizer_forward(h, leaves["quantizer.embedding.weight"], force_idx=force_idx, **qparams(cfg))
    xrec = decoder(quant @ leaves["post_quant.weight"].t() + leaves["post_quant.bias"], leaves, cfg, trace=dec_tr)
    loss, log = pixel_codebook_loss(qloss, img, xrec, **(loss_kw or {}))
    loss.backward()
    return dict(loss=loss.detach(), log=log, grads={k: v.grad for k, v in leaves.items() if v.grad is not None}, xrec=xrec.detach(),
                h=h.detach(), zq=quant.detach(), idx=idx, qloss=qloss.detach(), enc_trace=enc_tr, dec_trace=dec_tr)
