"""ViT encoder / decoder of the stage-1 tokenizer — parameter containers with the reference's constructor
signatures and state-dict layout (reference enhancing/modules/stage1/layers.py:153-217; key list in
SURVEY.md §8b), so released checkpoints load unchanged.

Unlike the reference these modules hold NO arithmetic: ``forward`` hands the tensors to the static HIP
schedule in ``enhancing.engine`` (patch-embed GEMM with fused bias + position table, pre-norm transformer
blocks on bf16 MFMA GEMMs + fused attention, final LayerNorm).  Sub-modules such as ``Attention`` exist to
reproduce the parameter tree (``transformer.layers.{i}.0.fn.to_qkv.weight`` ...), not to be called."""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np
import torch
import torch.nn as nn


def get_2d_sincos_pos_embed(embed_dim: int, grid_size) -> np.ndarray:
    """Fixed 2-D sin-cos table [gh*gw, embed_dim] (float64), reference layers.py:21-68.  The first half of the
    channels encodes the x (width) coordinate — meshgrid(w, h) puts w first (layers.py:30,43-44); within each
    half: [sin | cos] of pos * 10000^(-i/(D/4))."""
    gh, gw = (grid_size, grid_size) if not isinstance(grid_size, tuple) else grid_size
    xs, ys = np.meshgrid(np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32))

    def one_axis(dim: int, pos: np.ndarray) -> np.ndarray:
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        ang = pos.reshape(-1).astype(np.float64)[:, None] * omega[None, :]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    assert embed_dim % 4 == 0
    return np.concatenate([one_axis(embed_dim // 2, xs), one_axis(embed_dim // 2, ys)], axis=1)


def init_weights(m: nn.Module) -> None:
    """The reference's ``init_weights`` (layers.py:71-82) over this package's parameter containers: xavier-uniform Linear weights
    (as the official JAX ViT), zero Linear biases, LayerNorm (1, 0), xavier-uniform on the conv weights viewed [shape[0], -1] (the
    conv biases keep torch's default init).  Applied with ``self.apply`` at the END of ViTEncoder / ViTDecoder construction, exactly
    like the reference (layers.py:175,207) — see the seed-for-seed note on LinearParams."""
    if isinstance(m, LinearParams):
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, NormParams):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, PatchConvParams):
        w = m.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))


class LinearParams(nn.Module):
    """weight [out, in] (+ bias [out]) of an nn.Linear.

    Seed-for-seed parity with the reference: the reference CONSTRUCTS ``nn.Linear`` / ``nn.Conv2d`` /
    ``nn.ConvTranspose2d`` — whose default initialisers draw from the global RNG — and only afterwards re-draws the weights in
    ``self.apply(init_weights)`` order (layers.py:71-82,175,207).  To leave torch's RNG stream in the same state at every point,
    the containers here are filled by constructing the very same stock torch module and adopting its parameters; ``init_weights``
    then runs in the reference's ``apply`` order.  ``torch.manual_seed(s); ViTEncoder(...)`` therefore yields the reference's
    tensors bit-for-bit (tests/test_host_cpu.py::test_init_is_seed_for_seed_with_the_reference)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True) -> None:
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        donor = nn.Linear(in_features, out_features, bias=bias)
        self.weight = donor.weight
        self.bias = donor.bias if bias else None


class NormParams(nn.Module):
    """weight / bias of nn.LayerNorm(dim): ones / zeros (layers.py:77-79)."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class PatchConvParams(nn.Module):
    """weight [dim, C, p, p] + bias of Conv2d(C, dim, k=s=p) (bias [dim]) or ConvTranspose2d(dim, C, k=s=p) (bias [C]): both store
    the weight as [dim, C, p, p]; xavier on the [dim, C*p*p] view, torch-default bias (layers.py:80-82,169,204)."""

    def __init__(self, dim: int, channels: int, patch: Tuple[int, int], transposed: bool) -> None:
        super().__init__()
        donor = (nn.ConvTranspose2d(dim, channels, kernel_size=patch, stride=patch) if transposed
                 else nn.Conv2d(channels, dim, kernel_size=patch, stride=patch))
        assert tuple(donor.weight.shape) == (dim, channels, patch[0], patch[1])
        self.weight, self.bias = donor.weight, donor.bias


class Attention(nn.Module):
    """bias-free to_qkv [3*inner, dim], to_out [dim, inner] + bias; inner = heads * 64 (layers.py:108-120)."""

    def __init__(self, dim: int, heads: int = 8, dim_head: int = 64) -> None:
        super().__init__()
        if dim_head != 64:
            raise ValueError("the fused gfx950 attention kernel is specialised for dim_head = 64 (every reference config)")
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_qkv = LinearParams(dim, inner * 3, bias=False)
        self.to_out = LinearParams(inner, dim)


class FeedForward(nn.Module):
    """Linear(dim, hidden) - Tanh - Linear(hidden, dim): parameters live at net.0 / net.2 (layers.py:95-102)."""

    def __init__(self, dim: int, hidden_dim: int) -> None:
        super().__init__()
        self.net = nn.Sequential(LinearParams(dim, hidden_dim), nn.Identity(), LinearParams(hidden_dim, dim))


class PreNorm(nn.Module):
    def __init__(self, dim: int, fn: nn.Module) -> None:
        super().__init__()
        self.norm = NormParams(dim)
        self.fn = fn


class Transformer(nn.Module):
    """depth x [PreNorm(Attention), PreNorm(FeedForward)] + final LayerNorm (layers.py:135-150)."""

    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int) -> None:
        super().__init__()
        self.dim, self.depth, self.heads, self.mlp_dim = dim, depth, heads, mlp_dim
        self.layers = nn.ModuleList([
            nn.ModuleList([PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head)), PreNorm(dim, FeedForward(dim, mlp_dim))])
            for _ in range(depth)])
        self.norm = NormParams(dim)


def _pair(v) -> Tuple[int, int]:
    return v if isinstance(v, tuple) else (v, v)


class _ViTBase(nn.Module):
    def __init__(self, image_size, patch_size, dim: int, depth: int, heads: int, mlp_dim: int, channels: int, dim_head: int) -> None:
        super().__init__()
        self.image_size, self.patch_size = _pair(image_size), _pair(patch_size)
        ih, iw = self.image_size
        ph, pw = self.patch_size
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        if ph != pw or ih != iw:
            raise ValueError("the gfx950 patch kernels assume square images and patches (every reference config)")
        self.grid = (ih // ph, iw // pw)
        self.num_patches = self.grid[0] * self.grid[1]
        self.channels, self.dim = channels, dim
        self.patch_dim = channels * ph * pw
        self._engine = None  # set by enhancing.engine.Stage1Engine

    def _pos_table(self) -> nn.Parameter:
        pe = get_2d_sincos_pos_embed(self.dim, self.grid)
        return nn.Parameter(torch.from_numpy(pe).float().unsqueeze(0), requires_grad=False)

    def _require_engine(self):
        if self._engine is None:
            raise RuntimeError("this module runs only through the HIP engine: build the parent ViTVQ (or call "
                               "enhancing.engine.attach(module)) on a ROCm device first; there is no eager fallback")
        return self._engine


class ViTEncoder(_ViTBase):
    """reference layers.py:153-182: img [B,C,H,W] -> tokens [B, N, dim]."""

    def __init__(self, image_size: Union[Tuple[int, int], int], patch_size: Union[Tuple[int, int], int],
                 dim: int, depth: int, heads: int, mlp_dim: int, channels: int = 3, dim_head: int = 64) -> None:
        super().__init__(image_size, patch_size, dim, depth, heads, mlp_dim, channels, dim_head)
        self.to_patch_embedding = nn.Sequential(PatchConvParams(dim, channels, self.patch_size, transposed=False), nn.Identity())
        self.en_pos_embedding = self._pos_table()
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.apply(init_weights)

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        return self._require_engine().encoder_forward(img)


class ViTDecoder(_ViTBase):
    """reference layers.py:185-217: tokens [B, N, dim] -> img [B,C,H,W]."""

    def __init__(self, image_size: Union[Tuple[int, int], int], patch_size: Union[Tuple[int, int], int],
                 dim: int, depth: int, heads: int, mlp_dim: int, channels: int = 3, dim_head: int = 64) -> None:
        super().__init__(image_size, patch_size, dim, depth, heads, mlp_dim, channels, dim_head)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.de_pos_embedding = self._pos_table()
        self.to_pixel = nn.Sequential(nn.Identity(), PatchConvParams(dim, channels, self.patch_size, transposed=True))
        self.apply(init_weights)

    def forward(self, token: torch.Tensor) -> torch.Tensor:
        return self._require_engine().decoder_forward(token)

    def get_last_layer(self) -> nn.Parameter:
        return self.to_pixel[-1].weight
