"""LPIPS perceptual distance, ``lpips.LPIPS(net="vgg")`` of lpips 0.1.4 — the third-party dependency the reference pins (requirements.txt:2) and calls
at enhancing/losses/vqperceptual.py:29,43,74,115 — on this library's HIP kernels.

Algorithm (lpips/lpips.py ``LPIPS.forward``, lpips/pretrained_networks.py ``vgg16``; restated and cited line by line in ``oracle/lpips_oracle.py``):
    x        = (in - shift) / scale                                   ScalingLayer, in in [-1, 1]
    f_k      = torchvision vgg16.features up to relu1_2, relu2_2, relu3_3, relu4_3, relu5_3       (3x3 convs + ReLU, 2x2 max-pools between slices)
    n_k      = f_k / (||f_k||_channel + 1e-10)                         normalize_tensor
    d(in0, in1) = sum_k mean_{h,w} sum_c lin_k[c] * (n_k(in0) - n_k(in1))_c^2          NetLinLayer (1x1 conv, no bias; its Dropout is inactive: eval mode)
returned as [B, 1, 1, 1].  The module tree and state-dict keys are those of lpips 0.1.4 (``scaling_layer.shift``, ``net.slice1.0.weight`` ...,
``lin0.model.1.weight`` and the duplicate ``lins.0.model.1.weight``), so a state dict saved from the real package loads with ``strict=True``.

WEIGHTS.  lpips loads an ImageNet-pretrained torchvision VGG16 and its own learned lin layers; neither can be obtained in this environment (no
network, no lpips / torchvision wheel).  ``LPIPS(model_path=...)`` or the environment variable ``ENH_LPIPS_WEIGHTS`` names a ``torch.save``d state dict
in the lpips key layout; a checkpoint of the whole model that carries ``loss.perceptual_loss.*`` (as the reference's do) supplies them through the
parent's ``load_state_dict`` as well.  Without weights the module can be CONSTRUCTED (so that such a checkpoint can be loaded into it) but its forward
RAISES — the reference's ``lpips.LPIPS(net='vgg')`` fails offline too, and a silently random perceptual term would be logged as if it were the metric.
Random initialisation is an explicit opt-in (``pretrained=False`` or ``ENH_LPIPS_RANDOM_INIT=1``; used by tests and bench configurations): the trunk then
gets torchvision's own initialiser (kaiming-normal fan_out, zero bias) and the lin layers uniform [0, 1) weights (the learned ones are non-negative
too) from a PRIVATE generator (the global RNG is not consumed) — structurally right, differentiable and timed on the real topology, but its VALUES
are not the published metric: "parity unpinned" (SURVEY.md §8c).  ``weights_loaded`` says which of the two is running.

Execution: both images go through the trunk as one batch of 2B in channels-last bf16; the first convolution is fused with the scaling layer
(``enh_vgg_conv1``), the other twelve are implicit GEMMs on MFMA (``enh_conv3x3_nhwc_h16``: bias + ReLU fused; the input gradient is the same kernel on
flipped weights with the ReLU mask and the head's gradient fused into its epilogue), pooling / head are streaming kernels.  Gradients flow to ``in1``
(the reconstruction) only — the weights are frozen and ``in0`` is the data — exactly what the reference needs from this term.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _C

# torchvision vgg16.features: index -> (Cin, Cout) of every convolution, grouped by the lpips slices (pretrained_networks.py vgg16: [0,4) [4,9) [9,16) [16,23) [23,30))
_SLICES = [[(0, 3, 64), (2, 64, 64)],
           [(5, 64, 128), (7, 128, 128)],
           [(10, 128, 256), (12, 256, 256), (14, 256, 256)],
           [(17, 256, 512), (19, 512, 512), (21, 512, 512)],
           [(24, 512, 512), (26, 512, 512), (28, 512, 512)]]
_CHNS = [64, 128, 256, 512, 512]
_warned = False


class ScalingLayer(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])


class _ConvParams(nn.Module):
    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)


class _Slice(nn.Module):
    """parameters at the torchvision ``features`` indices (``net.slice3.12.weight`` ...)"""

    def __init__(self, convs) -> None:
        super().__init__()
        for idx, cin, cout in convs:
            self.add_module(str(idx), _ConvParams(cin, cout))


class _VGG16(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        for k, convs in enumerate(_SLICES):
            setattr(self, f"slice{k + 1}", _Slice(convs))


class NetLinLayer(nn.Module):
    """lpips.NetLinLayer: ``model = Sequential(Dropout(), Conv2d(C, 1, 1, bias=False))`` -> key ``model.1.weight`` [1, C, 1, 1]"""

    def __init__(self, chn_in: int) -> None:
        super().__init__()
        conv = nn.Module()
        conv.weight = nn.Parameter(torch.empty(1, chn_in, 1, 1), requires_grad=False)
        self.model = nn.Sequential(nn.Identity(), conv)


class _LPIPSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, in0, in1, normalize):
        out, saved = module._run_forward(in0, in1, normalize)
        ctx.module, ctx.saved, ctx.normalize = module, saved, normalize
        ctx.shape = in1.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        # ctx.saved is NOT dropped here: calculate_adaptive_factor (vqperceptual.py:94-103) differentiates nll_loss — which contains this term — with
        # retain_graph=True before the real backward runs through the same node a second time; the activations are freed with the graph
        d_in1 = ctx.module._run_backward(ctx.saved, gout.reshape(-1).float().contiguous(), ctx.normalize, ctx.shape)
        return None, None, d_in1, None


class LPIPS(nn.Module):
    def __init__(self, net: str = "vgg", verbose: bool = False, pretrained: bool = True, model_path: Optional[str] = None, **_ignored) -> None:
        super().__init__()
        if net not in ("vgg", "vgg16"):
            raise NotImplementedError("only net='vgg' (what the reference constructs, vqperceptual.py:29) is built")
        self.chns, self.L = list(_CHNS), len(_CHNS)
        self.scaling_layer = ScalingLayer()
        self.net = _VGG16()
        for k, c in enumerate(self.chns):
            setattr(self, f"lin{k}", NetLinLayer(c))
        self.lins = nn.ModuleList([getattr(self, f"lin{k}") for k in range(self.L)])   # lpips registers them twice: both key families exist
        self._dev: Dict[str, object] = {}
        self.weights_loaded = False       # True once an lpips-format state dict went in (directly or through a parent's load_state_dict)
        self.random_init = False          # True only on the explicit opt-in below
        path = model_path or os.environ.get("ENH_LPIPS_WEIGHTS")
        if path:
            sd = torch.load(path, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=True)
        elif not pretrained or os.environ.get("ENH_LPIPS_RANDOM_INIT", "0") not in ("", "0"):
            self._random_init()
        else:                              # constructed empty: forward raises until weights arrive (see the module docstring)
            with torch.no_grad():
                for p in self.parameters():
                    p.zero_()
        self._register_state_dict_hook(LPIPS._drop_random_weights)
        self.eval()

    @staticmethod
    def _drop_random_weights(module, state_dict, prefix, local_metadata):
        """a checkpoint must not launder a random trunk into "loaded weights": while the module runs on the explicit random initialisation its tensors
        are left out of every state dict it (or a parent: Trainer.fit saves model.state_dict()) produces — loading such a checkpoint leaves the
        perceptual term as constructed (empty: forward raises; or random again, by the same explicit opt-in)"""
        if module.random_init and not module.weights_loaded:
            for k in [k for k in state_dict if k.startswith(prefix)]:
                del state_dict[k]
        return state_dict

    def full_state_dict(self) -> Dict[str, torch.Tensor]:
        """every tensor under its lpips 0.1.4 key (both `lin{k}` and `lins.{k}` families), whatever its provenance — for tests and diagnostics; checkpoints
        go through state_dict(), which omits a randomly initialised trunk"""
        return {n: t.detach() for n, t in list(self.named_parameters(remove_duplicate=False)) + list(self.named_buffers(remove_duplicate=False))}

    # ---- parameters ----------------------------------------------------------------------------------------
    def _random_init(self) -> None:
        global _warned
        g = torch.Generator().manual_seed(0x1B1D5)   # private: the global RNG stream of the model's own initialisation is left untouched
        with torch.no_grad():
            for convs in _SLICES:
                for idx, cin, cout in convs:
                    p = self._conv(idx)
                    p.weight.normal_(0.0, (2.0 / (cout * 9)) ** 0.5, generator=g)     # torchvision VGG._initialize_weights: kaiming_normal_(fan_out, relu)
                    p.bias.zero_()
            for k in range(self.L):
                getattr(self, f"lin{k}").model[1].weight.uniform_(0.0, 1.0, generator=g)
        self.random_init = True
        if not _warned:
            warnings.warn("LPIPS: random initialisation requested (pretrained=False / ENH_LPIPS_RANDOM_INIT=1): the perceptual term runs on a randomly "
                          "initialised VGG16 — right topology and cost, values NOT the published metric (parity unpinned)")
            _warned = True

    def _conv(self, idx: int) -> _ConvParams:
        for k, convs in enumerate(_SLICES):
            for i, _, _ in convs:
                if i == idx:
                    return getattr(getattr(self.net, f"slice{k + 1}"), str(idx))
        raise KeyError(idx)

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        # reached both by self.load_state_dict and by a PARENT's (ViTVQ.init_from_ckpt loading a reference checkpoint that carries
        # loss.perceptual_loss.*), which never calls a child's load_state_dict override
        # weights_loaded only when EVERY tensor of the module arrives (a partial dict under strict=False leaves the missing ones zero), and a random
        # trunk can never arrive: state_dict() of a randomly initialised module omits its tensors (see _drop_random_weights)
        # lpips registers the five 1x1 heads twice (`lin{k}` attributes and the `lins` ModuleList): ONE family in the incoming dict fills both (they are
        # the same Parameters), so either counts as complete (a checkpoint carrying only one family left weights_loaded False although every
        # tensor had been copied)
        names = [n for n, _ in list(self.named_parameters(remove_duplicate=False)) + list(self.named_buffers(remove_duplicate=False))]
        core = {n for n in names if not n.startswith("lins.") and not n.startswith("lin")}
        fam_attr = {n for n in names if n.startswith("lin") and not n.startswith("lins.")}
        fam_list = {n for n in names if n.startswith("lins.")}
        have = set(state_dict)
        full = lambda fam: bool(fam) and {prefix + n for n in core | fam} <= have
        if full(fam_attr) or full(fam_list):
            self.weights_loaded, self.random_init = True, False
        self._dev.clear()
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def _apply(self, fn, *a, **k):
        self._dev.clear()
        return super()._apply(fn, *a, **k)

    def _weights_fingerprint(self):
        # (storage address, in-place version) of every parameter: any load / copy_ / optimizer-style write invalidates the packed operand cache
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _device_weights(self, device: torch.device, op=None) -> dict:
        """kernel operand forms of the frozen weights, built once per device and parameter version: tap-major bf16 [Cout][9*Cin] for the forward
        implicit GEMM and the flipped / transposed [Cin][9*Cout] for the input gradient"""
        from .op import conv_nhwc
        # 16-bit operand format of the trunk: what losses/op/conv_nhwc.py says when the forward runs (operand_dtype / ENH_LOSS_OPERANDS / the loss module
        # following its engine); the backward passes the format its saved features are in — it may run outside the forward's context
        op = conv_nhwc.OPERAND_DTYPE if op is None else op
        fp = (self._weights_fingerprint(), op)
        hit = self._dev.get(op)
        if hit is not None and hit.get("device") == device and hit.get("fingerprint") == fp:
            return hit
        d: Dict[str, object] = {"device": device, "fingerprint": fp, "fwd": {}, "bwd": {}, "bias": {}, "lin": [], "op": op}
        for convs in _SLICES:
            for idx, cin, cout in convs:
                p = self._conv(idx)
                w = p.weight.detach().to(device=device, dtype=torch.float32)
                d["bias"][idx] = p.bias.detach().to(device=device, dtype=torch.float32).contiguous()
                if idx == 0:
                    d["w0"] = w.contiguous()                                         # the first layer runs in f32 on the vector ALUs
                    continue
                d["fwd"][idx] = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(op).contiguous()
                d["bwd"][idx] = w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout).to(op).contiguous()
        for k in range(self.L):
            d["lin"].append(getattr(self, f"lin{k}").model[1].weight.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous())
        d["shift"] = self.scaling_layer.shift.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        d["scale"] = self.scaling_layer.scale.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        self._dev[op] = d
        return d

    # ---- lpips API -------------------------------------------------------------------------------------------
    def forward(self, in0: torch.Tensor, in1: torch.Tensor, retPerLayer: bool = False, normalize: bool = False) -> torch.Tensor:
        """d(in0, in1) as [B,1,1,1]; images in [-1,1], or in [0,1] with normalize=True (lpips' own flag).  Differentiable w.r.t. in1."""
        if retPerLayer:
            raise NotImplementedError("retPerLayer is not used by the reference")
        if not (self.weights_loaded or self.random_init):
            raise RuntimeError("LPIPS has no weights: pass model_path / set ENH_LPIPS_WEIGHTS to an lpips-format state dict (lpips 0.1.4, net='vgg'), load a "
                               "checkpoint that carries loss.perceptual_loss.*, set perceptual_weight: 0, or opt in to a randomly initialised trunk with "
                               "pretrained=False / ENH_LPIPS_RANDOM_INIT=1 (values then are NOT the published metric)")
        if not in1.is_cuda:
            raise RuntimeError("LPIPS runs only on a ROCm device: the HIP path has no CPU fallback")
        in0 = in0.detach().to(device=in1.device, dtype=torch.float32).contiguous()
        return _LPIPSFn.apply(self, in0, in1.to(torch.float32).contiguous(), bool(normalize))

    # ---- kernels ---------------------------------------------------------------------------------------------
    def _run_forward(self, in0, in1, normalize):
        B, C, H, W = in1.shape
        if C != 3 or in0.shape != in1.shape or H % 16 or W % 16:
            raise RuntimeError(f"LPIPS: expected two [B,3,H,W] batches with H, W multiples of 16, got {tuple(in0.shape)} and {tuple(in1.shape)}")
        dw, dev = self._device_weights(in1.device), in1.device
        op = dw["op"]
        x = torch.cat([in0, in1.detach()], 0)                      # one 2B batch through the trunk
        B2 = 2 * B
        feats: List[torch.Tensor] = []                              # post-ReLU output of every convolution, [2B, h, w, C] bf16
        acts: Dict[int, torch.Tensor] = {}
        h, w = H, W
        cur = None
        out = torch.empty(B, dtype=torch.float32, device=dev)
        for k, convs in enumerate(_SLICES):
            if k > 0:                                               # the max-pool opening slices 2..5
                pooled = torch.empty(B2, h // 2, w // 2, cur.shape[-1], dtype=op, device=dev)
                _C.maxpool2_nhwc(cur, B2, h, w, cur.shape[-1], pooled)
                h, w, cur = h // 2, w // 2, pooled
                acts[-k] = pooled
            for idx, cin, cout in convs:
                y = torch.empty(B2, h, w, cout, dtype=op, device=dev)
                if idx == 0:
                    _C.vgg_conv1(x, dw["w0"], dw["bias"][0], dw["shift"], dw["scale"], normalize, y)
                else:
                    _C.conv3x3_nhwc(cur, dw["fwd"][idx], B2, h, w, cin, cout, y, bias=dw["bias"][idx], mode=0)
                acts[idx] = y
                cur = y
            feats.append(cur)
            val_ws = torch.empty(B * h * w, dtype=torch.float32, device=dev)
            _C.lpips_head(cur, dw["lin"][k], B, h * w, cur.shape[-1], val_ws, out, accumulate=k > 0)
        return out.view(B, 1, 1, 1), (acts, feats, (B, H, W))

    def _run_backward(self, saved, gout, normalize, shape):
        acts, feats, (B, H, W) = saved
        op = feats[0].dtype      # the format the forward ran in
        dw, dev = self._device_weights(gout.device, op), gout.device
        rec = lambda t: t[B:]                                       # the reconstruction half of a [2B, ...] activation (contiguous slice)
        dims = [(H >> k, W >> k) for k in range(5)]
        # gradient of the head at every slice output, reconstruction half only
        head = []
        for k in range(5):
            h, w = dims[k]
            g = torch.empty(B, h, w, _CHNS[k], dtype=op, device=dev)
            _C.lpips_head_backward(feats[k], dw["lin"][k], gout, B, h * w, _CHNS[k], g)
            head.append(g)
        # walk the trunk backwards; gpre = gradient at a convolution's output BEFORE its ReLU
        gpre = None
        for k in range(4, -1, -1):
            h, w = dims[k]
            convs = _SLICES[k]
            for ci in range(len(convs) - 1, -1, -1):
                idx, cin, cout = convs[ci]
                y = rec(acts[idx])
                if ci == len(convs) - 1:
                    if k == 4:                                      # deepest slice output: only the head reaches it
                        gpre = _relu_mask(head[4], y)
                    # else: gpre for this layer was produced by the pool backward below (head[k] and the ReLU mask already applied)
                if idx == 0:
                    d_in1 = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
                    _C.vgg_conv1_backward(gpre, dw["w0"], dw["scale"], normalize, B, H, W, d_in1)
                    return d_in1
                if ci > 0:                                          # previous layer of the same slice: conv-transpose + ReLU mask of ITS output
                    prev = rec(acts[convs[ci - 1][0]])
                    nxt = torch.empty(B, h, w, cin, dtype=op, device=dev)
                    _C.conv3x3_nhwc(gpre, dw["bwd"][idx], B, h, w, cout, cin, nxt, mode=1, aux=prev)
                    gpre = nxt
                else:                                               # first layer of slice k > 0: its input is the pooled output of slice k-1
                    gp = torch.empty(B, h, w, cin, dtype=op, device=dev)
                    _C.conv3x3_nhwc(gpre, dw["bwd"][idx], B, h, w, cout, cin, gp, mode=2)
                    hp, wp = dims[k - 1]
                    yprev = rec(acts[_SLICES[k - 1][-1][0]])
                    nxt = torch.empty(B, hp, wp, cin, dtype=op, device=dev)
                    _C.maxpool2_nhwc_backward(yprev, gp, head[k - 1], B, hp, wp, cin, nxt)   # routes, adds the head's gradient, applies the ReLU mask
                    gpre = nxt
        raise AssertionError("unreachable")


def _relu_mask(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """g * (y > 0) for the deepest slice (one small tensor per step: [B, H/16, W/16, 512])"""
    return torch.where(y > 0, g, torch.zeros((), dtype=g.dtype, device=g.device)).contiguous()
