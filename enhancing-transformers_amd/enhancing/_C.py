"""ctypes binding of libenh_hip.so (the C ABI declared in include/enh_hip.h).

PyTorch tensors are used only as device-memory containers: every wrapper checks device / dtype /
contiguity (the reference's native ops do the same with CHECK_CUDA / CHECK_CONTIGUOUS,
enhancing/losses/op/fused_bias_act.cpp:9-15), passes raw ``data_ptr()`` values plus the current HIP
stream, and raises ``RuntimeError`` with ``enh_last_error()`` on a non-zero return code.

There is NO fallback: if the shared library is missing or a tensor is not on a ROCm device the call
fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("ENH_HIP_LIB", os.path.join(_PKG_ROOT, "lib", "libenh_hip.so"))

ACT_NONE, ACT_TANH, ACT_DTANH = 0, 1, 2
# Reductions across workgroups (LayerNorm dgamma / dbeta / fused bias sums, bias column sums, split-K weight gradients) run in their two-pass,
# fixed-order form by default: gradients are bit-reproducible from run to run (tests/test_parity_base_gpu.py).  ENH_DETERMINISTIC=0 selects the
# f32-atomic forms for A/B timing.
DETERMINISTIC = os.environ.get("ENH_DETERMINISTIC", "1") != "0"

_c = ctypes
_vp, _i64, _i32, _f32, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); must list every symbol include/enh_hip.h declares (checked by tests)
SIGNATURES = {
    "enh_last_error": (_c.c_char_p, []),
    "enh_abi_version": (_i32, []),
    "enh_vq_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "enh_vq_forward": (_i32, [_vp, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "enh_vq_backward": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "enh_vq_lookup": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "enh_layernorm_forward": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "enh_layernorm_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "enh_layernorm_backward_ws": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "enh_layernorm_backward_workspace_bytes": (_sz, [_i64, _i32]),
    "enh_gemm_h16": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _i64,
                            _i32, _vp, _vp, _i64, _i32, _vp]),
    "enh_gemm_h16_ws": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _i64,
                               _i32, _vp, _vp, _i64, _vp, _sz, _i32, _vp]),
    "enh_gemm_h16_workspace_bytes": (_sz, [_i32, _i32, _i64, _i64, _i64]),
    "enh_gemm_set_kernel": (_i32, [_i32]),
    "enh_gemm_set_scheduler": (_i32, [_i32]),
    "enh_set_cu_budget": (_i32, [_i32]),
    "enh_get_cu_budget": (_i32, []),
    "enh_debug_occupy_cus": (_i32, [_i32, _f32, _vp]),
    "enh_gemm_h16_variant": (_c.c_char_p, [_i32, _i32, _i64, _i64, _i64]),
    "enh_gemm_h16_variant_mode": (_c.c_char_p, [_i32, _i32, _i64, _i64, _i64, _i32]),
    "enh_gemm_h16_dtanh_colsum_workspace_bytes": (_c.c_size_t, [_i32, _i64, _i64, _i64]),
    "enh_gemm_h16_dtanh_colsum": (_i32, [_vp, _i64, _vp, _i64, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _c.c_size_t, _i32, _vp]),
    "enh_attention_set_kernel": (_i32, [_i32, _i32, _i32]),
    "enh_attention_forward": (_i32, [_vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i32, _vp]),
    "enh_attention_backward": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i32, _vp]),
    "enh_patchify": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_unpatchify_loss": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "enh_colsum_h16": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _i32, _vp]),
    "enh_colsum_h16_ws": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _vp, _sz, _i32, _vp]),
    "enh_colsum_h16_workspace_bytes": (_sz, [_i64, _i64]),
    "enh_cast_f32_h16": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "enh_cast_f32_h16_head_scaled": (_i32, [_vp, _vp, _i64, _i64, _f32, _i32, _vp]),
    "enh_cast_f32_h16_head_scaled_strided": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _f32, _i32, _i32, _vp]),
    "enh_crop_flip_u8": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "enh_resize_u8_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "enh_resize_u8": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "enh_fused_bias_act": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _f32, _vp]),
    "enh_channel_sum_f32": (_i32, [_vp, _i32, _i32, _i64, _vp, _i32, _vp]),
    "enh_upfirdn2d": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "enh_im2col": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "enh_col2im": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i32, _vp]),
    "enh_conv_nhwc_h16": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _f32, _f32, _vp, _i32, _vp]),
    "enh_conv_nhwc_h16_ws": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _sz, _i32, _vp]),
    "enh_conv_workspace_bytes": (_sz, [_vp]),
    "enh_conv_set_kernel": (_i32, [_i32]),
    "enh_conv_wgrad_workspace_bytes": (_sz, [_vp]),
    "enh_conv_wgrad_nhwc_h16": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "enh_conv_pack_weight": (_i32, [_vp, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_conv_unpack_wgrad": (_i32, [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "enh_blur_set_kernel": (_i32, [_i32]),
    "enh_debug_gemm_lab": (_i32, [_i32]),
    "enh_debug_gemm_order": (_i32, [_i32, _i32]),
    "enh_debug_gemm_splits": (_i32, [_i32]),
    "enh_blur_nhwc_h16": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_lrelu_gate_h16": (_i32, [_vp, _vp, _i64, _f32, _f32, _vp, _i32, _vp]),
    "enh_img_to_nhwc8": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_nhwc8_to_img": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_minibatch_stddev_nhwc": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_minibatch_stddev_nhwc_backward": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_gemm_f32": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "enh_attention_forward_f32": (_i32, [_vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "enh_attention_backward_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "enh_colsum_f32": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _vp]),
    "enh_patch_perm_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "enh_unpatchify_loss_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "enh_conv3x3_nhwc_h16": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "enh_vgg_conv1": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_vgg_conv1_backward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_maxpool2_nhwc_h16": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_maxpool2_nhwc_h16_backward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "enh_lpips_head": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _vp]),
    "enh_lpips_head_backward": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32, _vp]),
    "enh_split3_bf16": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "enh_split2_bf16": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "enh_gemm_bf16_split_fused": (_i32, [_i64, _i64, _i64]),
    "enh_gemm_bf16_split": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "enh_layernorm_forward_x3": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "enh_attention_forward_x3": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "enh_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _i32, _vp]),
    "enh_loss_scale_update": (_i32, [_vp, _vp, _vp, _f32, _f32, _i32, _vp]),
    "enh_nonfinite_flag": (_i32, [_vp, _i64, _vp, _vp]),
}

_LIB = None
ABI_VERSION = 17  # ENH_ABI_VERSION of the include/enh_hip.h these signatures were written against


def lib():
    """Loads libenh_hip.so (once).  Raises if it has not been built: there is no CPU / eager fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                               f"(or `make -C enhancing-transformers_amd/csrc`). There is no fallback path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == ABI drift: fail loudly
            fn.restype, fn.argtypes = res, args
        if L.enh_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {L.enh_abi_version()}, the bindings expect {ABI_VERSION}: rebuild the library")
        _LIB = L
        # kernel-family override for A/B measurements: explicit library state behind enh_gemm_set_kernel(); the environment is read HERE, in
        # the binding, never inside the library (reg = 0, pipe2 = 3, w256 = 7)
        sel = os.environ.get("ENH_GEMM_KERNEL")
        if sel:
            fam = {"reg": 0, "pipe2": 3, "w256": 7, "w256p": 8, "w256r": 9}.get(sel)
            if fam is None:
                raise RuntimeError(f"ENH_GEMM_KERNEL={sel!r}: expected reg | pipe2 | w256 | w256p | w256r")
            _check_rc = L.enh_gemm_set_kernel(fam)
            if _check_rc != 0:
                raise RuntimeError(L.enh_last_error().decode())
        sched = os.environ.get("ENH_GEMM_SCHEDULER")      # "static" | "dynamic" tile schedule of the persistent GEMMs (A/B)
        if sched:
            if sched not in ("static", "dynamic") or L.enh_gemm_set_scheduler(int(sched == "dynamic")) != 0:
                raise RuntimeError(f"ENH_GEMM_SCHEDULER={sched!r}: expected static | dynamic")
            _DYN_SCHEDULE[0] = sched == "dynamic"
        att = os.environ.get("ENH_ATTN_KERNEL")       # "fwd,dq,dkv" families, e.g. "1,1,1" (0 = library default; include/enh_hip.h enh_attention_set_kernel)
        if att:
            f, q, k = (int(x) for x in att.split(","))
            if L.enh_attention_set_kernel(f, q, k) != 0:
                raise RuntimeError(L.enh_last_error().decode())
            _ATT_FAMILY[:] = [f, q, k]
        conv = os.environ.get("ENH_CONV_KERNEL")      # A/B: "reg" register-staged everywhere | "t128" no 256-row kernels | "t256" 256-row wherever the shape allows
        if conv:
            if conv not in CONV_KERNELS or L.enh_conv_set_kernel(CONV_KERNELS[conv]) != 0:
                raise RuntimeError(f"ENH_CONV_KERNEL={conv!r}: expected one of {sorted(CONV_KERNELS)}")
    return _LIB


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().enh_last_error().decode()}")


def _p(t: Optional[torch.Tensor], dtype: Optional[torch.dtype] = None, name: str = "tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on a ROCm device (got {t.device}); the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and (t.dtype not in dtype if isinstance(dtype, tuple) else t.dtype != dtype):
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


F32, BF16, F16, I64, F64 = torch.float32, torch.bfloat16, torch.float16, torch.int64, torch.float64
H16 = (BF16, F16)      # the two 16-bit operand formats of the product path (include/enh_hip.h ENH_DT_BF16 / ENH_DT_F16)
DT_BF16, DT_F16, DT_F32 = 0, 1, 2


def _dt(*tensors) -> int:
    """the C ABI's dtype argument of a call, inferred from its 16-bit containers (torch.bfloat16 -> ENH_DT_BF16, torch.float16 -> ENH_DT_F16): all of one call's
    16-bit tensors must have the same format.  Calls without a 16-bit tensor pass the default (bf16)."""
    dts = {t.dtype for t in tensors if t is not None and t.dtype in H16}
    if len(dts) > 1:
        raise RuntimeError("the 16-bit operands of one call must all be bf16 or all be fp16")
    return DT_F16 if dts == {F16} else DT_BF16


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (torch's current stream is the stream every
    kernel of this library is enqueued on).  bench.py installs one to compute the roofline figures live."""

    def __init__(self) -> None:
        self.records = {}
        self.units = {}      # name -> "flop" (bf16 MFMA work), "flop_f32" (exact-f32 MFMA work) or "byte" (algorithmic HBM bytes)

    def run(self, name: str, work: float, fn, unit: str = "flop") -> None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.setdefault(name, []).append((s, e, work))
        self.units[name] = unit

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            out[name] = dict(launches=len(recs), total_ms=sum(ms), avg_ms=sum(ms) / len(ms), work=sum(w for _, _, w in recs), unit=self.units[name])
        return out


TIMER: Optional[KernelTimer] = None


def _timed(name: str, work: float, fn, unit: str = "flop") -> None:
    if TIMER is None:
        fn()
    else:
        TIMER.run(name, work, fn, unit)

# ------------------------------------------------------------------------------------------------
# quantizer
# ------------------------------------------------------------------------------------------------
_WS = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def vq_forward(z: torch.Tensor, codebook: torch.Tensor, beta: float, depth: int, use_norm: bool, want_bf16: bool = True, h16: torch.dtype = BF16):
    """z [M,32] f32, codebook [K,32] f32 -> (zq f32 [M,32], zq 16-bit copy (format h16) | None, idx i64 [M,depth], loss f32 [1])."""
    _p(z, F32, "z"); _p(codebook, F32, "codebook")  # device / dtype / contiguity first: fail before allocating
    M, d = z.shape
    K = codebook.shape[0]
    zq = torch.empty_like(z)
    zq16 = torch.empty(M, d, dtype=h16, device=z.device) if want_bf16 else None
    idx = torch.empty(M, depth, dtype=I64, device=z.device)
    loss = torch.empty(1, dtype=F32, device=z.device)
    L = lib()
    nb = L.enh_vq_workspace_bytes(M, K, depth)
    ws = _workspace(nb, z.device)
    # work model: 2*K*d FLOP per token per depth on the exact-f32 MFMA (157.3 TF peak); the call = vq_prep + vq_nn + vq_loss_finalize
    _timed("vq_forward (vq_prep + vq_nn_kernel + vq_loss_finalize)", 2.0 * M * K * d * depth,
           lambda: _check(L.enh_vq_forward(_p(z, F32, "z"), _p(codebook, F32, "codebook"), M, K, d, beta, depth, int(use_norm), _p(zq), _p(zq16),
                                           _p(idx), _p(loss), _p(ws), ws.numel(), _dt(zq16), _stream()), "enh_vq_forward"), unit="flop_f32")
    return zq, zq16, idx, loss


def vq_backward(z, codebook, idx, g_out, g_loss: float, g_loss_dev: Optional[torch.Tensor], beta: float, depth: int,
                use_residual: bool, use_norm: bool, d_codebook: torch.Tensor, want_bf16: bool = True, h16: torch.dtype = BF16):
    """Returns (dz f32, dz_bf16|None); ACCUMULATES into d_codebook [K,32] f32."""
    _p(z, F32, "z"); _p(codebook, F32, "codebook")
    M, d = z.shape
    K = codebook.shape[0]
    dz = torch.empty_like(z)
    dz16 = torch.empty(M, d, dtype=h16, device=z.device) if want_bf16 else None
    L = lib()
    nb = L.enh_vq_workspace_bytes(M, K, depth)
    ws = _workspace(nb, z.device)
    _check(L.enh_vq_backward(_p(z, F32, "z"), _p(codebook, F32, "codebook"), _p(idx, I64, "idx"), _p(g_out, F32, "g_out"),
                             float(g_loss), _p(g_loss_dev, F32, "g_loss_dev"), M, K, d, beta, depth, int(use_residual),
                             int(use_norm), _p(dz), _p(dz16), _p(d_codebook, F32, "d_codebook"), _p(ws), ws.numel(), _dt(dz16), _stream()),
           "enh_vq_backward")
    return dz, dz16


def vq_lookup(codebook, idx, use_norm: bool, want_bf16: bool = True, h16: torch.dtype = BF16):
    """idx [M,depth] i64 -> (sum_i n(E[idx_i]) f32 [M,32], bf16 copy)."""
    _p(idx, I64, "idx"); _p(codebook, F32, "codebook")
    M, depth = idx.shape
    K, d = codebook.shape
    out = torch.empty(M, d, dtype=F32, device=idx.device)
    out16 = torch.empty(M, d, dtype=h16, device=idx.device) if want_bf16 else None
    _check(lib().enh_vq_lookup(_p(codebook, F32, "codebook"), _p(idx, I64, "idx"), M, K, d, depth, int(use_norm), _p(out), _p(out16),
                               _dt(out16), _stream()), "enh_vq_lookup")
    return out, out16


# ------------------------------------------------------------------------------------------------
# layernorm
# ------------------------------------------------------------------------------------------------
def layernorm_forward(x, w, b, eps: float = 1e-5, y_bf16=None, y_f32=None, mean=None, rstd=None):
    M, D = x.shape
    _check(lib().enh_layernorm_forward(_p(x, F32, "x"), _p(w, F32, "w"), _p(b, F32, "b"), M, D, eps, _p(y_bf16, H16, "y_bf16"),
                                       _p(y_f32, F32, "y_f32"), _p(mean, F32, "mean"), _p(rstd, F32, "rstd"), _dt(y_bf16), _stream()),
           "enh_layernorm_forward")


def layernorm_backward(dy, x, w, mean, rstd, dres, dx_f32, dx_bf16, dw, db, dx_colsum=None):
    M, D = x.shape
    dy32, dy16 = (None, dy) if dy.dtype in H16 else (dy, None)   # upstream gradient: f32, or the dgrad GEMM's 16-bit output
    dt = _dt(dy16, dx_bf16)
    if not DETERMINISTIC:
        _check(lib().enh_layernorm_backward(_p(dy32, F32, "dy"), _p(dy16, H16, "dy_bf16"), _p(x, F32, "x"), _p(w, F32, "w"), _p(mean, F32, "mean"),
                                            _p(rstd, F32, "rstd"), _p(dres, F32, "dres"), M, D, _p(dx_f32, F32, "dx_f32"),
                                            _p(dx_bf16, H16, "dx_bf16"), _p(dw, F32, "dw"), _p(db, F32, "db"), _p(dx_colsum, F32, "dx_colsum"),
                                            dt, _stream()), "enh_layernorm_backward")
        return
    # deterministic form: per-workgroup column partials in a caller-owned workspace + a fixed-order second pass (no f32 atomics)
    nb = lib().enh_layernorm_backward_workspace_bytes(M, D)
    ws = _workspace(nb, x.device)
    # algorithmic HBM bytes per element: dy (2 or 4) + x 4 + dres 4 in, dx f32 4 + dx bf16 2 out  (16 B/elem with bf16 dy)
    bpe = (2 if dy16 is not None else 4) + 4 + (4 if dres is not None else 0) + (4 if dx_f32 is not None else 0) + (2 if dx_bf16 is not None else 0)
    _timed("ln_bwd_kernel", float(bpe) * M * D, lambda: _check(lib().enh_layernorm_backward_ws(_p(dy32, F32, "dy"), _p(dy16, H16, "dy_bf16"), _p(x, F32, "x"), _p(w, F32, "w"), _p(mean, F32, "mean"),
                                           _p(rstd, F32, "rstd"), _p(dres, F32, "dres"), M, D, _p(dx_f32, F32, "dx_f32"),
                                           _p(dx_bf16, H16, "dx_bf16"), _p(dw, F32, "dw"), _p(db, F32, "db"), _p(dx_colsum, F32, "dx_colsum"),
                                           _p(ws), ws.numel(), dt, _stream()), "enh_layernorm_backward_ws"), unit="byte")


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm(a, b, M: int, N: int, K: int, trans_a: bool = False, trans_b: bool = False, bias=None, act: int = ACT_NONE, aux=None,
         res=None, res_rows: int = 0, accumulate: bool = False, out_f32=None, out_bf16=None, lda: Optional[int] = None,
         ldb: Optional[int] = None, ldc: Optional[int] = None):
    """C[M,N] = epilogue(A(m,k) * B(n,k)); see include/enh_hip.h.  a / b are 2-D 16-bit tensors (row-major storage), both torch.bfloat16 or both
    torch.float16: the container dtype selects the MFMA operand format (ENH_DT_BF16 / ENH_DT_F16)."""
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    out = out_f32 if out_f32 is not None else out_bf16
    ldc = out.stride(0) if ldc is None else ldc
    # split-K calls (weight gradients; also any accumulate-into-f32 call with a long K and few tiles, e.g. the discriminator's 8192 -> 512 linear at a
    # small batch): partial slabs in a caller-owned workspace + a fixed-order second pass — deterministic, no f32 atomics.  (Until round 4 only the
    # weight-gradient layout got the workspace; the other layouts fell back to atomics and made the discriminator's forward differ by an ulp from call
    # to call — found by the graph-replay bit-identity test.)
    # Round 6: any f32-output call without an activation may be split (bias / residual / accumulate move into the second pass): what fills the chip at
    # 2 - 4 images per GPU, where the N = dim GEMMs are 96 - 192 tiles (csrc/gemm.hip gemm_splittable).
    ws, ws_bytes = None, 0
    if act == ACT_NONE and ldc == N and ((out_f32 is not None and out_bf16 is None) or
                                         (out_bf16 is not None and out_f32 is None and bias is None and res is None and not accumulate)):
        ws_bytes = lib().enh_gemm_h16_workspace_bytes(int(trans_a), int(trans_b), M, N, K)
        if ws_bytes:
            ws = _gemm_workspace(a.device, ws_bytes)
    dt = _dt(a, b, aux, out_bf16)
    args = (_p(a, H16, "A"), lda, int(trans_a), _p(b, H16, "B"), ldb, int(trans_b), M, N, K, _p(bias, F32, "bias"),
            act, _p(aux, H16, "aux"), aux.stride(0) if aux is not None else 0, _p(res, F32, "res"),
            res.stride(0) if res is not None else 0, res_rows if res is not None else 0, int(accumulate),
            _p(out_f32, F32, "out_f32"), _p(out_bf16, H16, "out_bf16"), ldc, _p(ws), ws_bytes if ws is not None else 0, dt, _stream())
    if TIMER is None:
        _check(lib().enh_gemm_h16_ws(*args), "enh_gemm_h16")
    else:  # label with the symbol rocprofv3 will report, e.g. "gemm_pipe2_kernel<BF16, false, true>" / "gemm_w256_kernel<F16, false, false, 1>"
        mode = _epi_mode_label(accumulate, ws is not None, out_f32 is not None, out_bf16 is not None, bias is not None, act, res is not None)
        # (a position-table residual, res_rows != M, is not the persistent kernel's case: ask with the generic mode)
        fam = lib().enh_gemm_h16_variant_mode(int(trans_a), int(trans_b), M, N, K, 0 if (res is not None and res_rows != M) else mode).decode()
        ot = _OT_NAME[dt]                                   # the kernels' first template argument: the operand type tag (csrc/common.h)
        targs = f"{ot}, {'true' if trans_a else 'false'}, {'true' if trans_b else 'false'}"
        dyn = "true" if _DYN_SCHEDULE[0] else "false"      # the persistent kernels' last template argument: the tile schedule (gemm_kernels.h DYN)
        if fam == "gemm_w256r_kernel":   # template <OT, TB, EPI, DYN>: A is never transposed there
            targs = f"{ot}, {'true' if trans_b else 'false'}, {mode}, {dyn}"
        elif fam == "gemm_w256p_kernel":
            targs += f", {mode}, {dyn}"
        elif fam == "gemm_w256_kernel":   # the epilogue mode is a template parameter (gemm_tiles.h epi_mode(), mirrored here for the label only)
            targs += f", {mode}"
        TIMER.run(f"{fam}<{targs}>", 2.0 * M * N * K, lambda: _check(lib().enh_gemm_h16_ws(*args), "enh_gemm_h16"))


def gemm_dtanh_colsum(a, b, M: int, N: int, K: int, aux, out_bf16, colsum_out, trans_b: bool = True, accumulate_colsum: bool = True):
    """out = (a b) * (1 - aux^2) -> bf16 and colsum_out (+)= column sums of out: enh_gemm_bf16_dtanh_colsum (the tanh' input gradient with the bias
    gradient of the Linear in front of the tanh taken in the GEMM's epilogue instead of by a second pass over `out`)"""
    nb = lib().enh_gemm_h16_dtanh_colsum_workspace_bytes(int(trans_b), M, N, K)
    ws = _workspace(nb, a.device)
    dt = _dt(a, b, aux, out_bf16)
    args = (_p(a, H16, "A"), a.stride(0), _p(b, H16, "B"), b.stride(0), int(trans_b), M, N, K, _p(aux, H16, "aux"), aux.stride(0),
            _p(out_bf16, H16, "out_bf16"), out_bf16.stride(0), _p(colsum_out, F32, "colsum"), int(accumulate_colsum), _p(ws), ws.numel(), dt, _stream())
    call = lambda: _check(lib().enh_gemm_h16_dtanh_colsum(*args), "enh_gemm_h16_dtanh_colsum")
    if TIMER is None:
        call()
    else:   # labelled with the GEMM kernel's symbol (the partial-row second pass and, off the tile grid, the column-sum kernel ride along)
        fam = lib().enh_gemm_h16_variant_mode(0, int(trans_b), M, N, K, 3).decode()
        dyn = ", true>" if _DYN_SCHEDULE[0] else ", false>"
        TIMER.run(f"{fam}<{_OT_NAME[dt]}, false, {'true' if trans_b else 'false'}" + ((", 3" + (dyn if fam == "gemm_w256p_kernel" else ">")) if "w256" in fam else ">"),
                  2.0 * M * N * K, call)


def set_cu_budget(n: int) -> None:
    """CUs the GEMM launches may count on (0 = all); see include/enh_hip.h"""
    _check(lib().enh_set_cu_budget(int(n)), "enh_set_cu_budget")


def get_cu_budget() -> int:
    return int(lib().enh_get_cu_budget())


def device_cus() -> int:
    """CUs of the current device"""
    L = lib()
    cur = int(L.enh_get_cu_budget())
    if _DEVICE_CUS[0] is None:
        L.enh_set_cu_budget(0)
        _DEVICE_CUS[0] = int(L.enh_get_cu_budget())
        L.enh_set_cu_budget(cur if cur != _DEVICE_CUS[0] else 0)
    return _DEVICE_CUS[0]


_DEVICE_CUS = [None]


_OT_NAME = {DT_BF16: "BF16", DT_F16: "F16"}      # operand type tags as they appear in the kernels' symbol names
_DYN_SCHEDULE = [True]      # mirrors the library's default (enh_gemm_set_scheduler), for timing labels only


CONV_KERNELS = {"auto": 0, "reg": 1, "t128": 2, "t256": 3}


def conv_set_kernel(name: str) -> None:
    """kernel family of the implicit-GEMM convolutions (enh_conv_set_kernel / enh_conv_wgrad_set_kernel are one switch here): auto | reg | t128 | t256"""
    _check(lib().enh_conv_set_kernel(CONV_KERNELS[name]), "enh_conv_set_kernel")


def gemm_set_scheduler(dynamic: bool) -> None:
    _check(lib().enh_gemm_set_scheduler(int(bool(dynamic))), "enh_gemm_set_scheduler")
    _DYN_SCHEDULE[0] = bool(dynamic)


def occupy_cus(n_wg: int, ms: float, stream=None) -> None:
    """measurement aid: hold n_wg CUs for ms milliseconds on `stream` (a torch stream; default: the current one)"""
    st = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    _check(lib().enh_debug_occupy_cus(int(n_wg), float(ms), st), "enh_debug_occupy_cus")


def _epi_mode_label(accumulate, have_ws, f32, bf16, bias, act, res) -> int:
    """EPI_* enum value gemm.hip's epi_mode() selects (0 generic, 1 bf16, 2 bf16+bias+tanh, 3 bf16+dtanh, 4 f32+bias+res, 5 f32, 6 split-K workspace,
    7 split-K atomics) — used only to label timings with the symbol name a profiler reports"""
    if accumulate and f32 and not bf16 and not bias and act == ACT_NONE and not res:
        return 6 if have_ws else 7    # (only when the shape is actually split; an unsplit accumulate call is generic)
    if not accumulate:
        if bf16 and not f32:
            if not bias and act == ACT_NONE and not res: return 1
            if bias and act == ACT_TANH and not res: return 2
            if not bias and act == ACT_DTANH and not res: return 3
        if f32 and not bf16 and act == ACT_NONE:
            if bias and res: return 4
            if not bias and not res: return 5
    return 0


_GEMM_WS = {}


def _gemm_workspace(device, nbytes: int):
    """one grow-only split-K workspace per (device, stream) (the library never allocates: SURVEY.md §8b ownership rule): the GEMMs of one stream run in
    order, so they share a buffer; the engine's weight-gradient side stream (small batches, engine/stage1.py) gets its own"""
    key = (device, torch.cuda.current_stream().cuda_stream)
    t = _GEMM_WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _GEMM_WS[key] = t
    return t


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
_ATT_FAMILY = [0, 0, 0]


def attention_set_kernel(fwd: int = 0, dq: int = 0, dkv: int = 0) -> None:
    """A/B aid: kernel family per pass (include/enh_hip.h enh_attention_set_kernel)"""
    _check(lib().enh_attention_set_kernel(fwd, dq, dkv), "enh_attention_set_kernel")
    _ATT_FAMILY[:] = [fwd, dq, dkv]


def attention_forward(qkv, B: int, N: int, H: int, scale: float, out, lse, q_prescaled: bool = False):
    """q_prescaled: the q third of qkv holds q * scale * log2(e) (include/enh_hip.h)"""
    # (labelled with the symbol rocprofv3 reports: family 5 — the default — serves pre-scaled q, family 1 everything else)
    fam = _ATT_FAMILY[0] or 5
    name = "attn_fwd_pre_kernel" if (fam == 5 and q_prescaled) else "attn_fwd_kernel"
    dt = _dt(qkv, out)
    _timed(f"{name}<{_OT_NAME[dt]}>", 4.0 * B * H * N * N * 64,
           lambda: _check(lib().enh_attention_forward(_p(qkv, H16, "qkv"), B, N, H, scale, int(q_prescaled), _p(out, H16, "out"), _p(lse, F32, "lse"),
                                                      dt, _stream()), "enh_attention_forward"))


def attention_backward(qkv, out, dout, lse, B: int, N: int, H: int, scale: float, dqkv, delta_ws, q_prescaled: bool = False):
    dt = _dt(qkv, out, dout, dqkv)
    _timed("attn_bwd (dq+dkv kernels)", 10.0 * B * H * N * N * 64,
           lambda: _check(lib().enh_attention_backward(_p(qkv, H16, "qkv"), _p(out, H16, "out"), _p(dout, H16, "dout"), _p(lse, F32, "lse"), B, N, H,
                                                       scale, int(q_prescaled), _p(dqkv, H16, "dqkv"), _p(delta_ws, F32, "delta_ws"), dt, _stream()),
                          "enh_attention_backward"))


# ------------------------------------------------------------------------------------------------
# data movement / loss / optimizer
# ------------------------------------------------------------------------------------------------
def patchify(img, p: int, out):
    B, C, H, W = img.shape
    _check(lib().enh_patchify(_p(img, F32, "img"), B, C, H, W, p, _p(out, H16, "patches"), _dt(out), _stream()), "enh_patchify")


def unpatchify_loss(pix, target, B: int, C: int, H: int, W: int, p: int, w_l1: float, w_l2: float, xrec, sums, dpix, grad_scale=None):
    """grad_scale: optional device scalar (f32 [1]) multiplying the gradient dpix only (the loss scale of the fp16 backward)"""
    _check(lib().enh_unpatchify_loss(_p(pix, F32, "pix"), _p(target, F32, "target"), B, C, H, W, p, w_l1, w_l2, _p(xrec, F32, "xrec"),
                                     _p(sums, F64, "sums"), _p(dpix, H16, "dpix"), _p(grad_scale, F32, "grad_scale"), _dt(dpix), _stream()), "enh_unpatchify_loss")


def colsum(x, M: int, N: int, out, accumulate: bool = False):
    """out[n] (+)= sum_m x[m, n]; deterministic two-pass form (per-chunk partials in a workspace, fixed-order second pass)"""
    if not DETERMINISTIC:
        _check(lib().enh_colsum_h16(_p(x, H16, "x"), M, N, x.stride(0), _p(out, F32, "out"), int(accumulate), _dt(x), _stream()), "enh_colsum_h16")
        return
    nb = lib().enh_colsum_h16_workspace_bytes(M, N)
    ws = _workspace(nb, x.device)
    _check(lib().enh_colsum_h16_ws(_p(x, H16, "x"), M, N, x.stride(0), _p(out, F32, "out"), int(accumulate), _p(ws), ws.numel(), _dt(x), _stream()), "enh_colsum_h16_ws")


def cast_bf16_head_scaled(x, y, n_scaled: int, alpha: float):
    """y = bf16(x * alpha) for the first n_scaled elements, bf16(x) for the rest"""
    _check(lib().enh_cast_f32_h16_head_scaled(_p(x, F32, "x"), _p(y, H16, "y"), x.numel(), n_scaled, alpha, _dt(y), _stream()), "enh_cast_f32_h16_head_scaled")


def cast_bf16_head_scaled_strided(x0, x_stride: int, y, n: int, n_scaled: int, alpha: float):
    """y [count, n] bf16 (contiguous) <- the blocks x0 + b * x_stride (x0: the first block, a view into the flat f32 store), q rows scaled by alpha"""
    count = y.shape[0]
    _check(lib().enh_cast_f32_h16_head_scaled_strided(_p(x0, F32, "x"), x_stride, _p(y, H16, "y"), y.stride(0), n, n_scaled, alpha, count, _dt(y), _stream()),
           "enh_cast_f32_h16_head_scaled_strided")


def cast_bf16(x, y):
    """y = round-to-nearest-even 16-bit image of x; y's container dtype (torch.bfloat16 / torch.float16) selects the format"""
    _check(lib().enh_cast_f32_h16(_p(x, F32, "x"), _p(y, H16, "y"), x.numel(), _dt(y), _stream()), "enh_cast_f32_h16")


def crop_flip_u8(src, meta, R: int):
    """src uint8 [B,Hs,Ws,3], meta int32 [B,3] = (y0, x0, flip) -> f32 [B,3,R,R] in [0,1] (enh_crop_flip_u8)"""
    B, Hs, Ws, _ = src.shape
    out = torch.empty(B, 3, R, R, dtype=F32, device=src.device)
    _check(lib().enh_crop_flip_u8(_p(src, torch.uint8, "src"), B, Hs, Ws, _p(meta, torch.int32, "meta"), R, _p(out), _stream()), "enh_crop_flip_u8")
    return out


def resize_u8(src, meta, bounds, weights, dst):
    """src uint8 [B,HS,WS,3] -> dst uint8 [B,HD,WD,3] (enh_resize_u8: PIL bilinear resize, per-image sizes / tables in meta, bounds, weights)"""
    B, HS, WS, _ = src.shape
    _, HD, WD, _ = dst.shape
    nb = lib().enh_resize_u8_workspace_bytes(B, HS, WD)
    ws = _workspace(nb, src.device)
    _check(lib().enh_resize_u8(_p(src, torch.uint8, "src"), B, HS, WS, _p(meta, torch.int32, "meta"), _p(bounds, torch.int32, "bounds"),
                               _p(weights, torch.int32, "weights"), _p(dst, torch.uint8, "dst"), HD, WD, _p(ws), ws.numel(), _stream()), "enh_resize_u8")
    return dst


def nonfinite_flag(x, flag):
    """flag[0] = 1.0 if x (f32, flat) holds an inf / nan; untouched otherwise (zero it once per step)"""
    _timed("nonfinite_flag_kernel", 4.0 * x.numel(),
           lambda: _check(lib().enh_nonfinite_flag(_p(x, F32, "x"), x.numel(), _p(flag, F32, "flag"), _stream()), "enh_nonfinite_flag"), unit="byte")


def adamw_step(p, g, m, v, p_bf16, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.99, eps: float = 1e-8,
               weight_decay: float = 1e-4, grad_scale: float = 1.0, skip_flag=None, loss_scale=None):
    # 30 B per parameter: p, g, m, v read (16) + p, m, v written (12) + the bf16 operand shadow written (2)
    _timed("adamw_kernel", (28.0 + (2.0 if p_bf16 is not None else 0.0)) * p.numel(),
           lambda: _check(lib().enh_adamw_step(_p(p, F32, "p"), _p(g, F32, "g"), _p(m, F32, "m"), _p(v, F32, "v"), _p(p_bf16, H16, "p_bf16"),
                                               p.numel(), step, lr, beta1, beta2, eps, weight_decay, grad_scale, _p(skip_flag, F32, "skip_flag"),
                                               _p(loss_scale, F32, "loss_scale"), _dt(p_bf16), _stream()), "enh_adamw_step"), unit="byte")


def loss_scale_update(scale, found_inf, tracker, growth: float = 2.0, backoff: float = 0.5, interval: int = 2000):
    """torch.cuda.amp.GradScaler.update() on the device (scale f32 [1], found_inf f32 [1], tracker int32 [1])"""
    _check(lib().enh_loss_scale_update(_p(scale, F32, "scale"), _p(found_inf, F32, "found_inf"), _p(tracker, torch.int32, "tracker"), growth, backoff, int(interval),
                                       _stream()), "enh_loss_scale_update")


# ------------------------------------------------------------------------------------------------
# discriminator native ops (reference enhancing/losses/op/*)
# ------------------------------------------------------------------------------------------------
def fused_bias_act(x, bias, ref, grad: int, alpha: float, scale: float):
    """fused.fused_bias_act(input, bias, refer, 3, grad, alpha, scale) -> new tensor (the op allocates its output, as the reference's does)."""
    _p(x, F32, "input")
    y = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    size_b = bias.numel() if bias is not None and bias.numel() else 0
    b = bias if size_b else None
    _check(lib().enh_fused_bias_act(_p(x, F32, "input"), _p(b, F32, "bias"), _p(ref if ref is not None and ref.numel() else None, F32, "refer"),
                                    _p(y), x.numel(), step_b, size_b, 3, grad, alpha, scale, _stream()), "enh_fused_bias_act")
    return y


def channel_sum(x):
    """sum over every axis except 1 of an f32 [B, C, ...] tensor -> [C]"""
    _p(x, F32, "x")
    B, C = x.shape[0], x.shape[1]
    inner = x.numel() // (B * C)
    out = torch.empty(C, dtype=F32, device=x.device)
    _check(lib().enh_channel_sum_f32(_p(x), B, C, inner, _p(out), 0, _stream()), "enh_channel_sum_f32")
    return out


def upfirdn2d(x, kernel, up_x: int, up_y: int, down_x: int, down_y: int, pad_x0: int, pad_x1: int, pad_y0: int, pad_y1: int):
    """upfirdn2d_op.upfirdn2d on x viewed [major, in_h, in_w]; returns [major, out_h, out_w]."""
    _p(x, F32, "input"); _p(kernel, F32, "kernel")
    major, in_h, in_w = x.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) // down_y
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) // down_x
    out = torch.empty(major, out_h, out_w, dtype=F32, device=x.device)
    _check(lib().enh_upfirdn2d(_p(x), _p(kernel), _p(out), major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                               _stream()), "enh_upfirdn2d")
    return out


def conv_out_size(n: int, k: int, stride: int, pad: int) -> int:
    return (n + 2 * pad - k) // stride + 1


_DT_OF = {BF16: DT_BF16, F16: DT_F16, F32: DT_F32}


def im2col(x, sb: int, sc: int, B: int, C: int, H: int, W: int, k: int, stride: int, pad: int, dtype: torch.dtype = BF16):
    """x f32, element (b,c,h,w) at b*sb + c*sc + h*W + w  ->  cols [B*Ho*Wo, Kp] in `dtype` (bf16 | fp16: MFMA operands; f32: the exact instrument),
    Kp = C*k*k rounded up to 8 (pad columns zero)."""
    _p(x, F32, "x")
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    ld = (C * k * k + 7) // 8 * 8
    cols = torch.empty(B * Ho * Wo, ld, dtype=dtype, device=x.device)
    _check(lib().enh_im2col(_p(x), sb, sc, B, C, H, W, k, stride, pad, Ho, Wo, _p(cols), ld, _DT_OF[dtype], _stream()), "enh_im2col")
    return cols


def col2im(dcols, B: int, C: int, H: int, W: int, k: int, stride: int, pad: int, out, sb: int, sc: int):
    """adjoint of im2col: dcols [B*Ho*Wo, Kp] (bf16 | fp16 | f32) -> f32 `out`, element (b,c,h,w) at b*sb + c*sc + h*W + w (overwritten)."""
    _p(dcols, (BF16, F16, F32), "dcols"); _p(out, F32, "dx")
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    _check(lib().enh_col2im(_p(dcols), dcols.stride(0), B, C, H, W, k, stride, pad, Ho, Wo, _p(out), sb, sc, _DT_OF[dcols.dtype], _stream()), "enh_col2im")
    return out


# ------------------------------------------------------------------------------------------------
# implicit-GEMM convolutions and element-wise kernels on channels-last bf16 activations (discriminator, LPIPS trunk)
# ------------------------------------------------------------------------------------------------
class ConvGeom(ctypes.Structure):
    """enh_conv_geom of include/enh_hip.h"""
    _fields_ = [(n, _c.c_int) for n in "B Hs Ws C Hm Wm gs oy0 ox0 nty ntx sty stx N HO WO os oph opw".split()]


def _geom(g) -> ConvGeom:
    return g if isinstance(g, ConvGeom) else ConvGeom(**{k: int(v) for k, v in g.items()})


def conv_nhwc(src, wt, geom, mode: int, bias=None, aux=None, add=None, p0: float = 0.0, p1: float = 1.0, out=None):
    """src [B,Hs,Ws,C], wt [N, taps*C] -> out [B,HO,WO,N] (allocated unless given), all in src's 16-bit format (bf16 | fp16); see enh_conv_nhwc_h16"""
    g = _geom(geom)
    if out is None:
        out = torch.empty(g.B, g.HO, g.WO, g.N, dtype=src.dtype, device=src.device)
    work = 2.0 * g.B * g.Hm * g.Wm * g.N * g.nty * g.ntx * g.C
    nb = lib().enh_conv_workspace_bytes(ctypes.byref(g))       # > 0: a small grid that the library splits over the contraction
    ws = _gemm_workspace(src.device, nb) if nb else None
    _timed("conv_igemm_kernel", work,
           lambda: _check(lib().enh_conv_nhwc_h16_ws(_p(src, H16, "src"), _p(wt, H16, "wt"), ctypes.byref(g), mode, _p(bias, F32, "bias"), _p(aux, H16, "aux"),
                                                     _p(add, H16, "add"), p0, p1, _p(out, H16, "out"), _p(ws), nb, _dt(src, wt, aux, add, out), _stream()), "enh_conv_nhwc_h16_ws"))
    return out


def conv_wgrad_nhwc(src, dy, geom):
    """-> f32 [N, taps*C]: sum over the pixels of dy of dy[pix, n] * gathered src[pix, (tap, c)]"""
    g = _geom(geom)
    dw = torch.empty(g.N, g.nty * g.ntx * g.C, dtype=F32, device=src.device)
    nb = lib().enh_conv_wgrad_workspace_bytes(ctypes.byref(g))
    ws = _gemm_workspace(src.device, nb) if nb else None
    work = 2.0 * g.B * g.Hm * g.Wm * g.N * g.nty * g.ntx * g.C
    _timed("conv_wgrad_igemm_kernel", work,
           lambda: _check(lib().enh_conv_wgrad_nhwc_h16(_p(src, H16, "src"), _p(dy, H16, "dy"), ctypes.byref(g), _p(dw), _p(ws), nb, _dt(src, dy), _stream()),
                          "enh_conv_wgrad_nhwc_h16"))
    return dw


def conv_pack_weight(w, scale: float, transposed: bool, kh0: int, kw0: int, kstep: int, nty: int, ntx: int, rows_padded: int, cols_padded: int,
                     dtype: torch.dtype = BF16):
    """w [Cout,Cin,k,k] f32 -> 16-bit operand image [rows_padded, nty*ntx*cols_padded] in `dtype` (bf16 | fp16)"""
    Cout, Cin, k, _ = w.shape
    out = torch.empty(rows_padded, max(nty * ntx * cols_padded, 8), dtype=dtype, device=w.device)
    if nty * ntx:
        _check(lib().enh_conv_pack_weight(_p(w, F32, "w"), Cout, Cin, k, scale, int(transposed), kh0, kw0, kstep, nty, ntx, rows_padded, cols_padded,
                                          _p(out), _dt(out), _stream()), "enh_conv_pack_weight")
    return out


def conv_unpack_wgrad(dwp, Cout: int, Cin: int, cin_padded: int, k: int, scale: float):
    dw = torch.empty(Cout, Cin, k, k, dtype=F32, device=dwp.device)
    _check(lib().enh_conv_unpack_wgrad(_p(dwp, F32, "dwp"), Cout, Cin, cin_padded, k, scale, _p(dw), _stream()), "enh_conv_unpack_wgrad")
    return dw


def blur_nhwc(x, kernel, pad0: int, pad1: int, flip: bool):
    """x [B,H,W,C] bf16, kernel [kh,kw] f32 -> [B, H+pad0+pad1-kh+1, W+pad0+pad1-kw+1, C] bf16"""
    B, H, W, C = x.shape
    kh, kw = kernel.shape
    out = torch.empty(B, H + pad0 + pad1 - kh + 1, W + pad0 + pad1 - kw + 1, C, dtype=x.dtype, device=x.device)
    _check(lib().enh_blur_nhwc_h16(_p(x, H16, "x"), _p(kernel, F32, "kernel"), B, H, W, C, kh, kw, pad0, pad1, pad0, pad1, int(flip), _p(out), _dt(x), _stream()),
           "enh_blur_nhwc_h16")
    return out


def blur_set_kernel(variant: int) -> None:
    """0 = per shape (the row-marching 4 x 4 kernel where it applies), 1 = the one-row kernel everywhere (A/B, tests)"""
    _check(lib().enh_blur_set_kernel(int(variant)), "enh_blur_set_kernel")


def lrelu_gate(g, ref, slope: float, scale: float):
    y = torch.empty_like(g)
    _check(lib().enh_lrelu_gate_h16(_p(g, H16, "g"), _p(ref, H16, "ref"), g.numel(), slope, scale, _p(y), _dt(g, ref), _stream()), "enh_lrelu_gate_h16")
    return y


def img_to_nhwc8(img, dtype: torch.dtype = BF16):
    B, C, H, W = img.shape
    out = torch.empty(B, H, W, 8, dtype=dtype, device=img.device)
    _check(lib().enh_img_to_nhwc8(_p(img, F32, "img"), B, C, H, W, _p(out), _dt(out), _stream()), "enh_img_to_nhwc8")
    return out


def nhwc8_to_img(src, C: int):
    B, H, W, _ = src.shape
    img = torch.empty(B, C, H, W, dtype=F32, device=src.device)
    _check(lib().enh_nhwc8_to_img(_p(src, H16, "src"), B, C, H, W, _p(img), _dt(src), _stream()), "enh_nhwc8_to_img")
    return img


def minibatch_stddev_nhwc(x, group: int, Cp: int):
    """x [B,H,W,C] bf16 -> [B,H,W,Cp] bf16: x, the slot's mean standard deviation in channel C, zeros above (enh_minibatch_stddev_nhwc)"""
    B, H, W, C = x.shape
    out = torch.empty(B, H, W, Cp, dtype=x.dtype, device=x.device)
    _check(lib().enh_minibatch_stddev_nhwc(_p(x, H16, "x"), B, H * W, C, Cp, group, _p(out), _dt(x), _stream()), "enh_minibatch_stddev_nhwc")
    return out


def minibatch_stddev_nhwc_backward(x, g, group: int):
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    _check(lib().enh_minibatch_stddev_nhwc_backward(_p(x, H16, "x"), _p(g, H16, "g"), B, H * W, C, g.shape[3], group, _p(dx), _dt(x, g), _stream()),
           "enh_minibatch_stddev_nhwc_backward")
    return dx


def colsum_nhwc(x):
    """sum over every axis but the last of a channels-last bf16 tensor -> f32 [C]; narrow C is folded so that a row of the kernel covers 512 columns"""
    C = x.shape[-1]
    M = x.numel() // C
    r = 1
    while C * r < 512 and M % (2 * r) == 0:
        r *= 2
    out = torch.empty(C * r, dtype=F32, device=x.device)
    colsum(x.view(M // r, C * r), M // r, C * r, out)
    return out.view(r, C).sum(0) if r > 1 else out


# ------------------------------------------------------------------------------------------------
# dtype-dispatching front-ends (bf16 product path / fp32 exact mode) used by the engine
# ------------------------------------------------------------------------------------------------
def mm(a, b, M: int, N: int, K: int, out, trans_a: bool = False, trans_b: bool = False, bias=None, act: int = ACT_NONE, aux=None, res=None,
       res_rows: int = 0, accumulate: bool = False):
    """C = epilogue(A B^T) into `out`; bf16 operands -> MFMA kernel (out may be bf16 or f32), f32 operands -> exact f32 kernel."""
    if a.dtype in H16:
        if out.dtype in H16:
            gemm(a, b, M, N, K, trans_a, trans_b, bias, act, aux, res, res_rows, accumulate, out_bf16=out)
        else:
            gemm(a, b, M, N, K, trans_a, trans_b, bias, act, aux, res, res_rows, accumulate, out_f32=out)
        return
    _check(lib().enh_gemm_f32(_p(a, F32, "A"), a.stride(0), int(trans_a), _p(b, F32, "B"), b.stride(0), int(trans_b), M, N, K, _p(bias, F32, "bias"), act,
                              _p(aux, F32, "aux"), aux.stride(0) if aux is not None else 0, _p(res, F32, "res"), res.stride(0) if res is not None else 0,
                              res_rows if res is not None else 0, int(accumulate), _p(out, F32, "out"), out.stride(0), _stream()), "enh_gemm_f32")


def ln_fwd(x, w, b, y, mean, rstd, y_extra_f32=None):
    if y.dtype in H16:
        layernorm_forward(x, w, b, 1e-5, y, y_extra_f32, mean, rstd)
    else:
        layernorm_forward(x, w, b, 1e-5, None, y, mean, rstd)


def ln_bwd(dy, x, w, mean, rstd, dres, dx, dx_operand, dw, db, dx_colsum=None):
    """dx_operand: the tensor the following GEMMs read (a bf16 copy in the product path, dx itself in exact mode)."""
    layernorm_backward(dy, x, w, mean, rstd, dres, dx, dx_operand if dx_operand.dtype in H16 else None, dw, db, dx_colsum)


def attn_fwd(qkv, B, N, H, scale, out, lse, q_prescaled: bool = False):
    if qkv.dtype in H16:
        attention_forward(qkv, B, N, H, scale, out, lse, q_prescaled)
    else:
        _check(lib().enh_attention_forward_f32(_p(qkv, F32, "qkv"), B, N, H, scale, _p(out, F32, "out"), _p(lse, F32, "lse"), _stream()), "enh_attention_forward_f32")


def attn_bwd(qkv, out, dout, lse, B, N, H, scale, dqkv, delta_ws, q_prescaled: bool = False):
    if qkv.dtype in H16:
        attention_backward(qkv, out, dout, lse, B, N, H, scale, dqkv, delta_ws, q_prescaled)
    else:
        _check(lib().enh_attention_backward_f32(_p(qkv, F32, "qkv"), _p(out, F32, "out"), _p(dout, F32, "dout"), _p(lse, F32, "lse"), B, N, H, scale,
                                                _p(dqkv, F32, "dqkv"), _p(delta_ws, F32, "delta_ws"), _stream()), "enh_attention_backward_f32")


# ------------------------------------------------------------------------------------------------
# x3 split-bf16 operands (parity-grade encoder forward; include/enh_hip.h "x3")
# ------------------------------------------------------------------------------------------------
def split3(x, y3, bias=None, act: int = ACT_NONE, order: int = 0, y_hi=None):
    """x f32 [M,K] -> y3 bf16 [M,3K] = [hi | lo | hi] (order 0) / [hi | hi | lo] (order 1) of f(x + bias); y_hi: optional bf16 [M,K] copy of the hi plane"""
    M, K = x.shape
    # algorithmic HBM bytes: 4 in + 6 out (+ 2)
    _timed("split3_kernel", (10.0 + (2.0 if y_hi is not None else 0.0)) * M * K,
           lambda: _check(lib().enh_split3_bf16(_p(x, F32, "x"), x.stride(0), M, K, _p(bias, F32, "bias"), act, order, _p(y3, BF16, "y3"), y3.stride(0),
                                                _p(y_hi, BF16, "y_hi"), y_hi.stride(0) if y_hi is not None else 0, _stream()), "enh_split3_bf16"), unit="byte")


def split2(x, hi, lo):
    _timed("split2_kernel", 8.0 * x.numel(),
           lambda: _check(lib().enh_split2_bf16(_p(x, F32, "x"), x.numel(), _p(hi, BF16, "hi"), _p(lo, BF16, "lo"), _stream()), "enh_split2_bf16"), unit="byte")


def gemm_split_fused(M: int, N: int, K: int) -> bool:
    """does enh_gemm_bf16_split serve this shape (the persistent 256 x 256 kernel)?  Otherwise: mm -> f32, then split2 / split3."""
    return bool(lib().enh_gemm_bf16_split_fused(M, N, K))


def _split_label(K: int, mode: int) -> str:
    """the symbol rocprofv3 reports for an enh_gemm_bf16_split call (the A-in-registers form serves an even number >= 6 of K stages)"""
    dyn = "true" if _DYN_SCHEDULE[0] else "false"
    nst = K // 64
    return f"gemm_w256r_kernel<BF16, false, {mode}, {dyn}>" if (nst % 2 == 0 and nst >= 6) else f"gemm_w256p_kernel<BF16, false, false, {mode}, {dyn}>"


def _poff(t: torch.Tensor, elems: int):
    return ctypes.c_void_p(t.data_ptr() + elems * t.element_size())


def gemm_split2(a, b, M: int, N: int, K: int, hi, lo):
    """hi = bf16(a b^T), lo = bf16(a b^T - hi): mm(...) -> f32 followed by split2, without the f32 round trip (bit-identical)"""
    _timed(_split_label(K, 8), 2.0 * M * N * K,
           lambda: _check(lib().enh_gemm_bf16_split(_p(a, BF16, "A"), a.stride(0), _p(b, BF16, "B"), b.stride(0), M, N, K, None, ACT_NONE,
                                                    _p(hi, BF16, "hi"), hi.stride(0), _p(lo, BF16, "lo"), lo.stride(0), None, 0, None, 0, _stream()), "enh_gemm_bf16_split"))


def gemm_split3_tanh(a, b, M: int, N: int, K: int, bias, y3, y_hi=None):
    """y3 [M, 3N] = the x3 row [hi | lo | hi] of tanh(a b^T + bias) (+ y_hi [M, N] = the hi plane): mm(...) -> f32 followed by split3(act = tanh), fused"""
    if not (y3.is_cuda and y3.is_contiguous() and y3.dtype == BF16 and y3.shape[-1] == 3 * N):
        raise RuntimeError("y3 must be a contiguous bf16 [M, 3N] device tensor")
    ld3 = y3.stride(0)
    _timed(_split_label(K, 9), 2.0 * M * N * K,
           lambda: _check(lib().enh_gemm_bf16_split(_p(a, BF16, "A"), a.stride(0), _p(b, BF16, "B"), b.stride(0), M, N, K, _p(bias, F32, "bias"), ACT_TANH,
                                                    _poff(y3, 0), ld3, _poff(y3, N), ld3, _poff(y3, 2 * N), ld3,
                                                    _p(y_hi, BF16, "y_hi"), y_hi.stride(0) if y_hi is not None else 0, _stream()), "enh_gemm_bf16_split"))


def ln_fwd_x3(x, w, b, y3, mean, rstd, y_bf16=None, y_f32=None):
    M, D = x.shape
    _check(lib().enh_layernorm_forward_x3(_p(x, F32, "x"), _p(w, F32, "w"), _p(b, F32, "b"), M, D, 1e-5, _p(y3, BF16, "y3"), _p(y_bf16, BF16, "y_bf16"),
                                          _p(y_f32, F32, "y_f32"), _p(mean, F32, "mean"), _p(rstd, F32, "rstd"), _stream()), "enh_layernorm_forward_x3")


def attention_forward_x3(qkv_hi, qkv_lo, B: int, N: int, H: int, scale: float, out3, out_bf16, lse):
    # executed MFMA work: three passes of the 4 N^2 64 algorithmic FLOP per (image, head)
    _timed("attn_fwd_x3_kernel", 3 * 4.0 * B * H * N * N * 64,
           lambda: _check(lib().enh_attention_forward_x3(_p(qkv_hi, BF16, "qkv_hi"), _p(qkv_lo, BF16, "qkv_lo"), B, N, H, scale, _p(out3, BF16, "out3"),
                                                         _p(out_bf16, BF16, "out_bf16"), _p(lse, F32, "lse"), _stream()), "enh_attention_forward_x3"))


def colsum_any(x, M, N, out, accumulate=False):
    if x.dtype in H16:
        colsum(x, M, N, out, accumulate)
    else:
        _check(lib().enh_colsum_f32(_p(x, F32, "x"), M, N, x.stride(0), _p(out, F32, "out"), int(accumulate), _stream()), "enh_colsum_f32")


def patchify_any(img, p, out):
    if out.dtype in H16:
        patchify(img, p, out)
    else:
        B, C, H, W = img.shape
        _check(lib().enh_patch_perm_f32(_p(img, F32, "img"), _p(out, F32, "patches"), B, C, H, W, p, 1, _stream()), "enh_patch_perm_f32")


def unpatchify_loss_any(pix, target, B, C, H, W, p, w_l1, w_l2, xrec, sums, dpix, grad_scale=None):
    if dpix is None or dpix.dtype in H16 or target is None:
        unpatchify_loss(pix, target, B, C, H, W, p, w_l1, w_l2, xrec, sums, dpix, grad_scale)
    else:
        _check(lib().enh_unpatchify_loss_f32(_p(pix, F32, "pix"), _p(target, F32, "target"), B, C, H, W, p, w_l1, w_l2, _p(xrec, F32, "xrec"),
                                             _p(sums, F64, "sums"), _p(dpix, F32, "dpix"), _stream()), "enh_unpatchify_loss_f32")


# ------------------------------------------------------------------------------------------------
# LPIPS (lpips 0.1.4, net="vgg"): channels-last bf16 activations
# ------------------------------------------------------------------------------------------------
def conv3x3_nhwc(x, wt, B: int, H: int, W: int, Cin: int, Cout: int, out, bias=None, mode: int = 0, aux=None, add=None):
    _timed("conv3x3_igemm_kernel", 2.0 * B * H * W * Cout * 9 * Cin,
           lambda: _check(lib().enh_conv3x3_nhwc_h16(_p(x, H16, "x"), _p(wt, H16, "wt"), B, H, W, Cin, Cout, _p(bias, F32, "bias"), mode, _p(aux, H16, "aux"),
                                                     _p(add, H16, "add"), _p(out, H16, "out"), _dt(x, wt, aux, add, out), _stream()), "enh_conv3x3_nhwc_h16"))
    return out


def vgg_conv1(img, w, bias, shift, scale, normalize: bool, out):
    B, _, H, W = img.shape
    _check(lib().enh_vgg_conv1(_p(img, F32, "img"), _p(w, F32, "w"), _p(bias, F32, "bias"), _p(shift, F32, "shift"), _p(scale, F32, "scale"), int(normalize), B, H, W,
                               _p(out, H16, "out"), _dt(out), _stream()), "enh_vgg_conv1")
    return out


def vgg_conv1_backward(gpre, w, scale, normalize: bool, B: int, H: int, W: int, dimg):
    _check(lib().enh_vgg_conv1_backward(_p(gpre, H16, "gpre"), _p(w, F32, "w"), _p(scale, F32, "scale"), int(normalize), B, H, W, _p(dimg, F32, "dimg"), _dt(gpre), _stream()),
           "enh_vgg_conv1_backward")
    return dimg


def maxpool2_nhwc(x, B: int, H: int, W: int, C: int, y):
    _check(lib().enh_maxpool2_nhwc_h16(_p(x, H16, "x"), B, H, W, C, _p(y, H16, "y"), _dt(x, y), _stream()), "enh_maxpool2_nhwc_h16")
    return y


def maxpool2_nhwc_backward(x, gy, add, B: int, H: int, W: int, C: int, gx):
    _check(lib().enh_maxpool2_nhwc_h16_backward(_p(x, H16, "x"), _p(gy, H16, "gy"), _p(add, H16, "add"), B, H, W, C, _p(gx, H16, "gx"), _dt(x, gy, add, gx), _stream()),
           "enh_maxpool2_nhwc_h16_backward")
    return gx


def lpips_head(feat, lin, B: int, HW: int, C: int, val_ws, out, accumulate: bool):
    _check(lib().enh_lpips_head(_p(feat, H16, "feat"), _p(lin, F32, "lin"), B, HW, C, _p(val_ws, F32, "val_ws"), _p(out, F32, "out"), int(accumulate), _dt(feat), _stream()),
           "enh_lpips_head")


def lpips_head_backward(feat, lin, gout, B: int, HW: int, C: int, dfeat1):
    _check(lib().enh_lpips_head_backward(_p(feat, H16, "feat"), _p(lin, F32, "lin"), _p(gout, F32, "gout"), B, HW, C, _p(dfeat1, H16, "dfeat1"), _dt(feat, dfeat1), _stream()),
           "enh_lpips_head_backward")
