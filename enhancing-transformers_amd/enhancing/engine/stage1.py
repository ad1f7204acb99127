"""Static HIP schedule of the stage-1 tokenizer: ViT encoder -> pre_quant -> VQ/RQ -> post_quant -> ViT decoder ->
pixel + codebook loss -> backward -> AdamW, for one process = one MI355X.

This replaces what the reference gets from ``pl.Trainer.fit`` + autograd over stock PyTorch ops
(reference main.py:51-61, vitvqgan.py:44-72,101-115,152-160): the forward and the hand-derived backward are
explicit sequences of C-ABI kernel launches (``enhancing._C``) on the current HIP stream, over

  * ONE flat fp32 parameter buffer (+ flat grad, Adam m / v, bf16 shadow) that the ``nn.Parameter``s of the
    module tree are views of — so ``state_dict`` / ``load_state_dict`` / checkpoints keep the reference's keys,
    AdamW is a single launch and DDP all-reduces contiguous slices;
  * a per-batch-size activation arena allocated once (the residual stream and LayerNorm statistics stay fp32,
    GEMM operands are bf16, attention saves only the row log-sum-exp).

PyTorch is used for device memory, streams and ``torch.distributed`` only.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _C

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
# precision modes of the engine: the two product paths ("bf16" / "fp16": 16-bit MFMA operands of that format, fp32 accumulation / residual stream /
# statistics / master weights) and the exact mode ("fp32": every operand fp32, vector-ALU kernels — the parity instrument)
OPERAND_DTYPE = {"bf16": BF16, "fp16": F16, "fp32": F32}
_ALIGN = 64  # elements; keeps every parameter view 16-byte aligned in both the fp32 and the bf16 buffer


class ParamStore:
    """Flattens the trainable parameters of a module tree into contiguous device buffers."""

    def __init__(self, module: nn.Module, device: torch.device, precision: str = "bf16", prefixes: Optional[Tuple[str, ...]] = None) -> None:
        """prefixes: restrict the store to parameters whose name starts with one of them (one store per optimizer)"""
        self.device = device
        self.precision = precision
        self.half = precision in ("bf16", "fp16")          # a product path: 16-bit operand shadow of the weights, rewritten by the AdamW kernel
        self.op_dtype = OPERAND_DTYPE[precision]
        mine = (lambda n: True) if prefixes is None else (lambda n: n.startswith(prefixes))
        params = [(n, p) for n, p in module.named_parameters() if p.requires_grad and mine(n)]
        self.names, self.offsets, total = self.layout(params)
        self.numel = total
        self.p = torch.zeros(total, dtype=F32, device=device)
        self.g = torch.zeros(total, dtype=F32, device=device)
        self.m = torch.zeros(total, dtype=F32, device=device)
        self.v = torch.zeros(total, dtype=F32, device=device)
        self.p16 = torch.zeros(total, dtype=self.op_dtype if self.half else BF16, device=device)
        self.w: Dict[str, torch.Tensor] = {}
        self.w16: Dict[str, torch.Tensor] = {}
        self.grad: Dict[str, torch.Tensor] = {}
        for n, p in params:
            off, cnt, shp = self.offsets[n]
            view = self.p[off:off + cnt].view(shp)
            view.copy_(p.data.to(device=device, dtype=F32))
            p.data = view                      # the nn.Parameter now aliases the flat buffer
            p.grad = self.g[off:off + cnt].view(shp)
            self.w[n], self.w16[n], self.grad[n] = view, self.p16[off:off + cnt].view(shp), p.grad
        # frozen tensors (position tables) just move to the device
        for n, p in module.named_parameters():
            if not p.requires_grad and mine(n):
                p.data = p.data.to(device=device, dtype=F32).contiguous()
                self.w[n] = p.data
        # GEMM operand view of every weight: the 16-bit shadow (product paths) or the fp32 master itself (exact mode)
        self.wa = self.w16 if self.half else self.w
        self.step_count = 0
        self.comm_done = False             # data-parallel: this step's gradient all-reduce has already been waited for (unscale_grads before the step)
        self.grads_unscaled = False        # fp16 engine: the flat gradient holds loss_scale x gradient from backward until unscale_grads() / the optimizer step
        self.version = 0                   # bumped whenever the fp32 masters may have changed: lazily rebuilt operand images (the x3 weights) compare it
        self.operand_hooks: List = []      # callables that rebuild derived GEMM operands (e.g. the towers' pre-scaled q | k | v weights) from the masters
        self.refresh_shadows()

    @staticmethod
    def layout(params):
        """flat layout of [(name, tensor)]: registration order, every parameter padded to _ALIGN elements -> (names, {name: (offset, numel, shape)}, total).
        Pure host arithmetic, shared with the CPU tests of the data-parallel bucket cover."""
        names: List[str] = []
        offsets: Dict[str, Tuple[int, int, torch.Size]] = {}
        total = 0
        for n, p in params:
            names.append(n)
            offsets[n] = (total, p.numel(), p.shape)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        return names, offsets, total

    @staticmethod
    def slice_from(names, offsets, prefix: str) -> Tuple[int, int]:
        offs = [offsets[n] for n in names if n.startswith(prefix)]
        begin = min(o for o, _, _ in offs)
        end = max((o + c + _ALIGN - 1) // _ALIGN * _ALIGN for o, c, _ in offs)
        return begin, end

    def refresh_shadows(self) -> None:
        """to be called after ANY write to the fp32 masters that did not go through Stage1Engine.optimizer_step (initial broadcast, weight loading)"""
        if self.half:
            _C.cast_bf16(self.p, self.p16)
        self.refresh_operands()

    def refresh_operands(self) -> None:
        self.version += 1
        for hook in self.operand_hooks:
            hook()

    def zero_grad(self) -> None:
        self.g.zero_()
        self.grads_unscaled = False      # (fp16 engine: see Stage1Engine.unscale_grads)
        self.comm_done = False

    def slice_of(self, prefix: str) -> Tuple[int, int]:
        """[begin, end) range of the flat buffers covered by parameters whose name starts with prefix."""
        return self.slice_from(self.names, self.offsets, prefix)


def backward_unit_order(enc_depth: int, dec_depth: int) -> List[str]:
    """the parameter-name prefixes Stage1Engine.backward_from announces to the gradient synchroniser, in announce order = the order in which the
    backward schedule finishes them (pixel head, decoder final norm, decoder layers last -> first, quantizer block, encoder likewise, patch embedding).
    Each prefix is one contiguous slice of the flat gradient buffer; together they must cover it without gaps (tests/test_ddp_cpu.py at base size)."""
    order = ["decoder.to_pixel.", "decoder.transformer.norm."]
    order += [f"decoder.transformer.layers.{i}." for i in range(dec_depth - 1, -1, -1)]
    order += ["post_quant.", "pre_quant.", "quantizer.", "encoder.transformer.norm."]
    order += [f"encoder.transformer.layers.{i}." for i in range(enc_depth - 1, -1, -1)]
    order += ["encoder.to_patch_embedding."]
    return order


class _Tower:
    """One pre-norm transformer stack (reference layers.py:135-150) as a static kernel schedule."""

    def __init__(self, store: ParamStore, prefix: str, dim: int, depth: int, heads: int, mlp: int, n_tok: int) -> None:
        self.s, self.prefix = store, prefix
        self.dim, self.depth, self.heads, self.mlp, self.n_tok = dim, depth, heads, mlp, n_tok
        self.inner = heads * 64
        self.scale = 64 ** -0.5
        self._bufs: Dict[Tuple[int, bool], dict] = {}
        t = prefix + "transformer."
        self.L = []
        for i in range(depth):
            q = f"{t}layers.{i}."
            self.L.append(dict(
                ln1_w=q + "0.norm.weight", ln1_b=q + "0.norm.bias", wqkv=q + "0.fn.to_qkv.weight",
                wout=q + "0.fn.to_out.weight", bout=q + "0.fn.to_out.bias",
                ln2_w=q + "1.norm.weight", ln2_b=q + "1.norm.bias", w1=q + "1.fn.net.0.weight", b1=q + "1.fn.net.0.bias",
                w2=q + "1.fn.net.2.weight", b2=q + "1.fn.net.2.bias"))
        self.lnf_w, self.lnf_b = t + "norm.weight", t + "norm.bias"
        self.first_bias_grad = None  # set by the engine: grad buffer of the bias that feeds layer 0's input (patch-embed / post_quant)
        # FORWARD operand of to_qkv in the product path: the bf16 image of the fp32 master whose q rows are multiplied by scale * log2(e) before the (single)
        # rounding, so that the attention kernels see log2-domain logits and feed -max / -lse through the MFMA C operand (include/enh_hip.h
        # enh_attention_forward, q_prescaled).  The BACKWARD GEMMs keep the plain shadow: the attention backward returns the gradient with respect to the
        # unscaled q, so neither the input gradient nor the weight gradient needs a correction.  Rebuilt whenever the masters change (store hook).
        # Known asymmetry: forward operand / alpha and the backward operand bf16(W_q) are two separate roundings of the same master, so they differ by
        # up to one bf16 ulp (2^-7 relative worst case, 2^-9 rms) — the size of the rounding either already carries against the master; bounded in
        # tests/test_ops_gpu.py::test_head_scaled_cast_and_the_forward_backward_operand_gap.  x3 and fp32 modes use the unscaled operand both ways.
        self.q_prescaled = store.half and os.environ.get("ENH_ATTN_PRESCALE", "1") != "0"
        self.wqkv_fwd: List[torch.Tensor] = []
        if self.q_prescaled:
            self._wqkv_fwd_all = torch.empty(depth, 3 * self.inner, dim, dtype=store.op_dtype, device=store.device)
            self.wqkv_fwd = [self._wqkv_fwd_all[i] for i in range(depth)]
            store.operand_hooks.append(self.refresh_qkv_operands)
            self.refresh_qkv_operands()

    def refresh_qkv_operands(self) -> None:
        """one launch for the tower when its layers sit equally spaced in the flat store (identical layers registered in order: always, today)"""
        alpha = self.scale * 1.4426950408889634
        src = [self.s.w[P["wqkv"]] for P in self.L]
        esz = src[0].element_size()
        steps = {(b.data_ptr() - a.data_ptr()) // esz for a, b in zip(src, src[1:])}
        if len(src) > 1 and len(steps) == 1 and min(steps) > 0 and min(steps) % 4 == 0 and all(t.is_contiguous() for t in src):
            _C.cast_bf16_head_scaled_strided(src[0], steps.pop(), self._wqkv_fwd_all, 3 * self.inner * self.dim, self.inner * self.dim, alpha)
            return
        for t, dst in zip(src, self.wqkv_fwd):
            _C.cast_bf16_head_scaled(t, dst, self.inner * self.dim, alpha)

    # ---- activation arena --------------------------------------------------------------------
    def bufs(self, B: int, save: bool) -> dict:
        key = (B, save)
        if key in self._bufs:
            return self._bufs[key]
        dev, M, H, N = self.s.device, B * self.n_tok, self.heads, self.n_tok
        BF16 = self.s.op_dtype  # activation-operand dtype of this precision mode (bf16 | fp16 | fp32)
        e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)
        n_layer_sets = self.depth if save else 1
        layers = []
        for _ in range(n_layer_sets):
            layers.append(dict(a1=e(M, self.dim, dt=BF16), qkv=e(M, 3 * self.inner, dt=BF16), o=e(M, self.inner, dt=BF16),
                               lse=e(B, H, N), x_mid=e(M, self.dim), a2=e(M, self.dim, dt=BF16), hid=e(M, self.mlp, dt=BF16),
                               mean1=e(M), rstd1=e(M), mean2=e(M), rstd2=e(M)))
        b = dict(layers=layers, x=[e(M, self.dim) for _ in range(self.depth + 1 if save else 2)],
                 xf16=e(M, self.dim, dt=BF16), xf32=e(M, self.dim), meanf=e(M), rstdf=e(M))
        if save:  # backward scratch, shared by all layers (in exact mode the "16" operand copies ARE the f32 tensors)
            exact = not self.s.half
            gA, gB = e(M, self.dim), e(M, self.dim)
            b.update(gA=gA, gA16=gA if exact else e(M, self.dim, dt=BF16), gB=gB, gB16=gB if exact else e(M, self.dim, dt=BF16),
                     dA=e(M, self.dim, dt=F32 if exact else BF16), dhid16=e(M, self.mlp, dt=BF16), do16=e(M, self.inner, dt=BF16),
                     dqkv16=e(M, 3 * self.inner, dt=BF16), delta=e(B, H, N))
        self._bufs[key] = b
        return b

    def input_buffer(self, B: int, save: bool) -> torch.Tensor:
        return self.bufs(B, save)["x"][0]

    # ---- x3 (split-bf16) operands: the parity-grade forward (csrc/x3.hip) -----------------------
    def x3_weights(self) -> List[dict]:
        """K-concatenated [hi | hi | lo] images of the layer weights (B operands of the three-pass products), rebuilt lazily after the masters changed"""
        if getattr(self, "_w3_version", None) != self.s.version:
            if not hasattr(self, "_w3"):
                e = lambda n, k: torch.empty(n, 3 * k, dtype=torch.bfloat16, device=self.s.device)
                self._w3 = [dict(wqkv=e(3 * self.inner, self.dim), wout=e(self.dim, self.inner), w1=e(self.mlp, self.dim), w2=e(self.dim, self.mlp))
                            for _ in range(self.depth)]
            for P, W in zip(self.L, self._w3):
                for k in ("wqkv", "wout", "w1", "w2"):
                    _C.split3(self.s.w[P[k]], W[k], order=1)
            self._w3_version = self.s.version
        return self._w3

    def x3_bufs(self, B: int) -> dict:
        """scratch of the x3 forward, shared by all layers: nothing here is read by the backward (it reads the hi planes saved in the layer arena)"""
        cache = self.__dict__.setdefault("_x3_bufs", {})
        if B not in cache:
            dev, M = self.s.device, B * self.n_tok
            e = lambda *shape, dt=torch.bfloat16: torch.empty(*shape, dtype=dt, device=dev)
            f32 = e(M * max(3 * self.inner, self.mlp), dt=F32)            # the qkv projection / fc1 pre-activation in f32 (never alive together)
            big3 = e(M * 3 * max(self.inner, self.mlp))                    # x3 rows of the attention output / of tanh(fc1)
            cache[B] = dict(a3=e(M, 3 * self.dim), qkv32=f32[:M * 3 * self.inner].view(M, 3 * self.inner), fc32=f32[:M * self.mlp].view(M, self.mlp),
                            qkv_lo=e(M, 3 * self.inner), o3=big3[:M * 3 * self.inner].view(M, 3 * self.inner), hid3=big3[:M * 3 * self.mlp].view(M, 3 * self.mlp),
                            xf3=e(M, 3 * self.dim))
        return cache[B]

    def forward_x3(self, B: int, save: bool, want_f32: bool = False) -> dict:
        """forward() with every product formed from split-bf16 operands (reference layers.py:118-132,145-150 in ~fp32 precision on the bf16 matrix
        cores).  With save=True the arena receives exactly what the bf16 forward would have saved (the hi planes), so backward() is unchanged —
        except that q is NOT pre-scaled here (recorded in the buffer dict)."""
        s, b, X, W3 = self.s, self.bufs(B, save), self.x3_bufs(B), self.x3_weights()
        if s.precision != "bf16" and save:
            raise RuntimeError("the x3 forward saves bf16 hi planes for a bf16 backward: training with an x3 tower needs precision='bf16'")
        M, dim, inner, mlp = B * self.n_tok, self.dim, self.inner, self.mlp
        fuse = os.environ.get("ENH_X3_FUSED_SPLIT", "1") != "0"      # A/B switch: the round-4 form (GEMM -> f32, split kernel) with 0
        # (asked per launch, not once per forward: the plan depends on the CU budget and the kernel-family override, which a collective's begin / end or an
        # A/B switch may change between two layers — the launcher would then refuse the fused form mid-forward: ADVICE r5)
        fused_qkv = lambda: fuse and _C.gemm_split_fused(M, 3 * inner, 3 * dim)
        fused_fc1 = lambda: fuse and _C.gemm_split_fused(M, mlp, 3 * dim)
        x = b["x"][0]
        for i, P in enumerate(self.L):
            A, W = b["layers"][i if save else 0], W3[i]
            if not save and A["qkv"].dtype != torch.bfloat16:      # (no-save arena of an fp16 engine: the buffer is scratch here, its bits are bf16 hi planes)
                A = dict(A, qkv=A["qkv"].view(torch.bfloat16))
            _C.ln_fwd_x3(x, s.w[P["ln1_w"]], s.w[P["ln1_b"]], X["a3"], A["mean1"], A["rstd1"], y_bf16=A["a1"] if save else None)
            if fused_qkv():      # hi / lo planes straight from the GEMM's epilogue (no f32 [M, 3 inner] round trip)
                _C.gemm_split2(X["a3"], W["wqkv"], M, 3 * inner, 3 * dim, A["qkv"], X["qkv_lo"])
            else:
                _C.mm(X["a3"], W["wqkv"], M, 3 * inner, 3 * dim, X["qkv32"])
                _C.split2(X["qkv32"], A["qkv"], X["qkv_lo"])
            _C.attention_forward_x3(A["qkv"], X["qkv_lo"], B, self.n_tok, self.heads, self.scale, X["o3"], A["o"] if save else None, A["lse"])
            _C.mm(X["o3"], W["wout"], M, dim, 3 * inner, A["x_mid"], bias=s.w[P["bout"]], res=x, res_rows=M)
            _C.ln_fwd_x3(A["x_mid"], s.w[P["ln2_w"]], s.w[P["ln2_b"]], X["a3"], A["mean2"], A["rstd2"], y_bf16=A["a2"] if save else None)
            if fused_fc1():
                _C.gemm_split3_tanh(X["a3"], W["w1"], M, mlp, 3 * dim, s.w[P["b1"]], X["hid3"], A["hid"] if save else None)
            else:
                _C.mm(X["a3"], W["w1"], M, mlp, 3 * dim, X["fc32"])
                _C.split3(X["fc32"], X["hid3"], bias=s.w[P["b1"]], act=_C.ACT_TANH, y_hi=A["hid"] if save else None)
            x_next = b["x"][i + 1] if save else b["x"][(i + 1) & 1]
            _C.mm(X["hid3"], W["w2"], M, dim, 3 * mlp, x_next, bias=s.w[P["b2"]], res=A["x_mid"], res_rows=M)
            x = x_next
        b["x_last"] = x
        _C.ln_fwd_x3(x, s.w[self.lnf_w], s.w[self.lnf_b], X["xf3"], b["meanf"], b["rstdf"], y_bf16=b["xf16"] if save else None,
                     y_f32=b["xf32"] if want_f32 else None)
        b["xf3"], b["x3"], b["q_prescaled"] = X["xf3"], True, False
        return b

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, B: int, save: bool, want_f32: bool = False) -> dict:
        """x[0] must already hold the tower input.  Returns the buffer dict; output = b['xf16'] (+ b['xf32'])."""
        s, b = self.s, self.bufs(B, save)
        M, dim, inner, mlp = B * self.n_tok, self.dim, self.inner, self.mlp
        x = b["x"][0]
        for i, P in enumerate(self.L):
            A = b["layers"][i if save else 0]
            _C.ln_fwd(x, s.w[P["ln1_w"]], s.w[P["ln1_b"]], A["a1"], A["mean1"], A["rstd1"])
            _C.mm(A["a1"], self.wqkv_fwd[i] if self.q_prescaled else s.wa[P["wqkv"]], M, 3 * inner, dim, A["qkv"])
            _C.attn_fwd(A["qkv"], B, self.n_tok, self.heads, self.scale, A["o"], A["lse"], self.q_prescaled)
            _C.mm(A["o"], s.wa[P["wout"]], M, dim, inner, A["x_mid"], bias=s.w[P["bout"]], res=x, res_rows=M)
            _C.ln_fwd(A["x_mid"], s.w[P["ln2_w"]], s.w[P["ln2_b"]], A["a2"], A["mean2"], A["rstd2"])
            _C.mm(A["a2"], s.wa[P["w1"]], M, mlp, dim, A["hid"], bias=s.w[P["b1"]], act=_C.ACT_TANH)
            x_next = b["x"][i + 1] if save else b["x"][(i + 1) & 1]
            _C.mm(A["hid"], s.wa[P["w2"]], M, dim, mlp, x_next, bias=s.w[P["b2"]], res=A["x_mid"], res_rows=M)
            x = x_next
        b["x_last"] = x
        _C.ln_fwd(x, s.w[self.lnf_w], s.w[self.lnf_b], b["xf16"], b["meanf"], b["rstdf"], b["xf32"] if want_f32 else None)
        if want_f32 and b["xf16"].dtype == F32:
            b["xf32"] = b["xf16"]
        b["x3"], b["q_prescaled"] = False, self.q_prescaled
        return b

    # ---- backward ----------------------------------------------------------------------------
    def backward(self, B: int, d_xf: torch.Tensor, on_layer_done=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """d_xf: grad wrt the final-LayerNorm output, f32 [M, dim].  Accumulates parameter grads; returns the grad
        wrt the tower input as (f32, bf16)."""
        s, b = self.s, self.bufs(B, True)
        M, dim, inner, mlp = B * self.n_tok, self.dim, self.inner, self.mlp
        g = s.grad
        q_pre = b.get("q_prescaled", self.q_prescaled)      # convention of the qkv tensor the LAST forward saved (the x3 forward stores unscaled q)
        gA, gA16, gB, gB16, dA = b["gA"], b["gA16"], b["gB"], b["gB16"], b["dA"]
        # every LN backward also emits the column sums of the residual-stream gradient it produces = the bias gradient of
        # the Linear (fc2 / to_out) that wrote that stream
        _C.ln_bwd(d_xf, b["x"][self.depth], s.w[self.lnf_w], b["meanf"], b["rstdf"], None, gA, gA16,
                  g[self.lnf_w], g[self.lnf_b], g[self.L[-1]["b2"]] if self.depth else None)
        if on_layer_done is not None:
            on_layer_done(f"{self.prefix}transformer.norm.")
        for i in range(self.depth - 1, -1, -1):
            P, A = self.L[i], b["layers"][i]
            # ---- MLP: x_out = fc2(tanh(fc1(a2))) + x_mid ----
            _C.mm(gA16, A["hid"], dim, mlp, M, g[P["w2"]], trans_a=True, trans_b=True, accumulate=True)
            if gA16.dtype in _C.H16:   # input gradient through the tanh + fc1's bias gradient (column sums of it) in one launch
                _C.gemm_dtanh_colsum(gA16, s.wa[P["w2"]], M, mlp, dim, A["hid"], b["dhid16"], g[P["b1"]], trans_b=True, accumulate_colsum=True)
            else:                               # exact-f32 mode
                _C.mm(gA16, s.wa[P["w2"]], M, mlp, dim, b["dhid16"], trans_b=True, act=_C.ACT_DTANH, aux=A["hid"])
                _C.colsum_any(b["dhid16"], M, mlp, g[P["b1"]], accumulate=True)
            _C.mm(b["dhid16"], A["a2"], mlp, dim, M, g[P["w1"]], trans_a=True, trans_b=True, accumulate=True)
            _C.mm(b["dhid16"], s.wa[P["w1"]], M, dim, mlp, dA, trans_b=True)
            _C.ln_bwd(dA, A["x_mid"], s.w[P["ln2_w"]], A["mean2"], A["rstd2"], gA, gB, gB16, g[P["ln2_w"]], g[P["ln2_b"]], g[P["bout"]])
            # ---- attention: x_mid = to_out(attn(to_qkv(a1))) + x_in ----
            _C.mm(gB16, A["o"], dim, inner, M, g[P["wout"]], trans_a=True, trans_b=True, accumulate=True)
            _C.mm(gB16, s.wa[P["wout"]], M, inner, dim, b["do16"], trans_b=True)
            _C.attn_bwd(A["qkv"], A["o"], b["do16"], A["lse"], B, self.n_tok, self.heads, self.scale, b["dqkv16"], b["delta"], q_pre)
            _C.mm(b["dqkv16"], A["a1"], 3 * inner, dim, M, g[P["wqkv"]], trans_a=True, trans_b=True, accumulate=True)
            _C.mm(b["dqkv16"], s.wa[P["wqkv"]], M, dim, 3 * inner, dA, trans_b=True)
            _C.ln_bwd(dA, b["x"][i], s.w[P["ln1_w"]], A["mean1"], A["rstd1"], gB, gA, gA16, g[P["ln1_w"]], g[P["ln1_b"]],
                      g[self.L[i - 1]["b2"]] if i > 0 else self.first_bias_grad)
            if on_layer_done is not None:
                on_layer_done(f"{self.prefix}transformer.layers.{i}.")
        return gA, gA16


class _AEFunction(torch.autograd.Function):
    """Autograd bridge over the static schedule: parameters are NOT inputs (their gradients are accumulated into the flat
    buffer that `param.grad` aliases, the way the fused training step does); the image receives no gradient."""

    @staticmethod
    def forward(ctx, engine, img, anchor):
        st = engine.forward_train(img)
        B, io = st["B"], engine._io_bufs(st["B"])
        _C.unpatchify_loss(st["pix"], None, B, engine.C, engine.size, engine.size, engine.patch, 0.0, 0.0, io["xrec"], None, None)
        ctx.engine, ctx.st = engine, st
        return io["xrec"].clone(), st["qloss"].view(()).clone()

    @staticmethod
    def backward(ctx, g_xrec, g_qloss):
        engine, st = ctx.engine, ctx.st
        B, io = st["B"], engine._io_bufs(st["B"])
        # fp16 engine: the caller backpropagates engine.scale_loss(loss) (torch.cuda.amp's scaler.scale(loss).backward() idiom: the incoming gradients
        # then carry the loss scale, which keeps them inside fp16's range when they are packed into the 16-bit operands below); optimizer_step /
        # unscale_grads divide it out of the parameter gradients again
        if g_xrec is None:
            io["dpix16"].zero_()
        else:
            _C.patchify_any(g_xrec.to(dtype=F32).contiguous(), engine.patch, io["dpix16"])
        g_dev = None if g_qloss is None else g_qloss.reshape(1).to(dtype=F32).contiguous()
        engine.backward_from(st, io["dpix16"], 1.0 if g_qloss is not None else 0.0, g_dev)
        return None, None, None


class _EncodeFn(torch.autograd.Function):
    """image -> h (quantizer input) over the static schedule, for quantizers that are NOT the fused kernel (GumbelQuantizer: plain torch between the two
    halves).  The state dict is shared with _DecodeFn through the engine (one outstanding forward)."""

    @staticmethod
    def forward(ctx, engine, img, anchor):
        st = engine.encode_train(img)
        engine._split_st = st
        ctx.engine, ctx.st = engine, st
        return st["h"].view(st["B"], engine.n_tok, engine.ed).clone()

    @staticmethod
    def backward(ctx, g_h):
        engine, st = ctx.engine, ctx.st
        dh = g_h.reshape(st["B"] * engine.n_tok, engine.ed).to(dtype=F32).contiguous()      # (carries the loss scale of the caller's engine.scale_loss(loss).backward())
        engine._check_serial(st)
        engine._check_scaled_accumulation()
        engine.backward_encoder(st, dh.to(engine.adt), announce_quantizer=False)
        return None, None, None


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, quant, anchor):
        st = engine._split_st
        B, io = st["B"], engine._io_bufs(st["B"])
        q32 = quant.detach().reshape(B * engine.n_tok, engine.ed).to(dtype=F32).contiguous()
        engine.decode_train(st, q32.to(dtype=engine.adt), zq32=q32)
        _C.unpatchify_loss(st["pix"], None, B, engine.C, engine.size, engine.size, engine.patch, 0.0, 0.0, io["xrec"], None, None)
        ctx.engine, ctx.st, ctx.qshape = engine, st, quant.shape
        return io["xrec"].clone()

    @staticmethod
    def backward(ctx, g_xrec):
        engine, st = ctx.engine, ctx.st
        io = engine._io_bufs(st["B"])
        _C.patchify_any(g_xrec.to(dtype=F32).contiguous(), engine.patch, io["dpix16"])      # (scaled by the caller: engine.scale_loss(loss).backward())
        engine._check_serial(st)
        engine._check_scaled_accumulation()
        dzq = engine.backward_decoder(st, io["dpix16"])
        return None, dzq.view(ctx.qshape).clone(), None


class Stage1Engine:
    """Binds a ``ViTVQ`` module tree to the HIP schedule on one device."""

    def __init__(self, model: nn.Module, device: Optional[torch.device] = None, precision: Optional[str] = None,
                 encoder_precision: Optional[str] = None, codes_precision: Optional[str] = None, decoder_precision: Optional[str] = None) -> None:
        """precision: "fp16" | "bf16" (product paths: 16-bit MFMA operands of that format, fp32 accumulation / residual stream / master weights) or
        "fp32" (exact mode for parity runs: every operand fp32, vector-ALU kernels); default from ENH_PRECISION, else "fp16" (since round 6: the
        single-pass mode that meets the 1e-3 parity clause; "bf16" is the round-1..5 default, ~3 % faster, ~5e-3).
        "fp16" is the reference's --use_amp dtype (main.py:25,52: Lightning precision=16): 11-bit significands bring the single-pass forward within 1e-3
        of the fp32 reference (bf16: ~5e-3) at the same MFMA rate; the backward runs on fp16 operands too, with the loss gradient multiplied by a
        power-of-two loss scale that lives ON THE DEVICE (`scale_t`; initial value ENH_LOSS_SCALE, default 2^16 — GradScaler's initial scale): the
        AdamW launch divides it out again, an inf / nan check of the flat gradient makes that launch a no-op (GradScaler.step's skip), and
        enh_loss_scale_update applies GradScaler.update (x 0.5 after an overflow, x 2 after ENH_LOSS_SCALE_INTERVAL = 2000 clean steps) — all without a
        host round trip, so the step stays HIP-graph capturable.  ENH_LOSS_SCALE_INTERVAL=0 keeps the scale from growing.
        Within the bf16 product path the ENCODER forward (patch embedding .. pre_quant, the part that decides the codes) can run on split-bf16
        ("x3") operands — three MFMA passes, ~1e-5 relative, codes equal to the fp32 reference's up to its own near-ties (csrc/x3.hip):
          encoder_precision  "bf16" | "x3": training / reconstruct / forward (ENH_ENCODER_PRECISION, default "bf16": the measured headline path)
          codes_precision    "bf16" | "x3": encode_codes, i.e. the tokens stage 2 consumes (ENH_CODES_PRECISION, default "x3")
          decoder_precision  "bf16" | "x3": post_quant .. to_pixel of training / reconstruct / decode (ENH_DECODER_PRECISION, default "bf16"); with both towers
                             on x3 the whole forward — codes, reconstruction, losses — is within ~1e-5 of the fp32 reference (the backward stays bf16)"""
        import os
        precision = precision or os.environ.get("ENH_PRECISION", "fp16")
        if precision not in OPERAND_DTYPE:
            raise ValueError(f"precision must be 'bf16', 'fp16' or 'fp32', got {precision!r}")
        self.precision = precision
        self.half = precision in ("bf16", "fp16")
        # per-part precision of the forward: "x3" or the engine's own single-pass operand format (spelled as `precision`; "bf16" is accepted as that
        # spelling under fp16 too, for the environment variables of earlier rounds).  Under fp16 the codes default to the single fp16 pass: it meets the
        # 1e-3 clause on its own; x3 stays available as the instrument (encode_codes(precision="x3")).
        self.encoder_precision = encoder_precision or os.environ.get("ENH_ENCODER_PRECISION", precision)
        self.codes_precision = codes_precision or os.environ.get("ENH_CODES_PRECISION", "x3" if precision == "bf16" else precision)
        self.decoder_precision = decoder_precision or os.environ.get("ENH_DECODER_PRECISION", precision)
        for name, v in (("encoder_precision", self.encoder_precision), ("codes_precision", self.codes_precision), ("decoder_precision", self.decoder_precision)):
            if v not in ("bf16", "fp16", "fp32", "x3"):
                raise ValueError(f"{name} must be 'x3' or the engine's precision, got {v!r}")
        if precision != "bf16" and "x3" in (self.encoder_precision, self.decoder_precision):
            raise ValueError("x3 towers in the TRAINING forward save bf16 hi planes for a bf16 backward: they need precision='bf16' (fp16 meets the tolerance in one pass)")
        self.adt = OPERAND_DTYPE[precision]
        # loss scale of the fp16 backward (1 elsewhere): a power of two, so scaling and unscaling are exact
        self.scaled = precision == "fp16"                       # the backward runs under a loss scale
        self._init_scale = float(os.environ.get("ENH_LOSS_SCALE", 65536.0)) if self.scaled else 1.0
        self.scale_growth_interval = int(os.environ.get("ENH_LOSS_SCALE_INTERVAL", 2000))      # torch.cuda.amp.GradScaler's default
        self.check_nonfinite = self.scaled and os.environ.get("ENH_NONFINITE_CHECK", "1") != "0"
        if not torch.cuda.is_available():
            raise RuntimeError("Stage1Engine needs a ROCm device (MI355X); the HIP path has no CPU fallback")
        _C.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.model = model
        enc, dec, q = model.encoder, model.decoder, model.quantizer
        # the autoencoder optimizer's parameter group (reference vitvqgan.py:154-158); a discriminator inside model.loss has its own store
        self.store = ParamStore(model, self.device, precision, prefixes=("encoder.", "decoder.", "pre_quant.", "post_quant.", "quantizer."))
        self.patch, self.size, self.C = enc.patch_size[0], enc.image_size[0], enc.channels
        self.n_tok = enc.num_patches
        self.pd = enc.patch_dim
        self.enc = _Tower(self.store, "encoder.", enc.dim, enc.transformer.depth, enc.transformer.heads, enc.transformer.mlp_dim, self.n_tok)
        self.dec = _Tower(self.store, "decoder.", dec.dim, dec.transformer.depth, dec.transformer.heads, dec.transformer.mlp_dim, self.n_tok)
        self.enc.first_bias_grad = self.store.grad["encoder.to_patch_embedding.0.bias"]
        self.dec.first_bias_grad = self.store.grad["post_quant.bias"]
        self.q = q
        self.ed = q.embed_dim
        self._io: Dict[int, dict] = {}
        enc._engine = dec._engine = self
        dec.get_last_layer()._enh_engine = self      # lets the loss module ask for ||d(.)/d last_layer|| (adaptive adversarial weight)
        self.world = 1
        self.comm = None  # enhancing.engine.ddp.GradSync when running data-parallel
        self.sync_grads = True  # False on all but the last micro-batch of a gradient-accumulation window (DDP's no_sync)
        # HIP-graph replay of the fused step (forward_backward_graphed): the step is a static launch sequence over pre-allocated buffers, ~75 launches
        # per transformer layer; at small per-GPU batches (the reference's large yaml trains at 2 images per GPU) the host cannot issue them as fast as
        # the GPU retires them.  Opt-in (ENH_GRAPHS=1 / engine.use_graphs = True); never used while per-kernel timing or gradient communication hooks
        # are active (both record events / issue collectives from the host inside the step).
        self.use_graphs = os.environ.get("ENH_GRAPHS", "0") == "1"
        self._graphs: Dict[tuple, tuple] = {}
        self.store.operand_hooks.append(self._refresh_x3_operands)
        self.found_inf = torch.zeros(1, dtype=F32, device=self.device) if self.check_nonfinite else None      # written by nonfinite_flag, read by the AdamW launch
        self.skipped_steps = None      # device counter of dropped steps (fp16): accumulated without a host sync
        if self.check_nonfinite:
            self.skipped_steps = torch.zeros(1, dtype=F32, device=self.device)
        # the loss scale and GradScaler's growth tracker, on the device (None for bf16 / fp32 engines: no scaling anywhere)
        self.scale_t = torch.full((1,), self._init_scale, dtype=F32, device=self.device) if self.scaled else None
        self._growth_tracker = torch.zeros(1, dtype=torch.int32, device=self.device) if self.scaled else None

    @property
    def loss_scale(self) -> float:
        """current loss scale as a host number (reads the device scalar: synchronises; for tests and logging — the step itself never needs it)"""
        return float(self.scale_t.item()) if self.scaled else 1.0

    # ---- helpers -----------------------------------------------------------------------------
    def _invalidate_saved(self) -> None:
        """every entry point that overwrites the per-batch-size io buffers (h, patches, pix, xrec) or the no-save arena makes an outstanding
        forward_train un-differentiable: bump the serial so its backward refuses instead of silently using stale h / patches"""
        self._fwd_serial = getattr(self, "_fwd_serial", 0) + 1

    def _io_bufs(self, B: int) -> dict:
        if B not in self._io:
            dev, M = self.device, B * self.n_tok
            e = lambda *shape, dt=F32: torch.empty(*shape, dtype=dt, device=dev)
            self._io[B] = dict(patches=e(M, self.pd, dt=self.adt), h=e(M, self.ed), pix=e(M, self.pd), xrec=e(B, self.C, self.size, self.size),
                               dpix16=e(M, self.pd, dt=self.adt), sums=torch.zeros(2, dtype=torch.float64, device=dev),
                               d_xf_dec=e(M, self.dec.dim), d_xf_enc=e(M, self.enc.dim), dzq=e(M, self.ed), bias_pix=e(self.pd),
                               g_bias_pix=e(self.pd))
        return self._io[B]

    def _check_img(self, img: torch.Tensor) -> torch.Tensor:
        if img.dim() != 4 or img.shape[1] != self.C or img.shape[2] != self.size or img.shape[3] != self.size:
            raise RuntimeError(f"expected images [B,{self.C},{self.size},{self.size}], got {tuple(img.shape)}")
        return img.to(device=self.device, dtype=F32).contiguous()

    # ---- forward pieces ----------------------------------------------------------------------
    def _x3_io(self, B: int) -> dict:
        """x3 operands outside the tower: the split patch / pre_quant weights (rebuilt lazily after the masters changed) and the split patches"""
        s = self.store
        X = self.__dict__.setdefault("_x3", dict(version=None, io={}))
        if X["version"] != s.version:
            if "wpe" not in X:
                X["wpe"] = torch.empty(self.enc.dim, 3 * self.pd, dtype=BF16, device=self.device)
                X["wpre"] = torch.empty(self.ed, 3 * self.enc.dim, dtype=BF16, device=self.device)
            _C.split3(s.w["encoder.to_patch_embedding.0.weight"].view(self.enc.dim, self.pd), X["wpe"], order=1)
            _C.split3(s.w["pre_quant.weight"], X["wpre"], order=1)
            if "wpost" not in X:
                X["wpost"] = torch.empty(self.dec.dim, 3 * self.ed, dtype=BF16, device=self.device)
                X["wpix"] = torch.empty(self.pd, 3 * self.dec.dim, dtype=BF16, device=self.device)
            _C.split3(s.w["post_quant.weight"], X["wpost"], order=1)
            # to_pixel's weight is stored [K = dim][N = C*p*p] (ConvTranspose2d): the K-concatenated operand is built from its transpose
            _C.split3(s.w["decoder.to_pixel.1.weight"].view(self.dec.dim, self.pd).t().contiguous(), X["wpix"], order=1)
            X["version"] = s.version
        if B not in X["io"]:
            M = B * self.n_tok
            X["io"][B] = dict(patches32=torch.empty(M, self.pd, dtype=F32, device=self.device), patches3=torch.empty(M, 3 * self.pd, dtype=BF16, device=self.device),
                              zq3=torch.empty(M, 3 * self.ed, dtype=BF16, device=self.device))
        return dict(wpe=X["wpe"], wpre=X["wpre"], wpost=X["wpost"], wpix=X["wpix"], **X["io"][B])

    def _encode_tokens(self, img: torch.Tensor, save: bool, want_f32: bool = False, x3: bool = False) -> dict:
        """patch-embed GEMM (+bias +pos table) -> encoder tower.  reference layers.py:177-182.  x3: on split-bf16 operands (product path only)."""
        B, s, io = img.shape[0], self.store, self._io_bufs(img.shape[0])
        M = B * self.n_tok
        if x3 and self.half:
            X = self._x3_io(B)
            _C.patchify_any(img, self.patch, X["patches32"])
            _C.split3(X["patches32"], X["patches3"], y_hi=io["patches"] if save else None)
            _C.mm(X["patches3"], X["wpe"], M, self.enc.dim, 3 * self.pd, self.enc.input_buffer(B, save), bias=s.w["encoder.to_patch_embedding.0.bias"],
                  res=s.w["encoder.en_pos_embedding"].view(self.n_tok, self.enc.dim), res_rows=self.n_tok)
            return self.enc.forward_x3(B, save, want_f32)
        _C.patchify_any(img, self.patch, io["patches"])
        w16 = s.wa["encoder.to_patch_embedding.0.weight"].view(self.enc.dim, self.pd)
        _C.mm(io["patches"], w16, M, self.enc.dim, self.pd, self.enc.input_buffer(B, save), bias=s.w["encoder.to_patch_embedding.0.bias"],
              res=s.w["encoder.en_pos_embedding"].view(self.n_tok, self.enc.dim), res_rows=self.n_tok)
        return self.enc.forward(B, save, want_f32)

    def _pre_quant(self, eb: dict, B: int) -> torch.Tensor:
        """h = pre_quant(final LayerNorm output) for the buffer dict a tower forward returned (vitvqgan.py:63)"""
        s, io = self.store, self._io_bufs(B)
        if eb.get("x3"):
            _C.mm(eb["xf3"], self._x3_io(B)["wpre"], B * self.n_tok, self.ed, 3 * self.enc.dim, io["h"], bias=s.w["pre_quant.bias"])
        else:
            _C.mm(eb["xf16"], s.wa["pre_quant.weight"], B * self.n_tok, self.ed, self.enc.dim, io["h"], bias=s.w["pre_quant.bias"])
        return io["h"]

    def _decode_tokens(self, zq16: torch.Tensor, B: int, save: bool, zq32: Optional[torch.Tensor] = None) -> torch.Tensor:
        """post_quant (+bias +pos table) -> decoder tower -> to_pixel GEMM.  reference vitvqgan.py:68-72, layers.py:209-214.  zq32 (the quantized tokens
        in f32) selects the x3 decoder: every product from split-bf16 operands (product path only)."""
        s, io, M = self.store, self._io_bufs(B), B * self.n_tok
        pp = self.patch * self.patch
        io["bias_pix"].view(self.C, pp).copy_(s.w["decoder.to_pixel.1.bias"].view(self.C, 1).expand(self.C, pp))
        if zq32 is not None and self.half:
            X = self._x3_io(B)
            _C.split3(zq32, X["zq3"])
            _C.mm(X["zq3"], X["wpost"], M, self.dec.dim, 3 * self.ed, self.dec.input_buffer(B, save), bias=s.w["post_quant.bias"],
                  res=s.w["decoder.de_pos_embedding"].view(self.n_tok, self.dec.dim), res_rows=self.n_tok)
            b = self.dec.forward_x3(B, save)
            _C.mm(b["xf3"], X["wpix"], M, self.pd, 3 * self.dec.dim, io["pix"], bias=io["bias_pix"])
            return io["pix"]
        _C.mm(zq16, s.wa["post_quant.weight"], M, self.dec.dim, self.ed, self.dec.input_buffer(B, save), bias=s.w["post_quant.bias"],
              res=s.w["decoder.de_pos_embedding"].view(self.n_tok, self.dec.dim), res_rows=self.n_tok)
        b = self.dec.forward(B, save)
        wpix16 = s.wa["decoder.to_pixel.1.weight"].view(self.dec.dim, self.pd)  # stored [K][N]
        _C.mm(b["xf16"], wpix16, M, self.pd, self.dec.dim, io["pix"], trans_b=True, bias=io["bias_pix"])
        return io["pix"]

    # ---- inference API (reference vitvqgan.py:44-90) -------------------------------------------
    @torch.no_grad()
    def encode_codes(self, img: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """precision: "x3" | "bf16" for this call (default: self.codes_precision = "x3": the codes stage 2 consumes follow the fp32 reference)"""
        self._invalidate_saved()
        img = self._check_img(img)
        B = img.shape[0]
        b = self._encode_tokens(img, save=False, x3=(precision or self.codes_precision) == "x3")
        h = self._pre_quant(b, B)
        _, _, idx, _ = _C.vq_forward(h, self.store.w["quantizer.embedding.weight"], float(self.q.beta), self.q.depth, self.q.use_norm, False)
        return idx.view(B, self.n_tok, self.q.depth) if self.q.use_residual else idx.view(B, self.n_tok)

    @torch.no_grad()
    def reconstruct(self, img: torch.Tensor):
        """forward without saving activations -> (xrec [B,C,H,W] f32, qloss scalar, indices)."""
        self._invalidate_saved()
        img = self._check_img(img)
        B, io = img.shape[0], self._io_bufs(img.shape[0])
        b = self._encode_tokens(img, save=False, x3=self.encoder_precision == "x3")
        h = self._pre_quant(b, B)
        exact = not self.half
        zq, zq16, idx, qloss = _C.vq_forward(h, self.store.w["quantizer.embedding.weight"], float(self.q.beta), self.q.depth, self.q.use_norm,
                                             want_bf16=not exact, h16=self.adt if self.half else BF16)
        pix = self._decode_tokens(zq if exact else zq16, B, save=False, zq32=zq if self.decoder_precision == "x3" else None)
        _C.unpatchify_loss(pix, None, B, self.C, self.size, self.size, self.patch, 0.0, 0.0, io["xrec"], None, None)
        idx = idx.view(B, self.n_tok, self.q.depth) if self.q.use_residual else idx.view(B, self.n_tok)
        return io["xrec"].clone(), qloss.view(()).clone(), idx

    @torch.no_grad()
    def decode_from_quant(self, quant: torch.Tensor) -> torch.Tensor:
        """decode(quant) for quant [B, N, embed_dim] f32 (reference vitvqgan.py:68-72)."""
        self._invalidate_saved()
        B = quant.shape[0]
        io = self._io_bufs(B)
        zq32 = quant.reshape(B * self.n_tok, self.ed).to(device=self.device, dtype=F32).contiguous()
        pix = self._decode_tokens(zq32.to(self.adt), B, save=False, zq32=zq32 if self.decoder_precision == "x3" else None)
        _C.unpatchify_loss(pix, None, B, self.C, self.size, self.size, self.patch, 0.0, 0.0, io["xrec"], None, None)
        return io["xrec"].clone()

    @torch.no_grad()
    def encoder_forward(self, img: torch.Tensor) -> torch.Tensor:
        self._invalidate_saved()
        img = self._check_img(img)
        b = self._encode_tokens(img, save=False, want_f32=True, x3=self.encoder_precision == "x3")
        return b["xf32"].view(img.shape[0], self.n_tok, self.enc.dim).clone()

    @torch.no_grad()
    def decoder_forward(self, tok: torch.Tensor) -> torch.Tensor:
        """ViTDecoder.forward(token) for token [B, N, dim] f32 (post_quant output) — reference layers.py:209-214."""
        self._invalidate_saved()
        B, s, io, M = tok.shape[0], self.store, self._io_bufs(tok.shape[0]), tok.shape[0] * self.n_tok
        x0 = self.dec.input_buffer(B, False)
        torch.add(tok.reshape(M, self.dec.dim).to(device=self.device, dtype=F32), s.w["decoder.de_pos_embedding"].view(self.n_tok, self.dec.dim).repeat(B, 1), out=x0)
        b = self.dec.forward(B, False)
        pp = self.patch * self.patch
        io["bias_pix"].view(self.C, pp).copy_(s.w["decoder.to_pixel.1.bias"].view(self.C, 1).expand(self.C, pp))
        _C.mm(b["xf16"], s.wa["decoder.to_pixel.1.weight"].view(self.dec.dim, self.pd), M, self.pd, self.dec.dim, io["pix"], trans_b=True,
              bias=io["bias_pix"])
        _C.unpatchify_loss(io["pix"], None, B, self.C, self.size, self.size, self.patch, 0.0, 0.0, io["xrec"], None, None)
        return io["xrec"].clone()

    # ---- training: forward (activations saved) / backward, used fused by training_step and split by autograd -----
    def encode_train(self, img: torch.Tensor) -> dict:
        """first half of forward_train: encoder + pre_quant with the activations saved -> state dict with h (quantizer input, f32 [M, embed_dim])"""
        img = self._check_img(img)
        B = img.shape[0]
        eb = self._encode_tokens(img, save=True, x3=self.encoder_precision == "x3")
        h = self._pre_quant(eb, B)
        self._fwd_serial = getattr(self, "_fwd_serial", 0) + 1
        return dict(img=img, B=B, eb=eb, h=h, serial=self._fwd_serial)

    def decode_train(self, st: dict, zq16: torch.Tensor, zq32: Optional[torch.Tensor] = None) -> dict:
        """second half: post_quant + decoder + to_pixel from the quantized tokens [M, embed_dim] (operand dtype of this precision mode), saved; zq32: the same
        tokens in f32 for the x3 decoder (decoder_precision)"""
        st.update(zq16=zq16, pix=self._decode_tokens(zq16, st["B"], save=True, zq32=zq32 if self.decoder_precision == "x3" else None))
        return st

    def forward_train(self, img: torch.Tensor) -> dict:
        """Forward with every activation the backward needs kept in the arena of this batch size.  ONE forward may be
        outstanding per batch size: a later forward_train overwrites the arena (checked through `_fwd_serial`)."""
        st = self.encode_train(img)
        E = self.store.w["quantizer.embedding.weight"]
        exact = not self.half
        zq, zq16, idx, qloss = _C.vq_forward(st["h"], E, float(self.q.beta), self.q.depth, self.q.use_norm, want_bf16=not exact, h16=self.adt if self.half else BF16)
        st.update(idx=idx, qloss=qloss)
        return self.decode_train(st, zq if exact else zq16, zq32=zq)

    def _check_serial(self, st: dict) -> None:
        if st["serial"] != self._fwd_serial:
            raise RuntimeError("backward called for a forward whose saved activations were overwritten by a later forward_train")

    def _check_scaled_accumulation(self) -> None:
        if self.scaled and self.store.grads_unscaled:
            raise RuntimeError("the flat gradient was unscaled (unscale_grads) and not zeroed since: a loss-scaled backward cannot accumulate onto it — call "
                               "zero_grad() first")

    def scale_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """loss * loss_scale (fp16 engine; the identity otherwise): what to call .backward() on when the loss is built on differentiable_forward /
        differentiable_encode + differentiable_decode — torch.cuda.amp's `scaler.scale(loss).backward()` under the reference's --use_amp (main.py:25,52).
        The upstream gradients then reach the 16-bit backward inside fp16's range (a mean-reduced pixel loss has |d loss / d xrec| ~ 1e-8, below fp16's
        smallest subnormal); |gradient| * loss_scale must stay below 65504 — an overflow sets found_inf and the optimizer step is dropped."""
        return loss * self.scale_t.to(loss.device).view(()) if self.scaled else loss

    @torch.no_grad()
    def unscale_grads(self) -> None:
        """fp16 engine: param.grad (views of the flat gradient) holds loss_scale x gradient after a backward — torch.cuda.amp's convention under the
        reference's --use_amp (GradScaler: gradients stay scaled until unscale_ / step).  optimizer_step divides the scale out inside the AdamW launch
        for free; call this (GradScaler.unscale_'s counterpart: one pass over the flat buffer) to READ true gradients before the step — gradient-norm
        logging, a custom optimizer, tests.  Idempotent until the next zero_grad(); a no-op for bf16 / fp32 engines."""
        if self.scaled and not self.store.grads_unscaled:
            if self.comm is not None and not self.store.comm_done:      # the reduced gradient is what gets unscaled; optimizer_step will not reduce again
                self.comm.finish()
                self.store.comm_done = True
            self.store.g.div_(self.scale_t)
            self.store.grads_unscaled = True

    def backward_decoder(self, st: dict, dpix16: torch.Tensor) -> torch.Tensor:
        """to_pixel + decoder tower + post_quant backward given dpix16 = d loss / d pix in the patch layout [M, C*p*p]; ACCUMULATES the parameter
        gradients, returns d loss / d (quantized tokens) as f32 [M, embed_dim] (an engine buffer)."""
        self._check_serial(st)
        B, s, io = st["B"], self.store, self._io_bufs(st["B"])
        M, g = B * self.n_tok, s.grad
        zq16 = st["zq16"]
        db = self.dec.bufs(B, True)
        notify = self.comm.layer_done if self.comm is not None and self.sync_grads else None
        d_xf = io["d_xf_dec"]
        wpix = "decoder.to_pixel.1.weight"
        _C.mm(db["xf16"], dpix16, self.dec.dim, self.pd, M, g[wpix].view(self.dec.dim, self.pd), trans_a=True, trans_b=True, accumulate=True)
        _C.colsum_any(dpix16, M, self.pd, io["g_bias_pix"], accumulate=False)
        g["decoder.to_pixel.1.bias"].add_(io["g_bias_pix"].view(self.C, -1).sum(1))
        _C.mm(dpix16, s.wa[wpix].view(self.dec.dim, self.pd), M, self.dec.dim, self.pd, d_xf)
        if notify:
            notify("decoder.to_pixel.")
        g0, g016 = self.dec.backward(B, d_xf, notify)
        _C.mm(g016, zq16, self.dec.dim, self.ed, M, g["post_quant.weight"], trans_a=True, trans_b=True, accumulate=True)
        _C.mm(g016, s.wa["post_quant.weight"], M, self.ed, self.dec.dim, io["dzq"], trans_b=True)
        if notify:
            notify("post_quant.")
        return io["dzq"]

    def backward_encoder(self, st: dict, dh16: torch.Tensor, announce_quantizer: bool = True) -> None:
        """pre_quant + encoder tower + patch embedding backward given d loss / d h [M, embed_dim] in the operand dtype; ACCUMULATES the gradients."""
        self._check_serial(st)
        B, s, io = st["B"], self.store, self._io_bufs(st["B"])
        M, g, eb = B * self.n_tok, s.grad, st["eb"]
        notify = self.comm.layer_done if self.comm is not None and self.sync_grads else None
        _C.mm(dh16, eb["xf16"], self.ed, self.enc.dim, M, g["pre_quant.weight"], trans_a=True, trans_b=True, accumulate=True)
        _C.colsum_any(dh16, M, self.ed, g["pre_quant.bias"], accumulate=True)
        d_xe = io["d_xf_enc"]
        _C.mm(dh16, s.wa["pre_quant.weight"], M, self.enc.dim, self.ed, d_xe, trans_b=True)
        if notify:
            notify("pre_quant.")
            if announce_quantizer:      # (a quantizer differentiated by torch autograd may still be accumulating: its slice is left to GradSync.finish())
                notify("quantizer.")
        e0, e016 = self.enc.backward(B, d_xe, notify)
        wpe = "encoder.to_patch_embedding.0.weight"
        _C.mm(e016, io["patches"], self.enc.dim, self.pd, M, g[wpe].view(self.enc.dim, self.pd), trans_a=True, trans_b=True, accumulate=True)
        if notify:
            notify("encoder.to_patch_embedding.")

    def backward_from(self, st: dict, dpix16: torch.Tensor, g_loss: float, g_loss_dev: Optional[torch.Tensor] = None) -> None:
        """Backward of forward_train given dpix16 = d loss / d pix in the patch layout [M, C*p*p] (16-bit operand) and the gradient
        flowing into the codebook loss (host scalar g_loss times optional device scalar).  ACCUMULATES into the flat grads.  With a loss scale
        (fp16) both incoming gradients carry it, and so does everything accumulated: see unscale_grads()."""
        self._check_serial(st)
        self._check_scaled_accumulation()
        dzq = self.backward_decoder(st, dpix16)
        exact = not self.half
        dh, dh16 = _C.vq_backward(st["h"], self.store.w["quantizer.embedding.weight"], st["idx"], dzq, g_loss, g_loss_dev, float(self.q.beta), self.q.depth,
                                  bool(self.q.use_residual), self.q.use_norm, self.store.grad["quantizer.embedding.weight"], want_bf16=not exact,
                                  h16=self.adt if self.half else BF16)
        self.backward_encoder(st, dh if exact else dh16)

    def forward_backward(self, img: torch.Tensor, w_l1: float = 0.0, w_l2: float = 1.0, codebook_weight: float = 1.0,
                         zero_grad: bool = True) -> dict:
        """Fused ViTVQ.training_step(optimizer_idx=0) + backward (vitvqgan.py:101-115) with
        loss = w_l1*L1 + w_l2*L2 + codebook_weight*qloss (vqperceptual.py:113-117,131-132, perceptual / adversarial weights 0).
        Gradients are ACCUMULATED into the flat grad buffer (zeroed first unless zero_grad=False: gradient accumulation,
        main.py:22,57)."""
        if zero_grad:
            self.store.zero_grad()
        st = self.forward_train(img)
        img, B, io = st["img"], st["B"], self._io_bufs(st["B"])
        io["sums"].zero_()
        # (fp16: the loss gradient enters the backward multiplied by the loss scale S, a DEVICE scalar — applied inside the pixel-loss gradient kernel and to the
        # codebook-loss gradient; the loss VALUES below are formed from the unscaled sums; optimizer_step divides the gradients by S again)
        _C.unpatchify_loss_any(st["pix"], img, B, self.C, self.size, self.size, self.patch, w_l1, w_l2, io["xrec"], io["sums"], io["dpix16"], grad_scale=self.scale_t)
        self.backward_from(st, io["dpix16"], codebook_weight, self.scale_t)
        numel = float(img.numel())
        l1 = (io["sums"][0] / numel).float()
        l2 = (io["sums"][1] / numel).float()
        ql = st["qloss"].view(())
        nll = w_l1 * l1 + w_l2 * l2
        return dict(loss=nll + codebook_weight * ql, quant_loss=ql, rec_loss=nll, loglaplace_loss=l1, loggaussian_loss=l2,
                    xrec=io["xrec"], indices=st["idx"], h=st["h"])

    def forward_backward_graphed(self, img: torch.Tensor, w_l1: float = 0.0, w_l2: float = 1.0, codebook_weight: float = 1.0,
                                 zero_grad: bool = True) -> dict:
        """forward_backward captured once per (batch shape, loss weights, zero_grad) into a HIP graph and replayed: one host call per step instead
        of ~1800 kernel launches (base).  The returned tensors are the graph's static outputs — valid until the next replay, like every engine
        buffer.  Falls back to the eager sequence whenever capture is not legitimate (timing hooks, data-parallel communication, fp32 host input)."""
        if not self.use_graphs or _C.TIMER is not None or self.comm is not None:
            return self.forward_backward(img, w_l1, w_l2, codebook_weight, zero_grad)
        key = (tuple(img.shape), float(w_l1), float(w_l2), float(codebook_weight), bool(zero_grad))
        entry = self._graphs.get(key)
        if entry is None and len(self._graphs) >= 8:      # ragged batch sizes: do not hoard one graph (and its private memory pool) per shape
            return self.forward_backward(img, w_l1, w_l2, codebook_weight, zero_grad)
        if entry is None:
            static_img = torch.empty(img.shape, dtype=F32, device=self.device)
            static_img.copy_(img)
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            saved_g = self.store.g.clone() if not zero_grad else None   # the warm-up passes must not be counted into an accumulation window
            with torch.cuda.stream(side):       # warm-up off the capture: library attribute setup, workspace growth, lazy buffers
                for _ in range(2):
                    self.forward_backward(static_img, w_l1, w_l2, codebook_weight, zero_grad)
            cur.wait_stream(side)
            if saved_g is not None:
                self.store.g.copy_(saved_g)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward_backward(static_img, w_l1, w_l2, codebook_weight, zero_grad)
            if saved_g is not None:
                self.store.g.copy_(saved_g)     # capture does not execute, but keep the invariant explicit
            keep = (list(_C._GEMM_WS.values()), list(_C._WS.values()))    # the workspaces whose addresses the graph has baked in must outlive it
            entry = self._graphs[key] = (graph, static_img, out, keep)
        graph, static_img, out, _ = entry
        static_img.copy_(img, non_blocking=True)
        graph.replay()
        self._invalidate_saved()
        return out

    @torch.no_grad()
    def last_layer_grad_norm(self, g_xrec: torch.Tensor) -> torch.Tensor:
        """|| d L / d decoder.to_pixel.weight ||_F for a loss whose gradient at the reconstruction is g_xrec [B,C,H,W], for the OUTSTANDING forward_train
        (the saved final-LayerNorm output is the other GEMM operand).  What VQLPIPSWithDiscriminator.calculate_adaptive_factor obtains with
        torch.autograd.grad(loss, last_layer) in the reference (vqperceptual.py:95-103); nothing is accumulated into the gradient buffers."""
        B = g_xrec.shape[0]
        io, db, M = self._io_bufs(B), self.dec.bufs(B, True), B * self.n_tok
        g32 = g_xrec.to(dtype=F32).contiguous()
        _C.patchify_any(g32 * self.scale_t if self.scaled else g32, self.patch, io["dpix16"])      # (scaled into fp16's range; divided out of the norm below)
        tmp = torch.zeros(self.dec.dim, self.pd, dtype=F32, device=self.device)
        _C.mm(db["xf16"], io["dpix16"], self.dec.dim, self.pd, M, tmp, trans_a=True, trans_b=True, accumulate=True)
        return tmp.norm() / self.scale_t.view(()) if self.scaled else tmp.norm()

    def differentiable_forward(self, img: torch.Tensor):
        """(xrec, qloss) connected to autograd: `.backward()` on any function of them runs backward_from and leaves the
        parameter gradients in `param.grad` (views of the flat buffer).  The reference's ViTVQ.forward contract
        (vitvqgan.py:44-48) for callers that bring their own loss module."""
        return _AEFunction.apply(self, img, self._anchor())

    def differentiable_encode(self, img: torch.Tensor) -> torch.Tensor:
        """h = pre_quant(encoder(img)) [B, N, embed_dim] connected to autograd; pair with differentiable_decode (a quantizer in plain torch in between)"""
        return _EncodeFn.apply(self, img, self._anchor())

    def differentiable_decode(self, quant: torch.Tensor) -> torch.Tensor:
        """xrec = decoder(post_quant(quant)) for the forward started by differentiable_encode, connected to autograd through `quant`"""
        return _DecodeFn.apply(self, quant, self._anchor())

    def _anchor(self) -> torch.Tensor:
        if getattr(self, "_anchor_t", None) is None:
            self._anchor_t = torch.zeros(1, device=self.device, requires_grad=True)  # makes autograd call our backward
        return self._anchor_t

    def optimizer_step(self, lr: float, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 1e-4, grad_scale: float = 1.0) -> None:
        """torch.optim.AdamW over the single parameter group of vitvqgan.py:153-160 (one fused launch)."""
        s = self.store
        if self.comm is not None:
            if not s.comm_done:
                self.comm.finish()
            s.comm_done = False
            grad_scale = grad_scale / self.comm.world
        s.step_count += 1
        skip = None
        if self.check_nonfinite:
            # GradScaler.step's found-inf skip + GradScaler.update (reference main.py:25,52 --use_amp), without a host round trip: one pass over the flat
            # gradient sets the flag, the AdamW launch reads it and writes nothing when it is set, and the scale is halved / doubled on the device.
            # skipped_steps counts the drops on the device (read it with .item() outside the step if wanted).  The host step count advances either way:
            # a dropped step then only shifts the bias correction by one step.
            skip = self.found_inf
            skip.zero_()
            _C.nonfinite_flag(s.g, skip)
            self.skipped_steps.add_(skip)
        _C.adamw_step(s.p, s.g, s.m, s.v, s.p16 if self.half else None, s.step_count, lr, betas[0], betas[1], eps, weight_decay,
                      grad_scale, skip_flag=skip, loss_scale=self.scale_t if (self.scaled and not s.grads_unscaled) else None)
        if self.check_nonfinite:
            _C.loss_scale_update(self.scale_t, skip, self._growth_tracker, 2.0, 0.5, self.scale_growth_interval)
        s.refresh_operands()      # operands derived from the masters (the towers' pre-scaled q | k | v weights, the x3 images: _refresh_x3_operands)

    def _refresh_x3_operands(self) -> None:
        """operand hook of the store (runs after EVERY write to the masters: optimizer_step, and refresh_shadows after a checkpoint load / broadcast — ADVICE
        r4): with an x3 TRAINING precision the split weight images are rebuilt eagerly.  They are otherwise rebuilt lazily by the next x3 forward — which a
        HIP-graph replay never runs on the host, so a replay after refresh_shadows() would read the images of the old weights."""
        if "x3" in (self.encoder_precision, self.decoder_precision) and self.precision == "bf16":      # (x3 training towers exist under bf16 only)
            if self.encoder_precision == "x3":
                self.enc.x3_weights()
            if self.decoder_precision == "x3":
                self.dec.x3_weights()
            for B in list(self.__dict__.get("_x3", {}).get("io", {})):
                self._x3_io(B)

    def train_step(self, img: torch.Tensor, lr: float, **loss_kw) -> dict:
        out = self.forward_backward(img, **loss_kw)
        self.optimizer_step(lr)
        return out
