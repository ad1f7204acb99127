"""The parity-grade encoder forward on split-bf16 ("x3") operands (csrc/x3.hip, engine/stage1.py::_Tower.forward_x3) — VERDICT r3 "next" 1.

north_star: "indices bit-exact, activations within 1e-3 rel of the reference PyTorch CPU path".  The single-pass bf16 encoder sits at h ~5.6e-3 and
flips ~2 % (up to 10 % on a spread codebook) of the end-to-end codes (tests/test_parity_base_gpu.py).  The x3 path carries every MFMA operand as
hi + lo bf16 planes and forms a_hi b_hi + a_lo b_hi + a_hi b_lo in the fp32 accumulator: asserted here at the BENCHMARKED widths
(base 768/12/12/3072, K = 8192; RQ depth 4; trained-like spread codebooks) against the fp32 CPU oracle:
    h rel <= 1e-4 (north_star: 1e-3; measured ~1e-5)      end-to-end code match >= 0.999
and, op by op, against fp64.
"""
import copy
import os

import numpy as np
import pytest
import torch

from util import bf16_floor, rel

pytestmark = pytest.mark.gpu

BASE = dict(image_size=256, patch_size=8, encoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072),
            decoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072), quantizer=dict(embed_dim=32, n_embed=8192))
H_TOL = 1e-4          # north_star asks 1e-3; the CPU emulation of the scheme gives 9.8e-6 at base depth
MATCH_MIN = 0.999     # VERDICT r3 "done" criterion


@pytest.fixture(scope="module")
def C():
    from enhancing import _C
    _C.lib()
    return _C


def _split(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi, lo


@pytest.mark.parametrize("order,act,with_bias,with_hi", [(0, 0, False, False), (0, 1, True, True), (1, 0, False, False), (1, 1, True, False)])
def test_split3_is_bitwise_the_definition(C, order, act, with_bias, with_hi):
    torch.manual_seed(0)
    M, K = 300, 200
    x = (torch.randn(M, K, device="cuda") * 3).contiguous()
    bias = torch.randn(K, device="cuda") if with_bias else None
    y3 = torch.empty(M, 3 * K, dtype=torch.bfloat16, device="cuda")
    yh = torch.empty(M, K, dtype=torch.bfloat16, device="cuda") if with_hi else None
    C.split3(x, y3, bias=bias, act=act, order=order, y_hi=yh)
    v = x + bias if with_bias else x
    if act:
        v = torch.tanh(v.double()).float()          # the kernel's tanhf is within an ulp of the correctly rounded value: compare hi + lo, not bits
        got = y3[:, :K].float() + (y3[:, K:2 * K] if order == 0 else y3[:, 2 * K:]).float()
        assert (got - v).abs().max().item() <= 2 ** -16
    else:
        hi, lo = _split(v)
        assert torch.equal(y3[:, :K], hi)
        assert torch.equal(y3[:, K:2 * K], lo if order == 0 else hi)
        assert torch.equal(y3[:, 2 * K:], hi if order == 0 else lo)
    assert torch.equal(y3[:, :K], y3[:, 2 * K:] if order == 0 else y3[:, K:2 * K])
    if with_hi:
        assert torch.equal(yh, y3[:, :K])


def test_split2_and_layernorm_x3(C):
    torch.manual_seed(1)
    x = torch.randn(1000, 768, device="cuda") * 2 + 0.3
    hi = torch.empty(1000, 768, dtype=torch.bfloat16, device="cuda"); lo = torch.empty_like(hi)
    C.split2(x, hi, lo)
    eh, el = _split(x)
    assert torch.equal(hi, eh) and torch.equal(lo, el)
    w, b = torch.randn(768, device="cuda"), torch.randn(768, device="cuda")
    e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="cuda")
    y16, y32, mean, rstd = e(1000, 768, dt=torch.bfloat16), e(1000, 768), e(1000), e(1000)
    C.layernorm_forward(x, w, b, 1e-5, y16, y32, mean, rstd)
    y3, z16, z32, mean3, rstd3 = e(1000, 3 * 768, dt=torch.bfloat16), e(1000, 768, dt=torch.bfloat16), e(1000, 768), e(1000), e(1000)
    C.ln_fwd_x3(x, w, b, y3, mean3, rstd3, y_bf16=z16, y_f32=z32)
    assert torch.equal(mean, mean3) and torch.equal(rstd, rstd3) and torch.equal(y32, z32) and torch.equal(y16, z16)
    fh, fl = _split(y32)
    assert torch.equal(y3[:, :768], fh) and torch.equal(y3[:, 768:1536], fl) and torch.equal(y3[:, 1536:], fh)


@pytest.mark.parametrize("M,N,K", [(2048, 2304, 768), (2048, 768, 3072), (2048, 768, 192), (2048, 32, 768), (384, 256, 128)])
def test_three_pass_product_on_concatenated_operands_vs_fp64(C, M, N, K):
    """ONE enh_gemm_h16 call with K' = 3K on [a_hi | a_lo | a_hi] x [b_hi | b_hi | b_lo] == the fp64 product to ~1e-5 (a single bf16 pass: ~3e-3)"""
    torch.manual_seed(2)
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda") * K ** -0.5
    a3 = torch.empty(M, 3 * K, dtype=torch.bfloat16, device="cuda"); b3 = torch.empty(N, 3 * K, dtype=torch.bfloat16, device="cuda")
    C.split3(a, a3, order=0); C.split3(b, b3, order=1)
    out = torch.empty(M, N, device="cuda")
    C.mm(a3, b3, M, N, 3 * K, out)
    ref = a.double() @ b.double().t()
    one = torch.empty(M, N, device="cuda")
    C.mm(a.to(torch.bfloat16), b.to(torch.bfloat16), M, N, K, one)
    e3, e1 = rel(out, ref), rel(one, ref)
    print(f"x3 product [{M}x{N}x{K}]: rel {e3:.2e} (single bf16 pass {e1:.2e})")
    assert e3 <= 2e-5 and e1 > 50 * e3


@pytest.mark.parametrize("B,N,H", [(2, 1024, 12), (1, 256, 8), (3, 64, 2), (1, 192, 1)])
def test_attention_x3_vs_fp64(C, B, N, H):
    torch.manual_seed(3)
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda")
    qkv[..., :H * 64] *= 2.0       # scores with some spread
    hi = torch.empty(B * N, 3 * H * 64, dtype=torch.bfloat16, device="cuda"); lo = torch.empty_like(hi)
    C.split2(qkv.view(B * N, -1), hi, lo)
    out3 = torch.empty(B * N, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
    out16 = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    C.attention_forward_x3(hi, lo, B, N, H, 0.125, out3, out16, lse)
    q, k, v = (t.view(B, N, H, 64).permute(0, 2, 1, 3).double() for t in qkv.chunk(3, dim=-1))
    s = q @ k.transpose(-1, -2) * 0.125
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, H * 64)
    D = H * 64
    got = out3[:, :D].float() + out3[:, D:2 * D].float()
    e = rel(got, ref)
    # the same kernel skeleton on single bf16 operands, for scale
    o1 = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda"); l1 = torch.empty_like(lse)
    C.attention_forward(qkv.view(B * N, -1).to(torch.bfloat16), B, N, H, 0.125, o1, l1)
    e1 = rel(o1, ref)
    print(f"attention x3 B={B} N={N} H={H}: out rel {e:.2e} (bf16 kernel {e1:.2e}), lse abs {(lse.double() - torch.logsumexp(s, -1)).abs().max().item():.1e}")
    assert e <= 3e-5
    assert (lse.double() - torch.logsumexp(s, -1)).abs().max().item() <= 1e-4       # (v_log_f32 / v_exp_f32 on values ~10: measured 3e-5)
    assert torch.equal(out3[:, :D], out3[:, 2 * D:]) and torch.equal(out16, out3[:, :D])
    assert rel(out16, ref) <= 1.15 * bf16_floor(ref)          # the hi plane alone is the bf16 rounding of the result


def _build(cfg, P, **kw):
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPS",
            "params": dict(codebook_weight=1.0, loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0)}
    m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
              AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
    m.precision = "bf16"      # x3 = split-bf16 operands: the instrument of the bf16 product path (the fp16 default of round 6 meets the tolerance in one pass)
    for k, v in kw.items():
        setattr(m, k, v)
    m.load_state_dict(P, strict=True)
    return m


def _case(label, cfg, B, seed, spread):
    import vitvq_oracle as O
    from test_parity_base_gpu import _spread_codebook
    torch.set_num_threads(min(32, max(torch.get_num_threads(), 8)))
    P = O.make_params(cfg, seed)
    x = O.make_images(seed + 1, B, cfg["image_size"], smooth=not spread)
    if spread:
        _spread_codebook(P, x, cfg, seed)
    m = _build(cfg, P)
    with torch.no_grad():
        _, _, o_idx, o_h = O.encode(x, P, cfg)
    codes = m.encode_codes(x).cpu()                       # default precision: x3
    codes_bf16 = m.encode_codes(x, precision="bf16").cpu()
    h = m.pre_quant_tokens(x, precision="x3").cpu()
    h_bf16 = m.pre_quant_tokens(x, precision="bf16").cpu()
    e_h, e_h16 = rel(h, o_h), rel(h_bf16, o_h)
    match, match16 = (codes == o_idx).float().mean().item(), (codes_bf16 == o_idx).float().mean().item()
    line = (f"== {label}: B={B}  x3 encoder: h rel {e_h:.2e}, end-to-end code match {match:.5f}   |   single-pass bf16 encoder: h rel {e_h16:.2e}, "
            f"match {match16:.4f}   (distinct codes in play {o_idx.unique().numel()}, mismatches {int((codes != o_idx).sum())} of {codes.numel()})")
    print("\n" + line)
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_x3.txt"), "a") as f:
            f.write(line + "\n")
    assert codes.shape == o_idx.shape and codes.dtype == torch.int64
    assert e_h <= H_TOL, e_h
    assert match >= MATCH_MIN, match
    return m, x, P


def test_x3_encoder_base_config2_vs_fp32_oracle():
    _case("imagenet_vitvq_base (config 2)", BASE, 2, 0, False)


def test_x3_encoder_base_rq4_config4_vs_fp32_oracle():
    cfg = copy.deepcopy(BASE)
    cfg["quantizer"].update(use_residual=True, num_quantizers=4)
    _case("imagenet_rqvae_base (config 4)", cfg, 2, 3, False)


def test_x3_encoder_base_with_a_trained_like_code_spread():
    _case("imagenet_vitvq_base (config 2), spread codebook", BASE, 2, 10, True)


def test_x3_encoder_base_rq4_with_a_trained_like_code_spread():
    cfg = copy.deepcopy(BASE)
    cfg["quantizer"].update(use_residual=True, num_quantizers=4)
    _case("imagenet_rqvae_base (config 4), spread codebook", cfg, 2, 13, True)


def test_x3_encoder_against_the_reference_golden_vectors(golden_dir):
    """the tiny model of tests/golden/vit_tiny.npz (outputs of the REFERENCE's own modules): x3 h within 1e-5-ish, every code equal"""
    import vitvq_oracle as O
    cfg = O.TINY_CFG
    P = O.make_params(cfg, seed=11)
    x = O.make_images(5, 2, cfg["image_size"])
    m = _build(cfg, P)
    g = np.load(f"{golden_dir}/vit_tiny.npz")
    h = m.pre_quant_tokens(x, precision="x3")
    codes = m.encode_codes(x)
    e_h = rel(h, torch.from_numpy(g["h"]))
    match = (codes.cpu().numpy() == g["idx"].astype(np.int64)).mean()
    print(f"tiny x3 encoder vs REFERENCE golden: h rel {e_h:.2e}, code match {match:.4f}")
    assert e_h <= 5e-5 and match == 1.0


def test_training_step_with_the_x3_encoder_forward():
    """encoder_precision = "x3": the training forward's encoder runs on split operands, the backward is the bf16 product path's on the saved hi planes
    (unscaled-q convention).  Loss / codes follow the fp32 oracle more closely than the bf16 forward; gradients stay within the bf16 path's bounds."""
    import vitvq_oracle as O
    cfg = O.TINY_CFG
    P = O.make_params(cfg, seed=11)
    x = O.make_images(5, 2, cfg["image_size"])
    m = _build(cfg, P, encoder_precision="x3")
    assert m.engine.encoder_precision == "x3"
    loss = m.training_step({"image": x}, 0, 0)
    o_loss, _, o_grads, _ = O.train_step_grads(x, P, cfg)
    m.engine.unscale_grads()      # fp16 engine: param.grad carries the loss scale until the step (or this call)
    errs = {k: rel(p.grad, o_grads[k]) for k, p in m.named_parameters() if k in o_grads}
    worst = max(errs, key=errs.get)
    print(f"x3-encoder train step: loss {loss.item():.6f} vs oracle {o_loss.item():.6f}; grads median rel {np.median(list(errs.values())):.2e}, worst {worst} {errs[worst]:.2e}")
    assert abs(loss.item() - o_loss.item()) <= 1e-2 * abs(o_loss.item())
    assert set(errs) == set(o_grads) and errs[worst] <= 3e-2, errs
    # an optimizer step invalidates the cached x3 weight images: the next forward must see the new masters
    h0 = m.pre_quant_tokens(x, precision="x3").clone()
    m.configure_optimizers()[0][0].step()
    h1 = m.pre_quant_tokens(x, precision="x3")
    P1 = {k: v.detach().cpu().float() for k, v in m.state_dict().items() if not k.startswith("loss.")}
    with torch.no_grad():
        o_h1 = O.encode(x, P1, cfg)[3]
    assert rel(h1, o_h1) <= 5e-5 and not torch.equal(h0, h1)


def test_full_x3_forward_reconstruction_and_loss_vs_fp32_oracle():
    """encoder_precision = decoder_precision = "x3": the WHOLE forward (codes, reconstruction, pixel and codebook loss) within ~1e-5 of the fp32 reference
    path — north_star's "recon tensors within 1e-3" met with two orders of margin on the bf16 matrix cores — at the base widths (B = 2) and, against the
    REFERENCE's own golden vectors, on the tiny model; the backward stays the bf16 product path's on the saved hi planes."""
    import vitvq_oracle as O
    P = O.make_params(BASE, 0)
    x = O.make_images(1, 2, BASE["image_size"])
    m = _build(BASE, P, encoder_precision="x3", decoder_precision="x3")
    with torch.no_grad():
        xrec, qloss = m(x)
        o_xrec, o_q = O.forward(x, P, BASE)
    e_x = rel(xrec, o_xrec)
    print(f"full x3 forward, base B=2: xrec rel {e_x:.2e}, qloss {qloss.item():.7f} vs {o_q.item():.7f}")
    assert e_x <= 1e-4 and abs(qloss.item() - o_q.item()) <= 1e-4 * abs(o_q.item())
    out = m.engine.forward_backward(x, w_l1=0.0, w_l2=1.0, codebook_weight=1.0)
    o_loss, _, o_grads, _ = O.train_step_grads(x, P, BASE)
    m.engine.unscale_grads()      # fp16 engine: param.grad carries the loss scale until the step (or this call)
    errs = {k: rel(p.grad, o_grads[k]) for k, p in m.named_parameters() if k in o_grads}
    worst = max(errs, key=errs.get)
    print(f"  train step: loss {out['loss'].item():.7f} vs {o_loss.item():.7f}; grads median rel {np.median(list(errs.values())):.2e}, worst {worst} {errs[worst]:.2e}")
    assert abs(out["loss"].item() - o_loss.item()) <= 1e-4 * abs(o_loss.item())
    assert errs[worst] <= 1.5e-2
    # the reference's own outputs (tiny model)
    cfg = O.TINY_CFG
    Pt = O.make_params(cfg, seed=11)
    xt = O.make_images(5, 2, cfg["image_size"])
    mt = _build(cfg, Pt, encoder_precision="x3", decoder_precision="x3")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vit_tiny.npz"))
    with torch.no_grad():
        xr, ql = mt(xt)
    print(f"  tiny vs REFERENCE golden: xrec rel {rel(xr, torch.from_numpy(g['xrec'])):.2e}, qloss {ql.item():.7f} vs {float(g['qloss']):.7f}")
    assert rel(xr, torch.from_numpy(g["xrec"])) <= 5e-5 and abs(ql.item() - float(g["qloss"])) <= 1e-5 * abs(float(g["qloss"]))


def test_x3_graph_replay_after_refresh_shadows_reads_the_new_weights():
    """the split weight images of an x3 TRAINING precision are rebuilt on the host side of the step; a HIP-graph replay never runs that
    code, so every write to the masters — optimizer_step AND refresh_shadows (checkpoint load, broadcast) — must rebuild them eagerly."""
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    P = O.make_params(cfg, seed=11)
    x = O.make_images(5, 2, cfg["image_size"]).cuda()
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPS",
            "params": dict(codebook_weight=1.0, loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0)}

    def build():
        m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(cfg["quantizer"]),
                  AttrDict.wrap(loss))
        m.precision = "bf16"
        m.load_state_dict(P)
        e = m.engine
        e.encoder_precision = e.decoder_precision = "x3"
        return m, e
    m, e = build()
    e.use_graphs = True
    a = float(e.forward_backward_graphed(x)["loss"])              # captures (weights W0)
    with torch.no_grad():
        e.store.p.mul_(1.05)                                      # an external write to the masters ...
    e.store.refresh_shadows()                                     # ... announced the documented way
    b = float(e.forward_backward_graphed(x)["loss"])              # replay: must see W1 in EVERY operand image, the split ones included
    m2, e2 = build()
    with torch.no_grad():
        e2.store.p.mul_(1.05)
    e2.store.refresh_shadows()
    ref = float(e2.forward_backward(x)["loss"])                   # eager, same weights
    assert a != b and b == ref, (a, b, ref)


def test_gemm_split2_is_bitwise_the_gemm_then_split2(C):
    """round 5: the qkv projection's hi / lo planes straight from the persistent GEMM's epilogue (enh_gemm_bf16_split) == enh_gemm_h16 (f32 out) followed
    by enh_split2_bf16, bit for bit (same accumulator, same definition hi = bf16(v), lo = bf16(v - hi)); leading dimensions respected"""
    torch.manual_seed(5)
    M, N, K = 8192, 2304, 3 * 768
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    assert C.gemm_split_fused(M, N, K)
    v = torch.empty(M, N, device="cuda")
    C.mm(a, b, M, N, K, v)
    eh = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); el = torch.empty_like(eh)
    C.split2(v, eh, el)
    hi = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda"); lo = torch.full_like(hi, 7.0)
    C.gemm_split2(a, b, M, N, K, hi, lo)
    torch.cuda.synchronize()
    assert torch.equal(hi, eh) and torch.equal(lo, el)
    assert not C.gemm_split_fused(2048, 768, K)          # too few tiles for the persistent kernel: the caller keeps the two-call form ...
    with pytest.raises(RuntimeError):
        C.gemm_split2(a[:2048], b[:768], 2048, 768, K, hi[:2048, :768].contiguous(), lo[:2048, :768].contiguous())     # ... and the entry point refuses


def test_gemm_split3_tanh_matches_the_gemm_then_split3(C):
    """fc1: y3 = [hi | lo | hi] of tanh(a b^T + bias) and the hi plane for the backward, from the GEMM epilogue.  hi + lo is the value to 2^-16 absolute
    against tanh in f64 (the bound the split3 kernel's own test uses) and agrees with the two-call form to the same level; planes are consistent."""
    torch.manual_seed(6)
    M, N, K = 8192, 3072, 3 * 768
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    assert C.gemm_split_fused(M, N, K)
    v = torch.empty(M, N, device="cuda")
    C.mm(a, b, M, N, K, v)
    ref = torch.tanh((v.double() + bias.double()))
    e3 = torch.empty(M, 3 * N, dtype=torch.bfloat16, device="cuda")
    C.split3(v, e3, bias=bias, act=C.ACT_TANH, order=0)
    y3 = torch.full((M, 3 * N), 7.0, dtype=torch.bfloat16, device="cuda"); yh = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    C.gemm_split3_tanh(a, b, M, N, K, bias, y3, yh)
    torch.cuda.synchronize()
    assert torch.equal(y3[:, :N], y3[:, 2 * N:]) and torch.equal(y3[:, :N], yh)
    got = y3[:, :N].double() + y3[:, N:2 * N].double()
    two = e3[:, :N].double() + e3[:, N:2 * N].double()
    err, err2 = (got - ref).abs().max().item(), (two - ref).abs().max().item()
    relerr = ((got - ref).abs() / ref.abs().clamp_min(1e-3)).max().item()
    print(f"fused tanh split: max |hi + lo - tanh| {err:.2e} (two-call form {err2:.2e}), max relative {relerr:.2e}; small-|x| share {(v.abs() < 0.12).float().mean().item():.3f}")
    assert err <= 2 ** -16 and relerr <= 3e-5
    # lo really is bf16(value - hi) of the kernel's own value: |value - hi - lo| <= half an ulp of lo
    assert (y3[:, :N].float() - yh.float()).abs().max().item() == 0.0
    # without the hi plane for the backward (save = False)
    y3b = torch.empty_like(y3)
    C.gemm_split3_tanh(a, b, M, N, K, bias, y3b, None)
    assert torch.equal(y3b, y3)


def test_x3_encode_with_fused_split_epilogues_equals_the_two_call_form(monkeypatch):
    """the base encoder at 8 images (M = 8192: both fused forms are served): h and codes with ENH_X3_FUSED_SPLIT=1 (default) vs the round-4 form"""
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    import vitvq_oracle as O
    cfg = dict(BASE)
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPS", "params": dict(codebook_weight=1.0, loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0)}
    torch.manual_seed(0)
    m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
    x = O.make_images(3, 8, 256).cuda()
    e = m.engine
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("ENH_X3_FUSED_SPLIT", flag)
        img = e._check_img(x)
        h = e._pre_quant(e._encode_tokens(img, save=False, x3=True), 8).clone()
        out[flag] = (h, e.encode_codes(x, precision="x3").clone())
    dh = rel(out["1"][0], out["0"][0])
    same = (out["0"][1] == out["1"][1]).float().mean().item()
    print(f"x3 encode, fused vs two-call split: h rel diff {dh:.2e}, code agreement {same:.6f}")
    assert dh <= 2e-5 and same >= 0.9995     # tanh is evaluated by another (equally accurate) formula in the epilogue: fp32 near-ties may move
