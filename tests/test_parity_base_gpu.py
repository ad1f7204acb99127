"""End-to-end parity of the MEASURED path (bf16 MFMA operands, fp32 accumulation / residual stream) at the BENCHMARKED
configurations — BASELINE.json configs 2, 4 and 5 at their real widths and depths (768/12/12/3072 towers, K = 8192, 256 px;
RQ depth 4; the large 512/8 + 1280/32 towers) — against the fp32 CPU oracle run with identical fp32 master weights.

What is reported (printed as a table per test; `pytest -s` shows it, the numbers are copied to DESIGN.md §4):
  * per-layer relative Frobenius error of the residual stream, encoder and decoder;
  * h (quantizer input), xrec, loss;
  * code match-rate END TO END (HIP codes from the bf16-operand h vs oracle codes from the fp32 h) and at the OP BOUNDARY
    (oracle quantizer fed the HIP path's own h: must be 1.0 — every index bit-exact);
  * one training step's parameter gradients (median / worst relative error).

Tolerances.  north_star asks 1e-3 rel for bf16 activations.  Rounding an exact tensor to bf16 already costs 1.66e-3
(util.bf16_floor), so that bound is unreachable for any quantity that passes through a bf16-stored GEMM operand; the bounds
asserted here are the measured errors of this path at these sizes x ~1.5 (see the constants), i.e. regression bounds — the
activations / gradients are compared with the oracle run DOWNSTREAM OF THE SAME CODES (oracle `force_idx`), so that arithmetic
error is not mixed with the O(1) effect of a flipped near-tie index; the freely-run oracle gives the end-to-end match-rate and the
"incl. index flips" reconstruction error.  The fp32 exact mode (test_model_gpu.py::test_exact_mode_*) is the instrument that meets 1e-3 (4e-7 measured).
"""
import copy

import numpy as np
import pytest
import torch

from util import rel

pytestmark = pytest.mark.gpu

BASE = dict(image_size=256, patch_size=8, encoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072),
            decoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072), quantizer=dict(embed_dim=32, n_embed=8192))
LARGE = dict(image_size=256, patch_size=8, encoder=dict(dim=512, depth=8, heads=8, mlp_dim=2048),
             decoder=dict(dim=1280, depth=32, heads=16, mlp_dim=5120), quantizer=dict(embed_dim=32, n_embed=8192))

# Regression bounds of the HEADLINE (single-pass bf16) path: 1.15 x the values measured on MI355X with the round-5 kernels (profiles/r05_parity_base_configs.txt;
# the round-2 bounds were ~1.5 x measured and would not have caught a regression of the measured path).  The kernels are bit-reproducible, so the
# measured values repeat exactly from run to run and box to box; the 15 % are room for deliberate kernel changes that move roundings, not for noise.
# per case: (max residual-stream error over the layers, h, xrec downstream of the same codes, worst parameter gradient, minimum end-to-end code match-rate)
MEASURED = {
    "base":        (5.07e-3, 5.57e-3, 5.68e-3, 8.99e-3, 0.9810),
    "rq4":         (5.30e-3, 4.04e-3, 5.75e-3, 8.42e-3, 0.9685),
    "base_spread": (4.81e-3, 7.46e-3, 5.10e-3, 1.00e-2, 0.9507),      # (gradients: the codebook's own bound is separate, see _run_case)
    "rq4_spread":  (5.03e-3, 4.43e-3, 5.70e-3, 9.00e-3, 0.8895),
    "large":       (5.42e-3, 5.63e-3, 6.12e-3, 9.63e-3, 0.9775),
}


def _tols(case, match_margin=0.01):
    st, h, xr, g, mt = MEASURED[case]
    return (1.15 * st, 1.15 * h, 1.15 * xr, 1.15 * g, mt - match_margin)


STREAM_TOL, H_TOL, XREC_TOL, GRAD_TOL, MATCH_MIN = _tols("base")      # (names kept for importers: the base case's bounds)


def _build(cfg, P):
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPS",
            "params": dict(codebook_weight=1.0, loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0)}
    m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
              AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
    m.precision = "bf16"      # this file holds the regression bounds of the bf16 product path; the fp16 default is tests/test_fp16_gpu.py's
    m.load_state_dict(P, strict=True)
    assert m.engine.precision == "bf16"
    return m


def _spread_codebook(P, x, cfg, seed, jitter=0.1):
    """A trained-like usage spread for the argmin (random init collapses to ~30 codes of 8192): the codebook becomes jittered,
    l2-normalised rows of the fp32 oracle's own h for this batch — every token then has a few near-duplicates of its own direction among thousands of
    close competitors, so nearly every token picks a DIFFERENT code and the top-2 gaps are small (the hard regime for index parity)."""
    import vitvq_oracle as O
    with torch.no_grad():
        h = O.encode(x, P, cfg)[3]
    g = torch.Generator().manual_seed(seed + 77)
    flat = torch.nn.functional.normalize(h.reshape(-1, h.shape[-1]), dim=-1)
    K, d = P["quantizer.embedding.weight"].shape
    pick = torch.randint(0, flat.shape[0], (K,), generator=g)
    P["quantizer.embedding.weight"] = torch.nn.functional.normalize(flat[pick] + jitter * torch.randn(K, d, generator=g) / d ** 0.5, dim=-1).contiguous()


def _run_case(label, cfg, B, seed, tols=None, spread=False, min_distinct=0):
    import vitvq_oracle as O
    torch.set_num_threads(min(32, max(torch.get_num_threads(), 8)))
    stream_tol, h_tol, xrec_tol, grad_tol, match_min = tols or (STREAM_TOL, H_TOL, XREC_TOL, GRAD_TOL, MATCH_MIN)
    P = O.make_params(cfg, seed)
    x = O.make_images(seed + 1, B, cfg["image_size"], smooth=not spread)
    if spread:
        _spread_codebook(P, x, cfg, seed)
    m = _build(cfg, P)
    eng = m.engine
    out = eng.forward_backward(x, w_l1=0.0, w_l2=1.0, codebook_weight=1.0)   # = ViTVQ.training_step(optimizer_idx=0) with this loss config
    torch.cuda.synchronize()
    loss = out["loss"]
    M = B * eng.n_tok
    io = eng._io_bufs(B)
    codes = (out["indices"].view(B, eng.n_tok, -1) if eng.q.use_residual else out["indices"].view(B, eng.n_tok)).cpu()
    # (1) the oracle run freely: end-to-end match-rate and reconstruction error INCLUDING near-tie index flips
    with torch.no_grad():
        o_q, o_ql, o_idx, o_h = O.encode(x, P, cfg)
        o_xrec_free = O.decode(o_q, P, cfg)
    match_e2e = (codes == o_idx).float().mean().item()
    e_x_free = rel(io["xrec"], o_xrec_free)
    # (2) op boundary: the oracle's quantizer on the HIP path's own h -> every index must be bit-exact
    _, _, idx_ob = O.quantizer_forward(io["h"].cpu().view(B, eng.n_tok, -1), P["quantizer.embedding.weight"], **O.qparams(cfg))
    match_ob = (codes == idx_ob).float().mean().item()
    # (3) the oracle's training step downstream of the SAME discrete codes: pure arithmetic error of activations and gradients
    ref = O.train_step_traced(x, P, cfg, force_idx=codes)
    rows = []
    for name, tower, tr in (("enc", eng.enc, ref["enc_trace"]), ("dec", eng.dec, ref["dec_trace"])):
        xs = tower.bufs(B, True)["x"]
        assert len(tr) == tower.depth + 1
        for i, t in enumerate(tr):
            rows.append((f"{name}.x[{i}]", rel(xs[i], t.reshape(M, -1))))
    e_h, e_x = rel(io["h"], ref["h"].reshape(M, -1)), rel(io["xrec"], ref["xrec"])
    errs = {k: rel(p.grad, ref["grads"][k]) for k, p in m.named_parameters() if k in ref["grads"]}
    worst = max(errs, key=errs.get)
    lines = [f"== {label}: B={B}, bf16 MFMA operands vs fp32 CPU oracle =="]
    lines += [f"  {n:12s} rel {e:.2e}" for n, e in rows]
    lines += [f"  h            rel {e_h:.2e}", f"  xrec         rel {e_x:.2e}  (same codes)   {e_x_free:.2e} (oracle run freely, incl. index flips)",
              f"  loss {loss.item():.6f} vs {ref['loss'].item():.6f}   qloss {out['quant_loss'].item():.6f} vs {ref['qloss'].item():.6f}",
              f"  code match-rate end-to-end {match_e2e:.4f}   at the op boundary (identical h) {match_ob:.6f}   distinct codes used {o_idx.unique().numel()}",
              f"  gradients (same codes): median rel {np.median(list(errs.values())):.2e}, worst {worst} {errs[worst]:.2e}",
              "  worst five: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:5])]
    print("\n" + "\n".join(lines))
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_base.txt"), "a") as f:
            f.write("\n".join(lines) + "\n")
    assert set(errs) == set(ref["grads"])
    # indices at the op boundary (identical quantizer input): bit-exact against the C oracle — the kernel's arithmetic twin, itself pinned to the REFERENCE's golden
    # indices in tests/test_ops_gpu.py — and against the torch restatement up to audited fp32 near-ties (two fp32 evaluations of quantizers.py:78-80 may order
    # two codes differently only if their exact distance gap is below (6n + 12) u = 1.22e-5: bench.py VQ_NEAR_TIE_BOUND; every mismatch is checked in fp64 at the
    # first depth where the code tuples part)
    import vq_oracle as VC
    qp = O.qparams(cfg)
    depth = int(qp["num_quantizers"]) if qp["use_residual"] else 1
    h_np = io["h"].cpu().view(-1, io["h"].shape[-1]).numpy()
    _, idx_c, _ = VC.forward(h_np, P["quantizer.embedding.weight"].numpy(), beta=qp["beta"], depth=depth, use_norm=qp["use_norm"])
    assert torch.equal(codes.reshape(-1, depth), torch.from_numpy(idx_c).reshape(-1, depth)), "indices must be bit-exact against the C oracle for identical quantizer input"
    if match_ob != 1.0:
        assert qp["use_norm"], "the near-tie bound is derived for l2-normalised codes"
        cg, co = codes.reshape(-1, depth), idx_ob.reshape(-1, depth)
        bad = (cg != co).any(-1).nonzero().view(-1)
        assert bad.numel() <= 1e-3 * cg.shape[0], f"{bad.numel()} tokens differ from the torch restatement"
        E64 = torch.nn.functional.normalize(P["quantizer.embedding.weight"].double(), dim=-1)
        for t in bad.tolist():
            d0 = int((cg[t] != co[t]).nonzero()[0])
            r = torch.from_numpy(h_np[t]).double()
            for d_ in range(d0):
                r = r - E64[cg[t, d_]]
            zn = torch.nn.functional.normalize(r, dim=-1)
            gap = abs(float(((zn - E64[cg[t, d0]]) ** 2).sum() - ((zn - E64[co[t, d0]]) ** 2).sum()))
            assert gap <= (6 * 32 + 12) * 2.0 ** -24, f"token {t}, depth {d0}: exact gap {gap:.3e} is not a near-tie"
        print(f"  op boundary vs the torch restatement: {bad.numel()} audited near-tie(s) of {cg.shape[0]} tokens")
    assert o_idx.unique().numel() >= min_distinct, f"only {o_idx.unique().numel()} distinct codes in play"
    assert max(e for _, e in rows) <= stream_tol, rows
    assert e_h <= h_tol and e_x <= xrec_tol
    assert abs(loss.item() - ref["loss"].item()) <= 1e-2 * abs(ref["loss"].item())
    assert match_e2e >= match_min
    if spread:
        # with near-duplicate codes z_q ~ z (codebook loss 3e-4): the codebook gradient is a difference of nearly equal unit vectors, so the 7e-3 error of
        # h is a ~8e-2 error of (z_q - z) — cancellation, not arithmetic (measured 8.4e-2); every other parameter keeps the common bound
        cb = errs.pop("quantizer.embedding.weight")
        assert cb <= 0.125, cb                     # measured 8.4e-2 (base) / 1.06e-1 (RQ-4)
        worst = max(errs, key=errs.get)
    assert errs[worst] <= grad_tol, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


def test_base_config2_bf16_vs_oracle():
    """BASELINE config 2: imagenet_vitvq_base.yaml towers — the configuration bench.py times."""
    _run_case("imagenet_vitvq_base (config 2)", BASE, 2, 0, tols=_tols("base"))


def test_base_rq4_config4_bf16_vs_oracle():
    """BASELINE config 4: RQ-VAE base, ResidualQuantizer depth 4, one shared codebook."""
    cfg = copy.deepcopy(BASE)
    cfg["quantizer"].update(use_residual=True, num_quantizers=4)
    _run_case("imagenet_rqvae_base (config 4)", cfg, 2, 3, tols=_tols("rq4"))


def test_base_config2_with_a_trained_like_code_spread():
    """config 2 with >= 1000 distinct codes in play (2048 tokens): the op-boundary match must STILL be exactly 1.0; the end-to-end rate (small top-2 gaps make
    it the worst case for the bf16-operand h) is held to the measured value - 0.01 like the others."""
    _run_case("imagenet_vitvq_base (config 2), spread codebook", BASE, 2, 10, tols=_tols("base_spread"), spread=True, min_distinct=1000)


def test_base_rq4_config4_with_a_trained_like_code_spread():
    cfg = copy.deepcopy(BASE)
    cfg["quantizer"].update(use_residual=True, num_quantizers=4)
    _run_case("imagenet_rqvae_base (config 4), spread codebook", cfg, 2, 13, tols=_tols("rq4_spread"), spread=True, min_distinct=1000)


def test_large_config5_towers_bf16_vs_oracle():
    """BASELINE config 5 towers: imagenet_vitvq_large.yaml (encoder 512/8/8/2048, decoder 1280/32/16/5120), AE step only."""
    _run_case("imagenet_vitvq_large towers (config 5)", LARGE, 1, 5, tols=_tols("large"))


def test_training_step_gradients_are_bit_reproducible():
    """two passes over the same batch from the same weights give the SAME gradient bits for EVERY parameter: weight gradients (two-pass split-K),
    LayerNorm dgamma / dbeta and the fused bias column sums (per-workgroup partials + fixed-order second pass), fc1 bias (two-pass column sums),
    attention (no atomics) and — since round 4 — the codebook gradient (owner-scans reduction in a fixed order, csrc/vq.hip; it used an LDS hash
    + f32 atomics before).  What a diff of two multi-GPU runs needs.  Also under the RQ-4 quantizer and with the dynamic GEMM tile schedule on."""
    import vitvq_oracle as O
    for rq in (False, True):
        cfg = copy.deepcopy(BASE)
        if rq:
            cfg["quantizer"].update(use_residual=True, num_quantizers=4)
        P = O.make_params(cfg, 3)
        x = O.make_images(4, 2, cfg["image_size"])
        m = _build(cfg, P)
        eng = m.engine
        runs = []
        for _ in range(3):
            eng.store.zero_grad()
            eng.forward_backward(x, w_l1=0.0, w_l2=1.0, codebook_weight=1.0)
            torch.cuda.synchronize()
            runs.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        diff = [k for k in runs[0] if not (torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]))]
        print(f"rq={rq}: parameters whose gradient bits differ between identical passes:", diff)
        assert not diff, diff
        assert runs[0]["quantizer.embedding.weight"].abs().sum().item() > 0
