"""Rounding-point ablation of the bf16 product path (TEST / LAB INFRASTRUCTURE — uses the CPU oracle; never imported by the package).

VERDICT r2 "next" 1(c): the measured path sits at h ~5e-3 relative to the fp32 reference where one bf16 rounding is 1.66e-3.  WHICH rounding points
contribute?  This script re-runs the fp32 oracle's encoder (oracle/vitvq_oracle.py, pinned to the reference) with the product path's bf16 rounding points
switched on ONE CLASS AT A TIME — exactly the tensors enhancing/engine/stage1.py stores or feeds to MFMA as bf16, everything else fp32 as in the engine:

    w     every GEMM weight operand (the bf16 shadow p16 the AdamW kernel rewrites)       patch  the patchified image (A operand of the patch-embed GEMM)
    ln    LayerNorm outputs a1 / a2 / the final norm (A operands of qkv, fc1, pre_quant)   qkv    the packed q, k, v the attention kernel reads
    p     softmax numerators exp(s - m) fed to the P.V MFMA (row sum kept in fp32)          o      attention output (A operand of to_out)
    hid   tanh(fc1) (A operand of fc2)

A bf16-operand MFMA with fp32 accumulation equals an fp32 matmul of the bf16-ROUNDED operand values up to summation order (1e-7), so rounding the
values in the fp32 oracle reproduces the product path's arithmetic error; "all" must therefore land on the error measured on the MI355X
(tests/test_parity_base_gpu.py prints it; profiles/r02_parity_base_configs.txt: h 5.6e-3 at base) — that equality is what validates the table.

    python tests/rounding_ablation.py [--batch 2] [--spread]        (CPU only, ~2 min at base dims)

--spread replaces the random-init codebook by jittered l2-normalised rows of h (a trained-like usage spread: >= 1000 distinct codes in play), the
regime in which the end-to-end code flip rate is meaningful (random init collapses to ~30 codes).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vitvq_oracle as O  # noqa: E402

CLASSES = ["w", "patch", "ln", "qkv", "p", "o", "hid"]


RB_DTYPE = torch.bfloat16      # --dtype: the 16-bit format of the rounding points (round 6: fp16 = the engine's default operand format, 11 significand bits against 8)


def rb(t, on):
    return t.to(RB_DTYPE).to(torch.float32) if on else t


def decoder_xrec(zq, P, cfg, on: set):
    """oracle post_quant + decoder + to_pixel (vitvq_oracle.decode) with the engine's rounding points: the quantized tokens (A operand of post_quant) count as "ln"-class
    activations, the rest as in encoder_h"""
    W = lambda k: rb(P[k], "w" in on)
    patch, heads, depth = cfg["patch_size"], cfg["decoder"]["heads"], cfg["decoder"]["depth"]
    x = rb(zq, "ln" in on) @ W("post_quant.weight").t() + P["post_quant.bias"] + P["decoder.de_pos_embedding"]
    B, N, _ = x.shape
    for i in range(depth):
        p = f"decoder.transformer.layers.{i}."
        a1 = rb(O.layer_norm(x, P[p + "0.norm.weight"], P[p + "0.norm.bias"]), "ln" in on)
        qkv = rb(a1 @ W(p + "0.fn.to_qkv.weight").t(), "qkv" in on)
        q, k, v = qkv.chunk(3, dim=-1)
        sp = lambda t: t.reshape(B, N, heads, 64).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        s_ = (q @ k.transpose(-1, -2)) * 64 ** -0.5
        e = torch.exp(s_ - s_.amax(-1, keepdim=True))
        att = (rb(e, "p" in on) @ v) / e.sum(-1, keepdim=True)
        o = rb(att.permute(0, 2, 1, 3).reshape(B, N, heads * 64), "o" in on)
        x = o @ W(p + "0.fn.to_out.weight").t() + P[p + "0.fn.to_out.bias"] + x
        a2 = rb(O.layer_norm(x, P[p + "1.norm.weight"], P[p + "1.norm.bias"]), "ln" in on)
        hid = rb(torch.tanh(a2 @ W(p + "1.fn.net.0.weight").t() + P[p + "1.fn.net.0.bias"]), "hid" in on)
        x = hid @ W(p + "1.fn.net.2.weight").t() + P[p + "1.fn.net.2.bias"] + x
    a = rb(O.layer_norm(x, P["decoder.transformer.norm.weight"], P["decoder.transformer.norm.bias"]), "ln" in on)
    w = W("decoder.to_pixel.1.weight")
    pix = a @ w.reshape(w.shape[0], -1) + P["decoder.to_pixel.1.bias"].repeat_interleave(patch * patch)
    return pix      # (the patch layout: the permutation to [B,C,H,W] does not change a Frobenius-relative error)


def encoder_h(img, P, cfg, on: set, trace=None):
    """oracle encoder + pre_quant (vitvq_oracle.encoder / encode) with the engine's bf16 rounding points applied for the classes in `on`."""
    W = lambda k: rb(P[k], "w" in on)
    patch, heads, depth = cfg["patch_size"], cfg["encoder"]["heads"], cfg["encoder"]["depth"]
    w = W("encoder.to_patch_embedding.0.weight")
    x = rb(O.patchify(img, patch), "patch" in on) @ w.reshape(w.shape[0], -1).t() + P["encoder.to_patch_embedding.0.bias"]
    x = x + P["encoder.en_pos_embedding"]
    B, N, _ = x.shape
    for i in range(depth):
        p = f"encoder.transformer.layers.{i}."
        a1 = rb(O.layer_norm(x, P[p + "0.norm.weight"], P[p + "0.norm.bias"]), "ln" in on)
        qkv = rb(a1 @ W(p + "0.fn.to_qkv.weight").t(), "qkv" in on)
        q, k, v = qkv.chunk(3, dim=-1)
        sp = lambda t: t.reshape(B, N, heads, 64).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        s = (q @ k.transpose(-1, -2)) * 64 ** -0.5
        e = torch.exp(s - s.amax(-1, keepdim=True))
        att = (rb(e, "p" in on) @ v) / e.sum(-1, keepdim=True)          # the kernel sums the UNROUNDED numerators in fp32 and rounds only the MFMA operand
        o = rb(att.permute(0, 2, 1, 3).reshape(B, N, heads * 64), "o" in on)
        x = o @ W(p + "0.fn.to_out.weight").t() + P[p + "0.fn.to_out.bias"] + x
        a2 = rb(O.layer_norm(x, P[p + "1.norm.weight"], P[p + "1.norm.bias"]), "ln" in on)
        hid = rb(torch.tanh(a2 @ W(p + "1.fn.net.0.weight").t() + P[p + "1.fn.net.0.bias"]), "hid" in on)
        x = hid @ W(p + "1.fn.net.2.weight").t() + P[p + "1.fn.net.2.bias"] + x
        if trace is not None:
            trace.append(x)
    a = rb(O.layer_norm(x, P["encoder.transformer.norm.weight"], P["encoder.transformer.norm.bias"]), "ln" in on)
    return a @ W("pre_quant.weight").t() + P["pre_quant.bias"]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def model_init_params(seed: int):
    """the weights bench.py's parity_mode block sees: configs/imagenet_vitvq_base.yaml through the package's own constructor (the reference's init, seed-for-seed:
    tests/test_host_cpu.py), on the CPU — no engine is bound, no kernel runs"""
    sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
    from enhancing.utils.general import get_config_from_file, initialize_from_config, set_seed
    set_seed(seed)
    cfg = get_config_from_file(os.path.join(ROOT, "configs", "imagenet_vitvq_base.yaml"))
    m = initialize_from_config(cfg.model)
    return {k: v.detach().float().clone() for k, v in m.state_dict().items() if not k.startswith("loss.")}


def run(batch: int, spread: bool, cfg=None, seed: int = 0, out=sys.stdout, init: str = "oracle", decoder_classes: bool = False):
    cfg = cfg or dict(image_size=256, patch_size=8, encoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072),
                      decoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072), quantizer=dict(embed_dim=32, n_embed=8192))
    torch.manual_seed(seed)
    P = model_init_params(seed) if init == "model" else O.make_params(cfg, seed)
    print(f"# parameters: {'the package constructor on configs/imagenet_vitvq_base.yaml (the reference init; what bench.py measures)' if init == 'model' else 'oracle.make_params (the test-suite init)'}", file=out)
    img = O.make_images(seed, batch, cfg["image_size"])
    with torch.no_grad():
        tr0 = []
        h0 = encoder_h(img, P, cfg, set(), tr0)
        h_ref = O.encode(img, P, cfg)[3]
        assert rel(h0, h_ref) < 1e-5, "the instrumented encoder must equal the oracle when nothing is rounded"
        E = P["quantizer.embedding.weight"]
        if spread:       # trained-like codebook: jittered normalised rows of h -> thousands of codes in play
            g = torch.Generator().manual_seed(seed + 1)
            flat = torch.nn.functional.normalize(h0.reshape(-1, h0.shape[-1]), dim=-1)
            pick = torch.randint(0, flat.shape[0], (E.shape[0],), generator=g)
            E = torch.nn.functional.normalize(flat[pick] + 0.35 * torch.randn(E.shape[0], E.shape[1], generator=g) / E.shape[1] ** 0.5, dim=-1)
        q = O.qparams(cfg)
        idx0 = O.quantizer_forward(h0, E, **q)[2]
        print(f"# base encoder, batch {batch} ({h0.shape[0] * h0.shape[1]} tokens), codebook {'spread (jittered h rows)' if spread else 'random init'}: "
              f"{idx0.unique().numel()} distinct codes in play", file=out)
        print(f"# {'rounded class':<14} {'h rel err':>10} {'resid L1':>10} {'resid L12':>10} {'code flips':>11}", file=out)
        rows = {}
        for name, on in [(c, {c}) for c in CLASSES] + [("all", set(CLASSES)), ("all but w", set(CLASSES) - {"w"}), ("all but ln", set(CLASSES) - {"ln"}),
                                                     ("w + ln", {"w", "ln"})]:
            tr = []
            h = encoder_h(img, P, cfg, on, tr)
            idx = O.quantizer_forward(h, E, **q)[2]
            flips = float((idx != idx0).float().mean())
            rows[name] = dict(h=rel(h, h0), l1=rel(tr[0], tr0[0]), l12=rel(tr[-1], tr0[-1]), flips=flips)
            r = rows[name]
            print(f"  {name:<14} {r['h']:>10.2e} {r['l1']:>10.2e} {r['l12']:>10.2e} {flips:>11.4f}", file=out)
        quad = sum(rows[c]["h"] ** 2 for c in CLASSES) ** 0.5
        print(f"# quadrature sum of the seven single-class errors: {quad:.2e} (all together: {rows['all']['h']:.2e}) -> the contributions are independent", file=out)
        # residual quantizer (BASELINE config 4: depth 4, one shared codebook): per-depth match of the all-classes-rounded h against the unrounded one
        q4 = dict(q, use_residual=True, num_quantizers=4)
        i0 = O.quantizer_forward(h0, E, **q4)[2]
        i1 = O.quantizer_forward(encoder_h(img, P, cfg, set(CLASSES)), E, **q4)[2]
        print(f"# RQ-4 on the same h: per-depth match {(i0 == i1).float().mean().item():.4f}, whole-tuple match {(i0 == i1).all(-1).float().mean().item():.4f}", file=out)
        # decoder side: xrec downstream of the SAME quantized tokens (the unrounded encoder's), every class rounded
        zq = O.encode(img, P, cfg)[0]
        x0 = decoder_xrec(zq, P, cfg, set())
        x1 = decoder_xrec(zq, P, cfg, set(CLASSES))
        rows["xrec"] = dict(h=rel(x1, x0))
        print(f"# decoder (post_quant .. to_pixel) with every class rounded, same codes: xrec rel err {rows['xrec']['h']:.2e}", file=out)
        if decoder_classes:       # which class carries the decoder's error ("the table that shows which class is the deliverable")
            dq = 0.0
            for c in [x for x in CLASSES if x != "patch"]:
                e = rel(decoder_xrec(zq, P, cfg, {c}), x0)
                dq += e * e
                print(f"  decoder, only {c:<6} rounded: xrec rel err {e:.2e}", file=out)
            print(f"# quadrature sum of the decoder's single-class errors: {dq ** 0.5:.2e}", file=out)
            print(f"# norms: |xrec| per pixel rms {float(x0.double().pow(2).mean().sqrt()):.3f}, mean {float(x0.mean()):.3f}", file=out)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--spread", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16", help="16-bit format of the rounding points")
    ap.add_argument("--init", choices=["oracle", "model"], default="oracle", help="oracle.make_params (the test-suite init) or the package constructor (bench.py's weights)")
    ap.add_argument("--decoder-classes", action="store_true", help="per-class table for the decoder side as well")
    a = ap.parse_args()
    RB_DTYPE = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    print(f"# rounding points in {a.dtype}")
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    run(a.batch, a.spread, init=a.init, decoder_classes=a.decoder_classes)
