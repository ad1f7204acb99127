"""bench.py — images/sec of the ViT-VQGAN-base 256x256 stage-1 AE training step on N MI355X (one process per GPU).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env, RCCL over xGMI); started bare (no launcher environment) with
--gpus N > 1 it re-executes itself under torch.distributed.run, one rank per GPU.  A "step" = one full AE
training step of configs/imagenet_vitvq_base.yaml on one synthetic ImageNet-shaped batch per GPU: forward + backward +
gradient all-reduce + fused AdamW, loss = 1.0*L2 + 1.0*codebook (LPIPS / GAN weights 0 — stated in config.workload).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects (rank 0, N = 1 unless noted):
  "roofline"   the kernel with the largest share of the step over ALL timed kernels (every GEMM instantiation, attention, convolutions, the quantizer,
               LayerNorm backward, AdamW — each priced against its own roofline: bf16 MFMA 2.5 PF, exact-f32 MFMA 157.3 TF, HBM 8 TB/s), timed live
               with HIP events on the launch stream inside the timed region; `traffic` = PMC bytes per launch from profiles/pmc_step.json, attached only
               when that file records the SAME config and per-GPU batch, else null.  "kernels" carries the same figures for every timed kernel.
  "vq"         the quantizer launch: tokens/s, exact-f32 TF/s and its fraction of 157.3 TF, algorithmic bytes, PMC bytes and GB/s against the HBM peak.
  "vq_match"   second half of BASELINE's metric: the argmin match-rate at the op boundary on ONE extra forward pass (h, indices and codebook snapshotted
               together), every mismatch audited in fp64 (side holding the exact argmin, gap in ulps) and ENFORCED against the derived near-tie bound —
               the process exits with code 3 on a violation; "vq_match_spread": the same kernel and inputs against a trained-like codebook
               (~1000 distinct codes in play instead of the handful synthetic training collapses to).  "vq_match_rate" = the spread leg (since round 4);
               both legs also under explicit names: "vq_match_rate_training_codebook" / "vq_match_rate_spread_codebook".
  "parity_mode"  north_star's parity clause priced per precision mode: images/s of encode_codes and of the AE train step for the headline engine (fp16 MFMA
               operands since round 6), the other single-pass 16-bit format (bf16), the split-bf16 "x3" instrument (three MFMA passes, ~1e-5) and the
               exact-fp32 engine mode; plus h error / end-to-end code match / free-running reconstruction error of each against the fp32 CPU oracle on a
               8-image sample (8192 tokens: the reference yaml's batch), taken at the constructor's weights before the first training step ("vs_fp32_cpu_oracle": the state every parity
               test pins) and again after the warm-up + timed steps ("..._after_training_steps"); 10 timed iterations per mode.
  "cpu_baseline"  the CPU oracle — a port of the reference's PyTorch path, oracle/vitvq_oracle.py — timed on the host cores on a bounded sample.
  "comm"       (N > 1) per rank: exposed communication of the timed steps (compute-stream wait and host wait), buckets, bytes reduced, un-announced elements;
               with the adversarial configs the discriminator's own bucketed all-reduce is reported separately ("discriminator").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FWD_GFLOP_PER_IMG_BASE = 426.44      # SURVEY.md §8(d) / BASELINE.md
STEP_TFLOP_PER_IMG_BASE = 1.279      # 1 fwd + 1 bwd = 3x forward
MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3         # exact-f32 MFMA = the vector rate (same guide, "Matrix cores")
HBM_PEAK_GBS = 8000.0                # spec; ~6300 achievable (same guide, "HBM")
# A mismatch between two fp32 evaluations of quantizers.py:78-80 is a legitimate near-tie iff the EXACT (fp64) distance gap of the two codes is below
# this bound.  Derivation (n = 32, u = 2^-24; unit vectors, so every partial sum has magnitude <= 1 and d <= 4): an implementation computes
# fl(fl(A + B_j) - 2 C_j) with A = |zn|^2, B_j = |en_j|^2, C_j = zn.en_j, each an n-term fp32 sum whose error is <= n u in ANY summation order
# (the reference's torch.sum / einsum blocking is not specified, the kernel's is include/enh_hip.h's contract); A's error is common to every j and
# cancels in the argmin; the two outer operations round by <= 2u and <= 4u.  So each computed d_j is within (n + 2n + 6) u of (exact + constant), two
# codes can swap order in one implementation only if their exact gap is <= 2 (3n + 6) u, and two implementations can disagree only if one of them
# swapped: gap <= (6n + 12) u = 204 u = 1.22e-5 = 51 ulp of the intermediate sum A + B ~ 2.  That is the worst case; the random-walk expectation for
# the largest gap seen over ~1e9 comparisons is ~1e-6 (4-6 ulp), which is what the runs show.  Anything above the bound is a kernel bug -> FAIL.
PARITY_IMAGES = 8        # images of the parity_mode sample (8192 tokens: the end-to-end code match-rate of a ~0.2 % flip rate needs more than 2048 tokens to be a number)
VQ_N = 32
VQ_NEAR_TIE_BOUND = (6 * VQ_N + 12) * 2.0 ** -24
ULP_OF_2 = 2.0 ** -22


def cpu_baseline(max_seconds: float = 40.0):
    """CPU oracle (port of the reference's PyTorch path) timed on the host cores: base config, batch 2, fwd + bwd + AdamW in fp32.
    The torch thread count is swept ONCE (one step per candidate) and recorded; the value is the MEDIAN of >= 3 post-warm-up steps at
    the best count."""
    import statistics
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vitvq_oracle as O
    cfg = dict(image_size=256, patch_size=8, encoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072),
               decoder=dict(dim=768, depth=12, heads=12, mlp_dim=3072), quantizer=dict(embed_dim=32, n_embed=8192))
    B = 2
    P = O.make_params(cfg, 0)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in P.items()}
    x = O.make_images(0, B, 256)
    state = {"step": 0}

    def one_step():
        state["step"] += 1
        t0 = time.time()
        _, _, grads, _ = O.train_step_grads(x, P, cfg)
        for k, g in grads.items():
            O.adamw_step(P[k], g, m[k], v[k], state["step"], 4.5e-6)
        return time.time() - t0

    ncpu = os.cpu_count() or 1
    env = os.environ.get("ENH_CPU_BASELINE_THREADS")
    # all hardware threads of the 2-socket GPU host are pathologically slow for this op mix (measured in round 1: 355 s per batch-4
    # step at 256 threads), so the sweep covers the moderate counts only
    cands = [int(env)] if env else sorted({c for c in (16, 32, 64) if c <= ncpu} or {ncpu})
    t_begin, sweep = time.time(), {}
    torch.set_num_threads(cands[0])
    one_step()  # warm-up (allocator, thread pool)
    for c in cands:
        torch.set_num_threads(c)
        sweep[c] = one_step()
        if time.time() - t_begin > max_seconds * 0.5:
            break
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    times = []
    while len(times) < 3 or (len(times) < 5 and time.time() - t_begin < max_seconds * 0.8):
        times.append(one_step())
    med = statistics.median(times)
    return {"value": B / med, "unit": "images/s", "cores": cores, "kind": "port",
            "kind_note": "port = oracle/vitvq_oracle.py, the CPU restatement of the reference's PyTorch path, pinned to the reference's own modules by "
                         "oracle/make_golden.py -> tests/golden; /root/reference does not exist on the GPU box, so the reference itself cannot be timed here",
            "sample": f"median of {len(times)} post-warm-up AE train steps (fwd+bwd+AdamW, fp32) of ViT-VQGAN-base at batch {B} (BASELINE.md plans batch 8: "
                      f"bounded to {B} so that the default run stays within minutes; images/s is per image either way); "
                      f"thread sweep (s/step): {', '.join(f'{c}: {t:.2f}' for c, t in sweep.items())}",
            "step_seconds": [round(t, 3) for t in times]}


def vq_match_rate(h_dev, idx_dev, codebook_dev, depth: int, use_residual: bool):
    """Second half of BASELINE.json's metric: the VQ argmin match-rate at the OP BOUNDARY — a quantizer input h produced on the GPU by the measured path
    (M = per-GPU batch x 1024 tokens), quantized by the reference's formula on the host (oracle, the checker) vs the indices the HIP kernel produced for
    the SAME h and the SAME codebook (both snapshotted from one forward pass with no optimizer step in between).  Every mismatch is audited in fp64:
    which side (if any) holds the exact argmin, and the exact distance gap between the two picks in ulps of the intermediate sum |zn|^2 + |en|^2 ~ 2.
    status = "FAIL" if any gap exceeds VQ_NEAR_TIE_BOUND (see its derivation above): main() then exits non-zero.  SURVEY.md §8(d) metric 2."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vitvq_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    h, idx, E = h_dev.detach().float().cpu(), idx_dev.detach().cpu().view(h_dev.shape[0], -1), codebook_dev.detach().float().cpu()
    M, mism, audits = h.shape[0], 0, []
    en64 = torch.nn.functional.normalize(E.double(), dim=-1)
    for s in range(0, M, 16384):
        zz = h[s:s + 16384]
        _, _, it = O.quantizer_forward(zz, E, 0.25, True, use_residual, depth if use_residual else None)
        it = it.view(zz.shape[0], -1)
        bad = (it != idx[s:s + 16384]).any(dim=1).nonzero().view(-1)
        mism += len(bad)
        if use_residual:
            continue       # (the audit below is for the single-stage quantizer the headline config uses)
        for j in bad.tolist()[:4096]:      # every mismatch is audited (a healthy run has tens); beyond 4096 per chunk the status below turns "unverified"
            zn = torch.nn.functional.normalize(zz[j:j + 1].double(), dim=-1)
            d = ((zn ** 2).sum(1, keepdim=True) + (en64 ** 2).sum(1) - 2 * zn @ en64.t()).view(-1)
            ref_i, hip_i, best = int(it[j, 0]), int(idx[s + j, 0]), int(torch.argmin(d))
            gap = abs(float(d[ref_i] - d[hip_i]))
            audits.append({"token": s + j, "reference_pick": ref_i, "hip_pick": hip_i, "fp64_argmin": best,
                           "fp64_argmin_side": "hip" if best == hip_i else ("reference" if best == ref_i else "neither"),
                           "gap_fp64": gap, "gap_ulps_of_2": round(gap / ULP_OF_2, 2)})
    worst = max((a_["gap_fp64"] for a_ in audits), default=0.0)
    sides = {k: sum(1 for a_ in audits if a_["fp64_argmin_side"] == k) for k in ("hip", "reference", "neither")}
    ok = worst <= VQ_NEAR_TIE_BOUND
    status = "FAIL" if not ok else ("ok" if len(audits) == mism else "unverified")     # unverified: mismatches exist that were not audited (residual quantizer / > 4096 per chunk)
    return {"value": 1.0 - mism / M, "tokens": M, "mismatches": mism, "distinct_codes": int(idx.unique().numel()), "codebook_size": int(E.shape[0]),
            "worst_mismatch_gap_fp64": worst, "worst_mismatch_gap_ulps_of_2": round(worst / ULP_OF_2, 2),
            "near_tie_bound_fp64": VQ_NEAR_TIE_BOUND, "near_tie_bound_ulps_of_2": round(VQ_NEAR_TIE_BOUND / ULP_OF_2, 1),
            "fp64_argmin_held_by": sides, "audited_count": len(audits), "audited": audits[:16], "status": status,
            "boundary": "identical quantizer input h and codebook (one extra forward pass after the timed region, no optimizer step in between); "
                        "HIP indices vs the reference formula (fp32, torch CPU) on the host"}


def parity_rows(model, eng, cfg, xs, headline_precision, keep_second_engine=False):
    """h error / end-to-end code match / free-running reconstruction error of the headline engine, the other 16-bit format and the x3 instrument against the
    fp32 CPU oracle (the reference's arithmetic) on the images xs, at the model's CURRENT weights.  Returns (rows, second model, second engine or None)."""
    import torch
    from enhancing.utils.general import initialize_from_config
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vitvq_oracle as O
    weights = {k: v for k, v in model.state_dict().items() if not k.startswith("loss.")}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    P = {k: v.detach().float().cpu() for k, v in weights.items()}
    ocfg = dict(image_size=cfg.model.params.image_size, patch_size=cfg.model.params.patch_size, encoder=dict(cfg.model.params.encoder),
                decoder=dict(cfg.model.params.decoder), quantizer=dict(cfg.model.params.quantizer))
    with torch.no_grad():
        _, _, o_idx, o_h = O.encode(xs.cpu(), P, ocfg)
        o_xrec = O.forward(xs.cpu(), P, ocfg)[0]
    relerr = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    par = {}

    def parity_of(name, m_, e_, prec=None):
        h = m_.pre_quant_tokens(xs, precision=prec).cpu()
        codes = e_.encode_codes(xs, precision=prec).cpu().view(o_idx.shape)
        par[name] = {"h_rel_err": relerr(h, o_h), "code_match_end_to_end": float((codes == o_idx).float().mean())}

    other = "bf16" if headline_precision == "fp16" else "fp16"

    def xrec_of(name, m_, e_):     # free-running (the mode's own codes) and downstream of the ORACLE's codes (the decoder's arithmetic alone: north_star's clause)
        with torch.no_grad():
            par[name]["xrec_rel_err_free_running"] = relerr(e_.reconstruct(xs)[0].detach().float().cpu(), o_xrec)
            par[name]["xrec_rel_err_same_codes"] = relerr(m_.decode_codes(o_idx.to(xs.device)).detach().float().cpu().view(o_xrec.shape), o_xrec)

    parity_of(headline_precision, model, eng, prec=headline_precision)      # (explicit: under the bf16 engine encode_codes DEFAULTS to the x3 encoder)
    xrec_of(headline_precision, model, eng)
    m2 = initialize_from_config(cfg.model)
    m2.precision = other
    m2.load_state_dict(weights, strict=False)
    e2 = m2.engine
    parity_of(other, m2, e2, prec=other)
    xrec_of(other, m2, e2)
    mb, eb = (model, eng) if headline_precision == "bf16" else (m2, e2)      # the bf16 engine hosts the x3 instrument
    parity_of("x3", mb, eb, prec="x3")
    eb.encoder_precision = eb.decoder_precision = "x3"
    try:
        xrec_of("x3", mb, eb)
    finally:
        eb.encoder_precision = eb.decoder_precision = "bf16"
    if keep_second_engine:
        return par, m2, e2
    del m2, e2
    torch.cuda.empty_cache()
    return par, None, None


def parity_mode_block(model, eng, cfg, batches, B, lr, dev, headline_precision, par_at_init=None, steps_done=0):
    """north_star's "indices bit-exact / activations within 1e-3 of the reference fp32 path" priced per precision mode, outside the timed region:
    encode-only and training throughput, and the parity each mode reaches against the fp32 CPU oracle on a PARITY_IMAGES-image sample (h = the quantizer input,
    xrec downstream of the oracle's own run, end-to-end code match).  Modes: "fp16" (one MFMA pass, fp16 operands: the headline since round 6 and the
    reference's --use_amp dtype), "bf16" (one pass, bf16 operands: the round-1..5 headline), "x3" (three passes on split-bf16 operands, ~1e-5: the
    instrument; needs the bf16 engine) and the exact-fp32 engine mode.  The headline engine is measured in place; the other 16-bit engine is a second
    model with the same weights.  The parity rows are taken TWICE: at the constructor's weights before the first training step (par_at_init: the state of
    every parity test in tests/ and of tests/rounding_ablation.py) and here, after the warm-up + timed steps have moved the weights."""
    import torch
    from enhancing import _C
    from enhancing.utils.general import initialize_from_config

    def rate(fn, n_img, iters):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return round(iters * n_img / (time.perf_counter() - t0), 1)

    N_IT = 10          # timed iterations per mode after one warm-up call
    out = {"encode_only_images_per_s": {}, "train_images_per_s": {}, "timed_iterations": N_IT, "headline": headline_precision}
    x = batches[0]
    xs = x[:PARITY_IMAGES].contiguous()
    weights = {k: v for k, v in model.state_dict().items() if not k.startswith("loss.")}
    other = "bf16" if headline_precision == "fp16" else "fp16"

    def tstep_of(e_):
        def f():
            e_.forward_backward(x, w_l1=0.0, w_l2=1.0, codebook_weight=1.0)
            e_.optimizer_step(lr)
        return f

    # ---- parity first (the timed training steps below move the weights) ----
    par_now, m2, e2 = parity_rows(model, eng, cfg, xs, headline_precision, keep_second_engine=True)
    mb, eb = (model, eng) if headline_precision == "bf16" else (m2, e2)
    out["vs_fp32_cpu_oracle"] = par_at_init if par_at_init is not None else par_now
    out["vs_fp32_cpu_oracle_sample_images"] = int(xs.shape[0])
    out["vs_fp32_cpu_oracle_state"] = ("the constructor's weights (reference init), before the first training step" if par_at_init is not None
                                                else f"after {steps_done} training steps on the synthetic batches")
    if par_at_init is not None:
        out["vs_fp32_cpu_oracle_after_training_steps"] = dict(par_now, steps=steps_done, lr=lr)
    out["vs_fp32_cpu_oracle_note"] = ("h: relative Frobenius error of the quantizer input; xrec free-running: a flipped near-tie code moves a whole token of the "
                                      "reconstruction (the same-codes figure is tests/test_fp16_gpu.py / test_parity_base_gpu.py); xrec same codes: the decoder "
                                      "alone, fed the oracle's codes (north_star's activation clause for the decoder side); the second block is the same sample after the "
                                      "warm-up + timed AdamW steps on synthetic noise (code usage has collapsed to a handful of codes by then — a state no test pins; "
                                      "tests/rounding_ablation.py --init model holds the per-class table of the first block)")
    # ---- throughput ----
    out["encode_only_images_per_s"][headline_precision] = rate(lambda: eng.encode_codes(x, precision=headline_precision), B, N_IT)
    out["encode_only_images_per_s"][other] = rate(lambda: e2.encode_codes(x, precision=other), B, N_IT)
    out["encode_only_images_per_s"]["x3"] = rate(lambda: eb.encode_codes(x, precision="x3"), B, N_IT)
    out["train_images_per_s"][other] = rate(tstep_of(e2), B, N_IT)
    eb.encoder_precision = "x3"
    try:
        out["train_images_per_s"]["x3_encoder_forward"] = rate(tstep_of(eb), B, N_IT)
        eb.decoder_precision = "x3"
        out["train_images_per_s"]["x3_whole_forward"] = rate(tstep_of(eb), B, N_IT)
    finally:
        eb.encoder_precision = eb.decoder_precision = "bf16"
    del m2, e2
    torch.cuda.empty_cache()
    # the exact-fp32 engine mode (vector-ALU kernels, csrc/exact_f32.hip): the parity instrument, timed at a small batch
    try:
        m32 = initialize_from_config(cfg.model)
        m32.precision = "fp32"
        m32.load_state_dict({k: v for k, v in weights.items()}, strict=False)
        e32, b32 = m32.engine, 8
        x8 = x[:b32].contiguous()
        out["encode_only_images_per_s"]["fp32_exact_mode"] = rate(lambda: e32.encode_codes(x8), b32, 2)

        def tstep32():
            e32.forward_backward(x8, w_l1=0.0, w_l2=1.0, codebook_weight=1.0)
            e32.optimizer_step(lr)
        out["train_images_per_s"]["fp32_exact_mode"] = rate(tstep32, b32, 1)
        out["fp32_exact_mode_batch"] = b32
        del m32, e32
        torch.cuda.empty_cache()
    except Exception as ex:      # the block is a report, not the product: never lose the headline line to it
        out["fp32_exact_mode_error"] = repr(ex)[:200]
    out["note"] = ("fp16 / bf16 = every MFMA operand in that 16-bit format, one pass (csrc/common.h operand type tags); x3 = every operand of the forward as hi + lo "
                   "bf16 planes, products a_hi b_hi + a_lo b_hi + a_hi b_lo in the fp32 accumulator (csrc/x3.hip; bf16 engine only).  encode_codes defaults to the "
                   "engine's own single pass under fp16 and to x3 under bf16.")
    return out


def kernel_rooflines(ks: dict, steps_timed: int, ms_per_step: float, pmc: dict):
    """every timed kernel with a work model, priced against ITS roofline: bf16 MFMA 2.5 PF, exact-f32 MFMA 157.3 TF, HBM 8 TB/s.  `traffic` comes from
    a committed PMC pass of the SAME config and batch (profiles/pmc_step.json, written by tools/rocpd_summary.py pmc) and is null otherwise."""
    peak = {"flop": (MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", "mfma", 1e12), "flop_f32": (MFMA_F32_PEAK_TFLOPS, "TFLOP/s", "mfma_f32", 1e12),
            "byte": (HBM_PEAK_GBS, "GB/s", "hbm", 1e9)}
    out = {}
    for k, v in ks.items():
        pk, unit, bound, div = peak[v["unit"]]
        ach = v["work"] / (v["total_ms"] * 1e-3) / div
        sym = k.split(" (")[0]
        tr = None
        if pmc:
            if k.startswith("attn_bwd"):          # one timed call = one dQ launch + one dK/dV launch: their bytes add
                parts = [rec.get("hbm_bytes_per_launch", 0.0) for name, rec in pmc.items() if name.startswith("attn_bwd")]
                tr = sum(parts) if parts else None
            elif k.startswith("vq_forward"):
                tr = pmc.get("vq_nn_kernel", {}).get("hbm_bytes_per_launch")
            else:                                  # exact symbol, or the instantiations of a template timed under one name: launch-weighted mean
                hits = [rec for name, rec in pmc.items() if name == sym or name.startswith(sym + "<")]
                n = sum(r.get("launches", 0) for r in hits)
                if hits and n:
                    tr = sum(r.get("hbm_bytes_per_launch", 0.0) * r.get("launches", 0) for r in hits) / n
        out[k] = {"kernel": k, "bound": bound, "achieved": round(ach, 1), "peak": pk, "unit": unit, "frac": round(ach / pk, 4), "traffic": tr,
                  "launches_timed": v["launches"], "avg_launch_ms": round(v["avg_ms"], 4),
                  "share_of_step": round((v["total_ms"] / steps_timed) / ms_per_step, 4)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("ENH_BENCH_BATCH", "128")), help="images per GPU per step")
    ap.add_argument("--config", type=str, default="imagenet_vitvq_base")
    ap.add_argument("--precision", choices=["fp16", "bf16"], default=os.environ.get("ENH_PRECISION", "fp16"),
                    help="16-bit MFMA operand format of the product path: fp16 (the reference's --use_amp dtype; meets the 1e-3 parity clause in one pass; "
                         "loss-scaled backward) or bf16")
    ap.add_argument("--timer-every", type=int, default=4, help="per-launch HIP-event timing (the roofline / kernels blocks) on every n-th step of the timed region "
                                                              "(1 = every step: ~3600 events per step cost ~3 %% of it; the sampled steps are inside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the parity_mode block (x3 / fp32 throughput and parity beside the headline)")
    ap.add_argument("--graphs", action="store_true", help="replay the fused AE step from a captured HIP graph (for launch-bound small batches); the per-kernel "
                                                          "timing of the roofline block then comes from two extra eager steps after the timed region")
    ap.add_argument("--grad-bf16", action="store_true", help="all-reduce the gradient buckets as bf16 (half the xGMI bytes); default fp32")
    ap.add_argument("--comm-cus", type=int, default=0, help="N > 1: CUs conceded to RCCL's channel workgroups while gradient buckets are in flight (sizes the "
                                                             "split-K / LayerNorm-backward launches; also caps NCCL_MAX_NCHANNELS to it); 0 = no reservation [default]")
    ap.add_argument("--grad-algo", choices=["allreduce", "rs_ag"], default="allreduce",
                    help="per bucket: one RCCL all-reduce (default) or reduce-scatter + all-gather (SURVEY.md 8e's direct exchange); same sums")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher environment re-executes itself under torch.distributed.run, one rank per GPU (as main.py does;
    # reference main.py:54-57 hands the same job to Lightning's DDP plugin).  Under a launcher (RANK set) WORLD_SIZE must equal --gpus.
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        port = os.environ.get("MASTER_PORT")
        if not port:      # a free port (two concurrent bare runs on one box must not meet at rendezvous); MASTER_PORT overrides
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = str(sk.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    import torch
    import torch.distributed as dist
    from enhancing import _C
    from enhancing.engine.ddp import GradSync, init_process_group_from_env
    from enhancing.utils.general import get_config_from_file, initialize_from_config, set_seed

    if args.comm_cus > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(args.comm_cus))      # RCCL: one workgroup per channel
    rank, local_rank, world = init_process_group_from_env("nccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("ENH_FORCE_DEVICE") is not None:  # test hook: several ranks on one GPU (with ENH_DIST_BACKEND=gloo)
        local_rank = int(os.environ["ENH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    set_seed(0)  # identical init on every rank (then broadcast anyway, as DDP does)
    cfg = get_config_from_file(os.path.join(ROOT, "configs", args.config + ".yaml"))
    pw = float(cfg.model.params.loss.get("params", {}).get("perceptual_weight", 1.0))      # VQLPIPS' default is 1.0 (vqperceptual.py:25)
    lpips_info = None
    if pw != 0.0 and not os.environ.get("ENH_LPIPS_WEIGHTS"):
        os.environ["ENH_LPIPS_RANDOM_INIT"] = "1"      # explicit opt-in (enhancing/losses/lpips.py): timing the real topology on random weights; reported below
    model = initialize_from_config(cfg.model)
    model.precision = args.precision
    if pw != 0.0:
        pl = model.loss.perceptual_loss
        lpips_info = {"perceptual_weight": pw, "weights_loaded": bool(pl.weights_loaded), "random_init": bool(pl.random_init)}
    eng = model.engine
    if world > 1:
        eng.comm = GradSync(eng.store, compress="bf16" if args.grad_bf16 else None, algo=args.grad_algo, comm_cus=args.comm_cus)
        eng.comm.broadcast_parameters(0)
        eng.store.refresh_shadows()
    B, size = args.batch, cfg.model.params.image_size
    # synthetic ImageNet-shaped batches, distinct per rank (seed + rank), resident in HBM before timing
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    nbatch = 2
    low = torch.rand(nbatch, B, 3, size // 16, size // 16, device=dev, generator=g)
    batches = [(torch.nn.functional.interpolate(low[i], size=(size, size), mode="bilinear") +
                0.05 * torch.randn(B, 3, size, size, device=dev, generator=g)).clamp_(0, 1).contiguous() for i in range(nbatch)]
    lr = 4.5e-6

    adversarial = hasattr(model.loss, "discriminator")   # --config imagenet_vitvq_base_adv: the reference's two-optimizer protocol
    if adversarial:
        model.train()
        model.learning_rate = lr
        opts, _ = model.configure_optimizers()
        if world > 1 and len(opts) > 1:       # the discriminator's own DDP: broadcast + bucketed all-reduce behind ITS backward (engine/ddp.py AutogradGradSync)
            dist.broadcast(opts[1].store.p, 0)
            opts[1].attach_sync(model.loss.discriminator, compress="bf16" if args.grad_bf16 else None, algo=args.grad_algo)

    def step(i):
        if adversarial:   # per optimizer: training_step (forward + backward) then its AdamW step, as Lightning 1.5 drives vitvqgan.py:101-127
            batch = {"image": batches[i % nbatch]}
            loss0 = model.training_step(batch, i, 0)
            opts[0].step()
            model.training_step(batch, i, 1)
            opts[1].step()
            return {"loss": loss0}
        out = eng.forward_backward_graphed(batches[i % nbatch], w_l1=0.0, w_l2=1.0, codebook_weight=1.0)   # eager unless --graphs
        eng.optimizer_step(lr)
        return out

    use_graphs = args.graphs and world == 1      # (the two-optimizer protocol replays one graph per optimizer and host-side variant: vitvqgan.py _graphed_training_step)
    eng.use_graphs = use_graphs
    par_at_init = None
    if world == 1 and rank == 0 and args.config == "imagenet_vitvq_base" and not args.no_parity_mode and not args.no_cpu_baseline and not adversarial:
        try:       # parity of the three modes at the constructor's weights (the state every parity test pins), before any training step; outside the timed region
            par_at_init = parity_rows(model, eng, cfg, batches[0][:PARITY_IMAGES].contiguous(), args.precision)[0]
        except Exception as ex:
            print(f"[bench] parity at init skipped: {ex!r}", file=sys.stderr)
    for i in range(args.warmup):
        out = step(i)
    timer = _C.KernelTimer()
    every = max(1, min(args.timer_every, args.steps))
    timed_steps = [i for i in range(args.steps) if (not use_graphs) and i % every == every - 1]      # live per-launch timing on these steps of the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_last_step_ms = 0.0
    for i in range(args.steps):
        _C.TIMER = timer if i in timed_steps else None
        t_last_step_ms = time.perf_counter() * 1e3      # (host clock of engine/ddp.py's per-bucket issue log)
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _C.TIMER = None
    loss = float(out["loss"])
    if use_graphs:      # kernel table for the roofline block: the same launch sequence, eager, outside the timed region
        eng.use_graphs = False
        _C.TIMER = timer
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        _C.TIMER = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    comm_info = None
    if world > 1:
        # per-rank exposed communication (so that the first real multi-GPU run explains itself) + the bucket accounting the DDP tests insist on
        mine = eng.comm.comm_wait_ms(last=args.steps)
        mine.update(rank=rank, bytes_reduced_per_step=eng.comm.bytes_reduced / max(args.warmup + args.steps, 1), gap_elems=eng.comm.gap_elems)
        # last timed step: when each bucket was handed to the collective library, as an offset from the step's start on the host clock
        mine["bucket_issue_timeline_last_step"] = eng.comm.issue_timeline(t_last_step_ms)
        if adversarial and len(opts) > 1 and opts[1].comm is not None:      # the second optimizer's collectives, reported separately
            dsync = opts[1].comm
            mine["discriminator"] = dict(dsync.comm_wait_ms(last=args.steps), bytes_reduced_per_step=dsync.bytes_reduced / max(args.warmup + args.steps, 1),
                                         gap_elems=dsync.gap_elems, n_params=int(dsync.store.g.numel()), n_buckets=len(dsync.buckets),
                                         bucket_issue_timeline_last_step=dsync.issue_timeline(t_last_step_ms))      # same clock as the autoencoder's list above: interleave to see the queue order
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        comm_info = {"backend": dist.get_backend(), "algo": args.grad_algo, "comm_cus": args.comm_cus, "bucket_dtype": "bf16" if args.grad_bf16 else "fp32",
                     "n_params": int(eng.store.g.numel()), "per_rank": allr,
                     "note": "stream_ms = time the compute stream waited for the collectives after backward (HIP events), host_ms = host time in wait()"}
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    img_per_s = args.steps * B * world / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    ks = timer.summary()
    steps_timed = 2 if use_graphs else len(timed_steps)
    # PMC traffic only from a pass over the SAME config and batch (profiles/pmc_step.json: {"config":, "batch":, "kernels": {symbol: {...}}})
    pmc, pmc_path = None, os.path.join(ROOT, "profiles", "pmc_step.json")
    if os.path.exists(pmc_path):
        try:
            rec = json.load(open(pmc_path))
            if rec.get("config") == args.config and int(rec.get("batch", -1)) == B:
                pmc = rec.get("kernels", {})
        except Exception:
            pmc = None
    roofs = kernel_rooflines(ks, steps_timed, ms_per_step, pmc)
    dom = max(roofs, key=lambda k: roofs[k]["share_of_step"])          # dominant kernel over ALL timed kernels (GEMM, attention, conv, VQ, LN, AdamW)
    is_base = args.config == "imagenet_vitvq_base"
    res = {
        "metric": "images/sec ViT-VQGAN-base 256px stage-1 train; VQ argmin match-rate" if is_base else f"images/sec {args.config} 256px stage-1 train",
        "value": round(img_per_s, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": (f"{args.config}.yaml two-optimizer step exactly as the reference drives it: autoencoder fwd + bwd through the StyleGAN2 "
                                f"discriminator (L2 + codebook + 0.1*vanilla GAN) + AdamW, then forward again + discriminator fwd/bwd on real and fake "
                                f"(lazy R1 every 16 batches) + AdamW" + (f"; LPIPS weight {pw}" if pw else "; LPIPS weight 0") if adversarial else
                                f"{args.config}.yaml AE training step (1 fwd + 1 bwd + grad all-reduce + AdamW), loss = 1.0*L2 + 1.0*codebook "
                                f"(LPIPS/GAN weights 0)") + f", K=8192 x 32 l2-normalised codes, fp32 master weights, {args.precision} MFMA operands / fp32 accumulate" +
                               (f", dynamic loss scale (now {int(eng.loss_scale)}; device-side inf/nan step skip + GradScaler update rule)" if args.precision == "fp16" else ""),
                   "per_gpu_batch": B, "global_batch": B * world, "image": f"{size}x{size}", "parallelism": f"dp{world}",
                   "hip_graph_replay": bool(use_graphs),
                   "loss_network_operands": (model.loss.loss_operands(model.decoder.get_last_layer()) if (adversarial or pw) and hasattr(model.loss, "loss_operands")
                                             else None)},
        "final_loss": loss,
        "step_mfma_frac": round(img_per_s / world * STEP_TFLOP_PER_IMG_BASE / MFMA_BF16_PEAK_TFLOPS, 4) if is_base else None,
        "roofline": roofs[dom],
        "kernel_timing": {"steps_with_per_launch_hip_events": steps_timed, "of_timed_steps": args.steps,
                          "note": "launch durations (roofline / kernels) are measured live with HIP events on the launch stream, on every "
                                  f"{every}-th step inside the timed region" if not use_graphs else "two eager steps after the graph-replayed timed region"},
        "kernels": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"], 2), "bound": roofs[k]["bound"], "achieved": roofs[k]["achieved"],
                        "unit": roofs[k]["unit"], "frac": roofs[k]["frac"], "share_of_step": roofs[k]["share_of_step"], "traffic": roofs[k]["traffic"]}
                    for k, v in sorted(ks.items())},
    }
    if lpips_info is not None:
        res["config"]["lpips"] = lpips_info
    vqk = next((k for k in roofs if k.startswith("vq_forward")), None)
    if vqk is not None:      # north_star: "wavefront-reduced nearest-neighbour HIP kernel evidenced by rocprof HBM GB/s vs peak"; SURVEY §8(d): fraction of 157.3 TF
        r, v = roofs[vqk], ks[vqk]
        tokens = B * (size // cfg.model.params.patch_size) ** 2
        vq = {"tokens_per_launch": tokens, "tokens_per_s": round(tokens / (v["avg_ms"] * 1e-3), 0), "f32_tflops": r["achieved"],
              "frac_of_f32_mfma_peak": r["frac"], "avg_launch_ms": r["avg_launch_ms"],
              "algorithmic_bytes_per_launch": tokens * 264 + 8192 * 32 * 4, "hbm_bytes_per_launch_pmc": r["traffic"],
              "hbm_gbs_pmc": round(r["traffic"] / (v["avg_ms"] * 1e-3) / 1e9, 1) if r["traffic"] else None, "hbm_peak_gbs": HBM_PEAK_GBS,
              "note": "compute-bound by design (1986 FLOP/B): the [M, 8192] distance matrix never touches HBM, so GB/s is a small fraction of peak"}
        res["vq"] = vq
    if comm_info is not None:
        res["comm"] = comm_info
    if world == 1 and not args.no_cpu_baseline:
        # checker legs (oracle on the host cores): the argmin match-rate on ONE extra forward pass (no optimizer step between the kernel's argmin and the
        # snapshot of its inputs), then the CPU baseline
        if not adversarial:
            chk = eng.forward_backward(batches[0], w_l1=0.0, w_l2=1.0, codebook_weight=1.0)
            torch.cuda.synchronize()
            q = model.quantizer
            h_chk, E_chk = chk["h"].clone(), eng.store.w["quantizer.embedding.weight"].clone()
            mr = vq_match_rate(h_chk, chk["indices"].clone(), E_chk, q.depth, bool(q.use_residual))
            res["vq_match_rate"] = mr["value"]
            res["vq_match"] = mr
            # the same kernel on the same 131 072 quantizer inputs against a codebook with a TRAINED-LIKE usage spread (jittered l2-normalised rows of h):
            # synthetic training collapses usage to a handful of codes (distinct_codes above), which exercises the 8192-way argmin very little
            if not q.use_residual:
                gcb = torch.Generator(device=dev).manual_seed(4321)
                hn = torch.nn.functional.normalize(h_chk.view(-1, h_chk.shape[-1]).float(), dim=-1)
                pick = torch.randint(0, hn.shape[0], (E_chk.shape[0],), device=dev, generator=gcb)
                E_sp = torch.nn.functional.normalize(hn[pick] + 0.1 * torch.randn(E_chk.shape, device=dev, generator=gcb) / E_chk.shape[1] ** 0.5, dim=-1).contiguous()
                _, _, idx_sp, _ = _C.vq_forward(h_chk.view(-1, h_chk.shape[-1]).contiguous(), E_sp, float(getattr(q, "beta", 0.25)), 1, True, want_bf16=False)
                torch.cuda.synchronize()
                ms = vq_match_rate(h_chk, idx_sp, E_sp, 1, False)
                ms["codebook"] = "jittered l2-normalised rows of the same h (trained-like spread); same kernel, same inputs"
                res["vq_match_spread"] = ms
                # headline match-rate = the leg that exercises the 8192-way argmin (~1000 distinct codes); the training-codebook leg stays in "vq_match"
                res["vq_match_rate"] = ms["value"]
                res["vq_match_rate_source"] = "vq_match_spread"
                # both legs under explicit names as well (the bare key changed meaning between rounds 3 and 4; rounds 1-3 reported the first)
                res["vq_match_rate_training_codebook"] = mr["value"]
                res["vq_match_rate_spread_codebook"] = ms["value"]
            if is_base and not args.no_parity_mode:
                res["parity_mode"] = parity_mode_block(model, eng, cfg, batches, B, lr, dev, args.precision, par_at_init, args.warmup + args.steps)
                res["parity_mode"]["train_images_per_s"][f"{args.precision} (headline)"] = round(img_per_s, 1)
        res["cpu_baseline"] = cpu_baseline()
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if "FAIL" in (res.get("vq_match", {}).get("status"), res.get("vq_match_spread", {}).get("status")):
        sys.exit(3)      # an index mismatch that is NOT an fp32 near-tie: the line above says which token


if __name__ == "__main__":
    main()
