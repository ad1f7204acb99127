"""Where do the GEMM's operand requests LAND?  ("change where the requests land, not when".)
Two experiments on the twelve GEMM roles of the base training step at B images per GPU, interleaved rounds on one box:

 (a) leading dimensions: every activation operand / output dense (row stride 1536 / 4608 / 6144 B = 12 / 36 / 48 128-byte lines) against the same
     call on buffers whose rows are padded by 64 elements (13 / 37 / 49 lines: odd, no common factor with any power-of-two channel interleave);
 (b) tile order of the forward / input-gradient forms (enh_debug_gemm_order): which patch of tiles the 32 workgroups of an XCD have in flight.

GPU box:  python tools/gemm_ld_lab.py [batch]        (LAB_ROUNDS, LAB_PAD env)"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROUNDS = int(os.environ.get("LAB_ROUNDS", "5"))
PAD = int(os.environ.get("LAB_PAD", "64"))
N_TOK, DIM, MLP = 1024, 768, 3072
M = B * N_TOK
dev = "cuda"
L = _C.lib()


def buf(rows, cols, pad, dtype=torch.bfloat16, scale=0.5, fn=None):
    """[rows, cols + pad] contiguous; the GEMM sees the first `cols` columns through its leading dimension"""
    t = torch.randn(rows, cols + pad, device=dev) * scale
    if fn is not None:
        t = fn(t)
    return t.to(dtype)


def make_case(kind, m, n, k, pad):
    """pad applies to ACTIVATION-side tensors only (A, outputs, saved tanh, residual): weights stay dense as in the engine"""
    if kind == "wgrad":      # dW[m][n] = sum_tokens dY[tok][m] X[tok][n]: both operands contraction-major, k = tokens
        a = buf(k, m, pad); b = buf(k, n, pad)
        c = torch.zeros(m, n, device=dev)
        return (lambda: _C.gemm(a, b, m, n, k, trans_a=True, trans_b=True, accumulate=True, out_f32=c, lda=m + pad, ldb=n + pad)), c
    tb = kind.startswith("dgrad")
    a = buf(m, k, pad)
    b = buf(k, n, 0, scale=0.1) if tb else buf(n, k, 0, scale=0.1)
    kw = dict(trans_b=tb, lda=k + pad, ldc=n + pad)
    if kind in ("fwd", "dgrad"):
        out = torch.empty(m, n + pad, dtype=torch.bfloat16, device=dev); kw["out_bf16"] = out
    elif kind == "fwd_tanh":
        out = torch.empty(m, n + pad, dtype=torch.bfloat16, device=dev); kw.update(out_bf16=out, bias=torch.randn(n, device=dev), act=_C.ACT_TANH)
    elif kind == "fwd_res":
        out = torch.empty(m, n + pad, device=dev); kw.update(out_f32=out, bias=torch.randn(n, device=dev), res=torch.randn(m, n + pad, device=dev), res_rows=m)
    elif kind == "dgrad_dtanh":
        out = torch.empty(m, n + pad, dtype=torch.bfloat16, device=dev)
        kw.update(out_bf16=out, act=_C.ACT_DTANH, aux=buf(m, n, pad, fn=torch.tanh))
    else:
        raise ValueError(kind)
    return (lambda: _C.gemm(a, b, m, n, k, **kw)), out


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


ROLES = [
    ("fwd qkv -> bf16", "fwd", M, 3 * DIM, DIM),
    ("fwd fc1 +bias+tanh -> bf16", "fwd_tanh", M, MLP, DIM),
    ("fwd fc2 +bias+res -> f32", "fwd_res", M, DIM, MLP),
    ("fwd out +bias+res -> f32", "fwd_res", M, DIM, DIM),
    ("dgrad qkv -> bf16", "dgrad", M, DIM, 3 * DIM),
    ("dgrad fc1 -> bf16", "dgrad", M, DIM, MLP),
    ("dgrad out -> bf16", "dgrad", M, DIM, DIM),
    ("dgrad fc2 * dtanh -> bf16", "dgrad_dtanh", M, MLP, DIM),
    ("wgrad qkv", "wgrad", 3 * DIM, DIM, M),
    ("wgrad out", "wgrad", DIM, DIM, M),
    ("wgrad fc1", "wgrad", MLP, DIM, M),
    ("wgrad fc2", "wgrad", DIM, MLP, M),
]


def exp_ld(iters=10):
    print(f"== (a) leading dimensions, batch {B}: dense vs rows padded by {PAD} elements; min / median ms over {ROUNDS} interleaved rounds, TF/s at the minimum")
    tot = {0: 0.0, PAD: 0.0}
    for name, kind, m, n, k in ROLES:
        cases = {p: make_case(kind, m, n, k, p)[0] for p in (0, PAD)}
        times = {p: [] for p in cases}
        for p, fn in cases.items():
            fn(); fn()
        torch.cuda.synchronize()
        for _ in range(ROUNDS):
            for p, fn in cases.items():
                times[p].append(timed(fn, iters))
        fl = 2.0 * m * n * k
        cells = []
        for p in cases:
            mn, md = min(times[p]), statistics.median(times[p])
            tot[p] += md
            cells.append(f"{'dense ' if p == 0 else 'padded'} {mn:6.3f}/{md:6.3f} ms {fl / mn / 1e9:5.0f} TF/s")
        d, q = statistics.median(times[0]), statistics.median(times[PAD])
        print(f"{name:28s} " + " | ".join(cells) + f" | padded/dense {q / d:5.3f}", flush=True)
        del cases
        torch.cuda.empty_cache()
    print(f"sum of medians (one launch per role): dense {tot[0]:.3f} ms, padded {tot[PAD]:.3f} ms, ratio {tot[PAD] / tot[0]:.4f}", flush=True)


ORDERS = [(0, 0), (8, 0), (4, 0), (16, 0), (32, 0), (2, 0), (8, 1), (4, 1), (16, 1), (64, 1)]     # (0, 0) = the library's per-shape default


def exp_order(iters=10):
    print(f"== (b) tile order of the forward / input-gradient forms, batch {B}: (grp_rows, col_fast); median ms over {ROUNDS} interleaved rounds "
          f"[(8, 0) = shipped: 8 x 4 patch per XCD; (g, 1): columns fastest = 32/nbn rows x all nbn column tiles]")
    print(f"{'role':28s} " + " ".join(f"{str(o):>9s}" for o in ORDERS))
    tot = {o: 0.0 for o in ORDERS}
    for name, kind, m, n, k in ROLES[:8]:
        fn, out = make_case(kind, m, n, k, 0)
        L.enh_debug_gemm_order(8, 0); fn(); torch.cuda.synchronize(); ref = out.clone()
        times = {o: [] for o in ORDERS}
        for o in ORDERS:
            L.enh_debug_gemm_order(*o); out.zero_(); fn(); fn(); torch.cuda.synchronize()
            assert torch.equal(out, ref), (name, o)
        for _ in range(ROUNDS):
            for o in ORDERS:
                L.enh_debug_gemm_order(*o)
                times[o].append(timed(fn, iters))
        for o in ORDERS:
            tot[o] += statistics.median(times[o])
        print(f"{name:28s} " + " ".join(f"{statistics.median(times[o]):9.4f}" for o in ORDERS), flush=True)
        del fn, out, ref
        torch.cuda.empty_cache()
    L.enh_debug_gemm_order(0, 0)
    print(f"{'sum':28s} " + " ".join(f"{tot[o]:9.4f}" for o in ORDERS), flush=True)


if __name__ == "__main__":
    which = os.environ.get("LAB_EXP", "ld,order").split(",")
    if "ld" in which:
        exp_ld()
    if "order" in which:
        exp_order()
