"""GEMM family lab: the persistent 256 x 256 kernels (family 8: w256p; family 9: w256r where it serves) against w256 (family 7) on the forms the
training step launches.
Dev tool, run on the GPU box:  python tools/gemm_p_lab.py [batch]   — bitwise comparison of every output + wall time in INTERLEAVED rounds
(the chip re-clocks by +-10 % between back-to-back measurements: single timings are not evidence)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FAMS = [(int(x), 0) for x in os.environ.get("LAB_FAMS", "7,8,9").split(",")]
ROUNDS = int(os.environ.get("LAB_ROUNDS", "4"))
N_TOK, DIM, MLP = 1024, 768, 3072
M = B * N_TOK
dev = "cuda"
L = _C.lib()


def set_family(f):
    fam = f[0] if isinstance(f, tuple) else f
    if L.enh_gemm_set_kernel(fam) != 0:
        raise RuntimeError(L.enh_last_error().decode())


def bf(*shape, scale=0.5):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def make_case(kind, m, n, k):
    """-> (callable running the GEMM into fresh outputs, output tensor getter)"""
    tb = kind.startswith("dgrad")
    a = bf(m, k)
    b = bf(k, n) if tb else bf(n, k)
    kw = dict(trans_b=tb)
    if kind in ("fwd", "dgrad"):
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev); kw["out_bf16"] = out
    elif kind == "fwd_f32":
        out = torch.empty(m, n, device=dev); kw["out_f32"] = out
    elif kind == "fwd_tanh":
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev); kw.update(out_bf16=out, bias=torch.randn(n, device=dev), act=_C.ACT_TANH)
    elif kind == "fwd_res":
        out = torch.empty(m, n, device=dev); kw.update(out_f32=out, bias=torch.randn(n, device=dev), res=torch.randn(m, n, device=dev), res_rows=m)
    elif kind == "dgrad_dtanh":
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        kw.update(out_bf16=out, act=_C.ACT_DTANH, aux=torch.tanh(bf(m, n).float()).to(torch.bfloat16))
    else:
        raise ValueError(kind)
    return (lambda: _C.gemm(a, b, m, n, k, **kw)), out


def check(kind, m, n, k):
    fn, out = make_case(kind, m, n, k)
    set_family(7); out.zero_(); fn(); torch.cuda.synchronize(); ref = out.clone()
    res = []
    for f in FAMS:
        if f[0] == 7 or f[1]:
            continue
        set_family(f); out.fill_(7.0); fn(); torch.cuda.synchronize()
        same = torch.equal(out, ref)
        bad = 0 if same else int((out != ref).sum())
        res.append(f"fam{f[0]}: {'bitwise equal' if same else f'MISMATCH {bad} elements, max |d| {(out.float() - ref.float()).abs().max().item():.3e}'}")
    print(f"check {kind:12s} M={m:6d} N={n:5d} K={k:5d}  " + "  ".join(res), flush=True)
    return all("equal" in r for r in res)


def bench(name, kind, m, n, k, iters=10):
    fn, out = make_case(kind, m, n, k)
    times = {f: [] for f in FAMS}
    for f in FAMS:
        set_family(f)
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for f in FAMS:
            set_family(f)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            times[f].append(s.elapsed_time(e) / iters)
    fl = 2.0 * m * n * k
    cells = []
    for f in FAMS:
        mn, md = min(times[f]), statistics.median(times[f])
        cells.append(f"fam{f[0]} {mn:6.3f}/{md:6.3f} {fl / mn / 1e9:5.0f}")
    print(f"{name:28s} " + " | ".join(cells), flush=True)


ok = True
# correctness on awkward shapes first: fewer tiles than CUs, a ragged last round, odd stage counts, every mode and both B layouts
for kind in ("fwd", "dgrad", "fwd_tanh", "dgrad_dtanh", "fwd_res", "fwd_f32"):
    for (m, n, k) in ((1024, 768, 192), (256 * 100, 768, 320), (256 * 37, 2304, 448), (256 * 50, 768, 384), (8192, 3072, 768)):
        ok &= check(kind, m, n, k)
print("ALL EQUAL" if ok else "SOME MISMATCH", flush=True)

print(f"batch {B}: family  min / median ms over {ROUNDS} interleaved rounds, TF/s at the minimum")
bench("fwd qkv -> bf16", "fwd", M, 3 * DIM, DIM)
bench("fwd fc1 +bias+tanh -> bf16", "fwd_tanh", M, MLP, DIM)
bench("fwd fc2 +bias+res -> f32", "fwd_res", M, DIM, MLP)
bench("fwd out +bias+res -> f32", "fwd_res", M, DIM, DIM)
bench("dgrad qkv -> bf16", "dgrad", M, DIM, 3 * DIM)
bench("dgrad fc1 -> bf16", "dgrad", M, DIM, MLP)
bench("dgrad out -> bf16", "dgrad", M, DIM, DIM)
bench("dgrad fc2 * dtanh -> bf16", "dgrad_dtanh", M, MLP, DIM)
set_family(-1)
