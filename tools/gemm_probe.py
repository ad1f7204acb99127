"""Dev probe (GPU box): our GEMM kernels vs torch.matmul (hipBLASLt) on the step's shapes, at two M, to separate kernel-structure limits from
HBM-streaming limits.  torch.matmul is a MEASURING STICK here only — the product path never calls it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

dev = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bf(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16)


def case(name, m, n, k, ta=False, tb=False, out="bf16", acc=False):
    a = bf(k, m) if ta else bf(m, k)
    b = bf(k, n) if tb else bf(n, k)
    o32 = torch.zeros(m, n, device=dev) if out == "f32" else None
    o16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if out == "bf16" else None
    t = timeit(lambda: _C.gemm(a, b, m, n, k, trans_a=ta, trans_b=tb, accumulate=acc, out_f32=o32, out_bf16=o16))
    A = a.t() if ta else a
    Bm = b if tb else b.t()
    ob = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    t2 = timeit(lambda: torch.matmul(A, Bm, out=ob))
    fl = 2 * m * n * k
    var = _C.lib().enh_gemm_h16_variant(int(ta), int(tb), m, n, k).decode().replace("gemm_", "").replace("_kernel", "")
    print(f"{name:12s} M={m:7d} N={n:5d} K={k:7d} ours[{var:5s}] {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF | hipblaslt(bf16 out) {t2*1e3:7.3f} ms {fl/t2/1e12:7.1f} TF", flush=True)


DIM, MLP = 768, 3072
for B in (16, 128):
    M = B * 1024
    print(f"--- tokens {M}")
    case("fwd qkv", M, 3 * DIM, DIM)
    case("fwd fc1", M, MLP, DIM)
    case("fwd fc2", M, DIM, MLP, out="f32")
    case("fwd fc2 b16", M, DIM, MLP, out="bf16")
    case("fwd out", M, DIM, DIM, out="f32")
    case("dgrad qkv", M, DIM, 3 * DIM, tb=True, out="f32")
    case("dgrad fc2", M, MLP, DIM, tb=True)
    case("dgrad fc1", M, DIM, MLP, tb=True, out="f32")
    case("wgrad qkv", 3 * DIM, DIM, M, ta=True, tb=True, out="f32", acc=True)
    case("wgrad fc1", MLP, DIM, M, ta=True, tb=True, out="f32", acc=True)
    case("wgrad fc2", DIM, MLP, M, ta=True, tb=True, out="f32", acc=True)
print("--- square")
case("4096^3", 4096, 4096, 4096)
case("8192^3", 8192, 8192, 8192)
