"""The split-K weight-gradient loop against two measurement-only forms of itself (enh_debug_gemm_lab): plain 16-byte fragment reads instead of the
transposing 8-byte ones, and no fragment reads at all.  GPU box: python tools/wgrad_lab.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

K = 131072
for name, M, N in (("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16)      # dY [tokens][out features]
    b = torch.randn(K, N, device="cuda").to(torch.bfloat16)      # X  [tokens][in features]
    c = torch.zeros(M, N, device="cuda")
    row = []
    for lab in (0, 1, 2, 3, 4, 5, 8, 0):
        _C.lib().enh_debug_gemm_lab(lab)
        t = timeit(lambda: _C.mm(a, b, M, N, K, c, trans_a=True, trans_b=True, accumulate=True))
        row.append(f"lab {lab}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:6.0f} TF/s")
    _C.lib().enh_debug_gemm_lab(0)
    print(f"{name:4s} [{M} x {N}] K = {K}: " + " | ".join(row))
