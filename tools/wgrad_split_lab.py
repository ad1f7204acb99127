"""Split-K plan of the weight gradients: the planner's slice count against slice counts that give every XCD WHOLE slices (8 x tiles-per-slice workgroups:
workgroup b runs on XCD b % 8 and the (slice, row, column) line is cut into 8 equal runs).  GPU box: python tools/wgrad_split_lab.py"""
import os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C
L = _C.lib()
K = int(os.environ.get("LAB_TOKENS", "131072"))

def timed(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

CANDS = [int(x) for x in os.environ["LAB_SPLITS"].split(",")] if os.environ.get("LAB_SPLITS") else None
for name, M, N, cands in (("qkv", 2304, 768, (0, 8, 9, 16)), ("out", 768, 768, (0, 24, 28, 16, 32)), ("fc1", 3072, 768, (0, 7, 8, 6)), ("fc2", 768, 3072, (0, 7, 8, 6))):
    cands = tuple(CANDS) if CANDS else cands
    a = torch.randn(K, M, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    c = torch.zeros(M, N, device="cuda")
    fn = lambda: _C.mm(a, b, M, N, K, c, trans_a=True, trans_b=True, accumulate=True)
    ref = None
    times = {s: [] for s in cands}
    for s in cands:
        L.enh_debug_gemm_splits(s); c.zero_(); fn(); torch.cuda.synchronize()
        if ref is None: ref = c.clone()
        err = float((c - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (name, s, err)
        fn()
    for _ in range(5):
        for s in cands:
            L.enh_debug_gemm_splits(s); times[s].append(timed(fn))
    tiles = (M // 256) * (N // 256)
    print(f"wgrad {name} [{M} x {N}], {tiles} tiles per slice: " + " | ".join(f"splits {s or 'auto'}: {statistics.median(times[s])*1e3:6.1f} us {2.0*M*N*K/statistics.median(times[s])/1e9:5.0f} TF/s" for s in cands), flush=True)
    del a, b, c
L.enh_debug_gemm_splits(0)
