"""LayerNorm backward at the BASELINE shape with DISTINCT buffers (real HBM traffic) — dev tool, run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

M, D = int(os.environ.get("MB_BATCH", "128")) * 1024, 768
dev = "cuda"
x, dy, dres = (torch.randn(M, D, device=dev) for _ in range(3))
dy16 = dy.to(torch.bfloat16)
w = torch.ones(D, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
dx, dx16 = torch.empty_like(x), torch.empty_like(dy16)
dw, db, dxs = (torch.zeros(D, device=dev) for _ in range(3))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ENH_LN"))
for name, g, nbytes in (("f32 dy", dy, 18), ("bf16 dy", dy16, 16)):
    t = timeit(lambda: _C.layernorm_backward(g, x, w, mean, rstd, dres, dx, dx16, dw, db, dxs))
    print(f"ln_bwd {name:8s} [{tag}] {t*1e6:8.1f} us  {M*D*nbytes/t/1e12:5.2f} TB/s", flush=True)
hb = (torch.randn(M, 3072, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(3072, device=dev)
t = timeit(lambda: _C.colsum(hb, M, 3072, o))
print(f"colsum [M,3072] bf16 {t*1e6:8.1f} us  {M*3072*2/t/1e12:5.2f} TB/s", flush=True)
