"""Dev probe: the attention kernels alone at the bench shape (B = 128, H = 12, N = 1024): timing, or a target for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C
B, N, H = int(os.environ.get("MB_BATCH", "128")), 1024, 12
dev = "cuda"
qkv = (torch.randn(B, N, 3 * H * 64, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty(B, N, H * 64, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, N, device=dev)
do = (torch.randn(B, N, H * 64, device=dev) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, device=dev)
def timeit(fn, iters=int(os.environ.get("ITERS", "10")), warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
fl = 4 * B * H * N * N * 64
tf = timeit(lambda: _C.attention_forward(qkv, B, N, H, 0.125, out, lse))
tb = timeit(lambda: _C.attention_backward(qkv, out, do, lse, B, N, H, 0.125, dqkv, delta))
print(f"attention fwd {tf*1e3:7.3f} ms {fl/tf/1e12:7.1f} TF/s | bwd {tb*1e3:7.3f} ms {2.5*fl/tb/1e12:7.1f} TF/s (algorithmic) | variant env: V1={os.environ.get('ENH_ATTN_V1')}")
