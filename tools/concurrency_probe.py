"""Dev probe: does running a store-heavy GEMM (dgrad / forward, 0.6-0.8 GB of output) CONCURRENTLY with a compute-heavy one (split-K weight gradient, tiny output)
on two streams beat running them back to back?  (Each w256 workgroup owns a CU, so the two kernels split the CUs between them.)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402
dev = "cuda"
M, DIM, MLP = 131072, 768, 3072
def bf(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
g16, hid, w2 = bf(M, DIM), bf(M, MLP), bf(DIM, MLP)
dhid = torch.empty(M, MLP, dtype=torch.bfloat16, device=dev)
gw2 = torch.zeros(DIM, MLP, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def wgrad(): _C.gemm(g16, hid, DIM, MLP, M, trans_a=True, trans_b=True, accumulate=True, out_f32=gw2)
def dgrad(): _C.gemm(g16, w2, M, MLP, DIM, trans_b=True, act=_C.ACT_DTANH, aux=hid, out_bf16=dhid)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def seq(): wgrad(); dgrad()
def conc():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa): wgrad()
    with torch.cuda.stream(sb): dgrad()
    cur.wait_stream(sa); cur.wait_stream(sb)
_C._gemm_workspace(torch.device("cuda", 0), 1 << 28)
print(f"wgrad alone {timeit(wgrad):.3f} ms   dgrad(dtanh) alone {timeit(dgrad):.3f} ms   sequential {timeit(seq):.3f} ms   concurrent {timeit(conc):.3f} ms")
