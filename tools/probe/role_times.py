"""Per-role GEMM times at the headline shapes (M = 131072 token rows), one launch per role in rotation over three buffer sets (so no launch finds its own
operands in the Infinity Cache), fp16 operands: which of the bias + residual launches (proj K = 768, fc2 K = 3072) carries the role's 0.36.
   python tools/probe/role_times.py [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

dev = torch.device("cuda")
F16 = torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
SETS = 3


def role(name, N, K, mode):
    sets = []
    for _ in range(SETS):
        a = (torch.randn(M, K, device=dev) * 0.5).to(F16)
        w = (torch.randn(N, K, device=dev) * 0.03).to(F16)
        bias = torch.randn(N, device=dev)
        if mode == "bias_res":
            res = torch.randn(M, N, device=dev)
            out = torch.empty(M, N, device=dev)
            fn = (lambda a=a, w=w, bias=bias, res=res, out=out: _C.mm(a, w, M, N, K, out, bias=bias, res=res, res_rows=M))
        elif mode == "tanh":
            out = torch.empty(M, N, device=dev, dtype=F16)
            fn = (lambda a=a, w=w, bias=bias, out=out: _C.mm(a, w, M, N, K, out, bias=bias, act=_C.ACT_TANH))
        elif mode == "plain16":
            out = torch.empty(M, N, device=dev, dtype=F16)
            fn = (lambda a=a, w=w, out=out: _C.mm(a, w, M, N, K, out))
        else:
            out = torch.empty(M, N, device=dev)
            fn = (lambda a=a, w=w, out=out: _C.mm(a, w, M, N, K, out))
        sets.append(fn)
    for f in sets:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 12
    e0.record()
    for i in range(reps):
        sets[i % SETS]()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * M * N * K
    by = 2.0 * M * K + 2.0 * N * K + {"bias_res": 8.0, "tanh": 2.0, "plain16": 2.0, "f32": 4.0}[mode] * M * N
    print(f"{name:28s} N={N:5d} K={K:5d} {mode:9s}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   algorithmic {by / 1e6:7.1f} MB -> {by / us / 1e3:6.2f} TB/s   "
          f"(MFMA floor at 1250 TF/s {fl / 1250e6:6.1f} us, HBM floor at 6.3 TB/s {by / 6.3e6:6.1f} us)", flush=True)


if __name__ == "__main__":
    role("proj fwd (bias + residual)", 768, 768, "bias_res")
    role("fc2 fwd (bias + residual)", 768, 3072, "bias_res")
    role("qkv fwd", 2304, 768, "plain16")
    role("fc1 + tanh", 3072, 768, "tanh")
    role("f32 out K=768", 768, 768, "f32")
    role("f32 out K=3072", 768, 3072, "f32")
