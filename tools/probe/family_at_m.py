"""Per-role GEMM time under the planner's family and under the 128 x 128 family (enh_gemm_set_kernel(3)) at a given token count: is the 256 x 256 kernel the right
choice when its tile count leaves a ragged second round (288 / 384 tiles on 256 CUs at 8 images)?   python tools/probe/family_at_m.py [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

dev = torch.device("cuda")
F16 = torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = _C.lib()


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def role(name, N, K, tb, mode):
    a = (torch.randn(M, K, device=dev) * 0.5).to(F16)
    b = (torch.randn((K, N) if tb else (N, K), device=dev) * 0.03).to(F16)
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    aux = torch.tanh(torch.randn(M, N, device=dev)).to(F16)
    o16, o32 = torch.empty(M, N, device=dev, dtype=F16), torch.empty(M, N, device=dev)
    fn = {"plain16": lambda: _C.mm(a, b, M, N, K, o16, trans_b=tb),
          "tanh": lambda: _C.mm(a, b, M, N, K, o16, trans_b=tb, bias=bias, act=_C.ACT_TANH),
          "dtanh": lambda: _C.mm(a, b, M, N, K, o16, trans_b=tb, act=_C.ACT_DTANH, aux=aux),
          "bias_res": lambda: _C.mm(a, b, M, N, K, o32, trans_b=tb, bias=bias, res=res, res_rows=M),
          "f32": lambda: _C.mm(a, b, M, N, K, o32, trans_b=tb)}[mode]
    out = []
    for fam in (-1, 3, 7):
        assert L.enh_gemm_set_kernel(fam) == 0
        try:
            out.append(timed(fn))
        except RuntimeError:
            out.append(float("nan"))
    L.enh_gemm_set_kernel(-1)
    fl = 2.0 * M * N * K
    print(f"{name:22s} N={N:5d} K={K:5d} tb={int(tb)} {mode:8s}: planner {out[0]:7.1f} us ({fl / out[0] / 1e6:6.1f} TF/s) | 128x128 {out[1]:7.1f} us | 256x256 one-tile {out[2]:7.1f} us", flush=True)


if __name__ == "__main__":
    print(f"M = {M}")
    role("qkv fwd", 2304, 768, False, "plain16")
    role("fc1 + tanh", 3072, 768, False, "tanh")
    role("proj fwd", 768, 768, False, "bias_res")
    role("fc2 fwd", 768, 3072, False, "bias_res")
    role("dgrad fc2 (tanh')", 3072, 768, True, "dtanh")
    role("dgrad fc1", 768, 3072, True, "plain16")
    role("dgrad proj", 768, 768, True, "plain16")
    role("dgrad qkv", 768, 2304, True, "plain16")
