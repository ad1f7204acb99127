"""forward split-K at small M: the same call with and without a workspace (enh_gemm_h16_ws: a null workspace = the unsplit plan).  python tools/probe/split_fwd.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

dev = torch.device("cuda")
F16 = torch.float16


def case(M, N, K, tb, epi):
    a = (torch.randn(M, K, device=dev) * 0.5).to(F16)
    b = (torch.randn((K, N) if tb else (N, K), device=dev) * 0.03).to(F16)
    bias, res, out = torch.randn(N, device=dev), torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
    kw = dict(bias=bias, res=res, res_rows=M) if epi else {}
    L = _C.lib()
    want = L.enh_gemm_h16_workspace_bytes(0, int(tb), M, N, K)
    ws = torch.empty(max(want, 16), dtype=torch.uint8, device=dev)

    def run(use_ws):
        saved = _C._gemm_workspace
        if not use_ws:
            _C._gemm_workspace = lambda d, n: None
        try:
            _C.gemm(a, b, M, N, K, trans_b=tb, out_f32=out, **kw)
        finally:
            _C._gemm_workspace = saved

    res_ = []
    for use_ws in (False, True):
        for _ in range(3):
            run(use_ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run(use_ws)
        e1.record()
        torch.cuda.synchronize()
        res_.append(e0.elapsed_time(e1) / 30 * 1e3)
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d} tb={int(tb)} {'bias+res' if epi else 'plain   '}: unsplit {res_[0]:7.1f} us ({fl / res_[0] / 1e6:6.1f} TF/s) | split ({want / (M * N * 4):.0f} slices) {res_[1]:7.1f} us "
          f"({fl / res_[1] / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    for (M, N, K, tb, epi) in [(2048, 1280, 5120, False, True), (2048, 1280, 5120, True, False), (2048, 1280, 3840, True, False), (2048, 1280, 1280, False, True),
                               (4096, 768, 3072, False, True), (4096, 768, 2304, True, False), (2048, 768, 3072, False, True), (1024, 1280, 5120, False, True)]:
        case(M, N, K, tb, epi)
