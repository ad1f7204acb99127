"""Do two GEMM launches on two HIP streams share the chip?  (round 6, small-batch question: can the weight-gradient GEMMs of a backward block run in the
CUs the token-gradient chain leaves idle at 8 / 2 images per GPU.)   python tools/probe/stream_overlap.py
For each pair (X on stream 1, Y on stream 2): time of X alone, Y alone, both back to back on one stream, both on two streams; 40 repetitions, HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

dev = torch.device("cuda")
F16 = torch.float16


def gemm_case(M, N, K, trans_a=False, trans_b=True, out_f32=True, accumulate=False):
    a = torch.randn((K, M) if trans_a else (M, K), device=dev).to(F16)      # _C.mm: C = A B^T; trans_a: A stored [K, M]; trans_b: B stored [K, N]
    b = torch.randn((K, N) if trans_b else (N, K), device=dev).to(F16)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if out_f32 else F16)
    return lambda: _C.mm(a, b, M, N, K, out, trans_a=trans_a, trans_b=trans_b, accumulate=accumulate)


def timed(fn, reps=40):
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


def pair(name, fx, fy):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def both_two_streams():
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1):
            fx()
        with torch.cuda.stream(s2):
            fy()
        main.wait_stream(s1); main.wait_stream(s2)

    def serial():
        fx(); fy()

    tx, ty, ts, tp = timed(fx), timed(fy), timed(serial), timed(both_two_streams)
    print(f"{name}: X {tx:7.1f} us | Y {ty:7.1f} us | one stream {ts:7.1f} us | two streams {tp:7.1f} us  (sum {tx + ty:7.1f}, max {max(tx, ty):7.1f})", flush=True)


if __name__ == "__main__":
    # base at 8 images (M = 8192): fc2 forward (96 tiles of 256^2 / 384 of 128^2) beside the fc1 weight gradient (split-K, fills the chip)
    pair("base  M=8192  fc2-fwd N=768 K=3072  ||  wgrad 768x3072 K=8192", gemm_case(8192, 768, 3072, trans_b=False), gemm_case(768, 3072, 8192, trans_a=True, trans_b=True, accumulate=True))
    pair("base  M=8192  qkv-dgrad N=768 K=2304 ||  wgrad 2304x768 K=8192", gemm_case(8192, 768, 2304), gemm_case(2304, 768, 8192, trans_a=True, trans_b=True, accumulate=True))
    pair("base  M=8192  proj N=768 K=768       ||  proj N=768 K=768", gemm_case(8192, 768, 768, trans_b=False), gemm_case(8192, 768, 768, trans_b=False))
    # large at 2 images (M = 2048)
    pair("large M=2048  fc2-fwd N=1280 K=5120 ||  wgrad 1280x5120 K=2048", gemm_case(2048, 1280, 5120, trans_b=False), gemm_case(1280, 5120, 2048, trans_a=True, trans_b=True, accumulate=True))
    pair("large M=2048  fc1 N=5120 K=1280     ||  wgrad 5120x1280 K=2048", gemm_case(2048, 5120, 1280, trans_b=False, out_f32=False), gemm_case(5120, 1280, 2048, trans_a=True, trans_b=True, accumulate=True))
