"""Per-layer timing of the discriminator's implicit-GEMM convolutions (forward / input gradient / weight gradient) at the real layer shapes.
Usage (GPU box): python tools/conv_bench.py [batch] [auto|reg|t128|t256]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing.losses.op import conv_nhwc as cn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    cn._C.conv_set_kernel(sys.argv[2])
    print(f"kernel family: {sys.argv[2]}")
# H, Cin, Cout, k, stride, pad (input spatial size H x H)
LAYERS = [(256, 8, 128, 1, 1, 0), (256, 128, 128, 3, 1, 1), (257, 128, 256, 3, 2, 0), (255, 128, 256, 1, 2, 0), (128, 256, 256, 3, 1, 1),
          (129, 256, 512, 3, 2, 0), (64, 512, 512, 3, 1, 1), (65, 512, 512, 3, 2, 0), (32, 512, 512, 3, 1, 1), (16, 512, 512, 3, 1, 1)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for H, Cin, Cout, k, s, p in LAYERS:
    x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device="cuda")
    Ho = (H + 2 * p - k) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda").to(torch.bfloat16)
    fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    tf = timeit(lambda: cn._fwd(x, w, 0.1, s, p))
    td = timeit(lambda: cn._dgrad(dy, w, 0.1, s, p, H, H, Cin))
    tw = timeit(lambda: cn._wgrad(x, dy, 0.1, s, p, k, Cin))
    print(f"B{B} {H:4d}^2 {Cin:4d}->{Cout:4d} k{k} s{s}: fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.0f} TF/s | dgrad {td*1e3:7.1f} us {fl/td/1e9:6.0f} | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:6.0f}")
# blur at the two largest sizes
kern = torch.tensor([1., 3., 3., 1.], device="cuda")
kern = kern[None] * kern[:, None]
kern = kern / kern.sum()
for variant in (1, 0):
    cn._C.blur_set_kernel(variant)
    for H, C in ((256, 128), (128, 256), (64, 512)):
        x = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
        t = timeit(lambda: cn._C.blur_nhwc(x, kern, 2, 2, False))
        print(f"blur variant {variant} {H}^2 x {C}: {t*1e3:7.1f} us  {2*x.numel()*2/t/1e6:6.0f} GB/s")
cn._C.blur_set_kernel(0)
