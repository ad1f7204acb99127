"""Where the time of a stride-2 input gradient goes: each parity class alone, and the SAME GEMM shape with dense output rows (tools/conv_s2_probe.py, GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C
from enhancing.losses.op import conv_nhwc as cn

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

B, H, Cin, Cout, k, stride, pad = 16, 257, 128, 256, 3, 2, 0
Ho = (H + 2 * pad - k) // stride + 1
dy = torch.randn(B, Ho, Ho, Cout, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, k, k, device="cuda")
out = torch.empty(B, H, H, Cin, dtype=torch.bfloat16, device="cuda")
tot = 0.0
for ph in range(2):
    kh0 = (ph + pad) % 2; nty = len(range(kh0, k, 2)); oy0 = (ph + pad - kh0) // 2; Hm = (H - ph + 1) // 2
    for pw in range(2):
        kw0 = (pw + pad) % 2; ntx = len(range(kw0, k, 2)); ox0 = (pw + pad - kw0) // 2; Wm = (H - pw + 1) // 2
        wt = cn._pack(w, 0.1, True, kh0, kw0, 2, nty, ntx, Cin, Cout)
        geom = dict(B=B, Hs=Ho, Ws=Ho, C=Cout, Hm=Hm, Wm=Wm, gs=1, oy0=oy0, ox0=ox0, nty=nty, ntx=ntx, sty=-1, stx=-1, N=Cin, HO=H, WO=H, os=2, oph=ph, opw=pw)
        t = timeit(lambda: _C.conv_nhwc(dy, wt, geom, 2, out=out))
        dense = dict(geom, HO=Hm, WO=Wm, os=1, oph=0, opw=0)
        outd = torch.empty(B, Hm, Wm, Cin, dtype=torch.bfloat16, device="cuda")
        td = timeit(lambda: _C.conv_nhwc(dy, wt, dense, 2, out=outd))
        fl = 2.0 * B * Hm * Wm * Cin * nty * ntx * Cout
        tot += t
        print(f"class ({ph},{pw}): {nty}x{ntx} taps, K = {nty*ntx*Cout:5d}, rows {B*Hm*Wm}: strided {t:7.1f} us {fl/t/1e6:6.0f} TF/s | dense rows, same GEMM {td:7.1f} us {fl/td/1e6:6.0f} TF/s")
print(f"sum of the four classes {tot:.1f} us")
