"""Split-K weight-gradient timing at the training step's four shapes (gemm_bf16_w256_kernel<true, true, EPI_WS> + the fixed-order reduce pass), per
kernel family given in LAB_FAMS (enh_gemm_set_kernel values; default: the per-shape choice only), with a bitwise comparison of the sums.
Dev tool, GPU box:  python tools/wgrad_lab.py [batch].  Round 3 used it for two A/B runs whose other arm has since been removed from the library
(profiles/r03_gemm_persistent_lab.txt sections 8, 9): the workgroup -> XCD mapping of split-K launches, and a five-slot ring of 32-deep stages."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LABS = [int(x) for x in os.environ.get("LAB_FAMS", "-1").split(",")]
ROUNDS = int(os.environ.get("LAB_ROUNDS", "4"))
TOK = B * 1024
L = _C.lib()
dev = "cuda"


def set_lab(fam):
    if L.enh_gemm_set_kernel(fam) != 0:
        raise RuntimeError(L.enh_last_error().decode())


def bf(*shape, scale=0.5):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


def case(name, n_out, k_in, iters=10):
    dy, x = bf(TOK, n_out, scale=0.1), bf(TOK, k_in)
    dw = torch.zeros(n_out, k_in, device=dev)
    fn = lambda: _C.gemm(dy, x, n_out, k_in, TOK, trans_a=True, trans_b=True, accumulate=True, out_f32=dw)
    outs = {}
    for lab in LABS:
        set_lab(lab); dw.zero_(); fn(); torch.cuda.synchronize(); outs[lab] = dw.clone()
    same = all(torch.equal(outs[LABS[0]], outs[l]) for l in LABS)
    times = {l: [] for l in LABS}
    for _ in range(ROUNDS):
        for lab in LABS:
            set_lab(lab)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(); s.record()
            for _ in range(iters):
                fn()
            e.record(); torch.cuda.synchronize()
            times[lab].append(s.elapsed_time(e) / iters)
    fl = 2.0 * n_out * k_in * TOK
    cells = [f"fam{l} {min(times[l]):6.3f}/{statistics.median(times[l]):6.3f} {fl / min(times[l]) / 1e9:5.0f}" for l in LABS]
    print(f"{name:10s} [{n_out:4d} x {k_in:4d}] {'bitwise equal' if same else 'MISMATCH'} | " + " | ".join(cells), flush=True)


print(f"tokens {TOK}: min / median ms (incl. the split-K reduce pass), TF/s at the minimum")
case("wgrad qkv", 2304, 768)
case("wgrad out", 768, 768)
case("wgrad fc1", 3072, 768)
case("wgrad fc2", 768, 3072)
set_lab(-1)
