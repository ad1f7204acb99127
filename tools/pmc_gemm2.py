"""fc2-forward (NT, K = 3072) and fc1-wgrad (TT) at M = 131072 with the kernel family chosen by ENH_GEMM_KERNEL."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C
M, DIM, MLP = 128 * 1024, 768, 3072
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
h, w2, x = bf(M, MLP), bf(DIM, MLP), bf(M, DIM)
o = torch.empty(M, DIM, device="cuda"); dw = torch.zeros(MLP, DIM, device="cuda")
for _ in range(3):
    _C.gemm(h, w2, M, DIM, MLP, out_f32=o)
    _C.gemm(h, x, MLP, DIM, M, trans_a=True, trans_b=True, accumulate=True, out_f32=dw)
torch.cuda.synchronize()
