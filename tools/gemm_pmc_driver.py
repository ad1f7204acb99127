"""A few launches of each GEMM role of the training step at the base shapes — the workload of tools/pmc_gemm_r3.sh (rocprofv3 --pmc passes). Dev tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

TOK, DIM, MLP = 131072, 768, 3072
dev = "cuda"


def bf(*s, scale=0.3):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


x, w_qkv, w_fc1 = bf(TOK, DIM), bf(3 * DIM, DIM), bf(MLP, DIM)
dy_qkv, dh = bf(TOK, 3 * DIM, scale=0.1), bf(TOK, MLP, scale=0.1)
out_qkv = torch.empty(TOK, 3 * DIM, dtype=torch.bfloat16, device=dev)
out_fc1 = torch.empty(TOK, MLP, dtype=torch.bfloat16, device=dev)
dx = torch.empty(TOK, DIM, dtype=torch.bfloat16, device=dev)
dw = torch.zeros(3 * DIM, DIM, device=dev)
bias = torch.randn(MLP, device=dev)
for _ in range(int(os.environ.get("ITERS", "3"))):
    _C.gemm(dy_qkv, x, 3 * DIM, DIM, TOK, trans_a=True, trans_b=True, accumulate=True, out_f32=dw)          # weight gradient (w256, split-K)
    _C.gemm(x, w_qkv, TOK, 3 * DIM, DIM, out_bf16=out_qkv)                                                    # forward qkv (w256r)
    _C.gemm(x, w_fc1, TOK, MLP, DIM, bias=bias, act=_C.ACT_TANH, out_bf16=out_fc1)                           # forward fc1 + tanh (w256r)
    _C.gemm(dh, w_fc1, TOK, DIM, MLP, trans_b=True, out_bf16=dx)                                              # input gradient fc1 (w256r, B stored [K][N])
torch.cuda.synchronize()
