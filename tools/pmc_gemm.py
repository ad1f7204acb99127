"""Runs each GEMM instantiation a few times at the bench shapes (base, per-GPU batch 128) so that rocprofv3 --pmc
passes can attribute HBM traffic per launch.  Usage on the GPU box (separate passes, counters only):
  rocprofv3 --pmc FETCH_SIZE  -d gpurun_out/pmc_fetch -- python tools/pmc_gemm.py
  rocprofv3 --pmc WRITE_SIZE  -d gpurun_out/pmc_write -- python tools/pmc_gemm.py
then tools/pmc_parse.py turns the counter CSVs into profiles/pmc_gemm.json."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402

M, DIM, MLP = 128 * 1024, 768, 3072
dev = "cuda"
bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
x, w1 = bf(M, DIM), bf(MLP, DIM)
h = torch.empty(M, MLP, dtype=torch.bfloat16, device=dev)
dx = torch.empty(M, DIM, device=dev)
dw = torch.zeros(MLP, DIM, device=dev)
for _ in range(3):
    _C.gemm(x, w1, M, MLP, DIM, out_bf16=h)                                   # NT  fc1 forward
    _C.gemm(h, w1, M, DIM, MLP, trans_b=True, out_f32=dx)                    # NN  fc1 dgrad
    _C.gemm(h, x, MLP, DIM, M, trans_a=True, trans_b=True, accumulate=True, out_f32=dw)  # TT  fc1 wgrad (split-K)
torch.cuda.synchronize()
