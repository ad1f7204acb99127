"""Dev probe: run torch.matmul (hipBLASLt) on the step's GEMM shapes so that `rocprofv3 --kernel-trace --stats` shows which kernels
(macro tile, workgroup, LDS) the vendor library picks.  Measuring stick only."""
import torch
dev = "cuda"
def bf(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
M = 131072
for (n, k) in ((2304, 768), (3072, 768), (768, 3072), (768, 768), (768, 2304)):
    a, b = bf(M, k), bf(n, k)
    o = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        torch.matmul(a, b.t(), out=o)
    bt = b.t().contiguous()  # [k, n]: the dgrad orientation
    for _ in range(3):
        torch.matmul(a, bt, out=o)
a, b = bf(4096, 4096), bf(4096, 4096)
for _ in range(3):
    torch.matmul(a, b.t())
torch.cuda.synchronize()
