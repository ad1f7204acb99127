"""Reproduces the ROCm graph-capture defect the library works around (csrc/common.cpp enh_zero_f32_launch): a hipMemsetAsync node captured inside a library call
re-executes only part of its range on replay.  With the fill kernel every replay equals the eager call; with hipMemsetAsync half of the 512 channel sums were
garbage from the second replay on (profiles/r04_graph_replay_findings.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
import torch
from enhancing import _C
_C.lib()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(2, 512, device="cuda", generator=g)
ref = _C.channel_sum(x).clone()
print("eager == x.sum(0):", torch.equal(ref, x.sum(0)), float((ref - x.sum(0)).abs().max()))
static_x = x.clone()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): _C.channel_sum(static_x)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = _C.channel_sum(static_x)
for rep in range(4):
    static_x.copy_(x * (rep + 1))
    gr.replay(); torch.cuda.synchronize()
    e = _C.channel_sum(x * (rep + 1))
    print("replay", rep, "equal to eager:", torch.equal(out, e), "n differing", int((out != e).sum()), "max abs diff", float((out - e).abs().max()))
