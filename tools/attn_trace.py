"""Cycle anatomy of the eight-wave antiphase attention forward (attention_v3.hip): s_memtime stamps of waves 0 (group A) and 4 (group B, same SIMD) of
workgroup 0 at the four edges of every period, at the bench shape with the whole chip busy.  python tools/attn_trace.py  -> table (+ gpurun_out/)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C
B, N, H = 128, 1024, 12
L = _C.lib()
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B, N, 3 * H * 64, device="cuda", generator=g) * 1.2)
qkv.view(B, N, 3, H * 64)[:, :, 0] *= 0.125 * 1.4426950408889634
qkv = qkv.to(torch.bfloat16)
out = torch.empty(B, N, H * 64, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, N, device="cuda")
nt = N // 64
tr = torch.zeros(2, nt, 4, dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = L.enh_debug_attention_fwd3_trace(_C._p(qkv), B, N, H, _C._p(out), _C._p(lse), _C._p(tr), _C._stream())
    assert rc == 0, L.enh_last_error()
torch.cuda.synchronize()
t = tr.cpu()
t0 = int(t[0, 0, 0])
lines = ["# s_memtime ticks (shader clock) relative to wave 0's first stamp; per period: vector segment | wait at barrier 1 | matrix segment | wait at barrier 2",
         f"# {'i':>2} | wave 0 (group A): {'start':>7} {'vector':>7} {'wait1':>6} {'matrix':>7} {'wait2':>6} | wave 4 (group B): {'start':>7} {'vector':>7} {'wait1':>6} {'matrix':>7} {'wait2':>6}"]
for i in range(nt):
    row = f"  {i:>2} |"
    for w in range(2):
        a, b, c, d = (int(x) for x in t[w, i])
        nxt = int(t[w, i + 1, 0]) if i + 1 < nt else d
        row += f"                  {a - t0:>7} {b - a:>7} {c - b:>6} {d - c:>7} {nxt - d:>6} |"
    lines.append(row)
for w in range(2):
    vec = [int(t[w, i, 1] - t[w, i, 0]) for i in range(2, nt)]
    mat = [int(t[w, i, 3] - t[w, i, 2]) for i in range(2, nt)]
    per = [int(t[w, i + 1, 0] - t[w, i, 0]) for i in range(2, nt - 1)]
    lines.append(f"# wave {4 * w}: median vector segment {sorted(vec)[len(vec) // 2]}, matrix segment {sorted(mat)[len(mat) // 2]}, period {sorted(per)[len(per) // 2]} ticks")
txt = "\n".join(lines)
print(txt)
od = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(od):
    open(os.path.join(od, "r04_attn_trace.txt"), "w").write(txt + "\n")
