"""Attention laboratory (GPU box): every kernel family of enh_attention_set_kernel at the bench shape (B = 128, H = 12, N = 1024) — correctness of a
sampled (image, head) pair against fp64 and wall time per pass.  Usage: python tools/attn_lab.py [fwd,dq,dkv ...]   e.g.  5,3,2 1,1,1"""
import os, sys, torch
DT = torch.float16 if os.environ.get('ATTN_DTYPE', 'fp16') == 'fp16' else torch.bfloat16      # operand format of the lab (round 6: fp16 = the headline)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C
B, N, H = int(os.environ.get("MB_BATCH", "128")), 1024, 12
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, N, 3 * H * 64, device=dev, generator=g) * 1.2).to(DT)
do = (torch.randn(B, N, H * 64, device=dev, generator=g)).to(DT)
out = torch.empty(B, N, H * 64, dtype=DT, device=dev)
lse = torch.empty(B, H, N, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, device=dev)
fl = 4 * B * H * N * N * 64


def timeit(fn, iters=int(os.environ.get("ITERS", "10")), warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def check(b, h):
    x = qkv[b].view(N, 3, H, 64)[:, :, h].double().clone()
    if PRE:
        x[:, 0] /= CP
    x.requires_grad_(True)
    q, k, v = x.unbind(1)
    s = (q @ k.t()) * 0.125
    o_ref = torch.softmax(s, -1) @ v
    o_ref.backward(do[b].view(N, H, 64)[:, h].double())
    o = out[b].view(N, H, 64)[:, h].float()
    d = dqkv[b].view(N, 3, H, 64)[:, :, h].float()
    return dict(out=rel(o, o_ref), lse=rel(lse[b, h], torch.logsumexp(s, -1)), dq=rel(d[:, 0], x.grad[:, 0]), dk=rel(d[:, 1], x.grad[:, 1]), dv=rel(d[:, 2], x.grad[:, 2]))


PRE = os.environ.get("PRE", "0") == "1"          # q_prescaled convention: the q third holds bf16(q * scale * log2e)
CP = 0.125 * 1.4426950408889634
if PRE:
    qkv.view(B, N, 3, H * 64)[:, :, 0] = (qkv.view(B, N, 3, H * 64)[:, :, 0].float() * CP).to(DT)
fams = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(5, 3, 2), (1, 1, 1)]
ROUNDS = int(os.environ.get("ROUNDS", "4"))      # families are timed in interleaved rounds: the chip's clock drifts with temperature / power by +-10 %
tfs, tbs, errs_of = {f: [] for f in fams}, {f: [] for f in fams}, {}
for rnd in range(ROUNDS):
    for fam in fams:
        _C.attention_set_kernel(*fam)
        if rnd == 0:
            out.fill_(float("nan")); dqkv.fill_(float("nan"))
            _C.attention_forward(qkv, B, N, H, 0.125, out, lse, q_prescaled=PRE)
            _C.attention_backward(qkv, out, do, lse, B, N, H, 0.125, dqkv, delta, q_prescaled=PRE)
            torch.cuda.synchronize()
            errs = [check(b, h) for b, h in ((0, 0), (B - 1, H - 1), (B // 2, 5))]
            errs_of[fam] = {k: max(e[k] for e in errs) for k in errs[0]}
        tfs[fam].append(timeit(lambda: _C.attention_forward(qkv, B, N, H, 0.125, out, lse, q_prescaled=PRE)))
        tbs[fam].append(timeit(lambda: _C.attention_backward(qkv, out, do, lse, B, N, H, 0.125, dqkv, delta, q_prescaled=PRE)))
for fam in fams:
    tf, tb = min(tfs[fam]), min(tbs[fam])
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{'prescaled q' if PRE else 'plain q'} family fwd,dq,dkv = {fam}: fwd min {tf*1e3:6.3f} med {med(tfs[fam])*1e3:6.3f} ms {fl/tf/1e12:6.1f} TF/s | "
          f"bwd min {tb*1e3:6.3f} med {med(tbs[fam])*1e3:6.3f} ms {2.5*fl/tb/1e12:6.1f} TF/s (algorithmic) | "
          + " ".join(f"{k} {v:.2e}" for k, v in errs_of[fam].items()), flush=True)
_C.attention_set_kernel(0, 0, 0)
