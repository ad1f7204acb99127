"""Bench-scale spot checks: the GEMM / attention / quantizer kernels at EXACTLY the grids bench.py launches (BASELINE config 2:
ViT-VQGAN base, 128 images per GPU -> M = 131072 tokens), where the XCD remap walks thousands of tiles, row offsets pass 2^31
bytes and split-K runs its full slice count.  A remap or 32-bit-offset bug that only appears at this size would pass every
small-shape test in test_ops_gpu.py and still print a plausible images/s.

Reference = fp64 torch matmul on the same device over SAMPLED output rows (plus the first and last tile rows), fed the same
bf16-representable operands; tolerances are the op-level ones (fp32 outputs 1e-5, bf16 outputs 1.15 x the bf16 rounding floor).
"""
import pytest
import torch

from util import bf16_floor, rel

pytestmark = pytest.mark.gpu

B_IMG, N_TOK, H, DIM, MLP = 128, 1024, 12, 768, 3072
M = B_IMG * N_TOK


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available()
    from enhancing import _C
    _C.lib()
    return _C


def _bf(shape, seed, scale=0.5):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


def _sample_rows(n, count, seed):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randint(0, n, (count,), generator=g)
    edge = torch.tensor([0, 1, 127, 128, 255, 256, n - 257, n - 256, n - 129, n - 128, n - 2, n - 1])
    return torch.unique(torch.cat([rows, edge.clamp_(0, n - 1)])).cuda()


# the six forward / dgrad contractions of one transformer layer at M = 131072 (engine/stage1.py _Tower.forward / backward)
FWD_DGRAD = [("fwd qkv", 3 * DIM, DIM, False, "bf16"), ("fwd fc1", MLP, DIM, False, "bf16"), ("fwd fc2", DIM, MLP, False, "f32"),
             ("fwd out", DIM, DIM, False, "f32"), ("dgrad qkv", DIM, 3 * DIM, True, "f32"), ("dgrad fc2", MLP, DIM, True, "bf16"),
             ("dgrad fc1", DIM, MLP, True, "bf16")]


@pytest.mark.parametrize("name,N,K,tb,out", FWD_DGRAD, ids=[c[0].replace(" ", "_") for c in FWD_DGRAD])
def test_gemm_bench_scale_rows(C, name, N, K, tb, out):
    a = _bf((M, K), 1)
    b = _bf((K, N) if tb else (N, K), 2, 0.05)
    o32 = torch.full((M, N), float("nan"), device="cuda") if out == "f32" else None
    o16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda") if out == "bf16" else None
    C.gemm(a, b, M, N, K, trans_b=tb, out_f32=o32, out_bf16=o16)
    rows = _sample_rows(M, 4096, 3)
    bd = b.double() if tb else b.double().t()
    ref = a[rows].double() @ bd
    got = (o32 if o32 is not None else o16)[rows]
    assert torch.isfinite((o32 if o32 is not None else o16).float()).all(), f"{name}: unwritten (NaN) outputs"
    e = rel(got.float(), ref)
    print(f"{name:10s} M={M} N={N} K={K} [{C.lib().enh_gemm_h16_variant(0, int(tb), M, N, K).decode()}]: rel {e:.2e} over {len(rows)} rows")
    if out == "f32":
        assert e <= 1e-5
    else:
        assert e <= 1.15 * bf16_floor(ref) + 1e-6


WGRAD = [("wgrad qkv", 3 * DIM, DIM), ("wgrad fc1", MLP, DIM), ("wgrad fc2", DIM, MLP), ("wgrad out", DIM, DIM)]


@pytest.mark.parametrize("name,NO,KI", WGRAD, ids=[c[0].replace(" ", "_") for c in WGRAD])
def test_gemm_bench_scale_wgrad_splitk(C, name, NO, KI):
    """dW[NO, KI] += dY^T X with the contraction over all 131072 tokens (split-K), accumulate into a non-zero dW."""
    dy = _bf((M, NO), 4, 0.1)
    x = _bf((M, KI), 5)
    g = torch.Generator(device="cuda").manual_seed(6)
    base = torch.randn(NO, KI, device="cuda", generator=g)
    dW = base.clone()
    C.gemm(dy, x, NO, KI, M, trans_a=True, trans_b=True, accumulate=True, out_f32=dW)
    rows = _sample_rows(NO, 192, 7)
    ref = dy[:, rows].double().t() @ x.double() + base[rows].double()
    e = rel(dW[rows], ref)
    print(f"{name:10s} NO={NO} KI={KI} tokens={M} [{C.lib().enh_gemm_h16_variant(1, 1, NO, KI, M).decode()}]: rel {e:.2e}")
    assert torch.isfinite(dW).all()
    assert e <= 2e-5   # fp32 accumulation over 131072-long sums, split-K partial order


def _attn_ref(qkv, scale):
    """qkv [N, 3, 64] fp64 of ONE (image, head) -> (out [N,64], lse [N])"""
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    s = (q @ k.t()) * scale
    return torch.softmax(s, -1) @ v, torch.logsumexp(s, -1)


def test_attention_bench_scale(C):
    """B = 128, H = 12, N = 1024 (1536 (image, head) pairs, 8 query blocks each): sampled pairs incl. the first and last vs fp64."""
    scale = 0.125
    qkv = _bf((B_IMG, N_TOK, 3 * H * 64), 8, 1.2)
    do = _bf((B_IMG, N_TOK, H * 64), 9, 1.0)
    out = torch.full((B_IMG, N_TOK, H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B_IMG, H, N_TOK), float("nan"), device="cuda")
    C.attention_forward(qkv, B_IMG, N_TOK, H, scale, out, lse)
    dqkv = torch.full((B_IMG, N_TOK, 3 * H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(B_IMG, H, N_TOK, device="cuda")
    C.attention_backward(qkv, out, do, lse, B_IMG, N_TOK, H, scale, dqkv, delta)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all() and torch.isfinite(dqkv.float()).all()
    g = torch.Generator().manual_seed(10)
    pairs = {(0, 0), (B_IMG - 1, H - 1), (B_IMG - 1, 0), (0, H - 1), (63, 5), (64, 6)}
    while len(pairs) < 24:
        pairs.add((int(torch.randint(0, B_IMG, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g))))
    worst = dict(out=0.0, lse=0.0, dq=0.0, dk=0.0, dv=0.0)
    for b, h in sorted(pairs):
        x = qkv[b].view(N_TOK, 3, H, 64)[:, :, h].double().clone().requires_grad_(True)
        o_ref, lse_ref = _attn_ref(x, scale)
        o_ref.backward(do[b].view(N_TOK, H, 64)[:, h].double())
        o = out[b].view(N_TOK, H, 64)[:, h].float()
        d = dqkv[b].view(N_TOK, 3, H, 64)[:, :, h].float()
        worst["out"] = max(worst["out"], rel(o, o_ref) / bf16_floor(o_ref.detach()))
        worst["lse"] = max(worst["lse"], rel(lse[b, h], lse_ref))
        for i, kname in enumerate(("dq", "dk", "dv")):
            worst[kname] = max(worst[kname], rel(d[:, i], x.grad[:, i]))
    print("attention at B=128 H=12 N=1024, worst over 24 (image, head) pairs:", {k: f"{v:.2e}" for k, v in worst.items()}, "(out in units of the bf16 floor)")
    assert worst["out"] <= 1.5 and worst["lse"] <= 1e-5
    assert max(worst["dq"], worst["dk"], worst["dv"]) <= 1e-2
