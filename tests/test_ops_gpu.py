"""Op-level parity tests of the HIP kernels (through the C ABI via ctypes) against the CPU oracle.

Tolerances (relative Frobenius error vs the fp64/fp32 oracle fed the SAME bf16-representable operands):
  * integer outputs (code indices): bit-exact;
  * fp32-output kernels: <= 1e-5 (measured 6e-8 .. 2e-7 on MI355X: only the fp32 summation order differs) — well inside the
    1e-3 the north-star asks for;
  * bf16-output kernels: <= 1.15 x the bf16 ROUNDING FLOOR of the exact result, computed in the test (util.bf16_floor; 1.66e-3
    for Gaussian data: no bf16-stored tensor can be closer to an fp32 reference than that, so 1e-3 is unattainable for them by
    construction; measured: GEMM / LayerNorm exactly at the floor);
  * fused attention (probabilities rounded to bf16 before P.V, bf16 output): <= 1.5 x floor (measured 1.10 x); its
    gradients (two chained bf16 roundings) <= 1e-2.
"""
import os
import numpy as np
import pytest
import torch

from util import bf16_floor, bf16r, rel

pytestmark = pytest.mark.gpu

F32_TOL, BF16_TOL, ATT_TOL = 1e-5, 2.5e-3, 5e-3   # BF16_TOL / ATT_TOL are absolute caps; the floor-relative asserts are the tight ones


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from enhancing import _C
    _C.lib()  # fail loudly if the HIP extension is missing
    return _C


# ---------------------------------------------------------------------------------------------
# quantizer
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,depth", [(1000, 8192, 1), (777, 500, 1), (640, 8192, 4), (96, 1024, 2)])
def test_vq_forward_bit_exact_vs_c_oracle(C, M, K, depth):
    import vitvq_oracle as O
    import vq_oracle as VC
    z, E, _ = O.make_vq_inputs(100 + M, M, K)
    zq_c, idx_c, loss_c = VC.forward(z.numpy(), E.numpy(), 0.25, depth)
    zq, zq16, idx, loss = C.vq_forward(z.cuda(), E.cuda(), 0.25, depth, True)
    assert torch.equal(idx.cpu(), torch.from_numpy(idx_c)), "code indices must be bit-exact"
    assert np.array_equal(zq.cpu().numpy().view(np.uint32), zq_c.view(np.uint32)), "z_q must be bit-exact"
    assert abs(loss.item() - float(loss_c)) <= 1e-6 * abs(float(loss_c))
    assert torch.equal(zq16.cpu(), zq.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("name", ["vq_k8192_m4096", "vq_k512_m1024", "rq4_k8192_m2048"])
def test_vq_forward_golden_reference(C, golden_dir, name):
    """indices / loss / z_q against vectors produced by the REFERENCE's VectorQuantizer (oracle/make_golden.py)."""
    import vitvq_oracle as O
    g = np.load(f"{golden_dir}/{name}.npz")
    M, K, depth = int(g["M"]), int(g["K"]), max(int(g["num_quantizers"]), 1)
    z, E, _ = O.make_vq_inputs(int(g["seed"]), M, K)
    zq, _, idx, loss = C.vq_forward(z.cuda(), E.cuda(), float(g["beta"]), depth, True)
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int64).reshape(M, depth)), "indices differ from the reference"
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    assert np.abs(zq.cpu().numpy()[:64] - g["zq_first"].reshape(64, 32)).max() <= 5e-7
    assert abs(zq.double().sum().item() - float(g["zq_sum"])) <= 1e-3


@pytest.mark.parametrize("M,K,depth,resid", [(1024, 1024, 1, False), (512, 2048, 4, True), (256, 512, 1, True)])
def test_vq_backward_vs_autograd(C, M, K, depth, resid):
    import vitvq_oracle as O
    z, E, g = O.make_vq_inputs(7 + depth, M, K)
    zt = z.clone().requires_grad_(True)
    Et = E.clone().requires_grad_(True)
    zq, loss, idx = O.quantizer_forward(zt, Et, 0.25, True, resid, depth if resid else None)
    gl = 0.7
    ((zq * g).sum() + gl * loss).backward()
    dE = torch.zeros(K, 32, device="cuda")
    idx_d = idx.view(M, -1).cuda()
    dz, dz16 = C.vq_backward(z.cuda(), E.cuda(), idx_d, g.cuda(), gl, None, 0.25, depth, resid, True, dE)
    assert rel(dz, zt.grad) <= F32_TOL
    assert rel(dE, Et.grad) <= F32_TOL
    assert rel(dz16.float(), zt.grad) <= BF16_TOL


def test_vq_full_size_match_rate(C):
    """BASELINE config-2 op vectors: M = 131072 tokens, K = 8192.  HIP == C oracle bit-exact; vs the plain
    PyTorch oracle every mismatch (if any) must be an fp32 near-tie (top-2 gap < 1e-6)."""
    import vitvq_oracle as O
    import vq_oracle as VC
    M, K = 131072, 8192
    z, E, _ = O.make_vq_inputs(1234, M, K)
    zq, _, idx, loss = C.vq_forward(z.cuda(), E.cuda(), 0.25, 1, True)
    idx = idx.cpu().view(-1)
    _, idx_c, _ = VC.forward(z.numpy(), E.numpy(), 0.25, 1)
    assert torch.equal(idx, torch.from_numpy(idx_c).view(-1))
    # plain-PyTorch oracle (the reference's formula) in chunks
    mism = 0
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    for s in range(0, M, 16384):
        zz = z[s:s + 16384]
        _, _, it = O.vq_quantize(zz, E)
        bad = (it != idx[s:s + 16384]).nonzero().view(-1)
        for j in bad.tolist():
            zn = torch.nn.functional.normalize(zz[j:j + 1].double(), dim=-1)
            en = torch.nn.functional.normalize(E.double(), dim=-1)
            d = ((zn ** 2).sum(1, keepdim=True) + (en ** 2).sum(1) - 2 * zn @ en.t()).view(-1)
            gap = (d[it[j]] - d[idx[s + j]]).abs().item()
            assert gap < 1e-6, f"token {s + j}: indices {it[j].item()} vs {idx[s + j].item()} differ with gap {gap}"
        mism += len(bad)
    rate = 1.0 - mism / M
    print(f"VQ argmin match-rate vs PyTorch oracle: {rate:.8f} ({mism} near-tie mismatches of {M})")
    assert rate >= 0.9999


def test_vq_lookup(C):
    import vitvq_oracle as O
    _, E, _ = O.make_vq_inputs(3, 64, 1024)
    idx = torch.randint(0, 1024, (300, 4), generator=torch.Generator().manual_seed(0))
    ref = O.l2norm(torch.nn.functional.embedding(idx, E)).sum(-2)
    out, out16 = C.vq_lookup(E.cuda(), idx.cuda(), True)
    assert rel(out, ref) <= 1e-6


# ---------------------------------------------------------------------------------------------
# layernorm
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,D", [(1027, 768), (64, 128), (300, 512), (130, 1280), (33, 2048)])
def test_layernorm_fwd_bwd(C, M, D):
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 2 + 0.5
    w = 1 + 0.1 * torch.randn(D, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(M, D, generator=g)
    dres = torch.randn(M, D, generator=g)
    xt, wt, bt = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xt, (D,), wt, bt, 1e-5)
    y.backward(dy)
    xd = x.cuda()
    y16 = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    y32 = torch.empty(M, D, device="cuda")
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    C.layernorm_forward(xd, w.cuda(), b.cuda(), 1e-5, y16, y32, mean, rstd)
    assert rel(y32, y) <= F32_TOL
    assert rel(y16.float(), y) <= 1.15 * bf16_floor(y) + 1e-6
    dx = torch.empty(M, D, device="cuda"); dx16 = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dxs = torch.zeros(D, device="cuda")
    C.layernorm_backward(dy.cuda(), xd, w.cuda(), mean, rstd, dres.cuda(), dx, dx16, dw, db, dxs)
    assert rel(dxs, (xt.grad + dres).double().sum(0)) <= F32_TOL
    assert rel(dx, xt.grad + dres) <= F32_TOL
    assert rel(dw, wt.grad) <= F32_TOL and rel(db, bt.grad) <= F32_TOL
    assert rel(dx16.float(), xt.grad + dres) <= BF16_TOL
    # the same op fed with the dgrad GEMM's bf16 output: identical to the f32 entry on the bf16-rounded gradient
    dy16 = dy.to(torch.bfloat16)
    xt2, wt2, bt2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xt2, (D,), wt2, bt2, 1e-5).backward(dy16.float())
    dw.zero_(); db.zero_(); dxs.zero_()
    C.layernorm_backward(dy16.cuda(), xd, w.cuda(), mean, rstd, dres.cuda(), dx, dx16, dw, db, dxs)
    assert rel(dx, xt2.grad + dres) <= F32_TOL and rel(dxs, (xt2.grad + dres).double().sum(0)) <= F32_TOL
    assert rel(dw, wt2.grad) <= F32_TOL and rel(db, bt2.grad) <= F32_TOL


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
def _mk(shape, g, scale=1.0, dt=torch.bfloat16):
    return (torch.randn(*shape, generator=g) * scale).to(dt).to(torch.float32)      # values exactly representable in the operand format


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 192, 768), (136, 32, 64), (384, 768, 3072), (128, 2304, 32),
                                   (520, 768, 192), (768, 512, 2048), (1288, 264, 64), (2048, 768, 192), (1024, 1280, 832), (4096, 2304, 768)])
def test_gemm_layouts(C, ta, tb, M, N, K):
    if ta and M % 8:
        pytest.skip("trans_a needs M % 8 == 0")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = _mk((M, K), g)
    B = _mk((N, K), g)
    ref = A.double() @ B.double().t()
    a = (A.t().contiguous() if ta else A).to(torch.bfloat16).cuda()
    b = (B.t().contiguous() if tb else B).to(torch.bfloat16).cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    C.gemm(a, b, M, N, K, trans_a=ta, trans_b=tb, out_f32=out, out_bf16=out16)
    assert rel(out, ref) <= F32_TOL, f"ta={ta} tb={tb}"
    assert rel(out16.float(), ref) <= 1.15 * bf16_floor(ref) + 1e-6


@pytest.mark.parametrize("kernel_shape", [(512, 384, 256, 128), (1024, 768, 192, 256)])
def test_gemm_epilogues(C, kernel_shape):
    g = torch.Generator().manual_seed(5)
    M, N, K, T = kernel_shape
    A, B = _mk((M, K), g, 0.5), _mk((N, K), g, 0.1)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    pos = torch.randn(T, N, generator=g)
    a, b = A.to(torch.bfloat16).cuda(), B.to(torch.bfloat16).cuda()
    base = A.double() @ B.double().t()
    # bias + tanh -> bf16
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    C.gemm(a, b, M, N, K, bias=bias.cuda(), act=C.ACT_TANH, out_bf16=o16)
    assert rel(o16.float(), torch.tanh(base + bias.double())) <= BF16_TOL
    # bias + residual (in place on the residual stream)
    x = res.clone().cuda()
    C.gemm(a, b, M, N, K, bias=bias.cuda(), res=x, res_rows=M, out_f32=x)
    assert rel(x, base + bias.double() + res.double()) <= F32_TOL
    # bias + position table (row index modulo T)
    o = torch.empty(M, N, device="cuda")
    C.gemm(a, b, M, N, K, bias=bias.cuda(), res=pos.cuda(), res_rows=T, out_f32=o)
    assert rel(o, base + bias.double() + pos.double().repeat(M // T, 1)) <= F32_TOL
    # tanh backward: v * (1 - h^2)
    h = bf16r(torch.tanh(torch.randn(M, N, generator=g)))
    C.gemm(a, b, M, N, K, act=C.ACT_DTANH, aux=h.to(torch.bfloat16).cuda(), out_bf16=o16)
    assert rel(o16.float(), base * (1 - h.double() ** 2)) <= BF16_TOL
    # accumulate
    o = res.clone().cuda()
    C.gemm(a, b, M, N, K, accumulate=True, out_f32=o)
    assert rel(o, base + res.double()) <= F32_TOL


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,tb", [(2048, 1280, 5120, False), (2048, 1280, 3840, True), (4096, 768, 3072, False), (1024, 768, 2304, True)])
def test_gemm_forward_split_k(C, M, N, K, tb, dt):
    """Round 6 (shipped batch sizes): an f32-output GEMM with few tiles and a long K (the N = dim GEMMs at 2 - 4 images per GPU: 96 - 192 tiles of
    128 x 128 on 512 slots) is split over K when the caller gives a workspace; bias / residual / position-table residual / accumulate are applied by the
    fixed-order second pass.  Every epilogue against fp64, bit-reproducible, and the plan really is a split (the workspace query is non-zero; at training
    sizes it is zero)."""
    g = torch.Generator().manual_seed(77)
    A, B = _mk((M, K), g, 0.5, dt), (_mk((K, N), g, 0.05, dt) if tb else _mk((N, K), g, 0.05, dt))
    bias, res, pos = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(256, N, generator=g)
    a, b = A.to(dt).cuda(), B.to(dt).cuda()
    base = A.double() @ (B.double() if tb else B.double().t())
    L = C.lib()
    assert L.enh_gemm_h16_workspace_bytes(0, int(tb), M, N, K) > 0
    assert L.enh_gemm_h16_workspace_bytes(0, int(tb), 131072, N, K) == 0
    outs = []
    for rep in range(2):
        o = torch.empty(M, N, device="cuda")
        C.gemm(a, b, M, N, K, trans_b=tb, out_f32=o)
        assert rel(o, base) <= F32_TOL
        x = res.clone().cuda()
        C.gemm(a, b, M, N, K, trans_b=tb, bias=bias.cuda(), res=x, res_rows=M, out_f32=x)          # in place on the residual stream
        assert rel(x, base + bias.double() + res.double()) <= F32_TOL
        p_ = torch.empty(M, N, device="cuda")
        C.gemm(a, b, M, N, K, trans_b=tb, bias=bias.cuda(), res=pos.cuda(), res_rows=256, out_f32=p_)
        assert rel(p_, base + bias.double() + pos.double().repeat(M // 256, 1)) <= F32_TOL
        acc = res.clone().cuda()
        C.gemm(a, b, M, N, K, trans_b=tb, bias=bias.cuda(), accumulate=True, out_f32=acc)
        assert rel(acc, base + bias.double() + res.double()) <= F32_TOL
        o16 = torch.empty(M, N, dtype=dt, device="cuda")                                          # a plain 16-bit output (the token-gradient GEMMs): summed in f32, packed once
        C.gemm(a, b, M, N, K, trans_b=tb, out_bf16=o16)
        assert rel(o16.float(), base) <= (BF16_TOL if dt == torch.bfloat16 else 3e-4)
        outs.append((o.clone(), x.clone(), p_.clone(), acc.clone(), o16.clone()))
    assert all(torch.equal(u, v) for u, v in zip(*outs))


@pytest.mark.parametrize("kind", ["fwd", "dgrad", "fwd_tanh", "dgrad_dtanh", "fwd_res", "fwd_f32"])
def test_persistent_gemm_is_bitwise_the_one_tile_kernel(C, kind):
    """gemm_w256p_kernel (one workgroup per CU walks the tiles; the next tile's operands are requested before this tile's stores) and
    gemm_w256r_kernel (the same with the A operand staged through registers) issue the same MFMA sequence per output element as
    gemm_w256_kernel and the same epilogue arithmetic: every output must be BIT-identical — on fewer tiles than CUs, on a ragged last round
    (300 tiles), on odd and minimal stage counts (w256r needs an even count >= 6: 448 / 64 = 7 falls back to w256p), in both B layouts and every
    fused mode; one case is also checked against fp64."""
    L = C.lib()
    g = torch.Generator().manual_seed(31)
    tb = kind.startswith("dgrad")
    try:
        for (m, n, k) in ((1024, 768, 192), (256 * 100, 768, 320), (256 * 37, 2304, 448), (256 * 50, 768, 384), (8192, 3072, 768)):
            if not os.environ.get("ENH_GEMM_KERNEL"):   # (a family override re-runs this file with that family pinned)
                want = "gemm_w256r_kernel" if (k // 64) % 2 == 0 and k // 64 >= 6 else "gemm_w256p_kernel"
                assert L.enh_gemm_h16_variant_mode(0, int(tb), 131072, n, k, 1).decode() == want    # the per-shape default at training sizes
            A, B = _mk((m, k), g, 0.5), _mk((n, k), g, 0.1)
            a = A.to(torch.bfloat16).cuda()
            b = (B.t().contiguous() if tb else B).to(torch.bfloat16).cuda()
            kw = dict(trans_b=tb)
            if kind in ("fwd", "dgrad"):
                out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); kw["out_bf16"] = out
            elif kind == "fwd_f32":
                out = torch.empty(m, n, device="cuda"); kw["out_f32"] = out
            elif kind == "fwd_tanh":
                out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); kw.update(out_bf16=out, bias=torch.randn(n, generator=g).cuda(), act=C.ACT_TANH)
            elif kind == "fwd_res":
                out = torch.empty(m, n, device="cuda"); kw.update(out_f32=out, bias=torch.randn(n, generator=g).cuda(), res=torch.randn(m, n, generator=g).cuda(), res_rows=m)
            else:
                out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
                kw.update(out_bf16=out, act=C.ACT_DTANH, aux=torch.tanh(torch.randn(m, n, generator=g)).to(torch.bfloat16).cuda())
            assert L.enh_gemm_set_kernel(7) == 0
            out.zero_(); C.gemm(a, b, m, n, k, **kw); torch.cuda.synchronize(); ref = out.clone()
            for fam in (8, 9):
                for dyn in (0, 1):       # static partition / tiles claimed from the per-XCD queues (round 4): the same bits
                    assert L.enh_gemm_set_kernel(fam) == 0 and L.enh_gemm_set_scheduler(dyn) == 0
                    out.fill_(7.0); C.gemm(a, b, m, n, k, **kw); torch.cuda.synchronize()
                    assert torch.equal(out, ref), f"family {fam} dyn {dyn} {kind} M={m} N={n} K={k}: {(out != ref).sum().item()} elements differ"
            if kind == "fwd_res" and m == 1024:
                assert rel(out, A.double() @ B.double().t() + kw["bias"].double().cpu() + kw["res"].double().cpu()) <= F32_TOL
    finally:
        L.enh_gemm_set_kernel(-1)
        L.enh_gemm_set_scheduler(1)


def test_dynamic_tile_schedule_under_cu_contention(C):
    """the claimed-tile schedule of the persistent GEMMs with a side-stream kernel HOLDING 24 CUs (the stand-in for a collective's channels,
    enh_debug_occupy_cus): workgroups that get their CU late find their queue empty — every tile is still computed exactly once (bit-identical output),
    over more launches than there are counter slots (64: each launch must leave its counters at zero), with and without a CU budget, for a
    register-staged (K = 768) and an LDS-DMA (K = 448) kernel and the tanh' mode that owns the whole LDS."""
    L = C.lib()
    g = torch.Generator().manual_seed(5)
    side = torch.cuda.Stream()
    try:
        for (m, n, k, mode) in ((256 * 40, 2304, 768, "bf16"), (256 * 23, 768, 448, "res"), (256 * 12, 3072, 768, "dtanh")):
            a = _mk((m, k), g, 0.5).to(torch.bfloat16).cuda()
            tb = mode == "dtanh"
            Bm = _mk((n, k), g, 0.1)
            b = (Bm.t().contiguous() if tb else Bm).to(torch.bfloat16).cuda()
            kw = dict(trans_b=tb)
            if mode == "bf16":
                out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); kw["out_bf16"] = out
            elif mode == "res":
                out = torch.empty(m, n, device="cuda"); kw.update(out_f32=out, bias=torch.randn(n, generator=g).cuda(), res=torch.randn(m, n, generator=g).cuda(), res_rows=m)
            else:
                out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
                kw.update(out_bf16=out, act=C.ACT_DTANH, aux=torch.tanh(torch.randn(m, n, generator=g)).to(torch.bfloat16).cuda())
            assert L.enh_gemm_set_kernel(7) == 0
            C.gemm(a, b, m, n, k, **kw); torch.cuda.synchronize(); ref = out.clone()
            assert L.enh_gemm_set_kernel(-1) == 0 and L.enh_gemm_set_scheduler(1) == 0
            for budget in (0, 232):
                C.set_cu_budget(budget)
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    C.occupy_cus(24, 400.0, side)
                for it in range(70):
                    out.fill_(3.0)
                    C.gemm(a, b, m, n, k, **kw)
                    if it % 23 == 0 or it == 69:
                        torch.cuda.synchronize()
                        assert torch.equal(out, ref), f"{mode} budget {budget} launch {it}: {(out != ref).sum().item()} elements differ"
                torch.cuda.synchronize()
    finally:
        C.set_cu_budget(0)
        L.enh_gemm_set_kernel(-1)
        L.enh_gemm_set_scheduler(1)


@pytest.mark.parametrize("kind", ["fwd", "fwd_tanh", "dgrad_dtanh", "fwd_res"])
def test_persistent_gemm_with_row_strides(C, kind):
    """leading dimensions larger than the row length (output, saved tanh output and residual stream living in wider buffers): the persistent kernels
    address rows through ldc / ldaux / ldres like the one-tile kernel — bit-identical results, and the columns beyond N stay untouched"""
    L = C.lib()
    g = torch.Generator().manual_seed(77)
    M, N, K, LD = 2048, 768, 768, 1024
    tb = kind.startswith("dgrad")
    a = _mk((M, K), g, 0.5).to(torch.bfloat16).cuda()
    B = _mk((N, K), g, 0.1)
    b = (B.t().contiguous() if tb else B).to(torch.bfloat16).cuda()
    kw = dict(trans_b=tb, ldc=LD)
    if kind == "fwd_res":
        out = torch.full((M, LD), 5.0, device="cuda"); kw.update(out_f32=out, bias=torch.randn(N, generator=g).cuda(), res=torch.randn(M, LD, generator=g).cuda(), res_rows=M)
    else:
        out = torch.full((M, LD), 5.0, dtype=torch.bfloat16, device="cuda"); kw["out_bf16"] = out
        if kind == "fwd_tanh":
            kw.update(bias=torch.randn(N, generator=g).cuda(), act=C.ACT_TANH)
        if kind == "dgrad_dtanh":
            kw.update(act=C.ACT_DTANH, aux=torch.tanh(torch.randn(M, LD, generator=g)).to(torch.bfloat16).cuda())
    try:
        assert L.enh_gemm_set_kernel(7) == 0
        C.gemm(a, b, M, N, K, **kw); torch.cuda.synchronize(); ref = out.clone()
        for fam in (8, 9):
            assert L.enh_gemm_set_kernel(fam) == 0
            out.fill_(5.0); C.gemm(a, b, M, N, K, **kw); torch.cuda.synchronize()
            assert torch.equal(out, ref), f"family {fam}: {(out != ref).sum().item()} elements differ"
        assert bool((out[:, N:] == 5.0).all()) and not bool((out[:, :N] == 5.0).all())
    finally:
        L.enh_gemm_set_kernel(-1)


@pytest.mark.parametrize("M,N,K", [(2048, 768, 192), (256 * 37, 3072, 768), (1000, 192, 256)])
def test_gemm_dtanh_with_fused_bias_gradient(C, M, N, K):
    """enh_gemm_h16_dtanh_colsum: C = (A B) * (1 - aux^2) exactly as enh_gemm_h16 with act = tanh' produces it (bit for bit), and the column sums of
    the STORED values as enh_colsum_h16_ws adds them (same inputs, another fixed summation order: f32 rounding only) — on the tile grid (epilogue
    partials + second pass) and off it (the two-call fallback); accumulate adds to the previous bias gradient; two runs are bit-identical."""
    g = torch.Generator().manual_seed(M + N + K)
    a = _mk((M, K), g, 0.5).to(torch.bfloat16).cuda()
    b = _mk((K, N), g, 0.1).to(torch.bfloat16).cuda()       # B stored [K][N] (trans_b): the input-gradient role
    h = torch.tanh(torch.randn(M, N, generator=g)).to(torch.bfloat16).cuda()
    ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    C.gemm(a, b, M, N, K, trans_b=True, act=C.ACT_DTANH, aux=h, out_bf16=ref)
    base = torch.randn(N, generator=g).cuda()
    runs = []
    for _ in range(2):
        out = torch.empty_like(ref)
        cs = base.clone()
        C.gemm_dtanh_colsum(a, b, M, N, K, h, out, cs, trans_b=True, accumulate_colsum=True)
        runs.append((out, cs))
    assert torch.equal(runs[0][0], ref)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    want = base.double() + ref.double().sum(0)
    assert rel(runs[0][1], want) <= 2e-6
    cs0 = torch.full((N,), 3.0, device="cuda")
    C.gemm_dtanh_colsum(a, b, M, N, K, h, out, cs0, trans_b=True, accumulate_colsum=False)
    assert rel(cs0, ref.double().sum(0)) <= 2e-6


def test_gemm_wgrad_splitk(C):
    """dW[N_out, K_in] += dY^T X over many tokens: the split-K (f32 atomics) path."""
    g = torch.Generator().manual_seed(9)
    tokens, n_out, k_in = 8192, 256, 192
    dY, X = _mk((tokens, n_out), g, 0.1), _mk((tokens, k_in), g)
    ref = dY.double().t() @ X.double()
    dW = torch.zeros(n_out, k_in, device="cuda")
    C.gemm(dY.to(torch.bfloat16).cuda(), X.to(torch.bfloat16).cuda(), n_out, k_in, tokens, trans_a=True, trans_b=True, accumulate=True, out_f32=dW)
    assert rel(dW, ref) <= 1e-4  # split-K: f32 atomics in arbitrary order over 8192-long sums
    C.gemm(dY.to(torch.bfloat16).cuda(), X.to(torch.bfloat16).cuda(), n_out, k_in, tokens, trans_a=True, trans_b=True, accumulate=True, out_f32=dW)
    assert rel(dW, 2 * ref) <= 1e-4


def test_gemm_wgrad_splitk_two_pass_is_deterministic(C):
    """256x256-tile split-K with the workspace: partial slabs + fixed-order second pass -> two runs are BIT-identical (the f32-atomic form is not),
    accumulate adds to the previous contents, and the workspace query matches the plan."""
    g = torch.Generator().manual_seed(10)
    tokens, n_out, k_in = 16384, 768, 768
    if not os.environ.get("ENH_GEMM_KERNEL"):   # (a family override re-runs this file with another kernel: then only the numerics are checked)
        assert C.lib().enh_gemm_h16_variant(1, 1, n_out, k_in, tokens).decode() == "gemm_w256_kernel"
    assert C.lib().enh_gemm_h16_workspace_bytes(1, 1, n_out, k_in, tokens) % (n_out * k_in * 4) == 0
    assert C.lib().enh_gemm_h16_workspace_bytes(1, 1, n_out, k_in, tokens) >= 2 * n_out * k_in * 4
    dY, X = _mk((tokens, n_out), g, 0.1).to(torch.bfloat16).cuda(), _mk((tokens, k_in), g).to(torch.bfloat16).cuda()
    ref = dY.double().t() @ X.double()
    base = torch.randn(n_out, k_in, generator=g).cuda()
    runs = []
    for _ in range(2):
        dW = base.clone()
        C.gemm(dY, X, n_out, k_in, tokens, trans_a=True, trans_b=True, accumulate=True, out_f32=dW)
        runs.append(dW)
    if not os.environ.get("ENH_GEMM_KERNEL"):
        assert torch.equal(runs[0], runs[1]), "two-pass split-K must be bit-reproducible"
    assert rel(runs[0], ref + base.double()) <= 1e-5


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_split_k_accumulate_is_bit_reproducible_in_every_layout(C, ta, tb):
    """an accumulate-into-f32 call whose few tiles cannot fill the chip is split along K — with the binding's workspace as partial slabs + a fixed-order second
    pass in EVERY operand layout (until round 4 only the weight-gradient layout: the discriminator's 8192 -> 512 linear at a small batch summed its K
    slices with f32 atomics and its output differed by an ulp from call to call)."""
    g = torch.Generator().manual_seed(9)
    M, N, K = 8, 512, 8192
    A, B = _mk((K, M) if ta else (M, K), g, 0.5), _mk((K, N) if tb else (N, K), g, 0.1)
    a, b = A.to(torch.bfloat16).cuda(), B.to(torch.bfloat16).cuda()
    assert C.lib().enh_gemm_h16_workspace_bytes(int(ta), int(tb), M, N, K) > 0      # the shape IS split
    outs = []
    for _ in range(12):
        o = torch.zeros(M, N, device="cuda")
        C.gemm(a, b, M, N, K, trans_a=ta, trans_b=tb, accumulate=True, out_f32=o)
        outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = (A.double().t() if ta else A.double()) @ (B.double() if tb else B.double().t())
    assert rel(outs[0], ref) <= F32_TOL


def test_gemm_rejects_bad_shapes(C):
    a = torch.zeros(8, 12, dtype=torch.bfloat16, device="cuda")
    o = torch.zeros(8, 8, device="cuda")
    with pytest.raises(RuntimeError):
        C.gemm(a, a, 8, 8, 12, out_f32=o)  # K % 8 != 0
    with pytest.raises(RuntimeError):
        C.gemm(a.cpu(), a, 8, 8, 8, out_f32=o)  # host tensor: no CPU fallback


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def _attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, N, H * 64), torch.logsumexp(s, dim=-1)


# kernel families per pass (include/enh_hip.h enh_attention_set_kernel): the library's default choice, the round-2 kernels, the software-pipelined
# round-3 kernels, and the round-2 skeletons with the statistics fed through the MFMA C operand
ATT_FAMILIES = [(0, 0, 0), (1, 1, 1), (1, 3, 2), (5, 1, 2), (5, 3, 1)]
LOG2E = 1.4426950408889634


@pytest.fixture(params=ATT_FAMILIES, ids=lambda f: "fam%d%d%d" % f)
def att_family(request, C):
    C.attention_set_kernel(*request.param)
    yield request.param
    C.attention_set_kernel(0, 0, 0)


def _prescale_q(qkv, H, scale):
    """the q_prescaled convention of include/enh_hip.h: the q third holds bf16(q * scale * log2e).  Returns (the tensor the kernels get, the UNSCALED
    fp64 qkv those bits represent — what the reference attention and its gradients are taken on)."""
    inner = H * 64
    dev = qkv.clone()
    dev[..., :inner] = bf16r(qkv[..., :inner] * (scale * LOG2E))
    ref = dev.double().clone()
    ref[..., :inner] /= (scale * LOG2E)
    return dev, ref


@pytest.mark.parametrize("pre", [False, True], ids=["plain", "prescaled"])
@pytest.mark.parametrize("B,N,H", [(2, 1024, 3), (3, 64, 2), (1, 256, 12), (2, 192, 1), (1, 128, 2), (1, 320, 1)])
def test_attention_forward_backward(C, att_family, B, N, H, pre):
    g = torch.Generator().manual_seed(B * 100 + N + H)
    qkv = bf16r(torch.randn(B, N, 3 * H * 64, generator=g) * 1.5)
    do = bf16r(torch.randn(B, N, H * 64, generator=g))
    scale = 64 ** -0.5
    qdev, qref = _prescale_q(qkv, H, scale) if pre else (qkv, qkv.double())
    qt = qref.clone().requires_grad_(True)
    ref, lse_ref = _attn_ref(qt, B, N, H, scale)
    ref.backward(do.double())
    qd = qdev.to(torch.bfloat16).cuda()
    out = torch.empty(B, N, H * 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    C.attention_forward(qd, B, N, H, scale, out, lse, q_prescaled=pre)
    assert rel(out.float(), ref) <= 1.5 * bf16_floor(ref)
    # lse: 1e-5 when the row sum is taken in fp32; family 2 sums the bf16 numerators on the matrix pipe (the normaliser of exactly what entered P V)
    assert rel(lse, lse_ref) <= 1e-5
    dqkv = torch.full((B, N, 3 * H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(B, H, N, device="cuda")
    C.attention_backward(qd, out, do.to(torch.bfloat16).cuda(), lse, B, N, H, scale, dqkv, delta, q_prescaled=pre)
    gq, gk, gv = qt.grad.view(B, N, 3, H * 64).unbind(2)
    dq, dk, dv = dqkv.float().view(B, N, 3, H * 64).unbind(2)
    assert rel(dq, gq) <= 2 * ATT_TOL and rel(dk, gk) <= 2 * ATT_TOL and rel(dv, gv) <= 2 * ATT_TOL, (rel(dq, gq), rel(dk, gk), rel(dv, gv))


@pytest.mark.parametrize("pre", [False, True], ids=["plain", "prescaled"])
@pytest.mark.parametrize("B,N,H", [(2, 64, 4), (2, 128, 4), (3, 192, 2), (4, 1024, 12)])
def test_attention_is_bit_reproducible_across_launches(C, att_family, B, N, H, pre):
    """Round 5: the round-4 forward took its row maxima through INLINE-ASM v_max3_f32 — invisible to the compiler's hazard recognizer, so the first one
    read the S product's destination registers before the matrix pipe had written them.  Any reference maximum yields a valid softmax (every parity test
    passed), but the result bits changed from launch to launch.  Every family, both conventions, forward and backward: identical bits on repeated launches
    with unrelated work in between."""
    g = torch.Generator(device="cuda").manual_seed(B + N + H)
    qkv = (torch.randn(B, N, 3 * H * 64, device="cuda", generator=g) * 1.2).to(torch.bfloat16)
    do = torch.randn(B, N, H * 64, device="cuda", generator=g).to(torch.bfloat16)
    runs = []
    for rep in range(5):
        out = torch.full((B, N, H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
        lse = torch.full((B, H, N), float("nan"), device="cuda")
        dqkv = torch.full_like(qkv, float("nan")); delta = torch.full((B, H, N), float("nan"), device="cuda")
        if rep % 2:
            torch.empty(1 << 24, device="cuda").normal_()        # unrelated work between the launches (other cache / clock state)
        C.attention_forward(qkv, B, N, H, 0.125, out, lse, q_prescaled=pre)
        C.attention_backward(qkv, out, do, lse, B, N, H, 0.125, dqkv, delta, q_prescaled=pre)
        torch.cuda.synchronize()
        runs.append((out, lse, dqkv))
    for k, name in enumerate(("out", "lse", "dqkv")):
        for r in runs[1:]:
            assert torch.equal(runs[0][k].view(torch.int16 if k != 1 else torch.int32), r[k].view(torch.int16 if k != 1 else torch.int32)), \
                f"{name}: {(runs[0][k] != r[k]).sum().item()} elements differ between two launches on the same input"


@pytest.mark.parametrize("pre", [False, True], ids=["plain", "prescaled"])
def test_attention_spiked_scores(C, att_family, pre):
    """keys dominating a row (force the running maximum / the pipelined kernels' reference maximum to jump mid-sweep, several times and in adjacent
    tiles, incl. the last one): the rescale path, checked row by row against fp64 (cdna_hip_programming.md T13: a wrong rescale order is silent on
    bounded random data and shows only in the rows that took the branch)."""
    B, N, H = 1, 512, 2
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, N, 3 * H * 64, generator=g)
    qkv[0, 5, :64] *= 6.0
    qkv[0, 300, H * 64:H * 64 + 64] = qkv[0, 5, :64] * 1.5   # key 300 of head 0 aligned with query 5
    # a staircase for query 70 of head 1: keys in tiles 1, 2, 3 and the last tile, each beating everything before it by a wide margin
    qv = qkv[0, 70, 64:128].clone()
    for key, gain in ((100, 2.0), (130, 4.0), (200, 7.0), (505, 11.0)):
        qkv[0, key, H * 64 + 64:H * 64 + 128] = qv * gain
    qkv = bf16r(qkv)
    qdev, qref = _prescale_q(qkv, H, 0.125) if pre else (qkv, qkv.double())
    ref, lse_ref = _attn_ref(qref, B, N, H, 0.125)
    out = torch.empty(B, N, H * 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    C.attention_forward(qdev.to(torch.bfloat16).cuda(), B, N, H, 0.125, out, lse, q_prescaled=pre)
    assert torch.isfinite(out.float()).all()
    assert rel(out.float(), ref) <= ATT_TOL
    assert rel(lse, lse_ref) <= 1e-5
    for q, h in ((5, 0), (70, 1)):          # the rows that took the branch, individually
        assert rel(out.float().cpu()[0, q, h * 64:(h + 1) * 64], ref[0, q, h * 64:(h + 1) * 64]) <= 2 * ATT_TOL, (q, h)
    # backward through the same spiked rows (lse comes from the kernel above)
    do = bf16r(torch.randn(B, N, H * 64, generator=g))
    qt = qref.clone().requires_grad_(True)
    r2, _ = _attn_ref(qt, B, N, H, 0.125)
    r2.backward(do.double())
    dqkv = torch.full((B, N, 3 * H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(B, H, N, device="cuda")
    C.attention_backward(qdev.to(torch.bfloat16).cuda(), out, do.to(torch.bfloat16).cuda(), lse, B, N, H, 0.125, dqkv, delta, q_prescaled=pre)
    assert torch.isfinite(dqkv.float()).all()
    assert rel(dqkv.float(), qt.grad) <= 3 * ATT_TOL, rel(dqkv.float(), qt.grad)


# ---------------------------------------------------------------------------------------------
# data movement, loss, reductions, optimizer
# ---------------------------------------------------------------------------------------------
def test_patchify_unpatchify_loss(C):
    import vitvq_oracle as O
    B, Cc, S, p = 3, 3, 64, 8
    g = torch.Generator().manual_seed(1)
    img = torch.rand(B, Cc, S, S, generator=g)
    out = torch.empty(B * (S // p) ** 2, Cc * p * p, dtype=torch.bfloat16, device="cuda")
    C.patchify(img.cuda(), p, out)
    assert torch.equal(out.cpu(), O.patchify(img, p).reshape(-1, Cc * p * p).to(torch.bfloat16))
    pix = torch.randn(B * (S // p) ** 2, Cc * p * p, generator=g)
    xrec = torch.empty(B, Cc, S, S, device="cuda")
    sums = torch.zeros(2, dtype=torch.float64, device="cuda")
    dpix = torch.empty_like(out)
    C.unpatchify_loss(pix.cuda(), img.cuda(), B, Cc, S, S, p, 0.3, 1.0, xrec, sums, dpix)
    ref = O.unpatchify(pix.view(B, -1, Cc * p * p), p, Cc, S, S)
    assert torch.equal(xrec.cpu(), ref)
    d = (ref - img).double()
    assert abs(sums[0].item() - d.abs().sum().item()) <= 1e-6 * d.abs().sum().item()
    assert abs(sums[1].item() - (d ** 2).sum().item()) <= 1e-6 * (d ** 2).sum().item()
    gref = O.patchify(((0.3 * torch.sign(d) + 2.0 * d) / d.numel()).float(), p).reshape(-1, Cc * p * p)
    assert rel(dpix.float(), gref) <= BF16_TOL


@pytest.mark.parametrize("M,N", [(5000, 192), (4099, 2304), (37, 3072), (1031, 520), (777, 130), (3, 8)])
def test_colsum_shapes(C, M, N):
    """wide kernel (N % 8 == 0: full, partial last 512-column block, fewer rows than one unrolled pass) and the 2-column fallback;
    overwrite and accumulate entry modes"""
    g = torch.Generator().manual_seed(M + N)
    x = bf16r(torch.randn(M, N, generator=g))
    xd = x.to(torch.bfloat16).cuda()
    out = torch.full((N,), 7.0, device="cuda")
    C.colsum(xd, M, N, out)
    ref = x.double().sum(0)
    assert rel(out, ref) <= F32_TOL
    C.colsum(xd, M, N, out, True)
    assert rel(out, 2 * ref) <= F32_TOL


def test_colsum_cast_adamw(C):
    import vitvq_oracle as O
    g = torch.Generator().manual_seed(2)
    x = bf16r(torch.randn(5000, 192, generator=g))
    out = torch.empty(192, device="cuda")
    C.colsum(x.to(torch.bfloat16).cuda(), 5000, 192, out)
    assert rel(out, x.double().sum(0)) <= F32_TOL
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = p.cuda(), gr.cuda(), m.cuda(), v.cuda()
    p16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for step in (1, 2, 3):
        O.adamw_step(p, gr, m, v, step, 4.5e-6)
        C.adamw_step(pd, gd, md, vd, p16, step, 4.5e-6)
    assert rel(pd, p) <= 1e-6 and rel(md, m) <= 1e-5 and rel(vd, v) <= 1e-5
    assert torch.equal(p16.cpu(), pd.cpu().to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------
# every GEMM kernel family on every shape it can serve (the per-shape default only exercises one of them)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sel,symbol", [("w256", "gemm_w256_kernel"), ("w256p", "gemm_w256_kernel"), ("w256r", "gemm_w256_kernel"), ("pipe2", "gemm_pipe2_kernel")])
def test_gemm_suite_under_each_kernel_family(sel, symbol):
    """the family override is process-global (enh_gemm_set_kernel, mapped from ENH_GEMM_KERNEL by the binding), so the GEMM tests are re-run
    in a child process per family; shapes a family cannot serve fall back to the per-shape choice"""
    import subprocess
    import sys
    env = dict(os.environ, ENH_GEMM_KERNEL=sel)
    probe = ("import sys; sys.path.insert(0, 'enhancing-transformers_amd'); from enhancing import _C; "
             "print(_C.lib().enh_gemm_h16_variant(0, 0, 4096, 4096, 4096).decode())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, "-c", probe], env=env, cwd=root, capture_output=True, text=True).stdout.strip() == symbol
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_ops_gpu.py", "-q", "-x", "-m", "gpu", "-k", "gemm and not under_each", "-p", "no:cacheprovider"],
                       env=env, cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


def test_head_scaled_cast_and_the_forward_backward_operand_gap(C):
    """to_qkv's FORWARD operand is bf16(alpha * W_q) (one rounding of the master, engine/stage1.py _Tower.refresh_qkv_operands) while the backward GEMMs
    read bf16(W_q): two roundings of the same fp32 number, so forward operand / alpha and backward operand differ by at most one bf16 ulp of W_q
    (relative 2^-7 worst case, 2^-9 rms) — the same size as the bf16 rounding each of them already carries against the master.  The cast itself is
    bit-exact against torch's round-to-nearest-even."""
    torch.manual_seed(5)
    inner, dim, alpha = 512, 768, 0.125 * 1.4426950408889634
    w = torch.randn(3 * inner, dim, device="cuda") * 0.02
    y = torch.empty(3 * inner, dim, dtype=torch.bfloat16, device="cuda")
    C.cast_bf16_head_scaled(w, y, inner * dim, alpha)
    torch.cuda.synchronize()
    expect = w.clone()
    expect[:inner] *= np.float32(alpha)
    assert torch.equal(y, expect.to(torch.bfloat16))
    # the strided form (every layer of a tower in one launch) writes the same bits
    store = torch.randn(5 * 3 * inner * dim + 64, device="cuda") * 0.02
    step = 3 * inner * dim + 16
    ys = torch.empty(3, 3 * inner, dim, dtype=torch.bfloat16, device="cuda")
    C.cast_bf16_head_scaled_strided(store[8:8 + 3 * inner * dim], step, ys, 3 * inner * dim, inner * dim, alpha)
    for b in range(3):
        one = torch.empty(3 * inner, dim, dtype=torch.bfloat16, device="cuda")
        C.cast_bf16_head_scaled(store[8 + b * step:8 + b * step + 3 * inner * dim].view(3 * inner, dim), one, inner * dim, alpha)
        assert torch.equal(ys[b], one)
    plain = w.to(torch.bfloat16).float()
    fwd_q = y[:inner].float() / np.float32(alpha)
    gap = (fwd_q - plain[:inner]).abs() / plain[:inner].abs().clamp_min(1e-30)
    assert float(gap.max()) <= 2.0 ** -7 and float(gap.pow(2).mean().sqrt()) <= 2.0 ** -8
    master_gap = (plain[:inner] - w[:inner]).abs() / w[:inner].abs().clamp_min(1e-30)      # what either operand already carries against the master
    assert float(gap.pow(2).mean().sqrt()) <= 2.0 * float(master_gap.pow(2).mean().sqrt())


def test_crop_flip_u8_device_transform():
    """enh_crop_flip_u8 (device-side crop + flip + ToTensor of the input pipeline, reference dataloader/imagenet.py:30-36) is bit-identical to the host path"""
    import numpy as np
    from enhancing import _C
    rs = np.random.RandomState(3)
    B, Hs, Ws, R = 5, 47, 61, 32
    src = rs.randint(0, 256, (B, Hs, Ws, 3)).astype(np.uint8)
    meta = np.stack([rs.randint(0, Hs - R + 1, B), rs.randint(0, Ws - R + 1, B), rs.randint(0, 2, B)], 1).astype(np.int32)
    out = _C.crop_flip_u8(torch.from_numpy(src).cuda(), torch.from_numpy(meta).cuda(), R).cpu().numpy()
    for b in range(B):
        y0, x0, flip = (int(v) for v in meta[b])
        w = src[b, y0:y0 + R, x0:x0 + R]
        w = w[:, ::-1] if flip else w
        assert np.array_equal(out[b], w.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)), b
    with pytest.raises(RuntimeError):
        _C.crop_flip_u8(torch.from_numpy(src).cuda(), torch.from_numpy(meta).cuda(), 64)      # window larger than the staging slot


def test_resize_u8_is_pillow_bit_for_bit():
    """enh_resize_u8 (device-side T.Resize of the input pipeline, reference dataloader/imagenet.py:31,49): a batch of ragged decoded images, shrinking and
    growing, training rule (shorter side -> R) and validation rule (exact (R, R)) — every output byte equals PIL.Image.resize(size, BILINEAR) and the CPU
    oracle; pixels outside an image's output rectangle stay untouched (zero)."""
    import numpy as np
    from PIL import Image
    import resize_oracle as RO
    from enhancing.dataloader.resize import resize_batch_u8
    rs = np.random.RandomState(5)
    sizes = [(500, 375), (333, 500), (100, 120), (64, 64), (480, 640), (257, 301), (37, 53)]
    HS, WS = max(s[0] for s in sizes), max(s[1] for s in sizes)
    src = np.zeros((len(sizes), HS, WS, 3), np.uint8)
    imgs = []
    for b, (h, w) in enumerate(sizes):
        imgs.append(rs.randint(0, 256, (h, w, 3)).astype(np.uint8))
        src[b, :h, :w] = imgs[b]
    for size in (256, (96, 128)):
        dst, outs = resize_batch_u8(torch.from_numpy(src).cuda(), sizes, size)
        dst = dst.cpu().numpy()
        for b, (ho, wo) in enumerate(outs):
            ref = np.array(Image.fromarray(imgs[b]).resize((wo, ho), Image.BILINEAR))
            assert np.array_equal(dst[b, :ho, :wo], ref), (size, b)
            assert np.array_equal(ref, RO.resize_u8(imgs[b], (ho, wo)))
            assert not dst[b, ho:].any() and not dst[b, :, wo:].any()
