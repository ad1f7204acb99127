"""GPU parity of the implicit-GEMM convolution stack ("equalised-lr conv 3x3/1x1 stride 1/2 as implicit GEMM on MFMA with
weight-grad switch for R1"): every kernel of enhancing/losses/op/conv_nhwc.py against plain torch ops on IDENTICAL bf16-representable operands.

Tolerances: bf16 outputs must sit at the bf16 rounding floor of the exact result (<= 1.25x the floor computed in the test); the f32 weight gradient
(exact bf16 products, f32 accumulation over up to ~1e5 pixels, fixed summation order) within 2e-5; pure data movement bit-exact."""
import pytest
import torch
import torch.nn.functional as F

from util import bf16_floor, bf16r, rel

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from enhancing import _C
    from enhancing.losses.op import conv_nhwc
    _C.lib()
    return conv_nhwc


@pytest.fixture(params=["auto", "t256"])
def family(request):
    """kernel family of the implicit-GEMM convolutions: the per-shape choice, and the 256-row kernels wherever the shape allows them (at test sizes the
    per-shape choice never picks them: fewer tiles than CUs)"""
    from enhancing import _C
    _C.conv_set_kernel(request.param)
    yield request.param
    _C.conv_set_kernel("auto")


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


# B, H, W, Cin (real), Cout, k, stride, pad   — the discriminator's layer shapes in small, plus ragged sizes / padded channels
CASES = [(2, 16, 16, 3, 128, 1, 1, 0), (3, 7, 9, 5, 64, 1, 1, 0), (1, 40, 33, 8, 16, 1, 1, 0),   # the 8-channel pointwise kernels (conv_pointwise.hip)
         (2, 16, 20, 64, 128, 3, 1, 1), (3, 17, 17, 64, 136, 3, 2, 0), (2, 15, 15, 40, 64, 1, 2, 0),
         (8, 4, 4, 513, 512, 3, 1, 1), (1, 70, 66, 24, 72, 3, 1, 1), (1, 35, 35, 96, 128, 3, 2, 0), (2, 33, 33, 128, 256, 3, 2, 0),
         # geometries the discriminator never uses: padded stride 2, stride 3 (nine parity classes), a 5 x 5 kernel
         (2, 10, 8, 24, 64, 3, 2, 1), (1, 11, 10, 16, 32, 3, 3, 1), (2, 6, 6, 8, 16, 5, 2, 2), (2, 12, 12, 64, 64, 3, 2, 1),
         # shapes the 256-row kernels take (C % 64 == 0, N % 128 == 0): 256 x 256 and 256 x 128 tiles, ragged last row tile, two column tiles, rows shorter
         # than the 8-pixel staging step, padded stride 2 (strided output rows in the input gradient)
         (2, 24, 24, 128, 256, 3, 1, 1), (1, 20, 12, 256, 512, 3, 1, 1), (5, 6, 5, 128, 128, 3, 1, 1), (2, 19, 18, 128, 256, 3, 2, 1),
         # ... and the 256-row weight gradient (grid width % 64 == 0, C % 128 == 0, Cout % 256 == 0, >= 16 K stages): ragged and whole column tiles, stride 2
         (2, 8, 64, 128, 256, 3, 1, 1), (3, 8, 64, 256, 256, 3, 1, 1), (2, 17, 129, 128, 256, 3, 2, 0)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", CASES)
def test_conv_forward_dgrad_wgrad(ops, family, B, H, W, Cin, Cout, k, s, p):
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    Cp = ops.pad8(Cin)
    x = bf16r(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g)
    scale = 1.0 / (Cin * k * k) ** 0.5
    ws = bf16r(w * scale)                                   # what the pack kernel hands to the MFMA
    xr, wr = x.clone().requires_grad_(True), ws.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=s, padding=p)
    dy = bf16r(torch.randn(yr.shape, generator=g))
    yr.backward(dy)
    xp = torch.zeros(B, H, W, Cp)
    xp[..., :Cin] = _nhwc(x)
    xd = xp.to(BF).cuda().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    y = ops.conv(xd, wd, scale, s, p)
    assert y.shape == (B, yr.shape[2], yr.shape[3], Cout) and y.dtype == BF
    y.backward(_nhwc(dy).to(BF).cuda())
    e_y, f_y = rel(_nchw(y.float()), yr), bf16_floor(yr.detach())
    gx = _nchw(xd.grad.float())
    e_x, f_x = rel(gx[:, :Cin], xr.grad), bf16_floor(xr.grad)
    # d/dw of conv(x, scale*w) = scale * (d/d ws)
    e_w = rel(wd.grad, scale * wr.grad)
    print(f"conv {B}x{H}x{W} {Cin}->{Cout} k{k} s{s} p{p}: y {e_y:.2e} (floor {f_y:.2e})  dx {e_x:.2e} (floor {f_x:.2e})  dw {e_w:.2e}")
    assert e_y <= 1.25 * f_y and e_x <= 1.25 * f_x and e_w <= 2e-5
    if Cp > Cin:
        assert not gx[:, Cin:].abs().sum().item()           # gradient of the zero padding channels: zero rows of the transposed weights


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(2, 16, 16, 64, 128, 3, 1, 1), (2, 17, 17, 32, 64, 3, 2, 0), (2, 15, 15, 32, 64, 1, 2, 0), (2, 17, 17, 128, 256, 3, 2, 0)])
def test_fused_epilogues_and_second_order(ops, family, B, H, W, Cin, Cout, k, s, p):
    """conv + bias + leaky-ReLU and conv + residual merge in one kernel: values, first derivatives, and the R1-style second-order term
    d/d(w, bias) of |d out / d x|^2 (differentiates THROUGH _Dgrad and _Gate) against torch autograd on the same operands"""
    from enhancing.losses.op import conv2d_gradfix
    g = torch.Generator().manual_seed(5 + Cin)
    x = bf16r(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g)
    b = torch.randn(Cout, generator=g)
    scale = 1.0 / (Cin * k * k) ** 0.5

    wsr = bf16r(w * scale).clone().requires_grad_(True)       # what the pack kernel hands to the MFMA; d/dw = scale * d/d(wsr)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.leaky_relu(F.conv2d(xr, wsr, stride=s, padding=p) + br.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    xd, wd, bd = _nhwc(x).to(BF).cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv_bias_lrelu(xd, wd, bd, scale, s, p)
    assert rel(_nchw(y.float()), yr) <= 1.25 * bf16_floor(yr.detach())
    # residual merge: alpha * (conv(x, scale*w) + add) with alpha folded into the weights
    add = bf16r(torch.randn(yr.shape, generator=g))
    zr = F.conv2d(x, bf16r(w * scale * 0.5), stride=s, padding=p) + 0.5 * add
    z = ops.conv_add(xd.detach(), wd.detach(), _nhwc(add).to(BF).cuda(), scale * 0.5, s, p, 0.5)
    assert rel(_nchw(z.float()), zr) <= 1.25 * bf16_floor(zr)
    # first order (all three gradients), then the second-order term
    gy = bf16r(torch.randn(yr.shape, generator=g))
    y.backward(_nhwc(gy).to(BF).cuda(), retain_graph=True)
    yr.backward(gy, retain_graph=True)
    e_w, e_b = rel(wd.grad, scale * wsr.grad), rel(bd.grad, br.grad)
    assert rel(_nchw(xd.grad.float()), xr.grad) <= 1.7 * bf16_floor(xr.grad) and e_w <= 5e-3 and e_b <= 5e-3, (e_w, e_b)   # g_pre AND dx are stored in bf16: two roundings
    wd.grad = None; wsr.grad = None; bd.grad = None
    with conv2d_gradfix.no_weight_gradients():
        gxd, = torch.autograd.grad(y, xd, _nhwc(gy).to(BF).cuda(), create_graph=True)
    gxr, = torch.autograd.grad(yr, xr, gy, create_graph=True)
    e1 = rel(_nchw(gxd.float()), gxr)
    gxd.float().square().sum().backward()
    gxr.square().sum().backward()
    e2 = rel(wd.grad, scale * wsr.grad)
    print(f"fused conv k{k} s{s}: dw {e_w:.2e} db {e_b:.2e} ; R1-style pass: dx {e1:.2e}, second-order dw {e2:.2e}")
    assert e1 <= 1.7 * bf16_floor(gxr.detach()) and e2 <= 2e-2      # the second-order term goes through two bf16-stored intermediates
    assert bd.grad is None or not bd.grad.abs().sum().item()        # the gate is piecewise constant: no second-order bias term


@pytest.mark.parametrize("pad", [(2, 2), (1, 1)])
@pytest.mark.parametrize("shape", [(2, 32, 32, 16), (3, 17, 23, 40), (1, 258, 257, 8)])
def test_blur_and_its_adjoint(ops, shape, pad):
    import disc_ops_oracle as DO
    g = torch.Generator().manual_seed(2)
    k1 = torch.tensor([1., 3., 3., 1.])
    kern = k1[None, :] * k1[:, None]
    kern = kern / kern.sum()
    kern[0, 1] += 0.01                                       # asymmetric, so that the flip conventions are actually tested
    x = bf16r(torch.randn(*shape, generator=g))
    xr = _nchw(x).clone().requires_grad_(True)
    yr = DO.upfirdn2d(xr, kern, pad=pad)
    gy = bf16r(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = x.to(BF).cuda().requires_grad_(True)
    y = ops.blur(xd, kern.cuda(), pad)
    y.backward(_nhwc(gy).to(BF).cuda())
    assert rel(_nchw(y.float()), yr) <= 1.25 * bf16_floor(yr.detach())
    assert rel(_nchw(xd.grad.float()), xr.grad) <= 1.25 * bf16_floor(xr.grad)


@pytest.mark.parametrize("shape,pad", [((2, 32, 32, 16), (2, 2)), ((3, 17, 23, 40), (1, 1)), ((1, 258, 257, 8), (2, 2)), ((2, 5, 9, 64), (2, 1)), ((16, 67, 66, 128), (2, 2)),
                                       ((1, 3, 2, 8), (2, 2))])
def test_marching_blur_equals_the_one_row_kernel_bit_for_bit(ops, shape, pad):
    """the 4 x 4 Blur as a column march (one load per input element and strip instead of four) adds every output's 16 products in the order of the
    one-row kernel: identical bits, for both flip conventions, ragged strips / column groups and padding on every side"""
    from enhancing import _C
    g = torch.Generator().manual_seed(11)
    kern = torch.rand(4, 4, generator=g).cuda()
    x = torch.randn(*shape, generator=g).to(BF).cuda()
    for flip in (False, True):
        _C.blur_set_kernel(1)
        try:
            ref = _C.blur_nhwc(x, kern, pad[0], pad[1], flip)
        finally:
            _C.blur_set_kernel(0)
        got = _C.blur_nhwc(x, kern, pad[0], pad[1], flip)
        assert got.shape == ref.shape and torch.equal(got, ref), (shape, pad, flip, (got.float() - ref.float()).abs().max().item())


def test_elementwise_pieces(ops):
    from enhancing import _C
    g = torch.Generator().manual_seed(3)
    a, r = bf16r(torch.randn(3, 9, 11, 24, generator=g)), bf16r(torch.randn(3, 9, 11, 24, generator=g))
    y = _C.lrelu_gate(a.to(BF).cuda(), r.to(BF).cuda(), 0.2, 1.7)
    want = (a * (torch.where(r > 0, 1.0, 0.2) * 1.7)).to(BF)       # the kernel's association: g * (gate * scale)
    assert torch.equal(y.cpu(), want)
    assert torch.equal(_C.lrelu_gate(a.to(BF).cuda(), None, 1.0, 0.5).cpu(), (a * 0.5).to(BF))
    img = torch.rand(2, 3, 13, 17, generator=g)
    n8 = _C.img_to_nhwc8(img.cuda())
    assert n8.shape == (2, 13, 17, 8) and torch.equal(n8[..., :3].cpu(), _nhwc(img).to(BF)) and not n8[..., 3:].float().abs().sum().item()
    back = _C.nhwc8_to_img(n8, 3)
    assert torch.equal(back.cpu(), bf16r(img))
    for shape in [(4, 64, 64, 128), (2, 5, 7, 40), (16, 4, 4, 512)]:
        t = bf16r(torch.randn(*shape, generator=g))
        assert rel(_C.colsum_nhwc(t.to(BF).cuda()), t.double().reshape(-1, shape[-1]).sum(0)) <= 1e-5


@pytest.mark.parametrize("B,group,C", [(8, 4, 512), (6, 3, 64), (16, 4, 512), (2, 2, 24)])
def test_minibatch_stddev(ops, B, group, C):
    g = torch.Generator().manual_seed(B + C)
    x = bf16r(torch.randn(B, 4, 4, C, generator=g))
    Cp = ops.pad8(C + 1)
    xr = x.clone().requires_grad_(True)
    yr = ops._stddev_torch(xr, group, Cp)
    gy = bf16r(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = x.to(BF).cuda().requires_grad_(True)
    y = ops.minibatch_stddev(xd, group)
    assert y.shape == (B, 4, 4, Cp) and torch.equal(y[..., :C].cpu(), x.to(BF)) and not y[..., C + 1:].float().abs().sum().item()
    assert rel(y[..., C].float(), yr[..., C]) <= 4e-3
    y.backward(gy.to(BF).cuda())
    assert rel(xd.grad.float(), xr.grad) <= 1.25 * bf16_floor(xr.grad)


def test_lowerings_agree_at_full_size(ops):
    """StyleDiscriminator(size=256) on the implicit-GEMM path against the explicit im2col lowering (the round-1 path, itself pinned to the reference's
    golden vectors at size 16): logits, d logits / d image and every parameter gradient of a d-loss step — both run bf16 operands, so they agree to the
    bf16 level, which a wrong 64-bit offset / tile-edge / parity-class bug at the real layer sizes would not"""
    from enhancing.engine.stage1 import ParamStore
    from enhancing.losses.layers import StyleDiscriminator, vanilla_d_loss
    torch.manual_seed(0)
    D = StyleDiscriminator(size=256)
    with torch.no_grad():
        for n, p in D.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape))
    dev = torch.device("cuda")
    D.to(dev)
    store = ParamStore(D, dev, precision="fp32")
    x = torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for low in ("igemm", "im2col"):
        D.lowering = low
        store.zero_grad()
        xi = x.clone().requires_grad_(True)
        logits = D(xi)
        loss = vanilla_d_loss(-logits, logits)
        loss.backward()
        res[low] = (logits.detach().clone(), xi.grad.clone(), {n: p.grad.clone() for n, p in D.named_parameters()})
    e_l, e_x = rel(res["igemm"][0], res["im2col"][0]), rel(res["igemm"][1], res["im2col"][1])
    e_p = {n: rel(res["igemm"][2][n], res["im2col"][2][n]) for n in res["igemm"][2]}
    worst = max(e_p, key=e_p.get)
    print(f"igemm vs im2col at 256px: logits {e_l:.2e}, dx {e_x:.2e}, worst parameter gradient {worst} {e_p[worst]:.2e}")
    assert e_l <= 3e-2 and e_x <= 0.2 and e_p[worst] <= 0.25
