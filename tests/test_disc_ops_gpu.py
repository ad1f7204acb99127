"""Discriminator native ops: HIP kernels vs the oracle restatement and the reference-produced golden vectors,
including first and second derivatives (the R1 penalty back-propagates through the discriminator's backward)."""
import os

import numpy as np
import pytest
import torch

from util import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from enhancing.losses import op
    return op


def _blur_kernel():
    k1 = torch.tensor([1., 3., 3., 1.])
    k = k1[None, :] * k1[:, None]
    return k / k.sum()


def test_golden_reference_vectors(ops, golden_dir):
    g = np.load(f"{golden_dir}/disc_ops.npz")
    rs = np.random.RandomState(int(g["seed"]))
    x = torch.from_numpy(rs.standard_normal((2, 3, 9, 7)).astype(np.float32)).cuda()
    k = _blur_kernel().cuda()
    for name, (up, down, pad) in {"blur_p22": (1, 1, (2, 2)), "blur_p11": (1, 1, (1, 1)), "up2": (2, 1, (2, 1)), "down2": (1, 2, (1, 1))}.items():
        out = ops.upfirdn2d(x, k, up=up, down=down, pad=pad)
        assert out.shape == g[name].shape and np.allclose(out.cpu().numpy(), g[name], atol=1e-6), name
    b = torch.from_numpy(rs.standard_normal(3).astype(np.float32)).cuda()
    assert np.allclose(ops.fused_leaky_relu(x, b, 0.2, 2 ** 0.5).cpu().numpy(), g["lrelu"], atol=1e-6)


@pytest.mark.parametrize("shape", [(4, 16, 33, 17), (2, 8), (3, 5, 12, 12)])
def test_fused_leaky_relu_first_and_second_derivative(ops, shape):
    import disc_ops_oracle as D
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g)
    b = torch.randn(shape[1], generator=g)
    go = torch.randn(*shape, generator=g)
    v = torch.randn(*shape, generator=g)  # direction for the second derivative

    def run(f, x, b, dev):
        x = x.to(dev).requires_grad_(True); b = b.to(dev).requires_grad_(True)
        y = f(x, b, 0.1, 1.7)
        gx, gb = torch.autograd.grad(y, (x, b), go.to(dev), create_graph=True)
        # second-order: d/dgo-style coupling as R1 does — differentiate (gx * v).sum() + gb.sum() w.r.t. the upstream grad carrier
        return y, gx, gb

    y, gx, gb = run(ops.fused_leaky_relu, x, b, "cuda")
    yo, gxo, gbo = run(D.fused_leaky_relu, x, b, "cpu")
    assert rel(y, yo) <= 1e-6 and rel(gx, gxo) <= 1e-6 and rel(gb, gbo) <= 1e-5
    # double backward through FusedLeakyReLUFunctionBackward: gradient of <gx, v> w.r.t. grad_output
    goc = go.cuda().requires_grad_(True)
    xc = x.cuda().requires_grad_(True)
    yc = ops.fused_leaky_relu(xc, b.cuda(), 0.1, 1.7)
    gxc, = torch.autograd.grad(yc, xc, goc, create_graph=True)
    gg, = torch.autograd.grad((gxc * v.cuda()).sum(), goc)
    gor = go.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = D.fused_leaky_relu(xr, b, 0.1, 1.7)
    gxr, = torch.autograd.grad(yr, xr, gor, create_graph=True)
    ggr, = torch.autograd.grad((gxr * v).sum(), gor)
    assert rel(gg, ggr) <= 1e-6


@pytest.mark.parametrize("pad", [(2, 2), (1, 1)])
@pytest.mark.parametrize("shape", [(2, 4, 32, 32), (3, 5, 17, 23)])
def test_blur_first_and_second_derivative(ops, shape, pad):
    import disc_ops_oracle as D
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g)
    k = _blur_kernel()
    xr = x.clone().requires_grad_(True)
    yr = D.upfirdn2d(xr, k, pad=pad)
    go = torch.randn(*yr.shape, generator=g)
    v = torch.randn(*shape, generator=g)
    gor = go.clone().requires_grad_(True)
    gxr, = torch.autograd.grad(yr, xr, gor, create_graph=True)
    ggr, = torch.autograd.grad((gxr * v).sum(), gor)
    xc = x.cuda().requires_grad_(True)
    yc = ops.upfirdn2d(xc, k.cuda(), pad=pad)
    goc = go.cuda().requires_grad_(True)
    gxc, = torch.autograd.grad(yc, xc, goc, create_graph=True)
    ggc, = torch.autograd.grad((gxc * v.cuda()).sum(), goc)
    assert yc.shape == yr.shape
    assert rel(yc, yr) <= 1e-6 and rel(gxc, gxr) <= 1e-6 and rel(ggc, ggr) <= 1e-6


def test_module_api(ops):
    m = ops.FusedLeakyReLU(8).cuda()
    assert m.bias.shape == (8,) and m.negative_slope == 0.2 and abs(m.scale - 2 ** 0.5) < 1e-12
    y = m(torch.randn(2, 8, 4, 4, device="cuda"))
    y.sum().backward()
    assert m.bias.grad is not None and m.bias.grad.shape == (8,)
