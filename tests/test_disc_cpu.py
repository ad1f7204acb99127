"""CPU tests for the discriminator row: the oracle against the golden vectors produced by the reference's own
layers.py, and the autograd wiring of the HIP-lowered discriminator with the kernels replaced by test-only torch stand-ins."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import disc_case, rel


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "disc_tiny.npz"))


def test_disc_oracle_matches_reference_golden(gold):
    """parameters regenerated from the seed through THIS package's module classes (same construction order as the reference's, so the
    same RNG stream) -> the oracle reproduces the reference's logits, R1 penalty, d-loss and gradients"""
    import disc_oracle as O
    D, real, fake = disc_case(gold)
    size = int(gold["size"])
    sd = {k: v.clone().requires_grad_("kernel" not in k) for k, v in D.state_dict().items()}
    d_loss, lr, lf, r1 = O.discriminator_loss(sd, size, real, fake, True)
    d_loss.backward()
    assert rel(lr, torch.from_numpy(gold["logits_real"])) <= 1e-5 and rel(lf, torch.from_numpy(gold["logits_fake"])) <= 1e-5
    assert abs(d_loss.item() - float(gold["d_loss"])) <= 1e-5 * abs(float(gold["d_loss"]))
    assert abs(r1.item() - float(gold["r1"])) <= 1e-4 * float(gold["r1"])
    for i, n in enumerate(gold["grad_names"]):
        assert abs(sd[str(n)].grad.double().norm().item() - gold["grad_norms"][i]) <= 1e-4 * gold["grad_norms"][i], n
    assert rel(sd["blocks.0.0.weight"].grad, torch.from_numpy(gold["g_rgb_w"])) <= 1e-4


@pytest.mark.parametrize("lowering", ["igemm", "im2col"])
@pytest.mark.parametrize("B", [8, 6, 2])
def test_discriminator_wiring_first_and_second_order(monkeypatch, B, lowering):
    """kernels replaced by exact torch stand-ins: logits, d logits / d image, and the parameter gradients of (R1 penalty + d-loss)
    — which differentiate THROUGH the first backward — must equal the fp32 oracle's to rounding"""
    import disc_oracle as O
    import hip_emulation
    hip_emulation.install(monkeypatch, exact=True)
    from enhancing import _C
    from enhancing.losses.layers import StyleDiscriminator
    from enhancing.losses.op import conv2d_gradfix
    # The igemm path folds the residual merge's 1/sqrt(2) into conv2's activation gain (one rounding where the reference has two), which moves later
    # pre-activations by an ulp; a leaky-ReLU gate whose pre-activation lies within ~1e-7 of zero then flips and the comparison "to rounding" is off
    # by that one gate (seen with seed 8: |pre-activation| 5e-10, one gate of 65536).  The stand-in records the smallest |activation| it produced and
    # the test asserts the margin, so a seed is only accepted if no such near-tie exists.
    margin = [float("inf")]
    inner = _C.conv_nhwc

    def conv_nhwc_recording(src, wt, geom, mode, **kw):
        out = inner(src, wt, geom, mode, **kw)
        if mode == 3:
            margin[0] = min(margin[0], float(out.detach().abs().min()))
        return out
    monkeypatch.setattr(_C, "conv_nhwc", conv_nhwc_recording)
    torch.manual_seed(B + 20 if lowering == "igemm" else B)
    D = StyleDiscriminator(size=16, lowering=lowering)
    with torch.no_grad():
        for n, p in D.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape))
    x = torch.rand(B, 3, 16, 16)
    x1 = x.clone().requires_grad_(True)
    y1 = D(x1)
    with conv2d_gradfix.no_weight_gradients():
        g1, = torch.autograd.grad(y1.sum(), x1, create_graph=True)
    (80 * g1.square().sum([1, 2, 3]).mean() + F.softplus(-y1).mean()).backward()
    sd = {k: v.detach().clone().requires_grad_("kernel" not in k) for k, v in D.state_dict().items()}
    x2 = x.clone().requires_grad_(True)
    y2 = O.discriminator(sd, x2, 16)
    g2, = torch.autograd.grad(y2.sum(), x2, create_graph=True)
    (80 * g2.square().sum([1, 2, 3]).mean() + F.softplus(-y2).mean()).backward()
    assert lowering != "igemm" or margin[0] > 2e-8, margin       # 0.2 x slope x the ~1e-7 an ulp moves a pre-activation
    assert rel(y1, y2) <= 1e-5 and rel(g1, g2) <= 1e-5
    for n, p in D.named_parameters():
        assert rel(p.grad, sd[n].grad) <= 2e-5, n


def test_layer_classes_keep_the_reference_nchw_api(monkeypatch):
    import disc_ops_oracle as DO
    import hip_emulation
    hip_emulation.install(monkeypatch, exact=True)
    from enhancing.losses.layers import ConvLayer, EqualLinear, StyleBlock
    torch.manual_seed(0)
    x = torch.randn(8, 16, 8, 8)
    blk = StyleBlock(16, 24)
    y = blk(x)
    s = 1 / (16 * 9) ** 0.5
    o = DO.fused_leaky_relu(F.conv2d(x, blk.conv1[0].weight * s, padding=1), blk.conv1[1].bias)
    o = DO.fused_leaky_relu(F.conv2d(DO.upfirdn2d(o, blk.conv2[0].kernel, pad=(2, 2)), blk.conv2[1].weight * s, stride=2), blk.conv2[2].bias)
    sk = F.conv2d(DO.upfirdn2d(x, blk.skip[0].kernel, pad=(1, 1)), blk.skip[1].weight * (1 / 4), stride=2)
    assert y.shape == (8, 24, 4, 4) and rel(y, (o + sk) / 2 ** 0.5) <= 1e-5
    c = ConvLayer(16, 8, 3, activate=False)   # bias on the convolution itself
    assert rel(c(x), F.conv2d(x, c[0].weight * s, c[0].bias, padding=1)) <= 1e-5
    lin = EqualLinear(40, 5, bias_init=0.3, lr_mul=0.5)
    z = torch.randn(3, 40)
    assert rel(lin(z), F.linear(z, lin.weight * lin.scale, lin.bias * lin.lr_mul)) <= 1e-5


def test_loss_module_generator_and_discriminator_sides(monkeypatch):
    """VQLPIPSWithDiscriminator.forward, both optimizer indices, against the oracle's restatement of vqperceptual.py:111-172: loss values,
    log keys, lazy R1 only on batches with batch_idx % do_r1_every == 0, disc_start gating"""
    import disc_oracle as O
    import hip_emulation
    hip_emulation.install(monkeypatch, exact=True)
    from enhancing.losses.vqperceptual import VQLPIPSWithDiscriminator
    torch.manual_seed(3)
    L = VQLPIPSWithDiscriminator(disc_start=5, loglaplace_weight=0.5, loggaussian_weight=1.0, perceptual_weight=0.0, adversarial_weight=0.1,
                                 disc_params={"size": 16}, do_r1_every=4)
    L.train()
    x, r = torch.rand(4, 3, 16, 16), torch.rand(4, 3, 16, 16)
    q = torch.tensor(0.3)
    sd = {k: v.detach().clone() for k, v in L.discriminator.state_dict().items()}
    for step, factor in ((0, 0), (7, 1)):
        rr = r.clone().requires_grad_(True)
        loss, log = L(q, x, rr, 0, step, 0, split="train")
        o_loss, o_g = O.generator_loss(sd, 16, q, x, r, 0.5, 1.0, 0.1, 1.0, factor)
        assert abs(loss.item() - o_loss.item()) <= 1e-5 and abs(log["train/g_loss"].item() - o_g.item()) <= 1e-5
        assert set(log) == {"train/total_loss", "train/quant_loss", "train/rec_loss", "train/loglaplace_loss", "train/loggaussian_loss",
                            "train/perceptual_loss", "train/g_loss"}
    for batch_idx, step, want_r1 in ((0, 7, True), (1, 7, False), (0, 0, False)):
        d_loss, log = L(q, x, r, 1, step, batch_idx, split="train")
        o_d, o_lr, o_lf, o_r1 = O.discriminator_loss(sd, 16, x, r, want_r1, 10.0, 4, 1 if step >= 5 else 0)
        assert abs(float(d_loss.detach()) - float(o_d.detach())) <= 1e-4 * max(1.0, abs(float(o_d.detach())))
        assert ("train/r1_reg" in log) == want_r1
        assert abs(log["train/logits_real"].item() - o_lr.mean().item()) <= 1e-5
        if want_r1:
            assert abs(log["train/r1_reg"].item() - o_r1.item()) <= 1e-4 * o_r1.item()
    L.eval()
    _, log = L(q, x, r, 1, 7, 0, split="val")
    assert "val/r1_reg" not in log and set(log) == {"val/disc_loss", "val/logits_real", "val/logits_fake"}


def test_packed_weight_cache_follows_the_weights(monkeypatch):
    """op/conv_nhwc.py caches the packed bf16 operand images of nn.Parameters between optimizer steps: reused while the weights are unchanged, rebuilt
    after an in-place torch update (autograd version), after invalidate_packed_weights() (what the fused AdamW calls: it writes through raw pointers) and
    for a different parameter that happens to get a recycled address; plain tensors (second-order weight-shaped gradients) are never cached"""
    import hip_emulation
    hip_emulation.install(monkeypatch, exact=True)
    from enhancing import _C
    from enhancing.losses.op import conv_nhwc
    calls = []
    real = _C.conv_pack_weight
    monkeypatch.setattr(_C, "conv_pack_weight", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    conv_nhwc.invalidate_packed_weights()
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(16, 8, 3, 3))
    x = torch.randn(2, 6, 6, 8)
    y0 = conv_nhwc.conv(x, w, 0.5, 1, 1)
    y1 = conv_nhwc.conv(x, w, 0.5, 1, 1)
    assert len(calls) == 1 and torch.equal(y0, y1)
    with torch.no_grad():
        w.mul_(2.0)                                       # in-place torch write: the version counter moves
    y2 = conv_nhwc.conv(x, w, 0.5, 1, 1)
    assert len(calls) == 2 and rel(y2, 2 * y0) <= 1e-6
    w.data.mul_(0.5)                                      # a writer the version counter does not see ...
    assert torch.equal(conv_nhwc.conv(x, w, 0.5, 1, 1), y2) and len(calls) == 2      # ... is served the stale image, which is why
    conv_nhwc.invalidate_packed_weights()                                            # raw-pointer writers must invalidate
    assert rel(conv_nhwc.conv(x, w, 0.5, 1, 1), y0) <= 1e-6 and len(calls) == 3
    conv_nhwc.conv(x, w.detach() * 1.0, 0.5, 1, 1)        # not a Parameter: packed every time
    conv_nhwc.conv(x, w.detach() * 1.0, 0.5, 1, 1)
    assert len(calls) == 5


@pytest.mark.parametrize("H,W,k,s,p", [(9, 11, 3, 1, 1), (9, 11, 3, 2, 0), (10, 8, 3, 2, 1), (7, 7, 1, 2, 0), (8, 9, 1, 1, 0), (11, 10, 3, 3, 1), (6, 6, 5, 2, 2),
                                       (5, 4, 3, 1, 0), (4, 4, 3, 2, 1)])
def test_conv_geometry_of_every_role_against_torch(monkeypatch, H, W, k, s, p):
    """the enh_conv_geom structs op/conv_nhwc.py builds for the forward, the input gradient (one launch per output parity class when stride > 1, classes
    without taps writing zeros) and the weight gradient, interpreted tap by tap by the test-only stand-in, against F.conv2d and its autograd — including
    strides, paddings and ragged sizes the discriminator itself never uses"""
    import hip_emulation
    hip_emulation.install(monkeypatch, exact=True)
    from enhancing.losses.op import conv_nhwc
    g = torch.Generator().manual_seed(H * 100 + W * 10 + k + s + p)
    B, Cin, Cout = 2, 5, 16
    Cp = conv_nhwc.pad8(Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, 0.3 * wr, stride=s, padding=p)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xp = torch.zeros(B, H, W, Cp)
    xp[..., :Cin] = x.permute(0, 2, 3, 1)
    xd, wd = xp.requires_grad_(True), w.clone().requires_grad_(True)
    y = conv_nhwc.conv(xd, wd, 0.3, s, p)
    assert rel(y.permute(0, 3, 1, 2), yr) <= 1e-6
    y.backward(dy.permute(0, 2, 3, 1).contiguous())
    assert rel(xd.grad[..., :Cin].permute(0, 3, 1, 2), xr.grad) <= 1e-6 and not xd.grad[..., Cin:].abs().sum().item()
    assert rel(wd.grad, wr.grad) <= 1e-6


@pytest.mark.parametrize("pad", [(2, 2), (1, 1), (2, 1), (0, 3), (-1, 2)])
def test_nhwc_blur_and_its_adjoint_against_the_ops_oracle(monkeypatch, pad):
    """conv_nhwc.blur = upfirdn2d(x, k, pad) with unit factors; its backward is the same kernel with the taps in the other order and pads kh-1-pad
    (negative pads = cropping included), checked against autograd through the oracle's upfirdn2d"""
    import disc_ops_oracle as DO
    import hip_emulation
    hip_emulation.install(monkeypatch, exact=True)
    from enhancing.losses.op import conv_nhwc
    g = torch.Generator().manual_seed(7)
    kern = torch.rand(4, 4, generator=g)
    x = torch.randn(2, 9, 10, 8, generator=g)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    crop0, crop1 = max(-pad[0], 0), max(-pad[1], 0)
    xc = xr[:, :, crop0:xr.shape[2] - crop1, crop0:xr.shape[3] - crop1]
    yr = DO.upfirdn2d(xc, kern, pad=(max(pad[0], 0), max(pad[1], 0)))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.clone().requires_grad_(True)
    y = conv_nhwc.blur(xd, kern, pad)
    assert rel(y.permute(0, 3, 1, 2), yr) <= 1e-6
    y.backward(gy.permute(0, 2, 3, 1).contiguous())
    assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) <= 1e-6
