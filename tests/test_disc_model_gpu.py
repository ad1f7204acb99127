"""GPU parity of the discriminator row: the convolution lowering (im2col / col2im / bf16 MFMA GEMM), the assembled
StyleGAN2 discriminator against the golden vectors produced by the REFERENCE's own layers.py (fp32), and the two-optimizer step protocol.

Tolerances.  Operator level: exact for the data movement, fp32-rounding for f32 outputs given IDENTICAL (bf16-representable) operands, the
bf16 rounding floor for bf16 outputs.  Model level, against the reference's fp32 arithmetic: the discriminator rounds every convolution
operand to bf16, and each rounding can flip leaky-ReLU gates of a random-initialised network, which moves gradients far more than values —
predicted with the kernels emulated in torch (tests/hip_emulation.py, exact=False): logits 8e-3, d logits/d image 6e-2, R1 3e-3, parameter
gradient norms 2e-2, single gradient tensors up to 1e-1.  The bounds below are about 2x those."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import bf16r, disc_case, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    from enhancing import _C
    _C.lib()
    return _C


GEOMS = [  # layout, B, C, H, W, k, stride, pad
    ("nchw", 2, 3, 16, 16, 1, 1, 0), ("cm", 2, 40, 12, 20, 3, 1, 1), ("cm", 3, 64, 17, 17, 3, 2, 0), ("cm", 2, 33, 15, 15, 1, 2, 0),
    ("cm", 8, 513, 4, 4, 3, 1, 1), ("nchw", 1, 5, 70, 70, 3, 1, 1), ("cm", 1, 96, 35, 35, 3, 2, 0)]


@pytest.mark.parametrize("layout,B,Cc,H,W,k,s,p", GEOMS)
def test_im2col_col2im(C, layout, B, Cc, H, W, k, s, p):
    g = torch.Generator().manual_seed(B * 100 + Cc)
    img = torch.randn(B, Cc, H, W, generator=g)
    x = (img if layout == "nchw" else img.permute(1, 0, 2, 3).contiguous()).cuda()
    sb, sc = (Cc * H * W, H * W) if layout == "nchw" else (H * W, B * H * W)
    cols = C.im2col(x, sb, sc, B, Cc, H, W, k, s, p)
    Ho, Wo = C.conv_out_size(H, k, s, p), C.conv_out_size(W, k, s, p)
    ref = F.unfold(img, k, padding=p, stride=s).permute(0, 2, 1).reshape(B * Ho * Wo, Cc * k * k)
    assert cols.shape == (B * Ho * Wo, (Cc * k * k + 7) // 8 * 8)
    assert torch.equal(cols[:, :Cc * k * k].cpu(), ref.to(torch.bfloat16))          # pure data movement + RNE rounding: bit-exact
    assert not cols[:, Cc * k * k:].float().abs().sum().item()                        # alignment columns are zero
    d = torch.randn(cols.shape, generator=g).to(torch.bfloat16)
    out = torch.full((B, Cc, H, W) if layout == "nchw" else (Cc, B, H, W), 7.0, device="cuda")
    C.col2im(d.cuda(), B, Cc, H, W, k, s, p, out, sb, sc)
    want = F.fold(d[:, :Cc * k * k].float().reshape(B, Ho * Wo, Cc * k * k).permute(0, 2, 1), (H, W), k, padding=p, stride=s)
    got = out.cpu() if layout == "nchw" else out.cpu().permute(1, 0, 2, 3)
    assert rel(got, want) <= 1e-6


@pytest.mark.parametrize("B,Cin,Cout,H,k,s,p", [(2, 3, 128, 16, 1, 1, 0), (2, 64, 128, 16, 3, 1, 1), (2, 64, 136, 17, 3, 2, 0), (8, 520, 64, 4, 3, 1, 1)])
def test_conv2d_values_and_gradients(C, B, Cin, Cout, H, k, s, p):
    """conv2d_gradfix.conv2d (the reference's NCHW signature) against F.conv2d on the same bf16-representable operands"""
    from enhancing.losses.op import conv2d_gradfix
    g = torch.Generator().manual_seed(Cin + Cout)
    x = bf16r(torch.randn(B, Cin, H, H, generator=g))
    w = bf16r(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    dy = bf16r(torch.randn(yr.shape, generator=g))
    yr.backward(dy)
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = conv2d_gradfix.conv2d(xd, wd, bias=bd, stride=s, padding=p)
    y.backward(dy.cuda())
    assert rel(y, yr) <= 1e-5 and rel(wd.grad, wr.grad) <= 1e-5 and rel(bd.grad, br.grad) <= 1e-5
    assert rel(xd.grad, xr.grad) <= 4e-3      # dcols leaves the dgrad GEMM in bf16: up to k*k rounded terms per input pixel


# (lowering, operand format) -> bounds (logits, dx of the R1 pass, r1, d_loss, gradient norms, generator-side dx, parameter gradients) against the REFERENCE's own
# golden (oracle/make_golden_disc.py runs /root/reference/enhancing/losses/layers.py): the product path (implicit GEMM, bf16 operands: leaky-ReLU gates of a random-init
# network flip under the operand rounding, see the emulation test below), the same arithmetic on fp16 operands through the im2col lowering (11 significand bits: ~8x fewer
# flips), and the fp32 INSTRUMENT (f32 columns, exact-f32 GEMM: no 16-bit rounding anywhere) — the mode in which f1 meets north_star's tolerance, as the fp32 engine mode
# does for the towers.  Bounds: 1.3 x the values measured on MI355X (profiles/r06_disc_parity.txt) for the two new modes.
DISC_MODES = {
    "igemm_bf16":  ("igemm", "bf16", (2e-2, 0.15, 2e-2, 1e-2, 6e-2, 0.15, 0.2)),
    "igemm_fp16":  ("igemm", "fp16", (2e-3, 3e-2, 8e-3, 4e-3, 8e-3, 2.5e-2, 6e-2)),       # the PRODUCT kernels (channels-last implicit GEMM, fused epilogues, 16-bit activations between layers) on fp16 operands (ABI v17): measured 1.2e-3, 2.1e-2, 5.6e-3, 2.3e-3, 5.0e-3, 1.5e-2, 4.0e-2 (final_conv bias; the others 1.1e-2) — 12x (logits) / 2.7x (gradients) under the bf16 row above
    "im2col_fp16": ("im2col", "fp16", (3e-3, 3e-2, 4e-3, 2e-3, 1.2e-2, 3e-2, 6e-2)),      # measured 2.1e-3, 2.0e-2, 2.4e-3, 9.8e-4, 7.4e-3, 2.0e-2, 4.5e-2 (final_conv bias; the others 1.0e-2)
    "im2col_fp32": ("im2col", "fp32", (1e-4, 1e-3, 1e-4, 1e-4, 1e-3, 1e-3, 1e-3)),        # measured 8.8e-6, 7.7e-4 (one leaky-ReLU gate of the random-init network sits within an fp32 ulp of zero), 2.0e-5, 8.0e-6, 5.8e-5, 2.8e-4, 1.1e-4: every quantity within north_star's 1e-3
}


@pytest.mark.parametrize("mode", list(DISC_MODES))
def test_discriminator_against_reference_golden(C, golden_dir, mode):
    from enhancing.engine.stage1 import ParamStore
    from enhancing.losses.layers import vanilla_d_loss
    from enhancing.losses.op import conv2d_gradfix
    lowering, operand, (b_logits, b_dx, b_r1, b_loss, b_norm, b_gf, b_grad) = DISC_MODES[mode]
    from enhancing.losses.op import conv_nhwc
    with conv2d_gradfix.operand_dtype(operand), conv_nhwc.operand_dtype("fp16" if operand == "fp16" else "bf16"):      # (the igemm path's linears run on conv2d_gradfix's GEMM node)
        _disc_golden_case(C, golden_dir, lowering, mode, b_logits, b_dx, b_r1, b_loss, b_norm, b_gf, b_grad, S=4096.0 if operand == "fp16" else 1.0)


def _disc_golden_case(C, golden_dir, lowering, mode, b_logits, b_dx, b_r1, b_loss, b_norm, b_gf, b_grad, S=1.0):
    """S: the loss networks' scale under fp16 operands (engine/optim.py LossScaler's initial value; what VQLPIPSWithDiscriminator.forward and
    ViTVQ.training_step apply): the R1 first-order pass and the d-loss backward run on S x their scalar and are divided back in f32"""
    from enhancing.engine.stage1 import ParamStore
    from enhancing.losses.layers import vanilla_d_loss
    from enhancing.losses.op import conv2d_gradfix
    G = np.load(os.path.join(golden_dir, "disc_tiny.npz"))
    D, real, fake = disc_case(G)
    D.lowering = lowering
    dev = torch.device("cuda")
    D.to(dev)
    store = ParamStore(D, dev, precision="fp32")
    real, fake = real.to(dev), fake.to(dev)
    x = real.clone().requires_grad_(True)
    lr_, lf_ = D(x), D(fake)
    e_logits = max(rel(lr_, torch.from_numpy(G["logits_real"])), rel(lf_, torch.from_numpy(G["logits_fake"])))
    with conv2d_gradfix.no_weight_gradients():
        gr, = torch.autograd.grad(lr_.sum() * S, x, create_graph=True)
    gr = gr / S
    r1 = gr.square().sum([1, 2, 3]).mean()
    d_loss = vanilla_d_loss(lf_, lr_) + 10 * 16 * r1 / 2
    store.zero_grad()
    (d_loss * S).backward()
    store.g.div_(S)
    e_dx = rel(gr, torch.from_numpy(G["dx_real"]))
    e_r1 = abs(r1.item() - float(G["r1"])) / float(G["r1"])
    e_loss = abs(d_loss.item() - float(G["d_loss"])) / abs(float(G["d_loss"]))
    P = dict(D.named_parameters())
    norms = {str(n): float(v) for n, v in zip(G["grad_names"], G["grad_norms"])}
    e_norm = max(abs(P[n].grad.double().norm().item() - v) / v for n, v in norms.items())
    e_t = {n: rel(P[n].grad, torch.from_numpy(G[k])) for n, k in (("final_conv.1.bias", "g_final_bias"), ("blocks.0.0.weight", "g_rgb_w"),
                                                                    ("final_linear.1.weight", "g_lin1_w"))}
    xf = fake.clone().requires_grad_(True)
    g_loss = vanilla_d_loss(D(xf))
    gf, = torch.autograd.grad(g_loss * S, xf)
    gf = gf / S
    e_gf = rel(gf, torch.from_numpy(G["g_fake"]))
    line = (f"discriminator [{mode}] vs reference golden: logits {e_logits:.2e}, dx {e_dx:.2e}, r1 {e_r1:.2e}, d_loss {e_loss:.2e}, grad norms {e_norm:.2e}, "
            f"grads {({k: float(f'{v:.2e}') for k, v in e_t.items()})}, generator-side dx {e_gf:.2e}")
    print(line)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "disc_parity.txt"), "a") as f:
            f.write(line + "\n")
    assert e_logits <= b_logits and e_dx <= b_dx and e_r1 <= b_r1 and e_loss <= b_loss and e_norm <= b_norm and e_gf <= b_gf, line
    assert max(e_t.values()) <= b_grad, e_t
    assert abs(g_loss.item() - float(G["g_loss"])) <= max(1e-2 if mode == "igemm_bf16" else 2e-3 * abs(float(G["g_loss"])), 1e-5)


def test_hip_discriminator_vs_the_same_rounding_points_in_torch(C, golden_dir, monkeypatch):
    """the loose bounds above (logits 2e-2, gradients 0.15 - 0.2 against the fp32 golden) mix two things — the bf16 ROUNDING POINTS
    the design chose (operands and stored activations in bf16: leaky-ReLU gates of a random-init network flip) and the error of the KERNELS themselves.
    tests/hip_emulation.py with exact=False reproduces exactly those rounding points in plain torch (fp32 accumulation, bf16 containers, the same
    geometry structs tap by tap).  Run on the golden case as it is, HIP and emulation disagree as much as each disagrees with fp32 (logits 9.9e-3,
    bias gradients 1e-1: measured, round 3) — they flip DIFFERENT gates, because a pre-activation within one summation-order ulp of zero decides a
    gate and a random-init network has thousands of those.  So the gates are taken out of the comparison instead: every activation bias is raised by
    +4 (pre-activations ~N(4, 1): a few gates in 10^5 are off, almost none near zero), the same state dict on both sides.  What is left is the
    kernels' arithmetic: logits 5e-3, every parameter gradient 2e-2 (first order, and the R1 penalty's second order through the backward pass), the
    image gradient 4e-2 (measured: logits 2.4e-3, parameter gradients <= 1.5e-2, median 1.2e-2, image gradient 3.1e-2, R1 2.0e-3, d-loss 6.9e-4)."""
    import copy
    import hip_emulation
    from enhancing.engine.stage1 import ParamStore
    from enhancing.losses.layers import StyleDiscriminator, vanilla_d_loss
    from enhancing.losses.op import conv2d_gradfix
    G = np.load(os.path.join(golden_dir, "disc_tiny.npz"))
    D, real, fake = disc_case(G)
    with torch.no_grad():
        for n, p_ in D.named_parameters():
            if n.endswith("bias") and p_.ndim == 1 and "final_linear.1" not in n:
                p_.add_(4.0)
    sd = {k: v.detach().clone() for k, v in D.state_dict().items()}

    def run(Dm, dev):
        x = real.to(dev).clone().requires_grad_(True)
        lr_, lf_ = Dm(x), Dm(fake.to(dev))
        with conv2d_gradfix.no_weight_gradients():
            gr, = torch.autograd.grad(lr_.sum(), x, create_graph=True)
        r1 = gr.square().sum([1, 2, 3]).mean()
        d_loss = vanilla_d_loss(lf_, lr_) + 10 * 16 * r1 / 2
        for p_ in Dm.parameters():
            p_.grad = None
        d_loss.backward()
        return dict(lr=lr_.detach().float().cpu(), lf=lf_.detach().float().cpu(), gr=gr.detach().float().cpu(), r1=float(r1), d_loss=float(d_loss),
                    grads={n: p_.grad.detach().float().cpu().clone() for n, p_ in Dm.named_parameters() if p_.grad is not None})

    dev = torch.device("cuda")
    Dg = copy.deepcopy(D).to(dev)
    store = ParamStore(Dg, dev, precision="fp32")
    store.zero_grad()
    hip = run(Dg, dev)
    hip_emulation.install(monkeypatch, exact=False)          # from here on enhancing._C's kernels are the torch stand-ins (CPU tensors)
    De = StyleDiscriminator(size=int(G["size"]))
    De.load_state_dict(sd)
    emu = run(De, torch.device("cpu"))
    e_logits = max(rel(hip["lr"], emu["lr"]), rel(hip["lf"], emu["lf"]))
    e_dx = rel(hip["gr"], emu["gr"])
    e_g = {n: rel(hip["grads"][n], emu["grads"][n]) for n in emu["grads"]}
    worst = max(e_g, key=e_g.get)
    line = (f"HIP discriminator vs torch emulation of the same bf16 rounding points: logits {e_logits:.2e}, d logits / d image {e_dx:.2e}, "
            f"r1 {abs(hip['r1'] - emu['r1']) / emu['r1']:.2e}, d_loss {abs(hip['d_loss'] - emu['d_loss']) / abs(emu['d_loss']):.2e}, "
            f"worst parameter gradient {worst} {e_g[worst]:.2e}, median {np.median(list(e_g.values())):.2e}")
    print(line)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "disc_vs_emulation.txt"), "a") as f:
            f.write(line + "\n" + "  per parameter: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(e_g.items(), key=lambda kv: -kv[1])[:12]) + "\n")
    assert set(hip["grads"]) == set(emu["grads"])
    assert e_logits <= 5e-3 and e_dx <= 4e-2
    assert abs(hip["r1"] - emu["r1"]) <= 2e-2 * emu["r1"] and abs(hip["d_loss"] - emu["d_loss"]) <= 5e-3 * abs(emu["d_loss"])
    assert e_g[worst] <= 2e-2, sorted(e_g.items(), key=lambda kv: -kv[1])[:5]


def test_two_optimizer_training_step_protocol(C):
    """ViTVQ.training_step with a discriminator in the loss (vitvqgan.py:101-127 under Lightning's toggle_optimizer): optimizer 0 trains
    only the autoencoder (through the discriminator's input gradient), optimizer 1 only the discriminator (R1 on batch 0), both step."""
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPSWithDiscriminator",
            "params": dict(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0, adversarial_weight=0.1, do_r1_every=2,
                           disc_params={"size": cfg["image_size"]})}
    m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
              AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
    m.load_state_dict({**O.make_params(cfg, seed=11), **{"loss." + k: v for k, v in m.loss.state_dict().items()}}, strict=True)
    m.train()
    opts, _ = m.configure_optimizers()
    assert len(opts) == 2
    ae, ds = m.engine.store, m.loss.disc_store(m.engine.device)
    batch = {"image": O.make_images(5, 2, cfg["image_size"])}
    for it in range(2):
        p_ae, p_d, g_d = ae.p.clone(), ds.p.clone(), ds.g.clone()
        l0 = m.training_step(batch, it, 0)
        assert torch.isfinite(l0) and ae.g.abs().sum().item() > 0 and torch.equal(ds.g, g_d)            # discriminator frozen
        assert "train/g_loss" in m.logged
        opts[0].step()
        ae.zero_grad()
        l1 = m.training_step(batch, it, 1)
        assert torch.isfinite(l1) and ds.g.abs().sum().item() > 0 and ae.g.abs().sum().item() == 0     # autoencoder untouched
        assert ("train/r1_reg" in m.logged) == (it % 2 == 0)
        m.logged.pop("train/r1_reg", None)
        opts[1].step()
        assert not torch.equal(ae.p, p_ae) and not torch.equal(ds.p, p_d)
    # the generator-side gradient really comes through the discriminator: with adversarial_weight -> 0 the AE gradient changes
    ae.zero_grad()
    m.training_step(batch, 0, 0)
    g_with = ae.g.clone()
    m.loss.adversarial_weight = 1e-12
    ae.zero_grad()
    m.training_step(batch, 0, 0)
    assert rel(g_with, ae.g) > 1e-3


def test_adaptive_adversarial_weight(C):
    """use_adaptive_adv (reference vqperceptual.py:95-103,125-126): d_weight = adversarial_weight * ||d nll/d last_layer|| / (||d g_loss/d last_layer|| + 1e-4).
    The engine's last-layer norm (gradient at the reconstruction x saved last-layer input) must equal the norm of the gradient its own backward writes
    for the same upstream gradient, and the logged d_weight must be the ratio of the two norms."""
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPSWithDiscriminator",
            "params": dict(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0, adversarial_weight=0.1, use_adaptive_adv=True,
                           disc_params={"size": cfg["image_size"]})}
    m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
              AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
    m.load_state_dict({**O.make_params(cfg, seed=11), **{"loss." + k: v for k, v in m.loss.state_dict().items()}}, strict=True)
    m.train()
    eng = m.engine
    x = O.make_images(5, 2, cfg["image_size"])
    # (1) the norm primitive against the engine's own backward
    # (upstream gradient at the size a mean-reduced loss produces: under the fp16 engine's static loss scale 2^16, |g| * 2^16 must stay below 65504)
    g = 1e-4 * torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(1)).cuda()
    eng.store.zero_grad()
    xrec, _ = m(x)
    n_fast = eng.last_layer_grad_norm(g).item()
    eng.scale_loss((xrec * g).sum()).backward()
    eng.unscale_grads()
    n_ref = m.decoder.get_last_layer().grad.norm().item()
    eng.store.zero_grad()
    assert abs(n_fast - n_ref) <= 2e-3 * n_ref, (n_fast, n_ref)
    # (2) the training step logs d_weight = 0.1 * ||d nll|| / (||d g|| + 1e-4), recomputed here from the two upstream gradients
    l0 = m.training_step({"image": x}, 0, 0)
    assert torch.isfinite(l0) and "train/d_weight" in m.logged
    xr, _ = m(x)
    xd = x.to(xr.device)
    nll = ((xr - xd) ** 2).mean()
    g_loss = m.loss.disc_loss(m.loss.discriminator(xr))
    gn, = torch.autograd.grad(nll, xr, retain_graph=True)
    gg, = torch.autograd.grad(g_loss, xr, retain_graph=True)
    want = 0.1 * eng.last_layer_grad_norm(gn) / (eng.last_layer_grad_norm(gg) + 1e-4)
    print("d_weight logged", m.logged["train/d_weight"].item(), "recomputed", want.item(), "norms", eng.last_layer_grad_norm(gn).item(), eng.last_layer_grad_norm(gg).item())
    # (two separate forward passes through the discriminator: its split-K f32 atomics make them differ in the last bits, and gate flips amplify that)
    assert abs(m.logged["train/d_weight"].item() - want.item()) <= 2e-2 * abs(want.item())


def test_adaptive_adversarial_weight_with_the_perceptual_term(C, lpips_random_init):
    """use_adaptive_adv with the reference's DEFAULT perceptual_weight (1.0).  calculate_adaptive_factor differentiates nll_loss —
    which contains the LPIPS term — with retain_graph=True (vqperceptual.py:94-103) and the real backward then runs through the same autograd node a
    second time: both optimizers' training steps must complete, log a finite d_weight, and give gradients to the autoencoder."""
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPSWithDiscriminator",
            "params": dict(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=1.0, adversarial_weight=0.1, use_adaptive_adv=True,
                           disc_params={"size": cfg["image_size"]})}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
                  AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
    assert m.loss.perceptual_loss.random_init and not m.loss.perceptual_loss.weights_loaded
    assert not any(k.startswith("loss.perceptual_loss.") for k in m.state_dict())      # the random trunk stays out of checkpoints
    m.load_state_dict({**O.make_params(cfg, seed=11), **{"loss." + k: v for k, v in m.loss.state_dict().items()}}, strict=False)
    assert m.loss.perceptual_loss.random_init and not m.loss.perceptual_loss.weights_loaded
    m.train()
    x = O.make_images(5, 2, cfg["image_size"])
    m.engine.store.zero_grad()
    l0 = m.training_step({"image": x}, 0, 0)
    assert torch.isfinite(l0) and torch.isfinite(m.logged["train/d_weight"]) and m.logged["train/d_weight"].item() > 0
    assert m.logged["train/perceptual_loss"].item() > 0
    gn = float(m.engine.store.g.float().norm())
    assert gn > 0 and gn == gn
    l1 = m.training_step({"image": x}, 0, 1)
    assert torch.isfinite(l1)


def test_fp16_loss_networks_scaled_discriminator_step(C, lpips_random_init):
    """the loss networks on fp16 operands (ENH_LOSS_OPERANDS=fp16 / conv_nhwc.operand_dtype): the discriminator's backward runs on LossScaler.scale_t x the
    loss (R1's first-order pass included), FlatAdamW.step checks inf / nan, unscales inside the AdamW launch and updates the scale — GradScaler's protocol,
    no host sync.  (1) two rounds of the two-optimizer protocol land within the operand-rounding distance of the bf16-operand run; (2) a scale that
    overflows fp16 skips the discriminator step, leaves parameters and moments untouched and halves the scale; the next step goes through."""
    import vitvq_oracle as O
    from enhancing.losses.op import conv_nhwc
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPSWithDiscriminator",
            "params": dict(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.1, adversarial_weight=0.1, use_adaptive_adv=True, do_r1_every=2,
                           disc_params={"size": cfg["image_size"]})}
    batch = {"image": O.make_images(5, 2, cfg["image_size"])}

    def build(operands):
        torch.manual_seed(0)
        m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
                  AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
        m.precision = "fp16"           # (pinned before the engine binds: the suite also runs under ENH_PRECISION=bf16)
        m.load_state_dict({**O.make_params(cfg, seed=11), **{"loss." + k: v for k, v in m.loss.state_dict().items()}}, strict=False)
        m.train()
        m.learning_rate = 1e-3
        m.loss.operands = operands
        return m, m.configure_optimizers()[0]

    def rounds(m, opts, n):
        out = []
        for it in range(n):
            l0 = m.training_step(batch, it, 0); opts[0].step()
            l1 = m.training_step(batch, it, 1)
            out.append((float(l0), float(l1), float(m.logged["train/r1_reg"]) if it % 2 == 0 else None, float(m.logged["train/d_weight"])))
            opts[1].step()
        return out

    mb, ob = build("bf16")
    ref = rounds(mb, ob, 2)
    assert not mb.loss.disc_store(mb.engine.device).loss_scaler.enabled
    if True:
        mf, of = build(None)          # default: the loss networks follow the engine (fp16 since round 6)
        assert mf.engine.precision == "fp16" and mf.loss.loss_operands(mf.decoder.get_last_layer()) == "fp16"
        ds = mf.loss.disc_store(mf.engine.device)
        p0 = ds.p.clone()
        got = rounds(mf, of, 2)
        sc = ds.loss_scaler
        assert sc.enabled and sc.value == 4096.0 and float(sc.found_inf) == 0.0 and int(sc.tracker) == 2
        assert not torch.equal(ds.p, p0)
        for (a0, a1, ar, aw), (b0, b1, br, bw) in zip(got, ref):
            assert abs(a0 - b0) <= 2e-2 * abs(b0) and abs(a1 - b1) <= 2e-2 * abs(b1) and abs(aw - bw) <= 0.1 * abs(bw), (got, ref)
            assert (ar is None) == (br is None) and (ar is None or abs(ar - br) <= 5e-2 * br), (got, ref)
        d_fp16, d_bf16 = ds.p - p0, mb.loss.disc_store(mb.engine.device).p - p0
        cos = float((d_fp16 * d_bf16).sum() / (d_fp16.norm() * d_bf16.norm()))
        print(f"discriminator update, fp16 vs bf16 loss networks after 2 rounds: cosine {cos:.4f}, losses {got} vs {ref}")
        assert cos >= 0.9          # (AdamW's first steps are sign-like: elements whose tiny gradient changes sign between the two roundings flip)
        # (2) overflow: the scale is forced above fp16's range
        sc.scale_t.fill_(2.0 ** 24)
        p1, m1, step1 = ds.p.clone(), ds.m.clone(), ds.step_count
        mf.training_step(batch, 2, 1)
        of[1].step()
        assert float(sc.found_inf) == 1.0 and torch.equal(ds.p, p1) and torch.equal(ds.m, m1) and sc.value == 2.0 ** 23 and int(sc.tracker) == 0
        sc.scale_t.fill_(4096.0)
        mf.training_step(batch, 3, 1)
        of[1].step()
        assert float(sc.found_inf) == 0.0 and not torch.equal(ds.p, p1) and torch.isfinite(ds.p).all()
    # pinned back to bf16 operands the scaler is the identity again
    mf.loss.operands = "bf16"
    l1 = mf.training_step(batch, 4, 1)
    assert not ds.loss_scaler.enabled and torch.isfinite(l1)


def test_graph_replay_of_the_two_optimizer_step_equals_the_eager_sequence(lpips_random_init):
    """the protocol every shipped config runs (autoencoder step through LPIPS + discriminator, then the discriminator step with the
    lazy R1 penalty) replayed from HIP graphs — one per (optimizer, host-side variant: R1 or not) — gives the SAME bits as the eager sequence: losses of
    every step, and both flat parameter buffers after four optimizer rounds (R1 on rounds 0 and 2)."""
    import warnings
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPSWithDiscriminator",
            "params": dict(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.1, adversarial_weight=0.1, do_r1_every=2,
                           disc_params={"size": cfg["image_size"]})}
    xs = [O.make_images(5 + i, 2, cfg["image_size"]) for i in range(2)]

    def run(graphs: bool):
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]),
                      AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss))
        m.load_state_dict({**O.make_params(cfg, seed=11), **{"loss." + k: v for k, v in m.loss.state_dict().items()}}, strict=False)
        m.train()
        m.learning_rate = 1e-3
        opts, _ = m.configure_optimizers()
        m.engine.use_graphs = graphs
        losses = []
        for i in range(4):
            b = {"image": xs[i % 2]}
            l0 = m.training_step(b, i, 0); opts[0].step()
            l1 = m.training_step(b, i, 1); opts[1].step()
            m.global_step += 1
            losses.append((l0.clone(), l1.clone(), m.logged["train/g_loss"].clone(), m.logged["train/perceptual_loss"].clone()))
        torch.cuda.synchronize()
        if graphs:
            assert len(m._step_graphs) == 3          # optimizer 0; optimizer 1 with and without R1
        return losses, m.engine.store.p.clone(), m.loss.disc_store(m.engine.device).p.clone()

    le, pe, de = run(False)
    lg, pg, dg = run(True)
    for i, (a, b) in enumerate(zip(le, lg)):
        assert all(torch.equal(u, v) for u, v in zip(a, b)), (i, [float(u) for u in a], [float(v) for v in b])
    assert torch.equal(pe, pg) and torch.equal(de, dg)
    assert float(le[0][0]) != float(le[3][0])


def test_discriminator_forward_is_bit_reproducible_after_weight_updates():
    """repeated calls on the same input and weights give the same bits — checked after a few optimizer steps (the nondeterminism of round 3's atomic split-K
    in the 8192 -> 512 linear only showed for some weight values)"""
    from enhancing.engine.optim import FlatAdamW
    from enhancing.engine.stage1 import ParamStore
    from enhancing.losses.layers import StyleDiscriminator
    torch.manual_seed(0)
    D = StyleDiscriminator(size=64).cuda()
    store = ParamStore(D, torch.device("cuda:0"), precision="fp32")
    opt = FlatAdamW(store, lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(2, 3, 64, 64, device="cuda", generator=g)
    for step in range(4):
        store.zero_grad()
        torch.nn.functional.softplus(D(x)).mean().backward()
        grads = store.g.clone()
        for _ in range(3):
            store.zero_grad()
            torch.nn.functional.softplus(D(x)).mean().backward()
            assert torch.equal(store.g, grads), f"step {step}: gradients differ between identical passes"
        opt.step()
        with torch.no_grad():
            ys = [D(x).clone() for _ in range(8)]
        assert all(torch.equal(ys[0], y) for y in ys[1:]), f"step {step}: logits differ between identical calls"
