"""CPU suite (-m "not gpu"): the oracle against the golden vectors produced by the REFERENCE (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

import vitvq_oracle as O


@pytest.mark.parametrize("name", ["vq_k8192_m4096", "vq_k512_m1024", "rq4_k8192_m2048"])
def test_torch_oracle_quantizer_vs_golden(golden_dir, name):
    g = np.load(f"{golden_dir}/{name}.npz")
    M, K, D = int(g["M"]), int(g["K"]), int(g["num_quantizers"])
    z, E, gout = O.make_vq_inputs(int(g["seed"]), M, K)
    zt = z.view(M // 64, 64, 32).clone().requires_grad_(True)
    Et = E.clone().requires_grad_(True)
    zq, loss, idx = O.quantizer_forward(zt, Et, float(g["beta"]), True, bool(g["use_residual"]), D or None)
    assert np.array_equal(idx.numpy().reshape(M, -1), g["idx"].astype(np.int64).reshape(M, -1))
    assert abs(loss.item() - float(g["loss"])) <= 1e-7
    assert np.array_equal(zq.detach().view(M, 32)[:64].numpy(), g["zq_first"].reshape(64, 32))
    loss.backward()
    assert np.allclose(Et.grad[g["dE_rows"]].numpy(), g["dE_sample"], rtol=0, atol=1e-9)
    assert abs(Et.grad.double().norm().item() - float(g["dE_norm"])) <= 1e-9
    dz = zt.grad if zt.grad is not None else torch.zeros_like(zt)
    assert abs(dz.double().norm().item() - float(g["dz_loss_norm"])) <= 1e-9
    assert int(g["n_used"]) == len(torch.unique(idx))


@pytest.mark.parametrize("name", ["vq_k8192_m4096", "vq_k512_m1024", "rq4_k8192_m2048"])
def test_c_oracle_quantizer_vs_golden(golden_dir, name):
    """the explicit-summation-order C restatement reproduces the reference's indices exactly on the golden cases."""
    import vq_oracle as VC
    g = np.load(f"{golden_dir}/{name}.npz")
    M, K, D = int(g["M"]), int(g["K"]), max(int(g["num_quantizers"]), 1)
    z, E, _ = O.make_vq_inputs(int(g["seed"]), M, K)
    zq, idx, loss = VC.forward(z.numpy(), E.numpy(), float(g["beta"]), D)
    assert np.array_equal(idx, g["idx"].astype(np.int64).reshape(M, D))
    assert abs(float(loss) - float(g["loss"])) <= 2e-7
    assert np.abs(zq[:64] - g["zq_first"].reshape(64, 32)).max() <= 5e-7


def test_c_oracle_edge_cases():
    import vq_oracle as VC
    rs = np.random.RandomState(0)
    E = rs.standard_normal((37, 32)).astype(np.float32)
    # a token equal to a code -> that code; duplicate codes -> lowest index wins (argmin first-index rule)
    E[20] = E[5]
    z = np.stack([E[5] * 3.0, E[36] * 0.5, np.zeros(32, np.float32)]).astype(np.float32)
    zq, idx, loss = VC.forward(z, E, 0.25, 1)
    assert idx[0, 0] == 5 and idx[1, 0] == 36
    assert np.isfinite(zq).all() and np.isfinite(loss)


def test_vit_tiny_oracle_vs_golden(golden_dir):
    g = np.load(f"{golden_dir}/vit_tiny.npz")
    cfg = O.TINY_CFG
    P = O.make_params(cfg, int(g["param_seed"]))
    x = O.make_images(int(g["image_seed"]), int(g["B"]), cfg["image_size"])
    loss, log, grads, xrec = O.train_step_grads(x, P, cfg)
    _, _, idx, h = O.encode(x, P, cfg)
    assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))
    assert np.allclose(h.detach().numpy(), g["h"], atol=2e-6)
    assert np.allclose(xrec.numpy(), g["xrec"], atol=2e-6)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    names = list(g["grad_names"])
    assert sorted(grads) == names
    for n, ref in zip(names, g["grad_norms"]):
        assert abs(grads[n].double().norm().item() - ref) <= 1e-5 * max(ref, 1e-8), n
    assert np.allclose(grads["quantizer.embedding.weight"].numpy(), g["g_codebook"], atol=1e-9)
    assert np.allclose(grads["pre_quant.weight"].numpy(), g["g_pre_quant_w"], rtol=1e-4, atol=1e-9)


def test_position_table_layout():
    """first half of the channels = x (width) coordinate, [sin | cos] per half."""
    pe = O.sincos_2d(16, 4, 4)
    t = 1 * 4 + 3  # token at y=1, x=3
    omega = 1.0 / 10000 ** (np.arange(4) / 4.0)
    assert np.allclose(pe[t, :4], np.sin(3 * omega)) and np.allclose(pe[t, 4:8], np.cos(3 * omega))
    assert np.allclose(pe[t, 8:12], np.sin(1 * omega)) and np.allclose(pe[t, 12:], np.cos(1 * omega))


def test_patchify_roundtrip_and_conv_equivalence():
    x = torch.rand(2, 3, 32, 32)
    p = O.patchify(x, 8)
    assert torch.equal(O.unpatchify(p, 8, 3, 32, 32), x)
    w = torch.randn(16, 3, 8, 8)
    ref = torch.nn.functional.conv2d(x, w, stride=8).flatten(2).transpose(1, 2)
    assert torch.allclose(p @ w.reshape(16, -1).t(), ref, atol=1e-4)
    wt = torch.randn(16, 3, 8, 8)
    tok = torch.randn(2, 16, 16)
    ref_t = torch.nn.functional.conv_transpose2d(tok.transpose(1, 2).reshape(2, 16, 4, 4), wt, stride=8)
    assert torch.allclose(O.unpatchify(tok @ wt.reshape(16, -1), 8, 3, 32, 32), ref_t, atol=1e-4)


def test_resize_oracle_is_pillow_bit_for_bit():
    """PIN of oracle/resize_oracle.py (the restatement of Pillow's antialiased bilinear resampler that the device-side Resize is checked against): equal to
    PIL.Image.resize(size, BILINEAR) itself — the call torchvision's T.Resize makes in the reference's transforms (enhancing/dataloader/imagenet.py:31,49)
    — on shrinking, growing, ragged and identity sizes; and the product's vectorised coefficient tables equal the oracle's scalar ones."""
    import numpy as np
    from PIL import Image
    import resize_oracle as RO
    from enhancing.dataloader import resize as R
    rs = np.random.RandomState(0)
    for (H, W, Ho, Wo) in [(500, 375, 341, 256), (333, 500, 256, 384), (100, 120, 256, 307), (64, 64, 256, 256), (480, 640, 256, 256), (257, 301, 256, 299),
                           (37, 53, 37, 20), (90, 31, 7, 31)]:
        img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        ref = np.array(Image.fromarray(img).resize((Wo, Ho), Image.BILINEAR))
        assert np.array_equal(RO.resize_u8(img, (Ho, Wo)), ref), (H, W, Ho, Wo)
        for a, b in ((W, Wo), (H, Ho)):
            b1, k1 = R.coeff_table(a, b)
            b2, k2 = RO.bilinear_coeffs(a, b)
            assert np.array_equal(b1, b2) and np.array_equal(k1, k2), (a, b)
    # torchvision's size rule for an int: shorter side -> size, the other int(size * long / short)
    assert R.output_size(500, 375, 256) == RO.torchvision_resize_size(500, 375, 256) == (341, 256)
    assert R.output_size(375, 500, 256) == (256, 341) and R.output_size(300, 200, (256, 256)) == (256, 256)
