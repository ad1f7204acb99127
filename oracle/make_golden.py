"""Pins the oracle against the REAL reference and writes tests/golden/*.npz.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
1. asserts oracle/vitvq_oracle.py == the reference's own modules
   (enhancing/modules/stage1/{quantizers,layers}.py imported by file path) on seeded inputs;
2. stores the REFERENCE's outputs as golden vectors (inputs are re-generated from seeds with
   numpy's MT19937, so only outputs / samples are stored -> small files).
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_loader as RL  # noqa: E402
import vitvq_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
torch.set_num_threads(8)


def ref_quantizer(Q, E, **kw):
    q = Q.VectorQuantizer(E.shape[1], E.shape[0], **kw)
    with torch.no_grad():
        q.embedding.weight.copy_(E)
    return q


def gold_quantizer(name, seed, M, K, kw):
    Q = RL.load_quantizers()
    z, E, g = O.make_vq_inputs(seed, M, K)
    z3 = z.view(M // 64, 64, 32)  # [B, N, d] like the model
    # --- reference ---
    q = ref_quantizer(Q, E, **kw)
    zr = z3.clone().requires_grad_(True)
    zq, loss, idx = q(zr)
    (zq * g.view_as(zq)).sum().backward(retain_graph=True)
    dz_out = zr.grad.clone(); zr.grad = None
    dE_out = q.embedding.weight.grad.clone() if q.embedding.weight.grad is not None else torch.zeros_like(E)
    q.embedding.weight.grad = None
    loss.backward()
    dz_loss = zr.grad.clone() if zr.grad is not None else torch.zeros_like(z3)
    dE_loss = q.embedding.weight.grad.clone()
    # --- oracle restatement ---
    zo = z3.clone().requires_grad_(True)
    Eo = E.clone().requires_grad_(True)
    zq2, loss2, idx2 = O.quantizer_forward(zo, Eo, beta=kw.get("beta", 0.25), use_norm=True,
                                           use_residual=kw.get("use_residual", False),
                                           num_quantizers=kw.get("num_quantizers"))
    assert torch.equal(idx, idx2), name
    assert torch.equal(zq.detach(), zq2.detach()) and torch.equal(loss.detach(), loss2.detach()), name
    loss2.backward()
    assert torch.allclose(Eo.grad, dE_loss, rtol=0, atol=0), name
    assert dE_out.abs().max() == 0  # straight-through: g_out never reaches the codebook
    # fp64 audit of the reference's own argmin (near-tie statistics)
    zn64 = torch.nn.functional.normalize(z.double(), dim=-1)
    if not kw.get("use_residual", False):
        en64 = torch.nn.functional.normalize(E.double(), dim=-1)
        d64 = (zn64 ** 2).sum(1, keepdim=True) + (en64 ** 2).sum(1) - 2 * zn64 @ en64.t()
        top2 = torch.topk(d64, 2, dim=1, largest=False)
        agree = (top2.indices[:, 0] == idx.view(-1)).float().mean().item()
        gap = (top2.values[:, 1] - top2.values[:, 0])
        print(f"  {name}: reference fp32 argmin == fp64 argmin on {agree*100:.4f}% ; min top-2 gap {gap.min().item():.3e}")
    samp = np.arange(0, K, max(K // 64, 1))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                        seed=seed, M=M, K=K, beta=kw.get("beta", 0.25),
                        use_residual=kw.get("use_residual", False), num_quantizers=kw.get("num_quantizers") or 0,
                        idx=idx.numpy().astype(np.int16), loss=loss.detach().numpy(),
                        zq_first=zq.detach().view(M, 32)[:64].numpy(), zq_sum=zq.detach().double().sum().numpy(),
                        dz_loss_first=dz_loss.view(M, 32)[:64].numpy(), dz_loss_norm=dz_loss.double().norm().numpy(),
                        dz_out_is_g=bool(torch.equal(dz_out, g.view_as(dz_out))),
                        dE_rows=samp, dE_sample=dE_loss[samp].numpy(), dE_norm=dE_loss.double().norm().numpy(),
                        n_used=len(torch.unique(idx)))
    print(f"  wrote {name}.npz  (loss {loss.item():.6f}, codes used {len(torch.unique(idx))})")


def gold_vit_tiny():
    L = RL.load_layers()
    Q = RL.load_quantizers()
    cfg = O.TINY_CFG
    P = O.make_params(cfg, seed=11)
    x = O.make_images(5, 2, cfg["image_size"])
    # --- reference modules wired exactly as ViTVQ.__init__/forward do (vitvqgan.py:35-48) ---
    enc = L.ViTEncoder(image_size=64, patch_size=8, **cfg["encoder"])
    dec = L.ViTDecoder(image_size=64, patch_size=8, **cfg["decoder"])
    quant = Q.VectorQuantizer(**cfg["quantizer"])
    pre = torch.nn.Linear(128, 32); post = torch.nn.Linear(32, 128)
    mods = {"encoder.": enc, "decoder.": dec, "quantizer.": quant, "pre_quant.": pre, "post_quant.": post}
    for pref, m in mods.items():
        sd = {k[len(pref):]: v for k, v in P.items() if k.startswith(pref)}
        missing = m.load_state_dict(sd, strict=True)
    # position tables: the restatement must equal what the reference constructs itself
    assert torch.equal(L.ViTEncoder(image_size=64, patch_size=8, **cfg["encoder"]).en_pos_embedding, P["encoder.en_pos_embedding"])
    h_enc = enc(x)
    h = pre(h_enc)
    zq, qloss, idx = quant(h)
    xrec = dec(post(zq))
    l2 = (xrec - x).pow(2).mean()
    loss = l2 + qloss
    loss.backward()
    ref_grads = {}
    for pref, m in mods.items():
        for k, v in m.named_parameters():
            if v.grad is not None:
                ref_grads[pref + k] = v.grad
    # --- oracle ---
    o_loss, o_log, o_grads, o_xrec = O.train_step_grads(x, P, cfg)
    o_q, o_ql, o_idx, o_h = O.encode(x, P, cfg)
    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert torch.equal(o_idx, idx), "tiny: code indices differ"
    assert rel(o_h, h) < 2e-6 and rel(o_xrec, xrec) < 2e-6, (rel(o_h, h), rel(o_xrec, xrec))
    assert abs(o_loss.item() - loss.item()) < 1e-6
    worst = max(rel(o_grads[k], ref_grads[k]) for k in ref_grads)
    assert set(o_grads) == set(ref_grads), set(o_grads) ^ set(ref_grads)
    assert worst < 5e-5, worst
    print(f"  vit_tiny: oracle == reference (h rel {rel(o_h, h):.2e}, xrec rel {rel(o_xrec, xrec):.2e}, worst grad rel {worst:.2e})")
    np.savez_compressed(os.path.join(GOLD, "vit_tiny.npz"), param_seed=11, image_seed=5, B=2,
                        h=h.detach().numpy(), idx=idx.numpy().astype(np.int16), xrec=xrec.detach().numpy(),
                        loss=loss.item(), qloss=qloss.item(), l2=l2.item(),
                        grad_names=np.array(sorted(ref_grads)),
                        grad_norms=np.array([ref_grads[k].double().norm().item() for k in sorted(ref_grads)]),
                        g_pre_quant_w=ref_grads["pre_quant.weight"].numpy(),
                        g_codebook=ref_grads["quantizer.embedding.weight"].numpy(),
                        g_qkv0=ref_grads["encoder.transformer.layers.0.0.fn.to_qkv.weight"].numpy(),
                        g_pixel_w=ref_grads["decoder.to_pixel.1.weight"].numpy())
    print("  wrote vit_tiny.npz")


if __name__ == "__main__":
    assert RL.available(), "needs /root/reference"
    os.makedirs(GOLD, exist_ok=True)
    gold_quantizer("vq_k8192_m4096", 1234, 4096, 8192, {})
    gold_quantizer("vq_k512_m1024", 99, 1024, 512, {"beta": 0.5})
    gold_quantizer("rq4_k8192_m2048", 4321, 2048, 8192, {"use_residual": True, "num_quantizers": 4})
    gold_vit_tiny()
    print("golden vectors written to", GOLD)


# NOTE: tests/golden/disc_ops.npz (discriminator native ops) is produced by oracle/make_golden_disc_ops.py: the reference's op/*.py cannot be
# imported (they JIT-compile CUDA sources at import), so its pure-Python fallbacks `upfirdn2d_native` and the CPU branch of
# `fused_leaky_relu` are extracted with `ast` and executed as they are.
