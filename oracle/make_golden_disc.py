"""Pins oracle/disc_oracle.py against the reference's OWN discriminator code and writes tests/golden/disc_tiny.npz.  Runs in the
build container only (needs /root/reference).

The reference's enhancing/losses/layers.py is loaded unmodified as a module of a synthetic package whose `.op` sub-package is
replaced by the pinned pure-PyTorch restatements (oracle/disc_ops_oracle.py; the real op/*.py JIT-compile CUDA sources at import) and
with `kornia` stubbed (only PatchDiscriminator-era helpers use it).  Inputs and parameters are regenerated from seeds by the tests;
only outputs are stored."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import disc_ops_oracle as DO  # noqa: E402
import disc_oracle as O  # noqa: E402

REF = os.environ.get("ENH_REFERENCE_ROOT", "/root/reference")
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_reference_layers():
    pkg = types.ModuleType("_ref_losses"); pkg.__path__ = []
    op = types.ModuleType("_ref_losses.op")

    class FusedLeakyReLU(nn.Module):   # same parameters as reference op/fused_act.py:93-108
        def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
            self.negative_slope, self.scale = negative_slope, scale

        def forward(self, input):
            return DO.fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)

    gradfix = types.ModuleType("_ref_losses.op.conv2d_gradfix")
    gradfix.conv2d = lambda input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1: F.conv2d(input, weight, bias, stride, padding, dilation, groups)
    op.FusedLeakyReLU, op.fused_leaky_relu, op.upfirdn2d, op.conv2d_gradfix = FusedLeakyReLU, DO.fused_leaky_relu, DO.upfirdn2d, gradfix
    kornia = types.ModuleType("kornia"); kf = types.ModuleType("kornia.filters"); kf.filter2d = None; kornia.filters = kf
    sys.modules.update({"_ref_losses": pkg, "_ref_losses.op": op, "_ref_losses.op.conv2d_gradfix": gradfix, "kornia": kornia, "kornia.filters": kf})
    spec = importlib.util.spec_from_file_location("_ref_losses.layers", os.path.join(REF, "enhancing/losses/layers.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_ref_losses.layers"] = mod
    spec.loader.exec_module(mod)
    return mod


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main(size=32, B=8, param_seed=21, data_seed=22):
    L = load_reference_layers()
    torch.manual_seed(param_seed)
    D = L.StyleDiscriminator(size=size)
    with torch.no_grad():   # biases start at zero in the reference; give them values so the golden case exercises them
        for n, p in D.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape))
    sd = {k: v for k, v in D.state_dict().items()}
    g = torch.Generator().manual_seed(data_seed)
    real = torch.rand(B, 3, size, size, generator=g)
    fake = (real + 0.1 * torch.randn(B, 3, size, size, generator=g)).clamp(0, 1)

    # --- the reference's own code: logits, d-loss with R1 exactly as vqperceptual.py:148-162 spells it, generator-side g_loss gradient
    D.train()
    x = real.clone().requires_grad_(True)
    logits_real = D(x)
    logits_fake = D(fake.detach())
    d_loss = L.vanilla_d_loss(logits_fake, logits_real)
    gradients, = torch.autograd.grad(outputs=logits_real.sum(), inputs=x, create_graph=True)
    r1 = gradients.square().sum([1, 2, 3]).mean()
    d_loss = d_loss + 10 * 16 * r1 / 2
    D.zero_grad()
    d_loss.backward()
    ref_grads = {n: p.grad.clone() for n, p in D.named_parameters()}
    xf = fake.clone().requires_grad_(True)
    g_loss = L.vanilla_d_loss(D(xf))
    g_fake, = torch.autograd.grad(g_loss, xf)

    # --- the restatement must agree
    sdp = {k: v.clone().requires_grad_(v.is_floating_point() and "kernel" not in k) for k, v in sd.items()}
    o_loss, o_real, o_fake, o_r1 = O.discriminator_loss(sdp, size, real, fake, True)
    o_loss.backward()
    worst = max(rel(sdp[n].grad, ref_grads[n]) for n in ref_grads)
    xf2 = fake.clone().requires_grad_(True)
    o_g = O.vanilla_d_loss(O.discriminator({k: v.detach() for k, v in sd.items()}, xf2, size))
    o_gfake, = torch.autograd.grad(o_g, xf2)
    print(f"disc_tiny: oracle vs reference: logits {rel(o_real, logits_real):.2e} / {rel(o_fake, logits_fake):.2e}, d_loss {abs(o_loss.item() - d_loss.item()):.2e}, "
          f"r1 {abs(o_r1.item() - r1.item()) / r1.item():.2e}, worst param grad {worst:.2e}, g_loss grad {rel(o_gfake, g_fake):.2e}")
    assert rel(o_real, logits_real) < 1e-5 and rel(o_fake, logits_fake) < 1e-5 and worst < 1e-4 and rel(o_gfake, g_fake) < 1e-5
    names = sorted(ref_grads)
    np.savez_compressed(os.path.join(GOLD, "disc_tiny.npz"), size=size, B=B, param_seed=param_seed, data_seed=data_seed,
                        logits_real=logits_real.detach().numpy(), logits_fake=logits_fake.detach().numpy(), d_loss=d_loss.item(), r1=r1.item(),
                        dx_real=gradients.detach().numpy(), g_loss=g_loss.item(), g_fake=g_fake.numpy(),
                        grad_names=np.array(names), grad_norms=np.array([ref_grads[n].double().norm().item() for n in names]),
                        g_final_bias=ref_grads["final_conv.1.bias"].numpy(), g_rgb_w=ref_grads["blocks.0.0.weight"].numpy(),
                        g_lin1_w=ref_grads["final_linear.1.weight"].numpy())
    print("wrote", os.path.join(GOLD, "disc_tiny.npz"))


if __name__ == "__main__":
    main()
