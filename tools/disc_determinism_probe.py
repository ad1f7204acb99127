"""debug aid: which discriminator outputs / gradients are not bit-reproducible from run to run (fixed inputs, fixed weights, one process)?"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("enhancing-transformers_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import torch.nn.functional as F
from enhancing.losses.layers import StyleDiscriminator
from enhancing.losses.op import conv2d_gradfix, conv_nhwc
from enhancing.engine.stage1 import ParamStore
torch.manual_seed(0)
D = StyleDiscriminator(size=64).cuda()
store = ParamStore(D, torch.device("cuda:0"), precision="fp32")
g = torch.Generator(device="cuda").manual_seed(1)
real = torch.rand(2, 3, 64, 64, device="cuda", generator=g)
fake = (real + 0.1 * torch.randn(2, 3, 64, 64, device="cuda", generator=g)).clamp(0, 1)

def once(r1: bool):
    store.zero_grad()
    conv_nhwc.invalidate_packed_weights()
    x = real.clone().requires_grad_(r1)
    lr_, lf_ = D(x), D(fake)
    loss = 0.5 * (F.softplus(-lr_).mean() + F.softplus(lf_).mean())
    if r1:
        with conv2d_gradfix.no_weight_gradients():
            gr, = torch.autograd.grad(lr_.sum(), x, create_graph=True)
        loss = loss + 80 * gr.square().sum([1, 2, 3]).mean()
    loss.backward()
    torch.cuda.synchronize()
    return dict(logits_real=lr_.detach().clone(), logits_fake=lf_.detach().clone(), loss=loss.detach().clone(),
                **{n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None})

for r1 in (False, True):
    ref = once(r1)
    bad = {}
    for it in range(30):
        cur = once(r1)
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                bad[k] = bad.get(k, 0) + 1
    print(f"R1={r1}: tensors that differed from the first run in 30 repetitions: {bad}")
