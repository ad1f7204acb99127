"""debug aid: reach a state where D(xrec) is not bit-reproducible, then find the first layer whose output differs between repeated calls"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("enhancing-transformers_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["ENH_LPIPS_RANDOM_INIT"] = "1"
warnings.simplefilter("ignore")
import torch
import vitvq_oracle as O
from enhancing.modules.stage1.vitvqgan import ViTVQ
from enhancing.utils.general import AttrDict
from enhancing.losses.op import conv_nhwc
import test_uninit_gpu as T
cfg = O.TINY_CFG
loss_cfg, _ = T.CASES["two_optimizer_lpips_disc_r1"]
torch.manual_seed(0)
m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss_cfg))
m.load_state_dict(O.make_params(cfg, seed=11), strict=False)
m.train(); m.learning_rate = 1e-4
opts, _ = m.configure_optimizers()
eng = m.engine
D = m.loss.discriminator

def layers(x):
    outs = []
    out = conv_nhwc.image_to_nhwc8(x); outs.append(("image_to_nhwc8", out))
    for bi, blk in enumerate(D.blocks):
        if hasattr(blk, "conv1"):      # StyleBlock: dissect
            o1 = blk.conv1.forward_nhwc(out) if hasattr(blk.conv1, "forward_nhwc") else None
            out = blk.forward_nhwc(out)
            if o1 is not None:
                outs.append((f"blocks.{bi}.conv1", o1))
        else:
            out = blk.forward_nhwc(out)
        outs.append((f"blocks.{bi}", out))
    B = out.shape[0]
    group = min(B, D.stddev_group); group = B // (B // group)
    out = conv_nhwc.minibatch_stddev(out, group); outs.append(("stddev", out))
    out = D.final_conv.forward_nhwc(out); outs.append(("final_conv", out))
    out = out.permute(0, 3, 1, 2).reshape(B, -1).float()
    h = D.final_linear[0](out); outs.append(("final_linear.0", h))
    lg = D.final_linear[1](h); outs.append(("final_linear.1", lg))
    return outs

found = False
for i in range(6):
    b = {"image": O.make_images(5 + i, 2, cfg["image_size"])}
    for oi, opt in enumerate(opts):
        m.training_step(b, i, oi); opt.step()
        with torch.no_grad():
            xrec, _, _ = eng.reconstruct(b["image"])
            ref = layers(xrec)
            for rep in range(10):
                cur = layers(xrec)
                diff = [(n, int((a.float() != c.float()).sum()), tuple(a.shape)) for (n, a), (_, c) in zip(ref, cur) if not torch.equal(a, c)]
                if diff:
                    print(f"step {i} opt {oi} rep {rep}: first differing layers: {diff[:4]}")
                    found = True
                    break
        if found:
            break
    if found:
        break
print("found" if found else "no non-reproducible forward in 6 steps")
