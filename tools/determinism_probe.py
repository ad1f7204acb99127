"""Two training sequences of the two-optimizer protocol from the same seed, snapshot after every training_step / optimizer step: first difference?"""
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
import test_uninit_gpu as T
cfg = O.TINY_CFG
loss_cfg, _ = T.CASES["two_optimizer_lpips_disc_r1"]

def run():
    torch.manual_seed(0)
    m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss_cfg))
    m.load_state_dict(O.make_params(cfg, seed=11), strict=False)
    m.train(); m.learning_rate = 1e-4
    opts, _ = m.configure_optimizers()
    eng = m.engine; ds = m.loss.disc_store(eng.device)
    snaps = []
    for i in range(3):
        b = {"image": O.make_images(5 + i, 2, cfg["image_size"])}
        for oi, opt in enumerate(opts):
            l = m.training_step(b, i, oi)
            torch.cuda.synchronize()
            snaps.append((f"step {i} opt {oi} after training_step", dict(loss=l.clone(), ag=eng.store.g.clone(), dg=ds.g.clone(),
                          **{k: v.clone() for k, v in m.logged.items() if torch.is_tensor(v)})))
            opt.step()
            torch.cuda.synchronize()
            snaps.append((f"step {i} opt {oi} after optimizer", dict(ap=eng.store.p.clone(), dp=ds.p.clone(), am=eng.store.m.clone(), dm=ds.m.clone())))
    return snaps

ref = run()
for trial in range(8):
    cur = run()
    first = None
    for (na, a), (nb, b) in zip(ref, cur):
        for k in a:
            if not torch.equal(a[k], b[k]):
                idx = (a[k] != b[k]).nonzero().view(-1)[:4].tolist() if a[k].dim() else []
                first = (na, k, int((a[k] != b[k]).sum()), idx)
                break
        if first:
            break
    print("trial", trial, "first difference:", first)
