"""debug aid: along a training sequence, repeat every training_step on its (fixed) weights and compare bits"""
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
torch.manual_seed(0)
m = ViTVQ("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(cfg["quantizer"]), AttrDict.wrap(loss_cfg))
m.load_state_dict(O.make_params(cfg, seed=11), strict=False)
m.train(); m.learning_rate = 1e-4
opts, _ = m.configure_optimizers()
eng = m.engine; ds = m.loss.disc_store(eng.device)
D = m.loss.discriminator
def snap(b, i, oi):
    l = m.training_step(b, i, oi); torch.cuda.synchronize()
    return dict(loss=l.clone(), ag=eng.store.g.clone(), dg=ds.g.clone(), **{k: v.clone() for k, v in m.logged.items() if torch.is_tensor(v)})
for i in range(4):
    b = {"image": O.make_images(5 + i, 2, cfg["image_size"])}
    for oi, opt in enumerate(opts):
        ref = snap(b, i, oi)
        bad = {}
        for rep in range(8):
            cur = snap(b, i, oi)
            for k in ref:
                if not torch.equal(ref[k], cur[k]):
                    bad[k] = bad.get(k, 0) + 1
        # the discriminator alone on this step's reconstruction
        with torch.no_grad():
            xrec, _, _ = eng.reconstruct(b["image"])
            lf = [D(xrec).clone() for _ in range(6)]
            xr2, _, _ = eng.reconstruct(b["image"])
        print(f"step {i} opt {oi}: not reproducible on fixed weights (8 reps): {bad}; D(xrec) x6 equal: {all(torch.equal(lf[0], t) for t in lf)}; reconstruct x2 equal: {torch.equal(xrec, xr2)}")
        opt.step()
