"""debug aid for tests/test_uninit_gpu.py: which discriminator-gradient elements depend on the allocator's history?"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("enhancing-transformers_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["ENH_LPIPS_RANDOM_INIT"] = "1"
import torch
import test_uninit_gpu as T
real_empty, real_like, real_new = torch.empty, torch.empty_like, torch.Tensor.new_empty
state = {"on": False}
torch.empty = lambda *a, **k: (T._poison(real_empty(*a, **k)) if state["on"] and k.get("device") not in (None, "cpu") or False else real_empty(*a, **k))
def empty(*a, **k):
    t = real_empty(*a, **k); return T._poison(t) if state["on"] and t.is_cuda else t
def empty_like(*a, **k):
    t = real_like(*a, **k); return T._poison(t) if state["on"] and t.is_cuda else t
def new_empty(self, *a, **k):
    t = real_new(self, *a, **k); return T._poison(t) if state["on"] and t.is_cuda else t
torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty
loss_cfg, kw = T.CASES["two_optimizer_lpips_disc_r1"]
warnings.simplefilter("ignore")
for trial in range(3):
    state["on"] = False
    a = T._run_steps(loss_cfg, 3, **kw)
    b_ = T._run_steps(loss_cfg, 3, **kw)
    state["on"] = True
    c = T._run_steps(loss_cfg, 3, **kw)
    state["on"] = False
    for name, x, y in (("clean vs clean", a, b_), ("clean vs poisoned", a, c)):
        d = (x[1][3] != y[1][3]).nonzero().view(-1).tolist()
        print(trial, name, "differing D-grad elements:", d[:10], [(float(x[1][3][i]), float(y[1][3][i])) for i in d[:4]])
from enhancing.losses.layers import StyleDiscriminator
from enhancing.engine.stage1 import ParamStore
D = StyleDiscriminator(size=64)
names, offsets, total = ParamStore.layout([(n, p) for n, p in D.named_parameters() if p.requires_grad])
for n in names:
    o, c, shp = offsets[n]
    print(n, o, o + c, tuple(shp))
