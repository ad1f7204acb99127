import sys, os, warnings
warnings.simplefilter("ignore")
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,R+"/enhancing-transformers_amd"); sys.path.insert(0,R+"/oracle"); sys.path.insert(0,R+"/tests")
import torch, lpips_oracle as LO
from util import rel
from enhancing.losses.lpips import LPIPS
g = torch.Generator().manual_seed(0)
B,S=2,64
in0 = torch.rand(B,3,S,S,generator=g); in1=(in0+0.1*torch.randn(B,3,S,S,generator=g)).clamp(0,1)
for only in [None,0,1,2,3,4]:
    m = LPIPS()
    if only is not None:
        with torch.no_grad():
            for k in range(5):
                if k!=only: getattr(m,f"lin{k}").model[1].weight.zero_()
        m._dev.clear()
    sd={k:v.clone() for k,v in m.full_state_dict().items()}
    x1=in1.clone().cuda().requires_grad_(True); d=m(in0.cuda(),x1,normalize=True); d.mean().backward()
    o1=in1.clone().requires_grad_(True); do=LO.lpips_distance(in0,o1,sd,normalize=True); do.mean().backward()
    a,b=x1.grad.cpu().double().flatten(), o1.grad.double().flatten()
    print(f"only slice {only}: value rel {rel(d,do):.2e} grad rel {rel(x1.grad,o1.grad):.3e} cos {float((a@b)/(a.norm()*b.norm())):.5f} norm ratio {float(a.norm()/b.norm()):.4f}")
