import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import ema as ema_lib
torch.manual_seed(0)
p = torch.nn.Parameter(torch.randn(5000, device="cuda:0") * 0.05)
e = ema_lib.ExponentialMovingAverage([p], decay=0.9999)
s0 = e.shadow_params[0].cpu().clone()
with torch.no_grad(): p.add_(torch.randn_like(p) * 1e-3)
e.update([p])
got = e.shadow_params[0].cpu()
omd = 1.0 - min(0.9999, 2 / 11)
exp = s0 - omd * (s0 - p.detach().cpu())
d = (got != exp).nonzero().reshape(-1)
print("differ", d.numel(), "omd32", np.float32(omd))
for i in d[:5].tolist():
    print(i, s0[i].item(), p[i].item(), got[i].item(), exp[i].item(), (s0[i] - p.detach().cpu()[i]).item())
# variants of the expression on the CPU
diff = s0 - p.detach().cpu()
v1 = s0 - torch.tensor(np.float32(omd)) * diff
v2 = torch.addcmul(s0, diff, torch.tensor(-np.float32(omd)))
print("v1==exp", torch.equal(v1, exp), "got==v1", torch.equal(got, v1), "got==fma variant", torch.equal(got, v2))
