"""Time of one score-network encoder launch at B=4096 (HIP events around 50 evaluations of CATE.forward's encoder are not
separable from the GEMMs, so this times the whole forward and the profile gives the split); prints ms per forward."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate  # noqa
from diffusionnag_amd.specs import CateDims
from tests.helpers import score_config
dev = "cuda:0"
m = mutils.create_model(score_config(CateDims(), dev)); m.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=0)); m.eval()
X = torch.randn(4096, 8, 7, device=dev); t = torch.full((1,), 500.0, device=dev); mk = utils.aug_mask(utils.nb201_adj()).to(dev)
for _ in range(5): m(X, t, mk)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): m(X, t, mk)
torch.cuda.synchronize(); print(f"forward {1e3 * (time.perf_counter() - t0) / 100:.4f} ms")
