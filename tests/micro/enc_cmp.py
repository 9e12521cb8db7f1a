"""Debug: sampling encoder (encoder8_kernel) against the generic encoder_kernel on a full batch, where do they differ."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate  # noqa
from diffusionnag_amd.specs import CateDims
from tests.helpers import score_config
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
m = mutils.create_model(score_config(CateDims(), dev)); m.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=0)); m.eval()
torch.manual_seed(0)
X = torch.randn(B, 8, 7, device=dev) * scale
t = torch.full((1,), 999.0, device=dev)
mk = utils.aug_mask(utils.nb201_adj()).to(dev)
ref_out, ref_c, _ = m(X, t, mk, return_concat=True, return_attn0=True)
for it in range(4):
    out, c, _ = m(X, t, mk, return_concat=True)
    torch.cuda.synchronize()
    e = (c - ref_c).abs() / ref_c.abs().max()
    bad = (e > 1e-5).nonzero()
    print(f"iter {it}: concat max rel err {float(e.max()):.2e}, out err {float((out-ref_out).abs().max()/ref_out.abs().max()):.2e}, bad entries {bad.shape[0]}")
    if bad.shape[0]:
        g = bad[:, 0]; print("  graphs", torch.unique(g)[:20].tolist(), "n graphs", torch.unique(g).numel())
        print("  nodes", torch.unique(bad[:, 1]).tolist(), " slices", torch.unique(bad[:, 2] // 64).tolist(), " cols%64", torch.unique(bad[:, 2] % 64)[:40].tolist())
