"""Per-phase cycle stamps of predictor_kernel (library built with -DDNAG_PRED_STAMPS; debug only)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate, digcn_meta  # noqa
from diffusionnag_amd.specs import PredictorDims
from tests.helpers import predictor_config
dev = torch.device("cuda:0")
B = 4096
X = torch.randn(B, 8, 7, device=dev); t = torch.full((1,), 500.0, device=dev)
m = utils.aug_mask(utils.nb201_adj()).to(dev)
clf = mutils.create_model(predictor_config(PredictorDims(), dev)); clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), 1)); clf.eval()
task = synthetic.synthetic_task(10, 20, 2).to(dev)
clf.sample_state = True; clf.D_mu = None
for _ in range(5): g = clf.input_grad(X, t, m, task=task)
torch.cuda.synchronize()
st = g.reshape(-1)[5 * 16 * 8 * 7: 5 * 16 * 8 * 7 + 64].cpu().tolist()      # block 5 = graphs 80..95 (16 per workgroup)
prev = 0
for i, v in enumerate(st):
    print(i, int(v), int(v - prev)); prev = v
    if i > 2 and v == 0: break
