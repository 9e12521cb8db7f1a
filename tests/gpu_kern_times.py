"""Run each per-evaluation kernel a few times at B=4096 (profile with rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate, digcn_meta  # noqa
from diffusionnag_amd.specs import CateDims, PredictorDims
from tests.helpers import score_config, predictor_config
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4096"))
shape = os.environ.get("SHAPE", "nb201")
dims = CateDims() if shape == "nb201" else CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
model = mutils.create_model(score_config(dims, dev)); model.load_state_dict(synthetic.cate_state_dict(dims, 0)); model.eval()
X = torch.randn(B, dims.max_node, dims.n_vocab, device=dev); t = torch.full((1,), 500.0, device=dev)
m = (utils.aug_mask(utils.nb201_adj()) if shape == "nb201" else utils.ofa_adj().unsqueeze(0)).to(dev)
for _ in range(20): model(X, t, m)
if shape == "nb201":
    clf = mutils.create_model(predictor_config(PredictorDims(), dev)); clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), 1)); clf.eval()
    task = synthetic.synthetic_task(10, 20, 2).to(dev)
    clf.sample_state = True; clf.D_mu = None
    for _ in range(20): clf.input_grad(X, t, m, task=task)
torch.cuda.synchronize()
