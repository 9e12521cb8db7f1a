"""Per-phase cycle stamps of encoder_kernel (library built with -DDNAG_ENC_STAMPS; debug only: device printf)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate  # noqa
from diffusionnag_amd.specs import CateDims
from tests.helpers import score_config
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shape = os.environ.get("SHAPE", "nb201")
dims = CateDims() if shape == "nb201" else CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
model = mutils.create_model(score_config(dims, dev)); model.load_state_dict(synthetic.cate_state_dict(dims, 0)); model.eval()
X = torch.randn(B, dims.max_node, dims.n_vocab, device=dev); t = torch.full((1,), 500.0, device=dev)
m = (utils.aug_mask(utils.nb201_adj()) if shape == "nb201" else utils.ofa_adj().unsqueeze(0)).to(dev)
for _ in range(3): model(X, t, m)
torch.cuda.synchronize()
