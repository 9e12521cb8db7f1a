"""Single-evaluation error of the score network against the reference's golden outputs: the sampling encoder
(encoder8_kernel) and, by asking for the attention probabilities, the generic encoder_kernel."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic
from diffusionnag_amd import cate  # noqa
from diffusionnag_amd.specs import CateDims
from tests.helpers import golden, score_config
dev = "cuda:0"
for mname in ("raw", "closure"):
    g = golden(f"cate_fwd_nb201_{mname}")
    m = mutils.create_model(score_config(CateDims(), dev)); m.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=int(g["seed"]), perturb_norms=True)); m.eval()
    X = torch.from_numpy(g["X"]).to(dev); t = torch.from_numpy(g["time_cond"]).to(dev)
    mk = torch.from_numpy(g["mask"]).to(dev).unsqueeze(0).repeat(X.shape[0], 1, 1)
    for name, kw in (("encoder8", dict(return_concat=True)), ("generic", dict(return_concat=True, return_attn0=True))):
        out, concat, _ = m(X, t, mk, **kw)
        eo = float((out.cpu() - torch.from_numpy(g["out"])).abs().max() / np.abs(g["out"]).max())
        ec = float((concat.cpu() - torch.from_numpy(g["concat"])).abs().max() / np.abs(g["concat"]).max())
        print(f"{mname:8s} {name:9s} out err {eo:.2e}  concat err {ec:.2e}")
