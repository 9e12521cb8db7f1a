"""Race screen for the head GEMMs' barrier / vmcnt structure: the same evaluation repeated many times at several batch
sizes must give bitwise identical outputs (an LDS-DMA read placed one phase early shows up as rare differing tiles)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate  # noqa
from diffusionnag_amd.specs import CateDims
from tests.helpers import score_config
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for shape, sizes in (("nb201", (64, 1000, 4096, 8192)), ("ofa", (13, 512, 2048))):
    dims = CateDims() if shape == "nb201" else CateDims(max_node=20, n_vocab=9, stagewise_pos=True)
    model = mutils.create_model(score_config(dims, dev)); model.load_state_dict(synthetic.cate_state_dict(dims, 0)); model.eval()
    m = (utils.aug_mask(utils.nb201_adj()) if shape == "nb201" else utils.ofa_adj().unsqueeze(0)).to(dev)
    for B in sizes:
        torch.manual_seed(B)
        X = torch.randn(B, dims.max_node, dims.n_vocab, device=dev); t = torch.full((1,), 321.0, device=dev)
        ref = model(X, t, m).clone()
        bad = 0
        for i in range(reps):
            out = model(X, t, m)
            if not torch.equal(out, ref):
                bad += 1
        print(f"{shape} B={B}: {reps} repeats, {bad} differ from the first", flush=True)
        assert torch.isfinite(ref).all()
        if bad:
            sys.exit(1)
print("race screen clean")
