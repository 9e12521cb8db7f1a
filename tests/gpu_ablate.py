"""Timing ablation of the encoder and surrogate kernels (debug env masks; outputs are wrong while a
phase is skipped).  python tests/gpu_ablate.py"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from diffusionnag_amd import mutils, synthetic, utils
from diffusionnag_amd import cate, digcn_meta  # noqa
from diffusionnag_amd.specs import CateDims, PredictorDims
from tests.helpers import score_config, predictor_config
import ctypes as C
from diffusionnag_amd import _lib

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4096"))
model = mutils.create_model(score_config(CateDims(), dev)); model.load_state_dict(synthetic.cate_state_dict(CateDims(), 0)); model.eval()
clf = mutils.create_model(predictor_config(PredictorDims(), dev)); clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), 1)); clf.eval()
X = torch.randn(B, 8, 7, device=dev); t = torch.full((1,), 500.0, device=dev); m = utils.aug_mask(utils.nb201_adj()).to(dev)
task = synthetic.synthetic_task(10, 20, 2).to(dev)
clf.sample_state = True; clf.D_mu = None

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

base = timeit(lambda: clf.input_grad(X, t, m, task=task))
print(f"predictor total {base:.1f} us")
for bit, name in enumerate(["layer0 fwd", "mix_fwd", "fwd GEMMs", "head fwd", "head bwd", "mix_bwd", "bwd GEMMs", "layer0 bwd"]):
    os.environ["DNAG_PRED_SKIP"] = str(1 << bit)
    v = timeit(lambda: clf.input_grad(X, t, m, task=task))
    print(f"  skip {name:12s}: {v:7.1f} us  (phase ~ {base - v:6.1f})")
os.environ["DNAG_PRED_SKIP"] = "255"
print(f"  skip all: {timeit(lambda: clf.input_grad(X, t, m, task=task)):.1f} us")
os.environ["DNAG_PRED_SKIP"] = "0"

# encoder alone is not exposed: time the whole forward and subtract
base = timeit(lambda: model(X, t, m))
print(f"cate forward total {base:.1f} us")
for bit, name in enumerate(["QKV", "attention", "O-proj", "FFN1", "FFN2", "ln_stats", "concat write"]):
    os.environ["DNAG_ENC_SKIP"] = str(1 << bit)
    v = timeit(lambda: model(X, t, m))
    print(f"  skip {name:12s}: {v:7.1f} us  (phase ~ {base - v:6.1f})")
os.environ["DNAG_ENC_SKIP"] = "127"
print(f"  skip all: {timeit(lambda: model(X, t, m)):.1f} us")
os.environ["DNAG_ENC_SKIP"] = "0"
