"""How long does the HOST take to enqueue one training step (no sync), against the GPU's time per step?"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from diffusionnag_amd import ema as ema_lib, losses, mutils, sde_lib, synthetic, utils
from diffusionnag_amd.specs import CateDims
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
B = 1024; dims = CateDims()
model = mutils.create_model(bench.score_config(dev, dims)); model.load_state_dict(synthetic.cate_state_dict(dims, seed=0)); model.dropout_p = 0.1
sde = sde_lib.VPSDE(0.1, 20.0, 1000)
g = torch.Generator().manual_seed(0)
x = (torch.nn.functional.one_hot(torch.randint(0, 7, (B, 8), generator=g), 7).float() * 2 - 1).to(dev)
mask = utils.nb201_adj().unsqueeze(0).repeat(B, 1, 1).to(dev)
cfg = bench.Cfg(optim=bench.Cfg(optimizer="Adam", lr=2e-5, beta1=0.9, eps=1e-8, weight_decay=0.0, warmup=1000, grad_clip=1.0))
state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()), step=0, ema=ema_lib.ExponentialMovingAverage(model.parameters(), decay=0.9999))
step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True)
model.train(); batch = (x, mask, mask)
for i in range(5): step_fn(state, batch)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for i in range(N): step_fn(state, batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, until the GPU is done {1e3 * (t2 - t0) / N:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20): step_fn(state, batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
