#!/usr/bin/env python
"""Benchmark of the sampling hot path: guided NAS-Bench-201 architecture samples/s for the FULL reverse
SDE (BASELINE.json configs[1]: predictor-guided sampling, batch 4096 per GPU, N = 1000 PC steps =
2000 score evaluations + 2000 guidance gradients per sample, VPSDE + reverse_diffusion + langevin).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete trajectory of one batch (weights, mask and D_mu resident in HBM, noise drawn
on the device by the Philox generator inside the timed region, as the reference draws it inside its
loop).  N > 1: every rank samples its own batch of the same size (weak scaling), no collective during
sampling, one RCCL all_gather of the samples at the end of each step.  Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from diffusionnag_amd import _lib, mutils, sampling, sde_lib, synthetic, utils  # noqa: E402
from diffusionnag_amd import cate as _cate, digcn_meta as _dm, predictor_cate as _pcm  # noqa: E402,F401
from diffusionnag_amd import distributed as D  # noqa: E402
from diffusionnag_amd.specs import CateDims, PredictorCateDims, PredictorDims  # noqa: E402

# algorithmic FLOPs (2 per MAC) of one guided sample (2000 score evaluations + guidance gradients): SURVEY.md section 8d
FLOP_PER_SAMPLE = {"nb201": 136.3e9, "ofa": 345.4e9}
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0     # same guide: BF16/F16 ~2.5 PF dense (a bare v_mfma_f32_16x16x32_f16 loop sustains ~1840 on this
                                  # chip on random operands, tests/micro/mfma_shapes.hip)
# The dominant kernel computes in split arithmetic (diffusionnag_amd/csrc/dnag_gemm_dma.hip.h): every fp32-accurate
# product costs THREE fp16 MFMAs, so `achieved` (ALGORITHMIC flops / time, as the contract defines it) can reach at
# most a third of the fp16 peak; `mfma_issued_*` gives the matrix-core utilisation proper.
SPLIT_MFMA_PER_PRODUCT = 3
# Fabric-side bytes of ONE launch of the dominant kernel (final.linears.1 at B=4096, NB201) from PMC counters:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE doubled as the guide's HBM
# section prescribes for gfx950 (profiles/r03f_pmc_traffic.json: 0.454 GB read + 0.006 GB written; algorithmic
# bytes 0.218 GB = A 201 MB + W 9.4 MB + the fused final.linears.2 partial sums 7.3 MB; the excess is the weight
# matrix fetched once per XCD and round of tiles, 8 x 3 x 9.4 MB, out of the Infinity Cache).  Counters cannot be collected inside this process, so the figure is quoted only for the shape it
# was measured on.
PMC_TRAFFIC_BYTES = {("nb201", 4096): 459930624}


class Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


OFA_DIMS = CateDims(max_node=20, n_vocab=9, stagewise_pos=True)


def score_config(device, d=CateDims()):
    ge = Cfg(n_layers=d.n_layers, d_model=d.d_model, n_head=d.n_head, d_ff=d.d_ff, dropout=0.1, n_vocab=d.n_vocab)
    return Cfg(device=device, data=Cfg(name="ofa" if d.stagewise_pos else "NASBench201", max_node=d.max_node,
                                       n_vocab=d.n_vocab, centered=True),
               model=Cfg(name="CATE", nonlinearity="swish", dropout=0.1, pos_enc_type=2, graph_encoder=ge),
               training=Cfg(continuous=True),
               sampling=Cfg(method="pc", predictor="reverse_diffusion", corrector="langevin", snr=0.16, n_steps_each=1,
                            probability_flow=False, noise_removal=True, regress=True, labels="max"))


def predictor_config(device):
    d = PredictorDims()
    ge = Cfg(initial_hidden=7, gcn_hidden=d.gcn_hidden, gcn_layers=d.gcn_layers, linear_hidden=d.linear_hidden)
    return Cfg(device=device, model=Cfg(name="MetaNeuralPredictor", nonlinearity="swish", input_type="DA", hs=d.hs,
                                        nz=d.nz, num_sample=d.num_sample, graph_encoder=ge))


def predictor_cate_config(device):
    d = PredictorCateDims()
    ge = Cfg(n_layers=d.n_layers, d_model=d.d_model, n_head=d.n_head, d_ff=d.d_ff, dropout=0.1, n_vocab=d.n_vocab)
    return Cfg(device=device, data=Cfg(name="ofa", max_node=d.max_node, n_vocab=d.n_vocab, centered=True),
               model=Cfg(name="MetaPredictorCATE", nonlinearity="swish", dropout=0.1, input_type="DA", hs=d.hs, nz=d.nz,
                         num_sample=d.num_sample, graph_encoder=ge))


def physical_cores():
    """Distinct (package, core) pairs this process may run on (from /proc/cpuinfo and the affinity mask); None if unknown."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, phys = set(), None, None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu, phys = int(v), None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id" and cpu in allowed:
                seen.add((phys, int(v)))
        return len(seen) or None
    except Exception:
        return None


def cpu_baseline(seconds=12.0):
    """The oracle (PyTorch-eager CPU restatement of the reference, validated against its golden vectors)
    timed on this box's host cores on a bounded sample of the same workload: guided NB201 PC steps
    at B=256, extrapolated to the 1000 steps of a trajectory (the cost of a step does not depend on t)."""
    from oracle import dnag_oracle as O
    B = 256
    sd = synthetic.cate_state_dict(CateDims(), seed=0)
    sdp = synthetic.predictor_state_dict(PredictorDims(), seed=1)
    D_mu = O.set_encode(sdp, synthetic.synthetic_task(10, 20, seed=2))
    sde = O.VPSDE(0.1, 20.0, 1000)
    mask = O.floyed(O.nb201_adj())
    kw = dict(pred_sd=sdp, D_mu=D_mu, classifier_scale=10000.0)
    tape = synthetic.noise_tape(16, (B, 8, 7), 1)
    O.pc_sample(sd, sde, mask, tape, n_steps=3, **kw)          # warm-up (oneDNN primitive creation)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and steps < 400:
        O.pc_sample(sd, sde, mask, tape, n_steps=8, **kw)
        steps += 8
    dt = time.perf_counter() - t0
    return {"value": B / (dt / steps * 1000.0), "unit": "samples/s", "cores": torch.get_num_threads(), "physical_cores": physical_cores(), "kind": "port",
            "sample": f"{steps} guided PC steps (2 score evals + 2 guidance grads each) at B={B} in {dt:.1f} s, "
                      f"x1000 steps per trajectory"}


def bench_train(a, emit=True):
    """BASELINE configs[2]: NAS-Bench-201 score-network DSM training step at batch 1024 on one GPU: the reference's
    whole step_fn (losses.py:204-227) -- time / noise draws, forward with dropout 0.1 (on-device Philox masks), loss,
    backward (dnag_cate_dsm_step), then gradient clipping + Adam with warm-up + EMA (dnag_optim_step) -- on resident
    inputs.  Algorithmic FLOPs: 3 x 63.36 MFLOP per graph (forward + data gradient + weight gradient)."""
    from diffusionnag_amd import ema as ema_lib, losses
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B = 1024 if a.batch_per_gpu == 4096 else a.batch_per_gpu
    steps = max(a.steps, 20)
    dims = CateDims()
    model = mutils.create_model(score_config(dev, dims))
    model.load_state_dict(synthetic.cate_state_dict(dims, seed=0))
    model.dropout_p = float(os.environ.get("DNAG_BENCH_DROPOUT", "0.1"))   # 0.1 is the workload; 0 times the no-dropout kernels (profiling only)
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    g = torch.Generator().manual_seed(0)
    ops = torch.randint(0, 7, (B, 8), generator=g)
    x = (torch.nn.functional.one_hot(ops, 7).float() * 2 - 1).to(dev)
    mask = utils.nb201_adj().unsqueeze(0).repeat(B, 1, 1).to(dev)
    # configs/tr_scorenet.py: Adam lr 2e-5, beta1 0.9, eps 1e-8, no weight decay, warm-up 1000 steps, grad_clip 1, EMA 0.9999
    cfg = Cfg(optim=Cfg(optimizer="Adam", lr=2e-5, beta1=0.9, eps=1e-8, weight_decay=0.0, warmup=1000, grad_clip=1.0))
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()), step=0,
                 ema=ema_lib.ExponentialMovingAverage(model.parameters(), decay=0.9999))
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True)
    model.train()
    batch = (x, mask, mask)
    for i in range(max(a.warmup, 3)):
        loss = step_fn(state, batch)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step_fn(state, batch)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    flop = 3 * 63.36e6 * B
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    line = {"metric": "DSM training steps/sec (forward + backward + optimiser), NB201", "value": steps / dt, "unit": "steps/s", "n_gpus": 1,
            "steps": steps, "warmup": max(a.warmup, 3), "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: NAS-Bench-201 score-network DSM training step, batch {B}, "
                                   "VPSDE beta 0.1-20, dropout 0.1 (Philox masks on device), raw adjacency mask, random-init weights, "
                                   "Adam lr 2e-5 with warm-up, grad_clip 1, EMA 0.9999",
                       "batch_per_gpu": B, "includes": "the reference's step_fn: draws of t and z, on-device weight packing, forward, loss, "
                                                      "backward, gradient clipping, Adam, EMA (one flat parameter buffer; no host sync)"},
            "graphs_per_s": B * steps / dt, "whole_step_tflops": flop * steps / dt / 1e12,
            "whole_step_frac_of_f32_mfma_peak": flop * steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "loss": float(loss.detach()), "params_finite": bool(torch.isfinite(flat).all()),
            "grad_norm_last": float(state["optimizer"].last_grad_norm)}
    if emit:
        print(json.dumps(line), flush=True)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="graphs per GPU; default 4096 at --gpus 1 (BASELINE configs[1]) and 8192 at --gpus N > 1 "
                         "(BASELINE configs[4]: 65 536 graphs as 8 shards of 8192, weak scaling)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short OFA and training side measurements (N = 1 only)")
    ap.add_argument("--sde-steps", type=int, default=1000, help="PC steps per trajectory (1000 = the full reverse SDE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--langevin", choices=["shard", "global"], default="shard",
                    help="batch of the Langevin step size (reference sampling.py:249-252): 'shard' = every GPU's batch on its own "
                         "(what one run of the reference per device does; no collective inside the trajectory), 'global' = all "
                         "shards as ONE batch (one 8 KB RCCL all_gather per corrector update, bit-identical to the unsharded run)")
    ap.add_argument("--workload", choices=["nb201", "ofa", "train"], default="nb201",
                    help="nb201 = BASELINE configs[1] (the headline metric); ofa = configs[3] (20x9 graphs, MetaPredictorCATE), "
                         "a side measurement that never replaces the headline line; train = configs[2] (DSM training step, "
                         "forward + backward, batch 1024), also a side measurement")
    a = ap.parse_args()
    if a.batch_per_gpu is None:
        a.batch_per_gpu = 4096 if a.gpus == 1 else 8192
    if a.workload == "train":
        return bench_train(a)
    line = bench_sampling(a, a.workload, a.steps, a.warmup)
    if line is not None and a.gpus == 1 and a.workload == "nb201" and not a.no_extra:
        # side workloads, driver-timed with the headline (short runs; they never replace `value`)
        ofa = bench_sampling(a, "ofa", 1, 1)
        tr = bench_train(a, emit=False)
        line["extra"] = {
            "ofa_guided_samples_per_s": ofa["value"], "ofa_ms_per_step": ofa["ms_per_step"], "ofa_workload": ofa["config"]["workload"],
            "ofa_whole_path_tflops": ofa["whole_path_tflops"],
            "train_steps_per_s": tr["value"], "train_ms_per_step": tr["ms_per_step"], "train_workload": tr["config"]["workload"],
            "train_whole_step_tflops": tr["whole_step_tflops"],
        }
    if line is not None:
        print(json.dumps(line), flush=True)
    if a.gpus > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def bench_sampling(a, workload, steps, warmup):
    """One sampling measurement; returns the JSON line on rank 0 (None elsewhere)."""

    rank, world, local = D.init_from_env()
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()   # also a one-rank group (tests)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B = a.batch_per_gpu
    total = B * world

    ofa = workload == "ofa"
    dims = OFA_DIMS if ofa else CateDims()
    N, V = dims.max_node, dims.n_vocab
    cfg = score_config(dev, dims)
    model = mutils.create_model(cfg)
    model.load_state_dict(synthetic.cate_state_dict(dims, seed=0))
    if ofa:
        clf = mutils.create_model(predictor_cate_config(dev))
        clf.load_state_dict(synthetic.predictor_cate_state_dict(PredictorCateDims(), seed=5))
        mask = synthetic.ofa_adjacency(N).to(dev)                # chain DAG (MobileNetV3/main_exp/diffusion/run_lib.py:147-153)
        cscale = 1000.0
    else:
        clf = mutils.create_model(predictor_config(dev))
        clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), seed=1))
        mask = utils.aug_mask(utils.nb201_adj())[0].to(dev)      # Floyd closure, as generate_archs_meta builds it
        cscale = 10000.0
    task = synthetic.synthetic_task(10, 20, seed=2).to(dev)
    sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    fn = sampling.get_sampling_fn(cfg, sde, (B, N, V), utils.get_data_inverse_scaler(cfg), 1e-3, conditional=True)
    if a.sde_steps != 1000:
        fn.max_steps = a.sde_steps                               # debugging only: the reported metric needs 1000
    start, _ = D.shard_batch(total, rank, world)
    D.configure_sampler_shard(fn, start, N, V, seed=1234)
    langevin = getattr(a, "langevin", "shard")
    if langevin == "global":
        if not grouped:
            raise SystemExit("--langevin global needs a torch.distributed group (launch with torch.distributed.run)")
        fn.langevin_batch = "global"
    torch.manual_seed(1000 + rank)                               # x0 is drawn on the CPU generator, like the reference

    def step():
        x = fn(model, mask, clf, classifier_scale=cscale, task=task)
        return D.gather_samples(x, total)

    def fence():
        torch.cuda.synchronize(dev)
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        out = step()
    lib = _lib.load()
    _lib.check(lib.dnag_timing_begin(16, 2048), "dnag_timing_begin")
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    import ctypes as C
    n_s, mean_ms, flops = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
    _lib.check(lib.dnag_timing_read(C.byref(n_s), C.byref(mean_ms), C.byref(flops)), "dnag_timing_read")
    lib.dnag_timing_end()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if grouped:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    finite = bool(torch.isfinite(out).all())

    line = None
    if rank == 0:
        value = total * steps / dt
        achieved = flops.value / (mean_ms.value * 1e-3) / 1e12 if mean_ms.value > 0 else 0.0
        flop_per_sample = FLOP_PER_SAMPLE[workload]
        line = {
            "metric": "guided arch samples/sec (full reverse SDE), " + ("OFA" if ofa else "NB201"), "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (matrix products on f16 hi/lo operand pairs, 22 significant bits, f32 accumulation)", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3]: MobileNetV3/OFA 20-node graphs, guided sampling (MetaPredictorCATE gradient), "
                                    f"batch {B} per GPU, VPSDE beta 0.1-20 N={a.sde_steps}, reverse_diffusion + langevin, "
                                    "snr 0.16, classifier_scale 1e3, chain mask, random-init weights (seeded), Philox noise on device")
                       if ofa else
                                   ((f"BASELINE configs[4]: NAS-Bench-201 predictor-guided sampling, {total} graphs sharded over {world} GPUs "
                                     "(one Langevin batch per shard, RCCL all_gather of the samples), " if world > 1 else
                                     "BASELINE configs[1]: NAS-Bench-201 predictor-guided sampling (meta-predictor gradient), ") +
                                    f"batch {B} per GPU, VPSDE beta 0.1-20 N={a.sde_steps}, reverse_diffusion + langevin, "
                                    "snr 0.16, classifier_scale 1e4, Floyd-closure mask, random-init weights (seeded), "
                                    "Philox noise on device"),
                       "batch_per_gpu": B, "global_batch": total, "pc_steps": a.sde_steps,
                       "score_evals_per_sample": 2 * a.sde_steps, "parallelism": f"batch-shard x{world}",
                       "langevin_batch": ("global: all shards as one batch, one all_gather of per-block norm sums per corrector update"
                                          if langevin == "global" else
                                          "per shard: every GPU's batch is its own Langevin batch, no collective inside the trajectory")},
            "roofline": {"bound": "mfma", "kernel": "gemm_dma_kernel<ELU,false,V,true> (final.linears.1, M x 1536 x 1536, with final.linears.2 fused into its epilogue; fp32-accurate "
                                                    "hi/lo split on v_mfma_f32_16x16x32_f16, 3 MFMAs per product)",
                         "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_MFMA_TFLOPS,
                         "mfma_issued_tflops": achieved * SPLIT_MFMA_PER_PRODUCT,
                         "mfma_issued_frac": achieved * SPLIT_MFMA_PER_PRODUCT / PEAK_F16_MFMA_TFLOPS,
                         "achieved_over_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": PMC_TRAFFIC_BYTES.get((workload, B)),
                         "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r03f_pmc_traffic.json)",
                         "launch_ms": mean_ms.value, "launches_timed": n_s.value, "flops_per_launch": flops.value,
                         "streams": ("the surrogate's gradient runs on the library's second stream beside the score network "
                                     "(dnag_lib.hip SideStream; DNAG_OVERLAP=0 disables): launch_ms is this kernel's time in "
                                     "that mix -- alone on the chip for NB201 (0.31 ms on one stream too), shared with the "
                                     "surrogate's kernels for OFA (0.80 ms on one stream)")},
            "whole_path_tflops": value * flop_per_sample * (a.sde_steps / 1000.0) / 1e12 / world,
            "whole_path_frac_of_f32_mfma_peak": value * flop_per_sample * (a.sde_steps / 1000.0) / 1e12 / world / PEAK_F32_MFMA_TFLOPS,
            "outputs_finite": finite,
        }
        if not a.no_cpu_baseline and world == 1 and not ofa:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
    return line


if __name__ == "__main__":
    main()
