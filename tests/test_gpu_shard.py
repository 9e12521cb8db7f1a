"""BASELINE configs[4]: 65 536 NAS-Bench-201 graphs as 8 shards of 8192.  On one GPU: the noise a shard
draws on the device equals its rows of ONE global Philox stream (keyed by the global element index), whichever rank
draws it, and the final gather runs through RCCL (a one-rank `nccl` group proves the library loads and the device path
of all_gather_into_tensor works).  The per-evaluation parity of an 8192-graph shard against the oracle is in
tests/test_gpu_parity.py::test_full_batch_guided_steps_match_oracle.  The exact-global-batch Langevin mode (SURVEY.md 8e,
"optional") is run by two processes sharing this GPU and compared bit for bit with the unsharded batch, and bench.py's
multi-GPU path (barrier, MAX all_reduce of the time, gather, JSON line) runs once through a one-rank `nccl` group."""

import os

import pytest
import torch

from diffusionnag_amd import cate as _cate  # noqa: F401
from diffusionnag_amd import distributed as D
from diffusionnag_amd import mutils, sampling, sde_lib, synthetic, utils
from diffusionnag_amd.specs import CateDims
from tests.helpers import score_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHARD, WORLD = 8192, 8


def _one_step(model, B, x0, stream_offset):
    """One predictor step without a corrector (no batch coupling): x_pred = x_mean + G z with z drawn on the device."""
    cfg = score_config(CateDims(), DEV)
    cfg.sampling.corrector = "none"
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), 1e-3)
    fn.noise_tape = {"x0": x0, "corrector": None, "predictor": None}     # given prior sample, Philox noise
    fn.max_steps, fn.record = 1, True
    fn.seed, fn.stream_offset = 77, stream_offset
    fn(model, utils.nb201_adj().unsqueeze(0))
    return fn.last_record["rec_x_mean"][0], fn.last_record["rec_x_pred"][0]


def test_shard_noise_is_a_slice_of_one_global_stream():
    total = SHARD * WORLD
    model = mutils.create_model(score_config(CateDims(), DEV))
    model.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=0, preset="contractive"))
    model.eval()
    x0 = torch.randn(total, 8, 7, generator=torch.Generator().manual_seed(5)).to(DEV)
    mean_all, pred_all = _one_step(model, total, x0, 0)                 # one 65 536-graph stream
    z_all = pred_all - mean_all
    assert abs(float(z_all.std()) / 0.1414 - 1.0) < 0.05               # G = sqrt(beta_max / N) = 0.141 at t = 1
    for r in (0, 7):
        lo, hi = D.shard_batch(total, r, WORLD)[0], sum(D.shard_batch(total, r, WORLD))
        assert (lo, hi) == (r * SHARD, (r + 1) * SHARD)
        mean_r, pred_r = _one_step(model, SHARD, x0[lo:hi].contiguous(), lo * 56)
        assert torch.equal(mean_r, mean_all[lo:hi])                     # per-graph arithmetic does not depend on the batch
        assert torch.equal(pred_r, pred_all[lo:hi]), f"rank {r}: shard noise is not rows [{lo}, {hi}) of the global stream"
    # a shard that forgets its offset draws rank 0's noise instead
    _, pred_bad = _one_step(model, SHARD, x0[7 * SHARD:].contiguous(), 0)
    assert not torch.equal(pred_bad, pred_all[7 * SHARD:])


def test_gather_runs_through_rccl_on_a_one_rank_group():
    import torch.distributed as dist
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        x = torch.randn(SHARD, 8, 7, device=DEV)
        out = D.gather_samples(x, SHARD)
        torch.cuda.synchronize()
        assert out.shape == x.shape and out.data_ptr() != x.data_ptr() and torch.equal(out, x)
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# exact-global-batch Langevin mode: two shards = the unsharded batch, bit for bit
# ------------------------------------------------------------------------------------------------
GB, GS = 64, 3          # graphs of the whole batch, PC steps.  Shards of 32: a multiple of the 16 graphs a surrogate workgroup
                        # shares its power-of-two operand scales over, so that a graph meets the same neighbours sharded or not


def _guided_run(B, tape, offset, mode):
    from diffusionnag_amd import digcn_meta as _dm  # noqa: F401
    from diffusionnag_amd.specs import PredictorDims
    from tests.helpers import predictor_config
    cfg = score_config(CateDims(), DEV)
    model = mutils.create_model(cfg)
    model.load_state_dict(synthetic.cate_state_dict(CateDims(), seed=0))
    clf = mutils.create_model(predictor_config(PredictorDims(), DEV))
    clf.load_state_dict(synthetic.predictor_state_dict(PredictorDims(), seed=1))
    sde = sde_lib.VPSDE(0.1, 20.0, 1000)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 8, 7), utils.get_data_inverse_scaler(cfg), 1e-3, conditional=True)
    fn.noise_tape = {k: v.to(DEV) for k, v in tape.items()}
    fn.max_steps, fn.record = GS, True
    fn.stream_offset = offset * 56
    fn.langevin_batch = mode
    out = fn(model.eval(), utils.aug_mask(utils.nb201_adj())[0], clf.eval(), classifier_scale=10000.0,
             task=synthetic.synthetic_task(10, 20, seed=2))
    r = fn.last_record
    return dict(out=out.cpu(), step=r["rec_step_size"].cpu(), x_corr=r["rec_x_corr"].cpu(), x_pred=r["rec_x_pred"].cpu(),
                calls=(fn.last_exchange.calls if fn.last_exchange is not None else 0))


def _shard_worker(rank, world, port, path):
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        tape = synthetic.noise_tape(GS, (GB, 8, 7), 91)
        lo, n = D.shard_batch(GB, rank, world)
        mine = {k: (v[lo:lo + n] if k == "x0" else v[:, lo:lo + n]).contiguous() for k, v in tape.items()}
        res = _guided_run(n, mine, lo, "global")
        torch.save(res, os.path.join(path, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_shards_reproduce_the_single_batch_step_size(tmp_path):
    """Two processes (gloo between them, both on this GPU) sample the two halves of a 64-graph batch in the
    exact-global-batch mode: every corrector update exchanges the per-block norm sums (dnag_pc_args.langevin_exchange)
    and reduces them in the order of the whole batch, so the step sizes -- and with them every state -- carry the same
    bits as ONE process sampling all 64 graphs; in the default per-shard mode they differ."""
    import socket
    import torch.multiprocessing as mp
    tape = synthetic.noise_tape(GS, (GB, 8, 7), 91)
    whole = _guided_run(GB, tape, 0, "shard")                   # one batch: its own mean IS the global mean
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    for r in range(2):
        assert parts[r]["calls"] == GS
        assert torch.equal(parts[r]["step"], whole["step"]), f"rank {r}: step sizes differ from the unsharded batch"
    for key in ("x_corr", "x_pred"):
        assert torch.equal(torch.cat([parts[0][key], parts[1][key]], dim=1), whole[key]), key
    assert torch.equal(torch.cat([parts[0]["out"], parts[1]["out"]]), whole["out"])
    # the default mode: the first shard alone is its own Langevin batch
    half = {k: (v[:GB // 2] if k == "x0" else v[:, :GB // 2]).contiguous() for k, v in tape.items()}
    alone = _guided_run(GB // 2, half, 0, "shard")
    assert not torch.equal(alone["step"], whole["step"])


def test_bench_sampling_runs_through_a_one_rank_nccl_group(capsys):
    """bench.py's --gpus N path on one GPU: the barrier, the MAX all_reduce of the time, the RCCL gather of the samples
    and the JSON line, with one 8192-graph shard (BASELINE configs[4]) and two PC steps; then once more in the
    exact-global-batch Langevin mode, whose per-update all_gather then runs through RCCL on the device."""
    import argparse
    import torch.distributed as dist
    import bench
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        lines = {}
        for mode in ("shard", "global"):
            a = argparse.Namespace(gpus=1, steps=1, warmup=1, batch_per_gpu=SHARD, no_extra=True, sde_steps=2,
                                   no_cpu_baseline=True, workload="nb201", langevin=mode)
            line = bench.bench_sampling(a, "nb201", 1, 1)
            assert line is not None and line["n_gpus"] == 1 and line["outputs_finite"]
            assert line["config"]["global_batch"] == SHARD and line["config"]["pc_steps"] == 2
            assert line["config"]["langevin_batch"].startswith("global" if mode == "global" else "per shard")
            assert line["value"] > 0 and line["roofline"]["launches_timed"] >= 1
            import json
            json.dumps(line)
            lines[mode] = line
    finally:
        dist.destroy_process_group()
