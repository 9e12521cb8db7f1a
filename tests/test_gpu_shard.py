"""BASELINE configs[4]: 65 536 NAS-Bench-201 graphs as 8 shards of 8192.  On one GPU: the noise a shard
draws on the device equals its rows of ONE global Philox stream (keyed by the global element index), whichever rank
draws it, and the final gather runs through RCCL (a one-rank `nccl` group proves the library loads and the device path
of all_gather_into_tensor works).  The per-evaluation parity of an 8192-graph shard against the oracle is in
tests/test_gpu_parity.py::test_full_batch_guided_steps_match_oracle_on_a_slice."""

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
