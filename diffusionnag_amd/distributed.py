"""Batch sharding of the sampler over the GPUs of one node (one process per GPU).

The reference is single-device.  Graphs are independent in the score
network, the surrogate and the predictor update, so the sample batch is cut into contiguous shards,
one per rank, and every rank runs the full reverse SDE on its shard with replicated weights.  The
only collective is the final gather of the ``[B/W, N, V]`` samples (RCCL over xGMI with the ``nccl``
backend; ``gloo`` on CPU for tests).  By default each shard is its own Langevin batch -- the step size
(reference ``sampling.py:249-252``) is a mean over the shard, exactly what running the reference
once per device would do.  ``sampler.langevin_batch = 'global'`` selects the exact-global-batch mode
(SURVEY.md 8e): one small all-gather of per-block norm sums per corrector update (``LangevinExchange``),
after which every rank reduces the same list in the same order, so the step size is bit-identical to that
of ONE batch holding all the graphs.  Philox noise is keyed by the global element index, so the noise a
graph sees does not depend on the number of ranks.  Per-graph arithmetic is independent of the batch except for
the power-of-two operand scales that the surrogate kernel derives per workgroup of 16 graphs: shard boundaries at
multiples of 16 graphs (8192 per GPU in BASELINE configs[4]) keep every bit identical to the unsharded run.
"""

from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns ``(rank, world_size, local_rank)``; a no-op returning ``(0, 1, 0)`` when not launched
    by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard ``(start, count)`` of ``total`` graphs for ``rank``; the first
    ``total % world`` ranks take one extra graph."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def configure_sampler_shard(sampler, start: int, n_node: int, n_vocab: int, seed: int) -> None:
    """Point a sampler built for the local shard shape at its slice of the global noise stream."""
    sampler.seed = seed
    sampler.stream_offset = start * n_node * n_vocab


def gather_samples(x_local: torch.Tensor, total: int) -> torch.Tensor:
    """All-gather the per-rank samples ``[count_r, N, V]`` into ``[total, N, V]`` in rank order.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    if not dist.is_initialized():
        if x_local.shape[0] != total:
            raise ValueError("single process: local shard must be the whole batch")
        return x_local
    world = dist.get_world_size()   # a one-rank group still runs the collective (RCCL on a GPU tensor)
    counts: List[int] = [shard_batch(total, r, world)[1] for r in range(world)]
    if x_local.shape[0] != counts[dist.get_rank()]:
        raise ValueError(f"rank {dist.get_rank()} holds {x_local.shape[0]} graphs, expected {counts[dist.get_rank()]}")
    cmax = max(counts)
    pad = torch.zeros((cmax,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]] = x_local
    out = torch.empty((world * cmax,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    out = out.view((world, cmax) + tuple(x_local.shape[1:]))
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)


class LangevinExchange:
    """The per-corrector-update exchange of the exact-global-batch Langevin mode (``include/dnag.h``,
    ``dnag_pc_args.langevin_exchange``).  ``local`` receives this shard's per-block sums of ``||grad_b||`` and
    ``||noise_b||`` (``[2][blocks]``, written by the exit kernel); ``__call__`` all-gathers every rank's ``local``
    into ``global_`` ``[world][2][blocks]`` in rank order, on the caller's stream.  With the ``nccl`` backend this is
    one RCCL ``all_gather`` of 8 KB per rank (8192 graphs) enqueued behind the exit kernel; other backends (``gloo``:
    tests) stage through the host."""

    def __init__(self, n_floats: int, device: torch.device):
        if not dist.is_initialized():
            raise RuntimeError("langevin_batch='global' needs an initialised torch.distributed process group")
        self.world = dist.get_world_size()
        self.local = torch.zeros(n_floats, dtype=torch.float32, device=device)
        self.global_ = torch.zeros(self.world * n_floats, dtype=torch.float32, device=device)
        self.calls = 0
        self.error: BaseException | None = None

    def __call__(self) -> None:
        self.calls += 1
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self.global_, self.local)      # ordered after the exit kernel on the current stream
            return
        host = self.local.cpu()                                           # synchronises with the stream
        out = torch.empty(self.world * host.numel(), dtype=torch.float32)
        dist.all_gather_into_tensor(out, host)
        self.global_.copy_(out)


def gather_partials(local: torch.Tensor) -> torch.Tensor:
    """What ``LangevinExchange`` does, for tensors on any device: every rank's ``local`` in rank order."""
    out = torch.empty(dist.get_world_size() * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out
