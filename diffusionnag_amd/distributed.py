"""Batch sharding of the sampler over the GPUs of one node (one process per GPU).

The reference is single-device.  Graphs are independent in the score
network, the surrogate and the predictor update, so the sample batch is cut into contiguous shards,
one per rank, and every rank runs the full reverse SDE on its shard with replicated weights.  The
only collective is the final gather of the ``[B/W, N, V]`` samples (RCCL over xGMI with the ``nccl``
backend; ``gloo`` on CPU for tests).  Each shard is its own Langevin batch -- the step size
(reference ``sampling.py:249-252``) is a mean over the shard, exactly what running the reference
once per device would do.  Philox noise is keyed by the global element index, so the noise a graph
sees does not depend on the number of ranks.
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
