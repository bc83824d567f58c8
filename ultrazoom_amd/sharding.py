"""Batch sharding across the GPUs of one node.

Images of a batch are independent (no cross-sample op anywhere in the path), so the batch is cut into
contiguous per-rank slices, every rank upscales its slice with its own replica of the weights, and the
only communication is ONE gather of the output images to a destination rank (RCCL over xGMI when the
process group's backend is "nccl"; "gloo" works for CPU tensors in the tests).
"""

from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_range(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [start, stop) of a batch for `rank`; earlier ranks take the remainder."""
    assert batch >= 0 and world_size > 0 and 0 <= rank < world_size
    base, extra = divmod(batch, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(batch: int, world_size: int) -> List[int]:
    return [shard_range(batch, world_size, r)[1] - shard_range(batch, world_size, r)[0] for r in range(world_size)]


def gather_outputs(local: Tensor, batch: int, dst: int = 0, group=None) -> Optional[Tensor]:
    """Gathers every rank's output slice on `dst` and returns the full [batch, ...] tensor there (None elsewhere).

    Slices may differ by one image; shorter ones are padded to the longest for the collective and trimmed
    afterwards, so a single `gather` is issued whatever the remainder.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(batch, world)
    assert local.shape[0] == sizes[rank], f"rank {rank}: local batch {local.shape[0]} != expected {sizes[rank]}"
    longest = max(sizes)
    send = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    send = send.contiguous()
    # gloo cannot gather device tensors: stage through host memory (rehearsals only; RCCL gathers in HBM)
    via_host = send.is_cuda and dist.get_backend(group) == "gloo"
    wire = send.cpu() if via_host else send
    if rank == dst:
        if len(set(sizes)) == 1:
            # equal slices: receive straight into the final tensor, no second copy
            full = torch.empty((batch,) + tuple(wire.shape[1:]), dtype=wire.dtype, device=wire.device)
            bufs = [full[i * longest : (i + 1) * longest] for i in range(world)]
            dist.gather(wire, gather_list=bufs, dst=dst, group=group)
        else:
            bufs = [torch.empty_like(wire) for _ in range(world)]
            dist.gather(wire, gather_list=bufs, dst=dst, group=group)
            full = torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
        return full.to(local.device) if via_host else full
    dist.gather(wire, gather_list=None, dst=dst, group=group)
    return None


def upscale_sharded(model, x: Tensor, dst: int = 0, group=None, gather: bool = True) -> Optional[Tensor]:
    """`x` is the FULL batch (same on every rank, or at least this rank's slice must be valid): each rank
    upscales its contiguous slice; with `gather` the outputs are collected on `dst`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(x.shape[0], world, rank)
    local = model.upscale(x[lo:hi])
    if not gather:
        return local
    return gather_outputs(local, x.shape[0], dst=dst, group=group)
