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


def upscale_sharded(model, x: Tensor, dst: int = 0, group=None, gather: bool = True, overlap_chunk: int = 0) -> Optional[Tensor]:
    """`x` is the FULL batch (same on every rank, or at least this rank's slice must be valid): each rank
    upscales its contiguous slice; with `gather` the outputs are collected on `dst`.
    `overlap_chunk` > 0: see `upscale_local_overlapped` (equal slices only; otherwise the plain path is taken)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(x.shape[0], world, rank)
    if gather and overlap_chunk > 0 and len(set(shard_sizes(x.shape[0], world))) == 1:
        return upscale_local_overlapped(model, x[lo:hi], dst=dst, group=group, chunk=overlap_chunk)
    local = model.upscale(x[lo:hi])
    if not gather:
        return local
    return gather_outputs(local, x.shape[0], dst=dst, group=group)


def upscale_local_overlapped(model, x_local: Tensor, dst: int = 0, group=None, chunk: int = 4) -> Optional[Tensor]:
    """Upscale + gather with the exchange hidden under the compute: this rank's slice (EVERY rank must hold the same
    number of images) is upscaled `chunk` images at a time, and each chunk's outputs start travelling to `dst` as an
    asynchronous gather while the next chunk is being computed -- only the last chunk's transfer is exposed.  The
    receive buffers are views of the final [world * n, ...] tensor, so nothing is copied twice.  With RCCL the
    collectives run on the backend's own stream, ordered after the producing kernels; `wait()` orders the caller's
    stream after them without blocking the host."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = x_local.shape[0]
    chunk = max(1, int(chunk))
    via_host = x_local.is_cuda and dist.get_backend(group) == "gloo"  # rehearsals: gloo moves host memory only
    full = None
    works, keep = [], []
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        y = model.upscale(x_local[c0:c1]).contiguous()
        wire = y.cpu() if via_host else y
        bufs = None
        if rank == dst:
            if full is None:
                full = torch.empty((world * n,) + tuple(wire.shape[1:]), dtype=wire.dtype, device=wire.device)
            bufs = [full[i * n + c0 : i * n + c1] for i in range(world)]
        works.append(dist.gather(wire, gather_list=bufs, dst=dst, group=group, async_op=True))
        keep.append(wire)  # the send buffer must outlive the transfer
    for w in works:
        w.wait()
    keep.clear()
    if full is None:
        return None
    return full.to(x_local.device) if via_host else full
