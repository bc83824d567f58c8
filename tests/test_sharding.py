"""Batch sharding + gather of outputs, world_size 2 over gloo on CPU."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ultrazoom_amd.sharding import gather_outputs, shard_range, shard_sizes


def test_shard_ranges_cover_the_batch():
    for batch in (0, 1, 2, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = shard_sizes(batch, world)
            assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1
    assert shard_sizes(128, 8) == [16] * 8  # BASELINE config 4: 16 images per GPU


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeModel:
    """Stands in for MewZoom on CPU: a per-image function, so any mix-up of slices is visible."""

    def upscale(self, x):
        return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) * 0.5


def _worker(rank, world, port, batch, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ultrazoom_amd.sharding import upscale_sharded

        g = torch.Generator().manual_seed(0)
        x = torch.rand(batch, 3, 5, 7, generator=g)
        out = upscale_sharded(_FakeModel(), x, dst=0)
        if rank == 0:
            want = _FakeModel().upscale(x)
            torch.save({"ok": bool(torch.equal(out, want)), "shape": tuple(out.shape)}, result_path)
        else:
            assert out is None
        # the overlapped variant (chunked asynchronous gathers into views of the final tensor) gives the same tensor
        out2 = upscale_sharded(_FakeModel(), x, dst=0, overlap_chunk=1)
        if rank == 0:
            assert torch.equal(out2, _FakeModel().upscale(x))
        else:
            assert out2 is None
        lo, hi = shard_range(batch, world, rank)
        full = gather_outputs(torch.full((hi - lo, 2), float(rank)), batch, dst=1)
        if rank == 1:
            sizes = shard_sizes(batch, world)
            want = torch.cat([torch.full((n, 2), float(r)) for r, n in enumerate(sizes)])
            assert torch.equal(full, want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5])
def test_upscale_sharded_world2_gloo(tmp_path, batch):
    path = tmp_path / "result.pt"
    mp.spawn(_worker, args=(2, _free_port(), batch, str(path)), nprocs=2, join=True)
    res = torch.load(path)
    assert res["ok"] and res["shape"] == (batch, 3, 10, 14)


def _worker_overlap(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ultrazoom_amd.sharding import upscale_local_overlapped

        per = 5  # chunks of 2, 2, 1
        x_local = torch.rand(per, 3, 4, 6, generator=torch.Generator().manual_seed(10 + rank))
        out = upscale_local_overlapped(_FakeModel(), x_local, dst=1, chunk=2)
        if rank == 1:
            want = torch.cat([_FakeModel().upscale(torch.rand(per, 3, 4, 6, generator=torch.Generator().manual_seed(10 + r)))
                              for r in range(world)])
            torch.save({"ok": bool(torch.equal(out, want)), "shape": tuple(out.shape)}, result_path)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_world2_gloo(tmp_path):
    path = tmp_path / "result.pt"
    mp.spawn(_worker_overlap, args=(2, _free_port(), str(path)), nprocs=2, join=True)
    res = torch.load(path)
    assert res["ok"] and res["shape"] == (10, 3, 8, 12)
