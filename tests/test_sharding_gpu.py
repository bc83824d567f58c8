"""The multi-GPU path of SURVEY.md section 8e made executable on ONE MI355X: the real MewZoom through `upscale_sharded` and
`upscale_local_overlapped` in two rank processes (gloo rendezvous, both ranks computing on cuda:0), compared bit for bit
with a single-process run; and an RCCL ("nccl") process group of world size 1 driving the same gather code on HBM tensors.
What this cannot show is the 1 -> 8 GPU scaling curve (xGMI transfers between distinct devices): that needs the driver's
multi-GPU node."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(upscale_ratio=4, primary_channels=32, primary_layers=2, secondary_channels=64, secondary_layers=2,
           tertiary_channels=128, tertiary_layers=2, quaternary_channels=256, quaternary_layers=4, hidden_ratio=2,
           num_deg_features=3)
BATCH, H, W = 6, 40, 56


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(dtype):
    from oracle import mewzoom_oracle as oracle  # parameter shapes only
    from ultrazoom_amd import MewZoom
    from ultrazoom_amd.synth import synth_state_dict

    m = MewZoom(**CFG)
    m.load_state_dict(synth_state_dict(oracle.parameter_shapes(CFG), seed=8))
    return m.to("cuda:0", dtype).eval()


def _rank_main(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ultrazoom_amd.sharding import shard_range, upscale_local_overlapped, upscale_sharded
        from ultrazoom_amd.synth import synth_image

        m = _model(torch.bfloat16)
        x = synth_image(BATCH, H, W, seed=10).to("cuda:0", torch.bfloat16)  # the full batch, identical on both ranks
        full = upscale_sharded(m, x, dst=0)                      # plain: one gather of the whole slice
        over = upscale_sharded(m, x, dst=0, overlap_chunk=2)     # chunked asynchronous gathers into views of the result
        lo, hi = shard_range(BATCH, world, rank)
        local = upscale_local_overlapped(m, x[lo:hi], dst=1, chunk=1)
        if rank == 0:
            assert local is None
            torch.save({"full": full.cpu(), "over": over.cpu()}, os.path.join(out_dir, "rank0.pt"))
        else:
            assert full is None and over is None
            torch.save({"local": local.cpu()}, os.path.join(out_dir, "rank1.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    from ultrazoom_amd.synth import synth_image

    # rank processes are forked from the clean fork server started in conftest.py (no exec from a GPU-initialised process)
    mp.start_processes(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="forkserver")
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    m = _model(torch.bfloat16)
    x = synth_image(BATCH, H, W, seed=10).to("cuda:0", torch.bfloat16)
    want = m.upscale(x).cpu()
    assert want.shape == (BATCH, 3, 4 * H, 4 * W)
    assert torch.equal(r0["full"], want), "sharded + gathered result differs from the single-process result"
    assert torch.equal(r0["over"], want), "overlapped chunked gather differs from the single-process result"
    assert torch.equal(r1["local"], want), "gather to a non-zero destination rank differs"


def test_rccl_world_size_one_gather_on_hbm():
    """RCCL init + the library's gather paths (plain and asynchronous chunked) on device tensors, world size 1."""
    from ultrazoom_amd.sharding import gather_outputs, upscale_local_overlapped, upscale_sharded
    from ultrazoom_amd.synth import synth_image

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        m = _model(torch.bfloat16)
        x = synth_image(4, H, W, seed=11).to("cuda:0", torch.bfloat16)
        want = m.upscale(x)
        got = upscale_sharded(m, x, dst=0)
        assert got.is_cuda and torch.equal(got, want)
        got2 = upscale_local_overlapped(m, x, dst=0, chunk=3)     # chunks of 3 + 1: two asynchronous RCCL gathers
        torch.cuda.synchronize()
        assert got2.is_cuda and torch.equal(got2, want)
        g = gather_outputs(want, 4, dst=0)
        assert g.data_ptr() != want.data_ptr() and torch.equal(g, want)
    finally:
        dist.destroy_process_group()
