#!/usr/bin/env python3
"""Headline benchmark: upscaled megapixels per second of MewZoom.upscale() on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3], the configuration the metric is quoted on): the 4X model with
96/192/384/768 channels and 8/8/8/16 layers (434 M parameters), bf16, 16 images of
1080x1920 per GPU -> 4320x7680 each.  (BASELINE's "1080p->4K" is not a 4X geometry; `--workload cfg3_540p`
runs the other reading, 540x960 -> 4K.)  Weights are hash-initialised and inputs synthetic: nothing else
ships with the reference.

A step = one upscale() of the rank's 16 images plus, when N > 1, the gather of all output images on rank 0
(the only collective of the path).  Inputs and weights are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline     : live HIP-event timing of the dominant kernel family (the 3x3 implicit-GEMM convolution):
                 algorithmic FLOPs of all its launches in one step / their summed device time, against the
                 dense bf16 MFMA peak of 2.5 PFLOP/s (`frac`) AND against what this very device sustains on a bare,
                 register-resident MFMA loop with random operands, measured in the same run by tools/microbench/mb_mfma
                 (`measured_peak`, `frac_of_measured_peak`).  `traffic` comes from committed rocprofv3 PMC passes and is
                 only quoted when they were taken with the kernel sources that are running (sha256 match).
  cpu_baseline : the CPU oracle (oracle/mewzoom_oracle.py, torch CPU fp32 on all host cores): `value` is timed on a
                 bounded sample of the SAME model as the GPU workload (with the GPU-vs-oracle PSNR / max-abs on that
                 sample), `cfg1` is BASELINE.md section 3's procedure (2X-48 model, 1x3x256x256, median of >= 5).
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from ultrazoom_amd import MewZoom  # noqa: E402
from ultrazoom_amd.sharding import upscale_local_overlapped  # noqa: E402
from ultrazoom_amd.synth import synth_image, synth_state_dict  # noqa: E402

MODELS = {
    "4x96": dict(upscale_ratio=4, primary_channels=96, primary_layers=8, secondary_channels=192, secondary_layers=8,
                 tertiary_channels=384, tertiary_layers=8, quaternary_channels=768, quaternary_layers=16,
                 hidden_ratio=2, num_deg_features=3),
    "2x48": dict(upscale_ratio=2, primary_channels=48, primary_layers=4, secondary_channels=96, secondary_layers=4,
                 tertiary_channels=192, tertiary_layers=4, quaternary_channels=384, quaternary_layers=8,
                 hidden_ratio=2, num_deg_features=3),
}
WORKLOADS = {
    # name: (model, images per GPU, H, W, description; {dtype} = the --dtype of the run)
    "cfg3_1080p": ("4x96", 16, 1080, 1920, "MewZoom-4X 96ch/40L {dtype}, 16 x 1080x1920 -> 4320x7680 per GPU"),
    "cfg3_540p": ("4x96", 16, 540, 960, "MewZoom-4X 96ch/40L {dtype}, 16 x 540x960 -> 2160x3840 (4K) per GPU"),
    "cfg2": ("2x48", 32, 540, 960, "MewZoom-2X 48ch/20L {dtype}, 32 x 540x960 -> 1080x1920 per GPU"),
}
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # dense MFMA, MI355X_MICROARCH.md


def parameter_shapes(cfg):
    m = MewZoom(**cfg)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask and cgroup quota, not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if "MZ_CPU_THREADS" in os.environ:
        n = int(os.environ["MZ_CPU_THREADS"])
    return max(1, n)


TRAFFIC_PROFILE = REPO / "profiles" / "r04_pmc_traffic_conv3x3.json"


def kernel_source_sha256() -> str:
    """Identity of the kernel build: sha256 over the sources of libmewzoom_hip.so (stable across rebuilds of the same tree,
    unlike the binary).  A PMC profile is only quoted next to a run of the sources it was taken with."""
    h = hashlib.sha256()
    for f in sorted((REPO / "ultrazoom_amd" / "csrc").glob("mz_*")):
        if f.suffix in (".hip", ".cpp", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


def traffic_from_profile(args):
    """HBM-side bytes per 3x3-kernel launch from the committed rocprofv3 PMC passes (tools/pmc_collect.sh +
    tools/pmc_traffic.py: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes, the same bench.py
    command on one 3-image micro-batch).  PMC counters cannot be collected from inside this process, so the figure is
    quoted only for the profiled workload AND only when the profile was taken with the library that is running now
    (sha256 recorded by pmc_traffic.py); otherwise null plus the reason."""
    if args.workload != "cfg3_1080p" or args.dtype != "bf16":
        return None, "no PMC profile for this workload"
    if not TRAFFIC_PROFILE.exists():
        return None, f"{TRAFFIC_PROFILE.name} not found"
    try:
        prof = json.loads(TRAFFIC_PROFILE.read_text())
        if prof.get("kernel_source_sha256") != kernel_source_sha256():
            return None, "stale PMC profile: it was taken with different kernel sources (ultrazoom_amd/csrc)"
        return prof, None
    except (OSError, KeyError, ValueError) as e:
        return None, f"unreadable PMC profile: {e}"


def measured_mfma_peak():
    """Runs tools/microbench/mb_mfma (bare bf16 MFMA loops, random operands, >= 2 s each) as a CHILD process and returns its
    JSON.  Must be called before this process touches the GPU (a process that has initialised HIP must not exec) and
    never under rocprofv3 (its preloaded library initialises HIP at start-up)."""
    exe = REPO / "tools" / "microbench" / "mb_mfma"
    if not exe.exists() or "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    try:
        out = subprocess.run([str(exe), "2.0"], capture_output=True, text=True, timeout=120, check=True).stdout
        return json.loads(out.strip().splitlines()[-1])
    except (OSError, subprocess.SubprocessError, ValueError, IndexError):
        return None


def cpu_baseline(cfg, sd, model, dtype, sample_hw):
    """Times the CPU oracle on a bounded sample and checks the GPU result against it."""
    from oracle import mewzoom_oracle as oracle  # checker / baseline only

    h, w = sample_hw
    x = synth_image(1, h, w, seed=99)
    cores = host_cores()
    torch.set_num_threads(cores)
    with torch.inference_mode():
        oracle.upscale(cfg, sd, x[:, :, : h // 2, : w // 2])  # warm-up (thread pool, allocator)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            want = oracle.upscale(cfg, sd, x)
            times.append(time.perf_counter() - t0)
    t = statistics.median(times)
    r = cfg["upscale_ratio"]
    mpix = h * r * w * r / 1e6
    got = model.upscale(x.to("cuda", dtype)).float().cpu()
    mse = (got.double() - want.double()).pow(2).mean().item()
    import math

    return {
        "value": mpix / t,
        "unit": "MPix/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"oracle upscale() of the SAME model as the GPU workload on 1x3x{h}x{w} fp32, 1 warm-up + median of 3, "
                  f"{t:.2f} s/iter, {oracle.flops_per_image(cfg, h, w) / t / 1e9:.0f} GFLOP/s "
                  "(BASELINE.md section 3's own shape is reported under 'cfg1')",
        "gpu_vs_oracle_psnr_db": 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf"),
        "gpu_vs_oracle_max_abs": (got - want).abs().max().item(),
        "cfg1": cpu_baseline_cfg1(cores),
    }, want


def cpu_baseline_cfg1(cores):
    """BASELINE.md section 3 as written: the 2X model (48/96/192/384 channels, 4/4/4/8 layers), x = 1x3x256x256, fp32,
    all host cores, >= 2 warm-ups, >= 5 iterations, median."""
    from oracle import mewzoom_oracle as oracle  # checker / baseline only

    cfg = MODELS["2x48"]
    sd = synth_state_dict(oracle.parameter_shapes(cfg), seed=1234)
    x = synth_image(1, 256, 256, seed=98)
    torch.set_num_threads(cores)
    with torch.inference_mode():
        for _ in range(2):
            oracle.upscale(cfg, sd, x)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            oracle.upscale(cfg, sd, x)
            times.append(time.perf_counter() - t0)
    t = statistics.median(times)
    return {
        "value": 512 * 512 / 1e6 / t,
        "unit": "MPix/s",
        "cores": cores,
        "sample": f"BASELINE configs[0]: MewZoom-2X 48ch/20L, 1x3x256x256 fp32, 2 warm-ups + median of 5, {t:.3f} s/iter, "
                  f"{oracle.flops_per_image(cfg, 256, 256) / t / 1e9:.0f} GFLOP/s",
    }


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process (one rank per GPU, rendezvous
    on 127.0.0.1), relay what they print -- rank 0's JSON line -- and return their exit code.  The parent never touches the
    GPU (a process that has initialised HIP must not exec, and need not: it only waits)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def timed_steps(step, sync, steps, warmup):
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return time.perf_counter() - t0


def secondary_readings(device, want_sample, sample_hw):
    """Short same-process runs of the other BASELINE readings, so that they are driver-timed too (rank 0, N = 1):
    cfg2 (BASELINE configs[1]) in bf16, the headline workload in fp16 (north_star's Target sentence), and the fp32
    verification mode against the CPU oracle on the cpu_baseline sample (north_star: <= 1e-3 max-abs)."""
    out = {}
    sds = {}  # hash-initialised weights per model (434 M parameters take seconds to generate: once)

    def run(workload, dtype_name, steps, warmup, u8=False):
        model_name, per_gpu, H, W, desc = WORKLOADS[workload]
        cfg = MODELS[model_name]
        dtype = DTYPES[dtype_name]
        if model_name not in sds:
            sds[model_name] = synth_state_dict(parameter_shapes(cfg), seed=1234)
        m = MewZoom(**cfg)
        m.load_state_dict(sds[model_name])
        m = m.to(device, dtype).eval()
        x = synth_image(per_gpu, H, W, seed=1000).to(device, dtype)
        if u8:  # what every caller of the reference does around upscale() (README.md:72-83), fused into the two ends of the path
            x = (x.float() * 255.0 + 0.5).clamp(0, 255).to(torch.uint8)
            fwd = m.upscale_uint8
            desc = desc + ", uint8 images in and out (upscale_uint8)"
        else:
            fwd = m.upscale
        el = timed_steps(lambda: fwd(x), lambda: torch.cuda.synchronize(device), steps, warmup)
        r = cfg["upscale_ratio"]
        h = m._engine.handle
        h.profile_enable(True)
        fwd(x)
        torch.cuda.synchronize(device)
        prof = h.profile_read()
        h.profile_enable(False)
        conv_tflops = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12 if prof["conv_ms"] > 0 else 0.0
        res = {"workload": desc.format(dtype=dtype_name), "value": per_gpu * H * r * W * r / 1e6 * steps / el, "unit": "MPix/s",
               "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3, "conv3x3_tflops": conv_tflops,
               "conv3x3_frac_of_peak": conv_tflops / PEAK_TFLOPS[dtype_name]}
        del m, x
        torch.cuda.empty_cache()
        return res

    out["cfg2"] = run("cfg2", "bf16", 5, 2)
    out["f16"] = run("cfg3_1080p", "f16", 5, 2)
    out["cfg3_540p"] = run("cfg3_540p", "bf16", 5, 2)   # BASELINE's "1080p->4K" read as "-> true 4K" 
    out["u8"] = run("cfg3_1080p", "bf16", 5, 2, u8=True)  # uint8 images at both ends of the headline workload
    if want_sample is not None:
        cfg = MODELS["4x96"]
        m = MewZoom(**cfg)
        m.load_state_dict(sds["4x96"])
        m = m.to(device, torch.float32).eval()
        h, w = sample_hw
        got = m.upscale(synth_image(1, h, w, seed=99).to(device, torch.float32)).float().cpu()
        out["f32_max_abs"] = (got - want_sample).abs().max().item()
        out["f32_sample"] = f"fp32 mode of the headline model vs the CPU oracle on 1x3x{h}x{w} (north_star tolerance: 1e-3 max-abs)"
        del m
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3_1080p", choices=list(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=list(DTYPES))
    ap.add_argument("--images-in-flight", type=int, default=0, help="micro-batch inside the library (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true", help="skip the on-box MFMA peak micro-benchmark (profiling runs)")
    ap.add_argument("--dump-launches", default="", help="write a per-launch CSV (shape, ms, TFLOP/s) of the profiled step")
    ap.add_argument("--no-gather", action="store_true", help="skip the output gather in the timed step (N > 1)")
    ap.add_argument("--images-per-gpu", type=int, default=0, help="override the workload's batch (profiling runs only)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short cfg2 / fp16 / fp32 side readings (N = 1)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous, barrier and JSON line only: no GPU work (launcher tests)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    # the on-box MFMA peak: a child process, started BEFORE this process initialises HIP (rank 0, single-GPU runs; never in a dry run,
    # which promises no GPU work at all)
    mfma_peak = measured_mfma_peak() if (rank == 0 and world == 1 and not args.no_microbench and not args.dry_run) else None
    # one rank per GPU; MZ_BENCH_BACKEND=gloo lets several ranks share one GPU for plumbing rehearsals on a 1-GPU box
    backend = os.environ.get("MZ_BENCH_BACKEND", "nccl")
    if args.dry_run:  # the launcher's plumbing without a GPU: rendezvous over gloo, one all-reduce, rank 0's line
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert int(t.item()) == world - 1
            dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "dry run", "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "dry_run": True}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    model_name, per_gpu, H, W, desc = WORKLOADS[args.workload]
    if args.images_per_gpu > 0:
        per_gpu = args.images_per_gpu
        desc += f" [batch overridden to {per_gpu}: profiling run, not a benchmark result]"
    desc = desc.format(dtype=args.dtype)
    cfg = MODELS[model_name]
    dtype = DTYPES[args.dtype]
    r = cfg["upscale_ratio"]

    sd = synth_state_dict(parameter_shapes(cfg), seed=1234)
    model = MewZoom(**cfg)
    model.load_state_dict(sd)
    model = model.to(device, dtype).eval()
    model.max_images_in_flight = args.images_in_flight
    # every rank gets different images (seeded by rank); same shape everywhere = weak scaling
    x = synth_image(per_gpu, H, W, seed=1000 + rank).to(device, dtype)
    global_batch = per_gpu * world
    do_gather = world > 1 and not args.no_gather

    # N > 1: the gather of a chunk's outputs travels while the next chunk is computed (only the last chunk's transfer
    # is exposed); the chunk equals the library's micro-batch so that no image runs alone
    chunk = args.images_in_flight if args.images_in_flight > 0 else 4
    if do_gather:
        model.max_images_in_flight = chunk

    def step():
        if do_gather:
            return upscale_local_overlapped(model, x, dst=0, chunk=chunk)
        return model.upscale(x)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def max_over_ranks(sec):
        if world > 1:
            t = torch.tensor([sec], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.item()
        return sec

    elapsed = max_over_ranks(timed_steps(step, sync, args.steps, args.warmup))
    # N > 1: the compute-only rate as well (the same steps without the gather), reported next to the headline value
    elapsed_no_gather = None
    if do_gather:
        elapsed_no_gather = max_over_ranks(timed_steps(lambda: model.upscale(x), sync, args.steps, 1))

    out_mpix_per_step = global_batch * (H * r) * (W * r) / 1e6
    value = out_mpix_per_step * args.steps / elapsed

    result = None
    if rank == 0:
        engine = model._engine
        flops_step = engine.handle.flops_per_image(H, W) * per_gpu  # this rank
        # ---- roofline leg: one extra, separately profiled step ----
        engine.handle.profile_enable(True)
        model.upscale(x)
        torch.cuda.synchronize(device)
        if args.dump_launches:
            engine.handle.profile_dump(args.dump_launches)
        prof = engine.handle.profile_read()
        engine.handle.profile_enable(False)
        conv_tflops = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12 if prof["conv_ms"] > 0 else 0.0
        peak = PEAK_TFLOPS[args.dtype]
        pmc, pmc_note = traffic_from_profile(args)
        alg_bytes_per_launch = prof["conv_bytes"] / max(1.0, prof["conv_launches"])
        measured_peak = mfma_peak["measured_peak_tflops"] if (mfma_peak and args.dtype != "f32") else None
        result = {
            "metric": f"upscaled MPix/sec, MewZoom-4X 1080p {args.dtype} batched inference" if args.workload.startswith("cfg3")
            else "upscaled MPix/sec",
            "value": value,
            "unit": "MPix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (hash-initialised weights, synthetic images; nothing ships with the reference)",
            "config": {
                "workload": desc,
                "global_batch": global_batch,
                "input": [H, W],
                "output": [H * r, W * r],
                "parallelism": f"batch-sharded x{world}, outputs gathered on rank 0" if do_gather else f"batch-sharded x{world}",
                "images_in_flight": args.images_in_flight,
            },
            "whole_path_tflops_per_gpu": flops_step * args.steps / elapsed / 1e12,
            "value_without_gather": out_mpix_per_step * args.steps / elapsed_no_gather if elapsed_no_gather else None,
            "roofline": {
                "bound": "mfma",
                "kernel": "3x3 implicit-GEMM convolution kernels (all 3x3 launches of one step, rank 0)",
                "achieved": conv_tflops,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": conv_tflops / peak,
                "measured_peak": measured_peak,
                "frac_of_measured_peak": conv_tflops / measured_peak if measured_peak else None,
                "measured_peak_detail": mfma_peak,
                # HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 PMC passes) next to the algorithmic
                # bytes per launch (each conv reads its input once and writes its output once, weights once)
                "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                "traffic_note": pmc_note or f"{TRAFFIC_PROFILE.name}: {pmc['launches']} launches of one {pmc.get('images', 3)}-image "
                                            "micro-batch forward, same kernel sources (sha256 match)",
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "traffic_over_algorithmic": pmc["traffic_bytes_per_launch"] / alg_bytes_per_launch if pmc and alg_bytes_per_launch else None,
                "traffic_bytes_per_step": pmc["traffic_bytes_per_launch"] * prof["conv_launches"] if pmc else None,
                "launches": prof["conv_launches"],
                "avg_launch_ms": prof["conv_ms"] / max(1.0, prof["conv_launches"]),
                "algorithmic_tflop_per_step": prof["conv_flops"] / 1e12,
                "conv_ms_per_step": prof["conv_ms"],
                "other_kernels_ms_per_step": prof["other_ms"],
                "algorithmic_GBps": prof["conv_bytes"] / (prof["conv_ms"] * 1e-3) / 1e9 if prof["conv_ms"] > 0 else 0.0,
            },
        }
        want_sample = None
        sample = (384, 640) if model_name == "4x96" else (540, 960)  # ~10-20 s of CPU work in total
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"], want_sample = cpu_baseline(cfg, sd, model, dtype, sample)
        if world == 1 and not args.no_secondary and args.workload == "cfg3_1080p" and args.dtype == "bf16" and args.images_per_gpu == 0:
            del x
            torch.cuda.empty_cache()
            result["secondary"] = secondary_readings(device, want_sample, sample)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
