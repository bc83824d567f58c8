#!/usr/bin/env python3
"""Diagnostic: where the waves of conv3t_kernel spend their cycles (tools/build_variant.sh diag -DMZ_DIAG).
MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=$PWD/ultrazoom_amd/libmewzoom_hip_diag.so python tools/stamp_probe_t.py"""
import ctypes, sys, os
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from gpu_util import alloc_act, op_conv
from ultrazoom_amd import _ffi
dt = torch.bfloat16
CASES = [(8, 540, 960, 96, 48, 2), (8, 540, 960, 96, 48, 1), (8, 540, 960, 192, 48, 2)]
if os.environ.get("STAMP_CASES"): CASES = eval(os.environ["STAMP_CASES"])
NAMES = {0: "plain step, chunk's first", 1: "plain step, later", 2: "epilogue step, chunk's first", 3: "epilogue step, later"}
for (B, H, W, cin, cout, silu) in CASES:
    x = torch.randn(B, cin // 8, H, W, 8, device="cuda").to(dt)
    w = torch.randn(cout, cin, 3, 3) * 0.02
    out = alloc_act(B, cout, H, W, dt)
    if silu == 2:  # fused conv2 + AdaptiveResidualMix
        xin = torch.randn(B, cout // 8, H, W, 8, device="cuda").to(dt)
        wm = (torch.randn(cout, 2 * cout, 1, 1) * 0.1).to("cuda", torch.float32).contiguous()
        wd = w.to("cuda", torch.float32).contiguous()
    for _ in range(int(os.environ.get("REPS", "3"))):
        if silu == 2:
            _ffi.check(_ffi.lib().mz_op_conv_mix(_ffi.dtype_code(dt), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(wd.data_ptr()),
                                                 ctypes.c_void_p(wm.data_ptr()), ctypes.c_float(0.3), ctypes.c_void_p(out.data_ptr()), B, H, W, cin, cout,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
        else:
            op_conv(dt, 0, x, None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
    buf = (ctypes.c_ulonglong * (16 * 64 * 8))()
    assert _ffi.lib().mz_debug_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    print(f"== {H}x{W} {cin}->{cout} {'fused mix' if silu == 2 else ('SiLU' if silu else 'plain')}: ideal MFMA cycles per tile {cin // 32 * 324 * 16}")
    for wv in (0, 4):
        r = a[wv * 32: wv * 32 + 32]
        n = max(r[1], 1)
        clock = f", in-kernel clock {r[28] / r[29] * 100.0:.0f} MHz over {r[28]} cycles" if r[29] > 0 else ""
        print(f"  wave {wv}: tiles {r[1]}, K loop (incl. its barrier waits) {r[0] // n} cycles per tile, final epilogue {r[2]}, phase start per tile {r[3] // n}{clock}")
        for c in (0, 1, 2, 3):
            k = max(r[24 + c], 1)
            print(f"      {NAMES[c]:30s} x{r[24 + c]:5d}: request + DMA issue {r[4 * c + 4] // k:5d}  epilogue {r[4 * c + 5] // k:5d}  vmcnt wait {r[4 * c + 6] // k:5d}  barrier {r[4 * c + 7] // k:5d}")
