#!/usr/bin/env python3
"""Diagnostic: where a compute wave of conv3q_kernel spends its cycles (K loop incl. barriers / accumulator zeroing / epilogue).
MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=$PWD/ultrazoom_amd/libmewzoom_hip_qstamp.so python tools/stamp_probe_q.py   (build: tools/build_variant.sh qstamp -DQ_STAMP
after adding the Q_STAMP hooks of the round-2 log; the hooks are not part of the shipped source)"""
import ctypes, sys, os
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from gpu_util import alloc_act, op_conv
from ultrazoom_amd import _ffi
dt = torch.bfloat16
for (B, H, W, cin, cout, silu) in [(3, 540, 960, 384, 192, 0), (3, 540, 960, 192, 384, 1), (3, 135, 240, 1536, 768, 0)]:
    x = torch.randn(B, cin // 8, H, W, 8, device="cuda").to(dt)
    w = torch.randn(cout, cin, 3, 3) * 0.02
    out = alloc_act(B, cout, H, W, dt)
    for _ in range(int(os.environ.get("REPS", "3"))):
        op_conv(dt, 0, x, None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
    buf = (ctypes.c_ulonglong * (16 * 64 * 8))()
    assert _ffi.lib().mz_debug_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    for wv in range(4):
        k, e, z, n, cyc, rt = a[wv * 8: wv * 8 + 6]
        n = max(n, 1)
        if wv == 0 and rt > 0:
            print(f"   in-kernel clock of the last launch: {cyc / rt * 100.0:.0f} MHz (s_memtime / s_memrealtime x 100 MHz)")
        print(f"{H}x{W} {cin}->{cout} silu={silu} wave {wv}: tiles {n}, per tile: K loop {k // n} cycles, zeroing {z // n}, epilogue {e // n}  (epilogue share {100.0 * e / (k + e + z):.1f} %, ideal MFMA cycles per tile {cin // 32 * 324 * 16})")
