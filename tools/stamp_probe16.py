#!/usr/bin/env python3
"""Diagnostic: run one 3x3 conv layer on the 16x16x32 kernel with the -DMZ_STAMP=2 build and print where the waves of
one workgroup spend each 32-channel chunk (cycles).
MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=.../libmewzoom_hip_stamp.so python tools/stamp_probe16.py B H W cin cout"""
import ctypes, sys, os
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from gpu_util import alloc_act, op_conv
from ultrazoom_amd import _ffi
B, H, W, cin, cout = 3, 270, 480, 768, 384
if len(sys.argv) > 5: B, H, W, cin, cout = map(int, sys.argv[1:6])
dt = torch.bfloat16
x = torch.randn(B, cin // 8, H, W, 8, device="cuda").to(dt)
w = torch.randn(cout, cin, 3, 3) * 0.02
out = alloc_act(B, cout, H, W, dt)
for _ in range(int(os.environ.get("REPS", "3"))):
    op_conv(dt, 0, x, None, w, 0.0, out, B, H, W, cin, cout, silu=1)
buf = (ctypes.c_ulonglong * (16 * 64 * 8))()
assert _ffi.lib().mz_debug_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 64, 8).astype(np.int64)
nch = cin // 32
t0 = a[0, 0, 0]
print("chunk | compute wave: arrive  bar1-wait  front  bar2-wait  back   | loaders (half 2c, 2c+1): wait-vmcnt bar issue")
rows = []
for c in range(min(40, 2 * nch)):
    row = []
    for wv in (0, 1, 4, 7):
        s = a[wv, c]
        row.append(f"w{wv}: @{s[0]-t0:7d} b{s[1]-s[0]:5d} F{s[2]-s[1]:5d} b{s[3]-s[2]:5d} K{s[4]-s[3]:5d}")
        rows.append((s[1]-s[0], s[2]-s[1], s[3]-s[2], s[4]-s[3]))
    for wv in (8, 9):
        for hlf in (0, 1):
            s = a[wv, 2 * c + hlf] if 2 * c + hlf < 64 else None
            if s is not None: row.append(f"L{wv}.{hlf}: vm{s[1]-s[0]:5d} b{s[2]-s[1]:5d} i{s[3]-s[2]:4d}")
    print(f"{c:3d} | " + " | ".join(row))
r = np.array(rows)
print("median cycles: barrier-1 wait %d, front %d, barrier-2 wait %d, back %d  (ideal MFMA per half for this wave: %d / %d)" % (
    np.median(r[:, 0]), np.median(r[:, 1]), np.median(r[:, 2]), np.median(r[:, 3]), 14 * 8 * 16, 13 * 8 * 16))
