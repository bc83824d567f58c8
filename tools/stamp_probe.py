#!/usr/bin/env python3
"""Diagnostic: run one 3x3 conv layer with the -DMZ_STAMP library build and print where one workgroup's waves
spend each K-stage (cycles).  MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=.../libmewzoom_hip_stamp.so python tools/stamp_probe.py"""
import ctypes, sys, os
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from gpu_util import alloc_act, op_conv
from ultrazoom_amd import _ffi
B, H, W, cin, cout = 3, 540, 960, 384, 192
if len(sys.argv) > 5: B, H, W, cin, cout = map(int, sys.argv[1:6])
dt = torch.bfloat16
x = torch.randn(B, cin // 8, H, W, 8, device="cuda").to(dt)
w = torch.randn(cout, cin, 3, 3) * 0.02
out = alloc_act(B, cout, H, W, dt)
for _ in range(int(os.environ.get("REPS", "3"))):
    op_conv(dt, 0, x, None, w, 0.0, out, B, H, W, cin, cout)
buf = (ctypes.c_ulonglong * (16 * 64 * 8))()
rc = _ffi.lib().mz_debug_read(buf)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 64, 8).astype(np.int64)
clk = a[15]
vals = []
for i in range(60):
    t0_, r0_, t1_, r1_ = clk[i, :4]
    if r1_ > r0_ > 0:
        vals.append((t1_ - t0_) / (r1_ - r0_) * 100.0)
if vals:
    print(f"in-kernel clock over the K loop (MHz): median {np.median(vals):.0f} min {min(vals):.0f} max {max(vals):.0f} n={len(vals)}; K-loop cycles median {np.median([c[2]-c[0] for c in clk[:60] if c[3]>c[1]>0]):.0f}")
ph = [(c[0] - c[5], c[2] - c[0], c[4] - c[2]) for c in clk[:60] if c[3] > c[1] > 0 and c[5] > 0 and c[4] > 0]
if ph:
    arr = np.array(ph)
    print(f"per-workgroup phases (cycles, median): entry->K-loop {np.median(arr[:,0]):.0f} | K-loop {np.median(arr[:,1]):.0f} | epilogue {np.median(arr[:,2]):.0f}")
if os.environ.get("CLOCK_ONLY"): sys.exit(0)
nst = min(cin // 16, 64)
t0 = a[0, 0, 0]
print("stage | wave: arrive(wait-start) waited-vmcnt barrier-wait issue compute   [cycles]")
for st in range(min(nst, 24)):
    row = []
    for wv in (0, 3, 4, 7, 8):
        s = a[wv, st]
        if wv == 8: row.append(f"L8: @{s[0]-t0:7d} vm{s[1]-s[0]:5d} bar{s[2]-s[1]:5d} iss{s[3]-s[2]:5d}")
        else: row.append(f"w{wv}: @{s[0]-t0:7d} vm{s[1]-s[0]:4d} bar{s[2]-s[1]:5d} iss{s[3]-s[2]:4d} cmp{s[4]-s[3]:5d}")
    print(f"{st:3d} | " + " | ".join(row))
