"""Scratch: where do conv3t's fused outputs differ from conv3s's?  usage: python tools/debug/conv3t_probe.py B H W cin cout wgs"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from gpu_util import DTYPES, alloc_act, from_act, q, to_act
from test_conv3t_gpu import op_conv_mix, rnd, wrnd

B, H, W, cin, cout, wgs = (int(v) for v in sys.argv[1:7])
dtype = DTYPES["bf16"]
hid = q(rnd((B, cin, H, W), 61), dtype); x = q(rnd((B, cout, H, W), 62), dtype)
w2 = q(wrnd((cout, cin, 3, 3), 63), dtype); wmix = q(rnd((cout, 2 * cout, 1, 1), 64, (3.0 / (2 * cout)) ** 0.5 * 1.7), dtype)
ha, xa = to_act(hid, dtype), to_act(x, dtype)
outs = {}
for name in ("t", "s"):
    os.environ.pop("MZ_NO_T", None)
    if name == "s": os.environ["MZ_NO_T"] = "1"
    if wgs: os.environ["MZ_PERSIST_WGS"] = str(wgs)
    out = alloc_act(B, cout, H, W, dtype)
    op_conv_mix(dtype, ha, xa, w2, wmix, 0.3, out, B, H, W, cin, cout)
    outs[name] = from_act(out, cout)
d = (outs["t"] - outs["s"]).abs()
bad = d > 0.05
print("bad elements:", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
if len(idx):
    import collections
    print("images:", collections.Counter(idx[:, 0].tolist()))
    print("channel planes (ch // 8):", sorted(collections.Counter((idx[:, 1] // 8).tolist()).items()))
    print("tile rows (y // 12):", sorted(collections.Counter((idx[:, 2] // 12).tolist()).items()))
    print("row in tile (y % 12):", sorted(collections.Counter((idx[:, 2] % 12).tolist()).items()))
    print("tile cols (x // 64):", sorted(collections.Counter((idx[:, 3] // 64).tolist()).items()))
    print("frag col ((x % 64) // 16):", sorted(collections.Counter(((idx[:, 3] % 64) // 16).tolist()).items()))
    print("pixel in frag (x % 16):", sorted(collections.Counter((idx[:, 3] % 16).tolist()).items()))
    print("first:", idx[:10].tolist())
    t = idx[0].tolist()
    print("t:", outs["t"][t[0], :, t[2], t[3]].tolist()[:48])
    print("s:", outs["s"][t[0], :, t[2], t[3]].tolist()[:48])
