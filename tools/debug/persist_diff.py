"""GPU box: where do the persistent and per-tile 3x3 kernels differ?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import torch.nn.functional as F
from gpu_util import DTYPES, alloc_act, from_act, op_conv, q, to_act
from test_ops_gpu import PERSIST_CASES, rnd, wrnd

for dt, dtype in DTYPES.items():
    for case in PERSIST_CASES:
        for wgs in (8, 16):
            B, H, W, cin, cout, silu = case
            x = q(rnd((B, cin, H, W), 11), dtype)
            w = q(wrnd((cout, cin, 3, 3), 12), dtype)
            outs = []
            for env in ({"MZ_PERSIST_WGS": str(wgs)}, {"MZ_NO_PERSIST": "1"}):
                os.environ.pop("MZ_PERSIST_WGS", None); os.environ.pop("MZ_NO_PERSIST", None)
                os.environ.update(env)
                out = alloc_act(B, cout, H, W, dtype)
                op_conv(dtype, 0, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
                outs.append(from_act(out, cout))
            d = (outs[0] - outs[1]).abs()
            nz = (d > 0).nonzero()
            want = F.conv2d(x, w, padding=1)
            if silu: want = F.silu(want)
            print(dt, case, wgs, "ndiff", nz.shape[0], "max", d.max().item(), "err_p", (outs[0]-want).abs().max().item(),
                  "err_t", (outs[1]-want).abs().max().item())
            if nz.shape[0]:
                print("   first", nz[:6].tolist(), "last", nz[-3:].tolist())
                ys = nz[:, 2].unique().tolist(); xs = nz[:, 3].unique().tolist(); cs = nz[:, 1].unique().tolist(); bs = nz[:,0].unique().tolist()
                print("   b", bs, "rows", ys[:20], "cols", xs[:40], "chans", cs[:40])
