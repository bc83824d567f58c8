"""GPU box: run one 3x3 conv layer N times on random data (run under rocprofv3 --kernel-trace --stats to get its time).
usage: layer_time.py B H W cin cout [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from gpu_util import alloc_act, op_conv
B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dtype = torch.bfloat16
torch.manual_seed(0)
x = (torch.rand(B, cin // 8, H, W, 8, device="cuda") * 2 - 1).to(dtype)
w = ((torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) * (3.0 / (9 * cin)) ** 0.5 * 1.7).float().contiguous()
out = alloc_act(B, cout, H, W, dtype)
for _ in range(n):
    op_conv(dtype, 0, x, None, w, 0.0, out, B, H, W, cin, cout, silu=int(os.environ.get("MZ_LAYER_SILU", "1")))
torch.cuda.synchronize()
print("done", out.float().abs().mean().item())
