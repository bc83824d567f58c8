"""GPU box: where does mix16b_kernel differ from mix16_kernel? (C = 192 AdaptiveResidualMix, in a dirty process)"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import torch
from gpu_util import DTYPES, alloc_act, from_act, op_conv, q, to_act
from test_ops_gpu import rnd, wrnd
dtype = torch.bfloat16
# dirty the allocator's memory with NaN / huge patterns
junk = [torch.full((64 << 20,), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(8)]
del junk
def run(B, H, W, c, knob, seed):
    if knob == "1": os.environ["MZ_NO_MIX16B"] = "1"
    else: os.environ.pop("MZ_NO_MIX16B", None)
    x = q(rnd((B, c, H, W), seed), dtype); z = q(rnd((B, c, H, W), seed + 1), dtype); w = q(wrnd((c, 2 * c, 1, 1), seed + 2), dtype)
    out = alloc_act(B, c, H, W, dtype); out.fill_(float("nan")); torch.cuda.synchronize()
    op_conv(dtype, 3, to_act(x, dtype), to_act(z, dtype), w, 0.37, out, B, H, W, 2 * c, c)
    return from_act(out, c).float()
run(1, 3, 50, 384, "0", 3)
nbad = 0
for (B, H, W) in [(1, 9, 40), (2, 33, 65), (3, 16, 16), (2, 64, 65)]:
    for rep in range(10):
        new = run(B, H, W, 192, "0", 7 + rep)
        old = run(B, H, W, 192, "1", 7 + rep)
        bad = ~((new == old) | (torch.isnan(new) & torch.isnan(old)))
        msg = f"{(B, H, W)} rep {rep}: differing {int(bad.sum())} of {bad.numel()}, nan new {int(torch.isnan(new).sum())} old {int(torch.isnan(old).sum())}"
        if bad.any():
            pix = bad.any(dim=1).flatten(); ch = bad.any(dim=3).any(dim=2).any(dim=0).flatten()
            msg += " | units " + str([i // 32 for i in range(0, pix.numel(), 32) if pix[i:i + 32].any()][:20])
            msg += " | planes " + str([i // 8 for i in range(0, 192, 8) if ch[i:i + 8].any()])
            idx = bad.nonzero()[0].tolist(); msg += f" | first {idx} new {new[tuple(idx)].item()} old {old[tuple(idx)].item()}"
            nbad += 1
            if nbad <= 6: print(msg)
print('LIB', os.environ.get('MEWZOOM_HIP_LIB', 'default'), 'runs with differences:', nbad, 'of 40')
