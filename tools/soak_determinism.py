#!/usr/bin/env python3
"""GPU box: repeats the forward of both bench models on fixed inputs and checks that every repetition returns the SAME BITS (a race in a
kernel's synchronisation -- a wait that leaves too much in flight, a slot reused too early -- shows up as a run that differs).
usage: python tools/soak_determinism.py [repetitions]"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
from bench import MODELS, parameter_shapes
from ultrazoom_amd import MewZoom
from ultrazoom_amd.synth import synth_image, synth_state_dict

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, shape, dtype in (("4x96", (3, 1080, 1920), torch.bfloat16), ("2x48", (8, 540, 960), torch.bfloat16), ("4x96", (1, 270, 480), torch.float16)):
    cfg = MODELS[name]
    m = MewZoom(**cfg); m.load_state_dict(synth_state_dict(parameter_shapes(cfg), seed=1234)); m = m.to("cuda", dtype).eval()
    x = synth_image(*shape, seed=77).to("cuda", dtype)
    ref = m.upscale(x).clone(); torch.cuda.synchronize()
    assert torch.isfinite(ref.float()).all()
    bad = 0; t0 = time.time()
    for i in range(reps):
        y = m.upscale(x)
        if not torch.equal(y, ref):
            bad += 1
            print(f"  repetition {i}: {(y != ref).sum().item()} elements differ")
    torch.cuda.synchronize()
    print(f"{name} {tuple(shape)} {str(dtype).split('.')[-1]}: {reps} repetitions, {bad} differ from the first ({time.time() - t0:.1f} s)")
    if bad: sys.exit(1)
print("deterministic")
