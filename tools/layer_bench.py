#!/usr/bin/env python3
"""Per-layer A/B timing of the 3x3 kernels on the cfg3 layer shapes: runs bench.py's model once per environment setting in ONE
process (interleaved rounds) and prints the per-shape device time from the library's HIP-event profile.
usage: python tools/layer_bench.py [rounds] -- env settings are the variants below."""
import csv, collections, os, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
from bench import MODELS, parameter_shapes
from ultrazoom_amd import MewZoom
from ultrazoom_amd.synth import synth_image, synth_state_dict

VARIANTS = {"base": {"MZ_NO_R": "1"}, "r": {}}
if os.environ.get("LAYER_BENCH_VARIANTS"):   # e.g. LAYER_BENCH_VARIANTS='{"a": {}, "b": {"MZ_NO_GEO40": "1"}}'
    import json
    VARIANTS = json.loads(os.environ["LAYER_BENCH_VARIANTS"])
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
batch = int(os.environ.get("LAYER_BENCH_BATCH", "0"))
if workload == "cfg2":
    cfg = MODELS["2x48"]
    x = synth_image(batch or 8, 540, 960, seed=1000).to("cuda", torch.bfloat16)
else:
    cfg = MODELS["4x96"]
    x = synth_image(batch or 3, 1080, 1920, seed=1000).to("cuda", torch.bfloat16)
sd = synth_state_dict(parameter_shapes(cfg), seed=1234)
models = {}
for name, env in VARIANTS.items():
    for k in [k for k in os.environ if k.startswith("MZ_") and not k.startswith("MZ_DEBUG")]:  # every kernel-selection knob
        os.environ.pop(k, None)
    os.environ.update(env)
    m = MewZoom(**cfg); m.load_state_dict(sd); m = m.to("cuda", torch.bfloat16).eval()
    m.upscale(x); torch.cuda.synchronize()   # engine (and its knobs) created under this environment
    models[name] = m
acc = {n: collections.defaultdict(list) for n in VARIANTS}
tot = {n: [] for n in VARIANTS}
for r in range(rounds):
    for name, m in models.items():
        h = m._engine.handle
        h.profile_enable(True)
        m.upscale(x); torch.cuda.synchronize()
        path = tempfile.mktemp(suffix=".csv")
        h.profile_dump(path)
        h.profile_read(); h.profile_enable(False)
        t = 0.0
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(path)):
            key = (row["kind"], row["H"], row["W"], row["cin"], row["cout"])
            per[key] += float(row["ms"]); t += float(row["ms"])
        for k, v in per.items():
            acc[name][k].append(v)
        tot[name].append(t)
        os.unlink(path)
first = next(iter(VARIANTS))
keys = list(acc[first].keys())
print(f"{'layer':40s}" + "".join(f"{n:>12s}" for n in VARIANTS) + "   ratio(last/first)")
for k in keys:
    vals = [min(acc[n][k]) for n in VARIANTS]
    print(f"{' '.join(k):40s}" + "".join(f"{v:12.3f}" for v in vals) + f"   {vals[-1] / vals[0]:.3f}")
vals = [min(tot[n]) for n in VARIANTS]
print(f"{'TOTAL ms per forward of ' + str(x.shape[0]) + ' images':40s}" + "".join(f"{v:12.3f}" for v in vals) + f"   {vals[-1] / vals[0]:.3f}")
