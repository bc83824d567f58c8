#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE implementation.

Runs only in the build container, where the reference checkout is mounted read-only at
/root/reference.  It never travels to the GPU box; only the small ``*.npz`` files it writes do.

How the reference is loaded: ``src/ultrazoom/model.py`` at the surveyed commit does not import on
Python 3.10 (PEP 695 ``type`` statement, ``typing.Self``) and cannot construct ``MewZoom`` because
of an undefined name in an ``assert``.  The file is therefore read as text,
three one-token substitutions are applied IN MEMORY and the result is exec'd into a fresh module.
Nothing from the reference is written to disk here.

Weights and inputs are not stored: both sides regenerate them from ``ultrazoom_amd.synth`` (an
integer hash), so a fixture holds just the case description and the reference's outputs.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))

from ultrazoom_amd.synth import hash_uniform, synth_image, synth_state_dict  # noqa: E402

REF_MODEL = Path("/root/reference/src/ultrazoom/model.py")

SUBSTITUTIONS = (
    ("from typing import Self", "from typing_extensions import Self"),
    ("type FeatureMapSize = ", "FeatureMapSize = "),
    ("qa_num_features > 0", "num_deg_features > 0"),
)


def load_reference() -> types.ModuleType:
    text = REF_MODEL.read_text()
    for old, new in SUBSTITUTIONS:
        assert text.count(old) == 1, f"expected exactly one occurrence of {old!r}"
        text = text.replace(old, new)
    mod = types.ModuleType("reference_ultrazoom_model")
    exec(compile(text, str(REF_MODEL), "exec"), mod.__dict__)
    return mod


def cfg(r, c, layers, hr=2, f=3):
    names = ("primary", "secondary", "tertiary", "quaternary")
    out = {"upscale_ratio": r, "hidden_ratio": hr, "num_deg_features": f}
    for n, ci, li in zip(names, c, layers):
        out[f"{n}_channels"] = ci
        out[f"{n}_layers"] = li
    return out


MODEL_CASES = {
    # name: (config, (B, H, W), weight seed, image seed, store-taps?, sample-only?)
    "g1_2x_c16": (cfg(2, (16, 32, 64, 128), (2, 2, 2, 2)), (1, 32, 32), 1, 1, True, False),
    "g2_odd_37x45": (cfg(2, (16, 32, 64, 128), (2, 2, 2, 2)), (2, 37, 45), 1, 2, False, False),
    "g2_odd_135x240": (cfg(2, (16, 32, 64, 128), (2, 2, 2, 2)), (1, 135, 240), 1, 3, False, True),
    "g3_4x_c16": (cfg(4, (16, 32, 64, 128), (2, 2, 2, 2)), (1, 24, 40), 2, 4, False, False),
    "g4_8x_c16": (cfg(8, (16, 32, 64, 128), (2, 2, 2, 2)), (1, 16, 16), 3, 5, False, False),
    "g5_hr1": (cfg(2, (16, 32, 64, 128), (2, 2, 2, 2), hr=1), (1, 24, 24), 4, 6, False, False),
    "g5_hr4": (cfg(2, (16, 32, 64, 128), (2, 2, 2, 2), hr=4), (1, 24, 24), 5, 7, False, False),
    "g5_layers_3254": (cfg(2, (16, 32, 64, 128), (3, 2, 5, 4)), (1, 24, 32), 6, 8, True, False),
    "g8_c24_f5": (cfg(2, (24, 40, 72, 136), (2, 2, 2, 2), f=5), (1, 21, 19), 7, 9, False, False),
    "g9_4x_c32": (cfg(4, (32, 64, 128, 256), (2, 2, 2, 4)), (1, 40, 56), 8, 10, False, True),
    "g7_cfg1_2x_c48": (cfg(2, (48, 96, 192, 384), (4, 4, 4, 8)), (1, 256, 256), 9, 11, False, True),
}

N_SAMPLES = 4096


def sample_indices(numel: int, seed: int) -> np.ndarray:
    u = hash_uniform(N_SAMPLES, 0xC0FFEE + seed)
    return np.minimum((u.astype(np.float64) * numel).astype(np.int64), numel - 1)


def run_model_case(ref, name, spec):
    config, (B, H, W), wseed, iseed, store_taps, sample_only = spec
    model = ref.MewZoom(**config)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, wseed)
    model.load_state_dict(sd)
    model.eval()
    x = synth_image(B, H, W, iseed)
    out = {}
    with torch.inference_mode():
        sr, qa = model.forward(x)
        up = model.upscale(x)
        out["qa"] = qa.numpy()
        if sample_only:
            flat = sr.reshape(-1).numpy()
            idx = sample_indices(flat.size, iseed)
            out["sr_idx"] = idx
            out["sr_samples"] = flat[idx]
            out["up_samples"] = up.reshape(-1).numpy()[idx]
            out["sr_stats"] = np.array(
                [sr.min().item(), sr.max().item(), sr.double().mean().item(), sr.double().sum().item()]
            )
            out["sr_chan_mean"] = sr.double().mean(dim=(0, 2, 3)).numpy()
        else:
            out["sr"] = sr.numpy()
            out["up"] = up.numpy()
        # The reference's OWN reduced-precision results (model.to(dtype) on CPU, same weights and image): the yardstick
        # for the 16-bit GPU kernels, whose error against the fp32 result is gated relative to this one.
        up32 = up
        for tag, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            low = ref.MewZoom(**config)
            low.load_state_dict(sd)
            low = low.to(dt).eval()
            sr_l, qa_l = low.forward(x.to(dt))
            up_l = low.upscale(x.to(dt))
            sr_lf, up_lf = sr_l.float(), up_l.float()
            out[f"ref_{tag}_err"] = np.array(
                [
                    (sr_lf - sr).abs().max().item(),                       # max-abs of forward() vs fp32
                    (up_lf.double() - up32.double()).pow(2).mean().item(),  # MSE of upscale() vs fp32
                    (qa_l.float() - qa).abs().max().item(),                # max-abs of the degradation features
                ]
            )
            bits = (lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16))  # exact 16-bit patterns
            if sample_only:
                out[f"ref_{tag}_sr_samples"] = bits(sr_l.reshape(-1)[torch.from_numpy(idx)])
            else:
                out[f"ref_{tag}_sr"] = bits(sr_l)
            out[f"ref_{tag}_qa"] = bits(qa_l)
            del low
        if store_taps:
            # sub-modules are called through .forward exactly as the reference does
            s = model.bicubic.forward(x)
            z0 = model.stem.forward(x)
            z1, z2, z3, z4, zq = model.unet.encoder.forward(z0)
            zd = model.unet.decoder.forward(z4, z3, z2, z1)
            zh = model.head.forward(zd)
            for k, v in dict(
                bicubic=s, stem=z0, enc1=z1, enc2=z2, enc3=z3, enc4=z4, unet=zd, head=zh
            ).items():
                out[f"tap_{k}"] = v.numpy()
    out["meta"] = np.array(
        json.dumps(
            {
                "config": config,
                "input": [B, H, W],
                "weight_seed": wseed,
                "image_seed": iseed,
                "shapes": {k: list(v) for k, v in shapes.items()},
                "num_params": int(model.num_params),
            }
        )
    )
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(f"{name}: sr {tuple(sr.shape)} range [{sr.min():.4f}, {sr.max():.4f}] qa {qa.flatten()[:3].tolist()}")


def run_op_cases(ref):
    """Per-operator fixtures."""
    out = {}
    x = synth_image(1, 9, 11, 21)
    for r in (2, 4, 8):
        up = torch.nn.Upsample(scale_factor=r, mode="bicubic")
        out[f"bicubic_r{r}"] = up(x).numpy()

    c = 16
    u = lambda shape, seed: torch.from_numpy(
        (2.0 * hash_uniform(int(np.prod(shape)), seed) - 1.0).reshape(shape).astype(np.float32)
    )
    with torch.inference_mode():
        mix = ref.AdaptiveResidualMix(c)
        sd = synth_state_dict({"conv.weight": (c, 2 * c, 1, 1), "alpha": ()}, 31)
        mix.load_state_dict(sd)
        a, b = u((2, c, 7, 9), 32), u((2, c, 7, 9), 33)
        out["mix"] = mix.forward(a, b).numpy()

        sp = ref.SubpixelConv2d(c, 8, 2)
        sp.load_state_dict(synth_state_dict({"conv.weight": (32, c, 3, 3)}, 34))
        out["subpixel"] = sp.forward(a).numpy()

        pc = ref.PixelCrush(c, 2 * c, 2)
        pc.load_state_dict(synth_state_dict({"conv.weight": (2 * c, c, 2, 2)}, 35))
        out["crush"] = pc.forward(a).numpy()  # 7x9 -> 3x4 (floors)

        ib = ref.InvertedBottleneck(c, 2)
        ib.load_state_dict(
            synth_state_dict({"conv1.weight": (2 * c, c, 3, 3), "conv2.weight": (c, 2 * c, 3, 3)}, 36)
        )
        out["bottleneck"] = ib.forward(a).numpy()

        qa = ref.QualityAssessor(c, 3)
        qa.load_state_dict(synth_state_dict({"conv.weight": (3, c, 3, 3), "conv.bias": (3,)}, 37))
        out["quality"] = qa.forward(a).numpy()

        fit = ref.Decoder.crop_feature_maps
        out["fit_pad"] = fit(a, (8, 10)).numpy()
        out["fit_pad2"] = fit(a, (9, 12)).numpy()
        out["fit_crop"] = fit(a, (5, 6)).numpy()
    np.savez_compressed(HERE / "g6_ops.npz", **out)
    print("g6_ops:", {k: v.shape for k, v in out.items()})


def run_validation_cases(ref):
    """Which constructor arguments the reference rejects (AssertionError)."""
    base = cfg(2, (16, 32, 64, 128), (2, 2, 2, 2))
    trials = {
        "ok": {},
        "ratio_3": {"upscale_ratio": 3},
        "ratio_1": {"upscale_ratio": 1},
        "ratio_16": {"upscale_ratio": 16},
        "primary_layers_1": {"primary_layers": 1},
        "secondary_layers_1": {"secondary_layers": 1},
        "tertiary_layers_0": {"tertiary_layers": 0},
        "quaternary_layers_1": {"quaternary_layers": 1},
        "hidden_ratio_3": {"hidden_ratio": 3},
        "hidden_ratio_8": {"hidden_ratio": 8},
        "primary_channels_3": {"primary_channels": 3},
        "primary_channels_2": {"primary_channels": 2},
        "num_deg_features_0": {"num_deg_features": 0},
    }
    result = {}
    for name, delta in trials.items():
        kw = dict(base)
        kw.update(delta)
        try:
            ref.MewZoom(**kw)
            result[name] = {"kwargs": kw, "raises": None}
        except Exception as e:  # noqa: BLE001
            result[name] = {"kwargs": kw, "raises": type(e).__name__}
    (HERE / "validation.json").write_text(json.dumps(result, indent=1, sort_keys=True))
    print("validation:", {k: v["raises"] for k, v in result.items()})


def run_checkpoint_case(ref):
    """g10: a TRAINING-style checkpoint of a tiny model as the reference itself writes it -- weight-norm
    parametrised (model.py:117-122), then LoRA adapters (model.py:124-129, 1361-1390) with non-zero B factors, keys
    prefixed like a torch.compile'd module -- and the plain weights `remove_parameterizations()` (model.py:131-139)
    bakes from it.  Pins `bake_state_dict`."""
    config = cfg(2, (8, 16, 16, 16), (2, 2, 2, 2))
    alpha, rank = 0.75, 2
    torch.manual_seed(1234)  # add_lora_adapters draws lora_a from the global RNG: independent of which cases ran before
    model = ref.MewZoom(**config)
    sd0 = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 21)
    model.load_state_dict(sd0)
    model.add_weight_norms()
    model.add_lora_adapters(rank, alpha)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith("lora_b"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.05)
            if k.endswith("original0"):
                v.mul_(1.0 + 0.1 * torch.rand(v.shape, generator=g))  # g no longer equals |v|
    raw = {"_orig_mod." + k: v.detach().clone() for k, v in model.state_dict().items()}
    model.remove_parameterizations()
    baked = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = {"lora_alpha": np.float32(alpha), "lora_rank": np.int32(rank), "config_json": np.array(json.dumps(config))}
    for k, v in raw.items():
        out["raw/" + k] = v.numpy()
    for k, v in baked.items():
        out["baked/" + k] = v.numpy()
    np.savez_compressed(HERE / "g10_checkpoint.npz", **out)
    print("g10_checkpoint", len(raw), "raw keys ->", len(baked), "baked keys")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    only = set(sys.argv[1:])
    for name, spec in MODEL_CASES.items():
        if only and name not in only:
            continue
        run_model_case(ref, name, spec)
    if not only or "ops" in only:
        run_op_cases(ref)
    if not only or "validation" in only:
        run_validation_cases(ref)
    if not only or "checkpoint" in only:
        run_checkpoint_case(ref)


if __name__ == "__main__":
    main()
