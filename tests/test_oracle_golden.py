"""The CPU oracle against the fixtures produced by the reference itself (tests/golden/make_golden.py).

This is what pins the oracle: every fixture is an output of the reference's own model.py.
"""

import json

import numpy as np
import pytest
import torch

from oracle import mewzoom_oracle as oracle
from golden_util import GOLDEN, MODEL_CASES, GoldenCase
from ultrazoom_amd.synth import hash_uniform, synth_image, synth_state_dict

# fp32 CPU vs fp32 CPU of the same maths in a different op order: round-off only.
TOL = 2e-5


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_case(name):
    case = GoldenCase(name)
    shapes = oracle.parameter_shapes(case.config)
    assert {k: list(v) for k, v in shapes.items()} == case.meta["shapes"]
    assert list(shapes) == list(case.meta["shapes"])  # same order as the reference's state_dict()
    assert sum(int(np.prod(s)) if s else 1 for s in shapes.values()) == case.meta["num_params"]
    taps = {}
    with torch.inference_mode():
        sr, qa = oracle.forward(case.config, case.weights(), case.image(), taps)
        up = oracle.upscale(case.config, case.weights(), case.image())
    errs = case.compare_sr(sr, up)
    assert errs["sr"] < TOL and errs["up"] < TOL, errs
    assert np.abs(qa.numpy() - case.data["qa"]).max() < TOL
    assert up.min() >= 0 and up.max() <= 1
    for key in case.data.files:
        if key.startswith("tap_"):
            got = taps[key[4:]].numpy()
            assert got.shape == case.data[key].shape
            assert np.abs(got - case.data[key]).max() < TOL, key


@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("f16", torch.float16)])
@pytest.mark.parametrize("name", MODEL_CASES)
def test_rounding_matched_mode_is_no_worse_than_the_reference_in_reduced_precision(name, tag, dtype):
    """The oracle's `storage=dtype` mode (fp32 arithmetic, tensors rounded where the GPU path rounds them) is the
    yardstick of the 16-bit GPU tests wherever no fixture exists.  Tie it to the reference: against the reference's fp32
    outputs it must err no more than the reference's OWN bf16 / fp16 run does (stored in the fixture), and it must
    stay within a small factor of it, so that it is neither a loose nor an arbitrary bar."""
    case = GoldenCase(name)
    ref_max, ref_mse, ref_qa = (float(v) for v in case.data[f"ref_{tag}_err"])
    with torch.inference_mode():
        sr, qa = oracle.forward(case.config, case.weights(), case.image(), storage=dtype)
    errs = case.compare_sr(sr, sr.clamp(0, 1))
    mse = case.mse_up(sr.clamp(0, 1))
    qa_err = np.abs(qa.numpy() - case.data["qa"]).max()
    assert errs["sr"] <= ref_max and mse <= ref_mse, (errs, ref_max, mse, ref_mse)
    assert errs["sr"] >= 0.2 * ref_max and mse >= 0.1 * ref_mse, "the matched mode must carry real rounding error"
    assert qa_err <= ref_qa + (2.0 ** -9 if tag == "bf16" else 2.0 ** -12)
    # the stored 16-bit outputs of the reference decode to the stored error figures
    key = f"ref_{tag}_sr_samples" if case.sampled else f"ref_{tag}_sr"
    low = torch.from_numpy(case.data[key].view(np.int16).copy()).view(dtype).float()
    want = torch.from_numpy(case.data["sr_samples"] if case.sampled else case.data["sr"])
    got_max = (low.reshape(-1) - want.reshape(-1)).abs().max().item()
    assert got_max <= ref_max + 1e-9 and (case.sampled or abs(got_max - ref_max) < 1e-9)


def test_ops():
    g = np.load(GOLDEN / "g6_ops.npz")
    x = synth_image(1, 9, 11, 21)
    for r in (2, 4, 8):
        got = oracle.bicubic_upsample(x, r).numpy()
        assert np.abs(got - g[f"bicubic_r{r}"]).max() < 1e-6
    c = 16
    u = lambda shape, seed: torch.from_numpy(
        (2.0 * hash_uniform(int(np.prod(shape)), seed) - 1.0).reshape(shape).astype(np.float32)
    )
    a, b = u((2, c, 7, 9), 32), u((2, c, 7, 9), 33)
    sd = synth_state_dict({"conv.weight": (c, 2 * c, 1, 1), "alpha": ()}, 31)
    assert np.abs(oracle.residual_mix(a, b, sd["conv.weight"], sd["alpha"]).numpy() - g["mix"]).max() < TOL
    w = synth_state_dict({"conv.weight": (32, c, 3, 3)}, 34)["conv.weight"]
    assert np.abs(oracle.subpixel_conv(a, w).numpy() - g["subpixel"]).max() < TOL
    w = synth_state_dict({"conv.weight": (2 * c, c, 2, 2)}, 35)["conv.weight"]
    assert np.abs(torch.nn.functional.conv2d(a, w, stride=2).numpy() - g["crush"]).max() < TOL
    sd = synth_state_dict({"conv.weight": (3, c, 3, 3), "conv.bias": (3,)}, 37)
    assert np.abs(oracle.quality_head(a, sd["conv.weight"], sd["conv.bias"]).numpy() - g["quality"]).max() < TOL
    assert np.array_equal(oracle.fit_to(a, (8, 10)).numpy(), g["fit_pad"])
    assert np.array_equal(oracle.fit_to(a, (9, 12)).numpy(), g["fit_pad2"])
    assert np.array_equal(oracle.fit_to(a, (5, 6)).numpy(), g["fit_crop"])


def test_validation_matches_reference():
    trials = json.loads((GOLDEN / "validation.json").read_text())
    for name, t in trials.items():
        if t["raises"] is None:
            oracle.validate_config(t["kwargs"])
        else:
            assert t["raises"] == "AssertionError"
            with pytest.raises(AssertionError):
                oracle.validate_config(t["kwargs"])


def test_flop_count_matches_survey():
    # verified there against torch.utils.flop_counter on the reference.
    names = ("primary", "secondary", "tertiary", "quaternary")
    def cfg(r, c, l):
        d = {"upscale_ratio": r, "hidden_ratio": 2, "num_deg_features": 3}
        for n, ci, li in zip(names, c, l):
            d[f"{n}_channels"], d[f"{n}_layers"] = ci, li
        return d
    f2 = oracle.flops_per_image(cfg(2, (48, 96, 192, 384), (4, 4, 4, 8)), 256, 256)
    assert abs(f2 / 1e9 - 261.64) < 0.05
    f4 = oracle.flops_per_image(cfg(4, (96, 192, 384, 768), (8, 8, 8, 16)), 1080, 1920)
    assert abs(f4 / 1e12 - 69.43) < 0.05
