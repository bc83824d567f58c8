"""Host-side behaviour of the drop-in MewZoom class (no GPU): parameter tree, HF round trip,
constructor validation, checkpoint baking, and the refusal to compute without an MI355X."""

import json

import pytest
import torch

from golden_util import GOLDEN, GoldenCase
from ultrazoom_amd import MewZoom, bake_state_dict
from ultrazoom_amd.synth import synth_state_dict


def test_state_dict_layout_matches_reference():
    for name in ("g1_2x_c16", "g3_4x_c16", "g4_8x_c16", "g5_layers_3254", "g8_c24_f5"):
        case = GoldenCase(name)
        m = MewZoom(**case.config)
        sd = m.state_dict()
        assert list(sd) == list(case.meta["shapes"])
        assert {k: list(v.shape) for k, v in sd.items()} == case.meta["shapes"]
        assert m.num_params == case.meta["num_params"]
        assert m.upscale_ratio == case.config["upscale_ratio"]
        m.load_state_dict(case.weights())  # strict


def test_constructor_raises_like_the_reference():
    trials = json.loads((GOLDEN / "validation.json").read_text())
    for name, t in trials.items():
        if t["raises"] is None:
            MewZoom(**t["kwargs"])
        else:
            with pytest.raises(AssertionError):
                MewZoom(**t["kwargs"])


def test_hf_round_trip(tmp_path):
    case = GoldenCase("g1_2x_c16")
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m.save_pretrained(tmp_path)
    cfg = json.loads((tmp_path / "config.json").read_text())
    assert {k: cfg[k] for k in case.config} == case.config
    m2 = MewZoom.from_pretrained(tmp_path)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_no_cpu_fallback():
    case = GoldenCase("g1_2x_c16")
    m = MewZoom(**case.config)
    with pytest.raises(RuntimeError, match="MI355X"):
        m.upscale(case.image())
    with pytest.raises(RuntimeError, match="MI355X"):
        m.forward(case.image())


def test_bake_weight_norm_checkpoint():
    case = GoldenCase("g1_2x_c16")
    sd = case.weights()
    raw = {}
    for k, v in sd.items():
        if k.endswith("conv.weight") or k.endswith("conv1.weight") or k.endswith("conv2.weight"):
            base = k[: -len(".weight")]
            norm = v.flatten(1).norm(dim=1).reshape(-1, 1, 1, 1)
            raw["_orig_mod." + base + ".parametrizations.weight.original0"] = norm.clone()
            raw["_orig_mod." + base + ".parametrizations.weight.original1"] = v * 3.0  # direction only
        else:
            raw["_orig_mod." + k] = v
    baked = bake_state_dict(raw)
    assert set(baked) == set(sd)
    for k in sd:
        assert torch.allclose(baked[k], sd[k], atol=1e-6), k
    m = MewZoom(**case.config)
    m.load_training_checkpoint(raw)


def test_bake_reference_checkpoint_weight_norm_and_lora():
    """g10: a checkpoint written by the reference itself (weight norm + LoRA adapters, `_orig_mod.` prefixes) and the
    weights its own `remove_parameterizations()` produced from it."""
    import json
    from pathlib import Path

    import numpy as np

    d = np.load(Path(__file__).parent / "golden" / "g10_checkpoint.npz")
    raw = {k[len("raw/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("raw/")}
    want = {k[len("baked/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("baked/")}
    alpha = float(d["lora_alpha"])
    with pytest.raises(ValueError, match="lora_alpha"):
        bake_state_dict(raw)
    baked = bake_state_dict(raw, lora_alpha=alpha)
    assert list(baked) and set(baked) == set(want)
    for k, v in want.items():
        assert baked[k].shape == v.shape, k
        assert torch.allclose(baked[k], v, atol=2e-7, rtol=1e-6), (k, (baked[k] - v).abs().max().item())
    m = MewZoom(**json.loads(str(d["config_json"])))
    m.load_training_checkpoint(raw, lora_alpha=alpha)
    for k, v in m.state_dict().items():
        assert torch.allclose(v, want[k], atol=2e-7, rtol=1e-6), k
    # a wrong alpha must not pass silently
    off = bake_state_dict(raw, lora_alpha=alpha * 2)
    assert any((off[k] - want[k]).abs().max().item() > 1e-3 for k in want if k.endswith("conv.weight"))


def _g10():
    import numpy as np

    d = np.load(GOLDEN / "g10_checkpoint.npz")
    raw = {k[len("raw/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("raw/")}
    want = {k[len("baked/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("baked/")}
    return json.loads(str(d["config_json"])), raw, want, int(d["lora_rank"]), float(d["lora_alpha"])


def test_reference_checkpoint_recipe_runs_unchanged():
    """The statements of the reference's own loader (test_compare.py:32-45, validate.py:57-67) against the drop-in class:
    add_weight_norms() [+ add_lora_adapters()] -> strip `_orig_mod.` -> load_state_dict() -> remove_parameterizations()."""
    cfg, raw, want, rank, alpha = _g10()
    model = MewZoom(**cfg)
    model.add_weight_norms()
    model.add_lora_adapters(rank, alpha)
    state_dict = dict(raw)
    # Compensate for compiled state dict.  (test_compare.py:39-41)
    for key in list(state_dict.keys()):
        state_dict[key.replace("_orig_mod.", "")] = state_dict.pop(key)
    model.load_state_dict(state_dict)
    model.remove_parameterizations()
    model.eval()
    assert set(model.state_dict()) == set(want)
    for k, v in model.state_dict().items():
        assert torch.allclose(v, want[k], atol=2e-7, rtol=1e-6), k
    # LoRA tensors without add_lora_adapters(): the reference rejects the unexpected keys; so does this class
    with pytest.raises(RuntimeError, match="add_lora_adapters"):
        m2 = MewZoom(**cfg)
        m2.add_weight_norms()
        m2.load_state_dict(state_dict)
    # interface completeness (model.py:104-147)
    model.enable_activation_checkpointing()
    before = model.stem.conv.weight.clone()
    model.initialize_weights()
    assert not torch.equal(before, model.stem.conv.weight)
    with pytest.raises(AssertionError):
        model.add_lora_adapters(0, 1.0)


def test_module_copies_and_pickles():
    import copy
    import io
    import pickle

    from ultrazoom_amd import _ffi

    case = GoldenCase("g1_2x_c16")
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone._engine is None
        for (k1, v1), (k2, v2) in zip(m.state_dict().items(), clone.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2)
    buf = io.BytesIO()
    torch.save(m, buf)
    # the raw handle itself must refuse to be copied (two owners of one mz_handle would free it twice)
    h = _ffi.Handle(case.config, _ffi.MZ_F32)
    with pytest.raises(TypeError):
        copy.deepcopy(h)
    with pytest.raises(TypeError):
        pickle.dumps(h)
    h.close()
    m.refresh_weights()  # a no-op without an engine
