"""Drop-in boundary behaviour on a real MI355X: checkpoint ingestion through the HIP path, CUDA-stream semantics of the
engine, module copies after a forward, explicit weight refresh."""

import copy
import json

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, GoldenCase
from oracle import mewzoom_oracle as oracle
from ultrazoom_amd import MewZoom
from ultrazoom_amd.synth import synth_image

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3


def _g10():
    d = np.load(GOLDEN / "g10_checkpoint.npz")
    raw = {k[len("raw/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("raw/")}
    baked = {k[len("baked/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("baked/")}
    return json.loads(str(d["config_json"])), raw, baked, int(d["lora_rank"]), float(d["lora_alpha"])


@pytest.mark.parametrize("recipe", ["load_training_checkpoint", "reference_recipe"])
def test_training_checkpoint_through_the_hip_path(recipe):
    """SURVEY 8f N2 on the GPU: a checkpoint written by the reference itself (weight norm + LoRA, `_orig_mod.` keys;
    primary_channels = 8, so every level runs the padded-channel path: 8 -> 16) is ingested and run through the HIP
    kernels; the result must equal the oracle fed the weights the REFERENCE's own remove_parameterizations() baked."""
    cfg, raw, baked, rank, alpha = _g10()
    m = MewZoom(**cfg)
    if recipe == "load_training_checkpoint":
        m.load_training_checkpoint(raw, lora_alpha=alpha)
    else:  # test_compare.py:36-45, statement for statement
        m.add_weight_norms()
        m.add_lora_adapters(rank, alpha)
        state_dict = dict(raw)
        for key in list(state_dict.keys()):
            state_dict[key.replace("_orig_mod.", "")] = state_dict.pop(key)
        m.load_state_dict(state_dict)
        m.remove_parameterizations()
    m = m.to("cuda").eval()
    x = synth_image(2, 45, 52, seed=5)
    with torch.inference_mode():
        want_sr, want_qa = oracle.forward(cfg, baked, x)
    sr, qa = m.forward(x.cuda())
    err = (sr.cpu() - want_sr).abs().max().item()
    qerr = (qa.cpu() - want_qa).abs().max().item()
    print(f"g10 checkpoint via {recipe}: f32 max-abs {err:.3e}, qa {qerr:.3e}")
    assert err <= F32_TOL and qerr <= F32_TOL
    up = m.upscale(x.cuda())
    assert (up.cpu() - want_sr.clamp(0, 1)).abs().max().item() <= F32_TOL


def test_engine_builds_and_runs_on_a_side_stream():
    """the engine's first use (zero page / stem-weight zero fill + weight packing) and the forward itself
    on a NON-default stream, then the same model on the default stream and on a second side stream: the cached
    workspace is handed from stream to stream in order."""
    case = GoldenCase("g3_4x_c16")
    x = case.image().cuda()
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda").eval()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        y1 = m.upscale(x)          # builds the engine on s1
        y1b = m.upscale(x)
    y0 = m.upscale(x)              # default stream, same workspace
    with torch.cuda.stream(s2):
        y2 = m.upscale(x)
    torch.cuda.synchronize()
    errs = case.compare_sr(m.forward(x)[0], y1)
    assert errs["sr"] <= F32_TOL and errs["up"] <= F32_TOL, errs
    assert torch.equal(y1, y1b) and torch.equal(y1, y0) and torch.equal(y1, y2)


def test_deepcopy_after_forward_and_refresh_weights():
    case = GoldenCase("g1_2x_c16")
    x = case.image().cuda()
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda").eval()
    y = m.upscale(x)
    clone = copy.deepcopy(m)       # the live engine handle must not travel
    assert clone._engine is None and m._engine is not None
    assert torch.equal(clone.upscale(x), y)
    # a .data write is invisible to the version counter: refresh_weights() re-packs
    original = m.stem.conv.bias.detach().clone()
    with torch.no_grad():
        m.stem.conv.bias.data.add_(0.25)
    stale = m.upscale(x)
    assert torch.equal(stale, y), "documented behaviour: .data writes need refresh_weights()"
    m.refresh_weights()
    fresh = m.upscale(x)
    assert not torch.equal(fresh, y)
    # an ordinary in-place update IS detected
    with torch.no_grad():
        m.stem.conv.bias.copy_(original)
    assert torch.equal(m.upscale(x), y)
    # a model moved under inference_mode has no version counters: it still runs
    with torch.inference_mode():
        mi = MewZoom(**case.config)
        mi.load_state_dict(case.weights())
        mi = mi.to("cuda").eval()
    assert torch.equal(mi.upscale(x), y)
