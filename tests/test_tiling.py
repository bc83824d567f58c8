"""Exact image-level tiling."""

import pytest
import torch

from golden_util import GoldenCase
from ultrazoom_amd import MewZoom
from ultrazoom_amd.synth import synth_image
from ultrazoom_amd.tiling import receptive_field, upscale_tiled


def test_receptive_field_bounds():
    small = GoldenCase("g1_2x_c16").config
    assert receptive_field(small) % 8 == 0
    # 4 levels x (1 + 1) blocks x 2 convs: 4 * (1 + 2 + 4 + 8) = 60 pixels from the blocks alone
    assert 60 < receptive_field(small) <= 104
    big = dict(small, upscale_ratio=4, primary_layers=8, secondary_layers=8, tertiary_layers=8, quaternary_layers=16)
    assert receptive_field(big) >= 360  # ">= 360 LR px" for the 4X / 40-layer model


def test_halo_smaller_than_receptive_field_is_refused():
    cfg = GoldenCase("g1_2x_c16").config
    m = MewZoom(**cfg)
    with pytest.raises(ValueError, match="receptive field"):
        upscale_tiled(m, torch.zeros(1, 3, 64, 64), tile=(32, 32), halo=16)


def test_tiling_with_the_cpu_oracle_as_the_model():
    """The slicing logic itself, run on the CPU with the oracle standing in for the model: tiled == untiled up to the
    float32 summation-order noise of the CPU convolution library (whose blocking depends on the tensor size)."""
    from oracle import mewzoom_oracle as oracle

    case = GoldenCase("g1_2x_c16")

    class OracleModel:
        _cfg = case.config

        def upscale(self, x):
            return oracle.upscale(case.config, case.weights(), x)

    x = synth_image(1, 150, 210, seed=12)
    full = OracleModel().upscale(x)
    tiled = upscale_tiled(OracleModel(), x, tile=(64, 96))
    assert tiled.shape == full.shape
    assert (tiled - full).abs().max().item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_tiled_equals_untiled_bit_for_bit(dt):
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
    case = GoldenCase("g1_2x_c16")
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda", dtype).eval()
    x = synth_image(2, 203, 277, seed=13).to("cuda", dtype)  # odd sizes: floors and pads at every level
    full = m.upscale(x)
    tiled = upscale_tiled(m, x, tile=(64, 120))
    assert torch.equal(tiled, full)


@pytest.mark.gpu
def test_tiled_4x_model_equals_untiled():
    case = GoldenCase("g3_4x_c16")
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda", torch.float16).eval()
    x = synth_image(1, 171, 232, seed=14).to("cuda", torch.float16)
    assert torch.equal(upscale_tiled(m, x, tile=(80, 80)), m.upscale(x))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g2_odd_135x240", "g7_cfg1_2x_c48"])
def test_tiled_upscale_against_the_reference_fixtures(name):
    """N4 against the REFERENCE's own outputs (not against the untiled HIP result): `upscale_tiled` of the fixture's input in fp32
    must meet the fixture within the north_star tolerance (1e-3 max-abs).  g2: odd sizes (floors and zero pads at every level) cut
    into 64 x 96 tiles; g7: the BASELINE configs[0] model (48 channels / 20 layers) on 256 x 256, sampled fixture."""
    case = GoldenCase(name)
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda", torch.float32).eval()
    x = case.image().to("cuda", torch.float32)
    halo = receptive_field(case.config)
    tile = (64, 96)
    assert x.shape[-2] > tile[0] or x.shape[-1] > tile[1], "the image must really be cut"
    up = upscale_tiled(m, x, tile=tile)
    err = case.compare_sr(up, up)["up"]     # the clamped result against the reference's upscale()
    assert err <= 1e-3, f"{name}: tiled fp32 upscale deviates from the reference fixture by {err:.3e} (halo {halo})"
