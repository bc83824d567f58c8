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


def _really_cut(H, W, tile, halo):
    """True when some tile is cut on BOTH sides of BOTH axes: the second tile of an axis starts at `tile`, so its slice
    [tile - halo, 2 tile + halo) must lie strictly inside the image -- else "tile + halo" reaches a true border (or is simply the
    whole axis) and the cut is not tested."""
    return all(t > halo and n > 2 * t + halo for t, n in zip(tile, (H, W)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g2_odd_135x240", "g7_cfg1_2x_c48"])
def test_tiled_upscale_against_the_reference_fixtures(name):
    """N4 against the REFERENCE's own outputs (not against the untiled HIP result): `upscale_tiled` of the fixture's input in fp32
    must meet the fixture within the north_star tolerance (1e-3 max-abs).  g2: odd sizes (floors and zero pads at every level);
    g7: the BASELINE configs[0] model (48 channels / 20 layers) on 256 x 256, sampled fixture.  The fixtures are SMALLER than a tile plus
    two receptive-field halos, so a cut removes real pixels on at most one side of an axis (g7: every "tile" is the whole image): these
    cases pin the slicing arithmetic to the reference; cuts on both sides of both axes are `test_tiled_two_axis_cuts_against_the_oracle`."""
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


@pytest.mark.gpu
def test_tiled_two_axis_cuts_against_the_oracle():
    """N4 with REAL cuts on both axes, against the pinned CPU oracle (not against the untiled HIP result): the BASELINE configs[0] model
    (2X, 48 / 96 / 192 / 384 channels, 4 / 4 / 4 / 8 layers: receptive field 224 px) on 1 x 3 x 744 x 1000 in tiles of 256 x 384 with
    the full halo.  744 > 2 * 256 + 224 and 1000 > 2 * 384 + 224: the middle tile row / column is cut on both sides (its slice is
    704 x 832, strictly inside the image), 3 x 3 tiles, none of them sees the whole image on either axis.  fp32 HIP path vs oracle.upscale of the whole image: <= 1e-3 max-abs (north_star)."""
    from oracle import mewzoom_oracle as oracle

    case = GoldenCase("g7_cfg1_2x_c48")  # its weights: the hash initialiser with the fixture's seed
    cfg, sd = case.config, case.weights()
    H, W, tile = 744, 1000, (256, 384)
    halo = receptive_field(cfg)
    assert _really_cut(H, W, tile, halo), (H, W, tile, halo)
    calls = []
    m = MewZoom(**cfg)
    m.load_state_dict(sd)
    m = m.to("cuda", torch.float32).eval()

    class Spy:  # records the slices upscale_tiled() really hands to the model
        _cfg = m._cfg

        def upscale(self, t):
            calls.append(tuple(t.shape[-2:]))
            return m.upscale(t)

    x = synth_image(1, H, W, seed=31)
    up = upscale_tiled(Spy(), x.to("cuda", torch.float32), tile=tile).float().cpu()
    assert len(calls) == 9 and all(h < H and w < W for h, w in calls), calls  # no slice spans an axis
    assert (tile[0] + 2 * halo, tile[1] + 2 * halo) in calls                    # the centre tile carries four cut edges
    with torch.inference_mode():
        want = oracle.upscale(cfg, sd, x)
    err = (up - want).abs().max().item()
    assert err <= 1e-3, f"two-axis tiled fp32 upscale deviates from the oracle by {err:.3e}"
    assert torch.equal(up, m.upscale(x.to("cuda", torch.float32)).float().cpu()), "tiled must equal untiled bit for bit"


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("no_r", [False, True])
def test_tiled_equals_untiled_96_channels_hidden_ratio_2(dt, no_r, monkeypatch):
    """The headline model's channel width: C = 96 with hidden_ratio 2 runs the fused conv2 + mix on conv3r_kernel (or, MZ_NO_R=1, on
    conv3s_kernel) -- two kernels that agree to <= 1 ulp only.  Which of them runs must not depend on the size of the tensor, or a
    100-pixel-wide tile would differ from the same pixels inside a 1920-wide image: tiled == untiled bit for bit, in both settings.
    (Slices of 200 / 296 / 224 columns against a 328-column image: where the 8 x 48 tiles of conv3r pad fewer pixels than conv3s's
    8 x 64 tiles differs from slice to slice -- 240 against 256, 336 against 320.)"""
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[dt]
    if no_r:
        monkeypatch.setenv("MZ_NO_R", "1")
    else:
        monkeypatch.delenv("MZ_NO_R", raising=False)
    cfg = dict(upscale_ratio=2, primary_channels=96, primary_layers=2, secondary_channels=96, secondary_layers=2,
               tertiary_channels=96, tertiary_layers=2, quaternary_channels=96, quaternary_layers=2, hidden_ratio=2, num_deg_features=3)
    from ultrazoom_amd.synth import synth_state_dict

    m = MewZoom(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd)
    m = m.to("cuda", dtype).eval()
    halo = receptive_field(cfg)
    H, W, tile = 2 * 104 + halo + 16, 2 * 104 + halo + 24, (104, 104)
    assert _really_cut(H, W, tile, halo)
    x = synth_image(1, H, W, seed=32).to("cuda", dtype)
    assert torch.equal(upscale_tiled(m, x, tile=tile), m.upscale(x))
