"""CPU oracle for the MewZoom upscale path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch CPU tensor ops, fp32 or fp64) of the
forward pass of the reference's ``MewZoom`` model.  It exists so the HIP path can be checked
against something that runs without a GPU.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package ``ultrazoom_amd`` never
does (``tests/test_layout.py`` enforces that).

Parity status: PINNED.  ``tests/golden/make_golden.py`` ran the reference's own
``src/ultrazoom/model.py`` in the build container and stored its outputs as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against every one of them.

The functions take a flat ``dict[str, Tensor]`` keyed exactly like the reference's
``state_dict()`` plus the 11 constructor kwargs, so any checkpoint the
reference can load can be fed here unchanged.

Reference citations are ``src/ultrazoom/model.py:<line>`` in andrewdalpino/UltraZoom v0.3.0.
"""

from __future__ import annotations

from math import ceil, floor, log2
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

CONFIG_KEYS = (
    "upscale_ratio",
    "primary_channels",
    "primary_layers",
    "secondary_channels",
    "secondary_layers",
    "tertiary_channels",
    "tertiary_layers",
    "quaternary_channels",
    "quaternary_layers",
    "hidden_ratio",
    "num_deg_features",
)

BICUBIC_A = -0.75


# --------------------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------------------


def validate_config(cfg: dict) -> None:
    """Constructor-time checks; the reference raises AssertionError for each of these."""
    # model.py:67-69
    assert cfg["upscale_ratio"] in {2, 4, 8}, "Upscale ratio must be one of {2, 4, 8}."
    # model.py:218-222 (FanOutProjection: 3 < primary_channels)
    assert cfg["primary_channels"] > 3, "Output channels must be greater than input channels."
    # model.py:265-275
    for name in ("primary", "secondary", "tertiary", "quaternary"):
        assert cfg[f"{name}_layers"] > 1, f"Number of {name} layers must be greater than 1."
        assert cfg[f"{name}_channels"] > 0, "Number of channels must be greater than 0."
    # model.py:738
    assert cfg["hidden_ratio"] in {1, 2, 4}, "Hidden ratio must be either 1, 2, or 4."
    # model.py:356-358 (as intended by the author: the QA head needs at least one feature)
    assert cfg["num_deg_features"] > 0, "Number of quality assessor features must be greater than 0."


def stage_plan(cfg: dict) -> Tuple[List[int], List[int], List[int]]:
    """Channels per level and encoder/decoder block counts per level.

    model.py:277-300: the encoder gets ceil(L/2) blocks of a level, the decoder floor(L/2).
    Index 0 is the full-resolution (primary) level, index 3 the coarsest (quaternary).
    """
    names = ("primary", "secondary", "tertiary", "quaternary")
    channels = [cfg[f"{n}_channels"] for n in names]
    enc = [ceil(cfg[f"{n}_layers"] / 2) for n in names]
    dec = [floor(cfg[f"{n}_layers"] / 2) for n in names]
    return channels, enc, dec


def parameter_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """Every state_dict key of the reference model and its shape."""
    validate_config(cfg)
    ch, enc, dec = stage_plan(cfg)
    hr = cfg["hidden_ratio"]
    shapes: Dict[str, Tuple[int, ...]] = {}

    def block(prefix: str, c: int) -> None:
        shapes[f"{prefix}.convnet.conv1.weight"] = (hr * c, c, 3, 3)  # model.py:742-744
        shapes[f"{prefix}.convnet.conv2.weight"] = (c, hr * c, 3, 3)  # model.py:746-748
        shapes[f"{prefix}.skip.alpha"] = ()  # model.py:807 (a module's own parameters precede its children's)
        shapes[f"{prefix}.skip.conv.weight"] = (c, 2 * c, 1, 1)  # model.py:805

    c0 = ch[0]
    shapes["stem.conv.weight"] = (c0, 3, 1, 1)  # model.py:224
    shapes["stem.conv.bias"] = (c0,)

    for s in range(4):
        for i in range(enc[s]):
            block(f"unet.encoder.stage{s + 1}.{i}", ch[s])  # model.py:360-386
    for s in range(3):
        shapes[f"unet.encoder.downsample{s + 1}.conv.weight"] = (ch[s + 1], ch[s], 2, 2)  # :388-390
    shapes["unet.encoder.qa_head.conv.weight"] = (cfg["num_deg_features"], ch[3], 3, 3)  # :1010
    shapes["unet.encoder.qa_head.conv.bias"] = (cfg["num_deg_features"],)

    # Decoder stage1 is the coarsest level: model.py:290-300 passes quaternary first.
    for d in range(4):
        lvl = 3 - d
        for i in range(dec[lvl]):
            block(f"unet.decoder.stage{d + 1}.{i}", ch[lvl])  # model.py:541-567
    for d in range(3):
        cin, cout = ch[3 - d], ch[2 - d]
        shapes[f"unet.decoder.upsample{d + 1}.conv.weight"] = (4 * cout, cin, 3, 3)  # :569-571,900-909
    for d in range(3):
        cout = ch[2 - d]
        shapes[f"unet.decoder.skip{d + 1}.alpha"] = ()
        shapes[f"unet.decoder.skip{d + 1}.conv.weight"] = (cout, 2 * cout, 1, 1)  # :573-575

    n_head = int(log2(cfg["upscale_ratio"]))  # model.py:945
    for i in range(n_head):
        block(f"head.layers.{i}.refiner", c0)  # model.py:981
        cout = 3 if i == n_head - 1 else c0  # model.py:947-954
        shapes[f"head.layers.{i}.upscale.conv.weight"] = (4 * cout, c0, 3, 3)  # model.py:983
    return shapes


# --------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------


def cubic_weights(t: float, a: float = BICUBIC_A) -> List[float]:
    """The four cubic-convolution tap weights for fractional offset t."""

    def near(u: float) -> float:  # |u| <= 1
        return ((a + 2.0) * u - (a + 3.0)) * u * u + 1.0

    def far(u: float) -> float:  # 1 < |u| < 2
        return ((a * u - 5.0 * a) * u + 8.0 * a) * u - 4.0 * a

    return [far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)]


def bicubic_upsample(x: Tensor, r: int) -> Tensor:
    """``Upsample(scale_factor=r, mode="bicubic")`` (model.py:71,156) written as a polyphase filter.

    Output index d = r*k + p uses source coordinate (p + 0.5)/r - 0.5 + k; taps sit at
    k + f - 1 .. k + f + 2 with f = floor of the phase offset, indices clamped to the image.
    Implemented separably (rows then columns) with gathers, in the dtype of ``x``.
    """
    B, C, H, W = x.shape

    def taps(n: int):
        d = torch.arange(n * r)
        k = d // r
        p = d % r
        idx = torch.empty(n * r, 4, dtype=torch.long)
        wgt = torch.empty(n * r, 4, dtype=x.dtype)
        for phase in range(r):
            src = (phase + 0.5) / r - 0.5
            f = floor(src)
            t = src - f
            w = cubic_weights(t)
            sel = p == phase
            for i in range(4):
                idx[sel, i] = (k[sel] + f - 1 + i).clamp(0, n - 1)
                wgt[sel, i] = w[i]
        return idx, wgt

    iy, wy = taps(H)
    ix, wx = taps(W)
    # columns (width) first, then rows
    xw = x[:, :, :, ix]  # B,C,H,rW,4
    xw = (xw * wx).sum(-1)
    xh = xw[:, :, iy, :]  # B,C,rH,4,rW
    out = (xh * wy[:, :, None]).sum(-2)
    return out


def _rq(t: Tensor, storage) -> Tensor:
    """Round to the 16-bit storage type and back (identity when ``storage`` is None)."""
    return t if storage is None else t.to(storage).to(t.dtype)


def residual_mix(x: Tensor, z: Tensor, w_mix: Tensor, alpha: Tensor, storage=None) -> Tensor:
    """AdaptiveResidualMix.forward, model.py:826-839."""
    beta = torch.sigmoid(F.conv2d(torch.cat([x, z], dim=1), w_mix))
    w = torch.sigmoid(alpha) * beta
    return _rq((1 - w) * x + w * z, storage)


def res_block(x: Tensor, p: Dict[str, Tensor], prefix: str, storage=None) -> Tensor:
    """EncoderBlock / DecoderBlock forward, model.py:507-511 + 773-778."""
    h = _rq(F.silu(F.conv2d(x, p[f"{prefix}.convnet.conv1.weight"], padding=1)), storage)
    z = _rq(F.conv2d(h, p[f"{prefix}.convnet.conv2.weight"], padding=1), storage)
    return residual_mix(x, z, p[f"{prefix}.skip.conv.weight"], p[f"{prefix}.skip.alpha"], storage)


def subpixel_conv(x: Tensor, w: Tensor) -> Tensor:
    """SubpixelConv2d.forward with ratio 2, model.py:926-930."""
    return F.pixel_shuffle(F.conv2d(x, w, padding=1), 2)


def fit_to(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """Decoder.crop_feature_maps, model.py:650-689: centre-crop or zero-pad (extra on bottom/right)."""
    h, w = x.shape[2:]
    th, tw = int(size[0]), int(size[1])
    if h > th:
        s = (h - th) // 2
        x = x[:, :, s : s + th, :]
    elif h < th:
        top = (th - h) // 2
        x = F.pad(x, (0, 0, top, th - h - top))
    if w > tw:
        s = (w - tw) // 2
        x = x[:, :, :, s : s + tw]
    elif w < tw:
        left = (tw - w) // 2
        x = F.pad(x, (left, tw - w - left, 0, 0))
    return x


def film_conv(x: Tensor, w: Tensor, gamma: Tensor, beta: Tensor, silu: bool = False) -> Tensor:
    """NO REFERENCE COUNTERPART: the v0.3.0 snapshot has no ControlModule / FiLM.  The build's own
    statement of the per-channel modulation its optional epilogue computes, for the GPU test of that epilogue only:
    act(gamma[b, c] * conv3x3(x, w)[b, c] + beta[b, c]) with gamma, beta of shape [B, C]."""
    y = F.conv2d(x, w, padding=1) * gamma[:, :, None, None] + beta[:, :, None, None]
    return F.silu(y) if silu else y


def quality_head(z4: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """QualityAssessor.forward, model.py:1026-1032: conv3x3 + bias, global spatial mean."""
    return F.conv2d(z4, w, b, padding=1).mean(dim=(2, 3))


# --------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------


def forward(cfg: dict, params: Dict[str, Tensor], x: Tensor, taps: dict | None = None, storage=None):
    """MewZoom.forward, model.py:149-164: returns (sr_unclamped, z_qa).

    ``taps`` (optional dict) receives intermediate tensors for layer-by-layer checks.

    ``storage`` (None, torch.bfloat16 or torch.float16) selects the ROUNDING-MATCHED variant used to check the
    16-bit GPU kernels tightly: the arithmetic stays fp32 (as the MFMA accumulators are), but the image, every
    parameter and every tensor the GPU path keeps in HBM (stem output, SiLU(conv1), conv2, each mix, PixelCrush,
    each sub-pixel convolution, the QA convolution, the final image) is rounded to ``storage`` where the GPU path
    rounds it.  With ``storage=None`` this is the plain restatement of the reference that the fixtures pin; the
    rounded variant differs from it only by those explicit roundings.
    """
    validate_config(cfg)
    assert x.dim() == 4 and x.shape[1] == 3, "expected a (B, 3, H, W) tensor"
    ch, enc, dec = stage_plan(cfg)
    r = cfg["upscale_ratio"]
    p = {k: _rq(v.to(x.dtype), storage) for k, v in params.items()}
    x = _rq(x, storage)

    def keep(name: str, t: Tensor) -> None:
        if taps is not None:
            taps[name] = t

    s = bicubic_upsample(x, r)  # model.py:156
    keep("bicubic", s)
    z = _rq(F.conv2d(x, p["stem.conv.weight"], p["stem.conv.bias"]), storage)  # model.py:158, 239-242
    keep("stem", z)

    # Encoder, model.py:461-484
    feats = []
    for lvl in range(4):
        if lvl > 0:
            z = _rq(F.conv2d(z, p[f"unet.encoder.downsample{lvl}.conv.weight"], stride=2), storage)  # model.py:881-882
        for i in range(enc[lvl]):
            z = res_block(z, p, f"unet.encoder.stage{lvl + 1}.{i}", storage)
        feats.append(z)
        keep(f"enc{lvl + 1}", z)
    if storage is None:
        z_qa = quality_head(feats[3], p["unet.encoder.qa_head.conv.weight"], p["unet.encoder.qa_head.conv.bias"])
    else:  # the GPU path stores the bias-free convolution, then reduces in fp32 and adds the bias
        z_qa = _rq(F.conv2d(feats[3], p["unet.encoder.qa_head.conv.weight"], padding=1), storage).mean(dim=(2, 3))
        z_qa = _rq(z_qa + p["unet.encoder.qa_head.conv.bias"], storage)

    # Decoder, model.py:691-724
    z = feats[3]
    for d in range(4):
        lvl = 3 - d
        if d > 0:
            z = _rq(subpixel_conv(z, p[f"unet.decoder.upsample{d}.conv.weight"]), storage)
            z = fit_to(z, feats[lvl].shape[2:])
            z = residual_mix(
                feats[lvl], z, p[f"unet.decoder.skip{d}.conv.weight"], p[f"unet.decoder.skip{d}.alpha"], storage
            )
        for i in range(dec[lvl]):
            z = res_block(z, p, f"unet.decoder.stage{d + 1}.{i}", storage)
    keep("unet", z)

    # Head, model.py:968-972 / 997-1001
    n_head = int(log2(r))
    for i in range(n_head):
        z = res_block(z, p, f"head.layers.{i}.refiner", storage)
        z = subpixel_conv(z, p[f"head.layers.{i}.upscale.conv.weight"])
        if i + 1 < n_head:  # the last sub-pixel convolution stays in fp32 registers until the image is stored
            z = _rq(z, storage)
    keep("head", z)

    assert s.shape == z.shape, "Input and residual must have the same shape."  # model.py:790
    return _rq(s + z, storage), z_qa  # model.py:162-164 (rounding commutes with the clamp: 0 and 1 are representable)


def upscale(cfg: dict, params: Dict[str, Tensor], x: Tensor, storage=None) -> Tensor:
    """MewZoom.upscale, model.py:166-179."""
    with torch.inference_mode():
        sr, _ = forward(cfg, params, x, storage=storage)
        return torch.clamp(sr, 0, 1)


def predict_degredation(cfg: dict, params: Dict[str, Tensor], x: Tensor) -> Tensor:
    """MewZoom.predict_degredation (sic), model.py:181-192."""
    with torch.inference_mode():
        return forward(cfg, params, x)[1]


def flops_per_image(cfg: dict, H: int, W: int) -> int:
    """Algorithmic FLOPs (2 x conv MACs, nothing else) of one forward on an H x W input.

    SURVEY.md section 8(d); exact, including floor effects of odd sizes.
    """
    ch, enc, dec = stage_plan(cfg)
    hr = cfg["hidden_ratio"]
    F_ = cfg["num_deg_features"]
    sizes = [(H, W)]
    for _ in range(3):
        h, w = sizes[-1]
        sizes.append((h // 2, w // 2))
    macs = 0
    macs += 3 * ch[0] * H * W
    blk = lambda c: (18 * hr + 2) * c * c
    for lvl in range(4):
        h, w = sizes[lvl]
        macs += (enc[lvl] + dec[lvl]) * blk(ch[lvl]) * h * w
    for lvl in range(3):
        h, w = sizes[lvl + 1]
        macs += 4 * ch[lvl] * ch[lvl + 1] * h * w  # PixelCrush
        macs += 9 * ch[lvl + 1] * 4 * ch[lvl] * h * w  # decoder sub-pixel conv (runs at the coarse size)
        he, we = sizes[lvl]
        macs += 2 * ch[lvl] * ch[lvl] * he * we  # decoder skip mix
    h, w = sizes[3]
    macs += 9 * ch[3] * F_ * h * w
    n_head = int(log2(cfg["upscale_ratio"]))
    h, w = H, W
    for i in range(n_head):
        cout = 3 if i == n_head - 1 else ch[0]
        macs += blk(ch[0]) * h * w + 9 * ch[0] * 4 * cout * h * w
        h, w = 2 * h, 2 * w
    return 2 * macs
