"""Exact image-level tiling for inputs whose activations do not fit in HBM (BASELINE configs[4] is the
"tiled single-image" case).  The reference has no tiling code: this is host-side slicing around `MewZoom.upscale`.

Why it is exact.  Every output pixel depends on a bounded window of the input (3x3 convolutions at four resolutions,
2x2 stride-2 PixelCrush, PixelShuffle, the 4-tap bicubic skip): `receptive_field(config)` low-resolution pixels on
each side.  A tile is upscaled together with a halo of at least that many pixels and only its core is kept, so what
the zero padding / index clamping at an artificial cut changes never reaches a kept pixel.  Tile origins are
multiples of 8 so that the three stride-2 levels, their floors at odd sizes and the decoder's bottom/right zero
padding (model.py:650-689) fall on the same pixels as in the whole image; cuts at the true image border keep the
true border.  The kernels accumulate in an order that does not depend on where a pixel sits in its tensor, so
tiled and untiled results agree bit for bit, and the tiled fp32 result meets the reference's fixtures (tests/test_tiling.py)."""

from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor


def receptive_field(config: dict) -> int:
    """Upper bound of the one-sided receptive field of `upscale` in low-resolution pixels, rounded up to a multiple
    of 8 (block counts per stage as model.py:277-300: ceil(L/2) encoder + floor(L/2) decoder blocks of two 3x3
    convolutions each)."""
    layers = [config[f"{n}_layers"] for n in ("primary", "secondary", "tertiary", "quaternary")]
    r = 0.0
    for s, L in enumerate(layers):
        blocks = math.ceil(L / 2) + L // 2
        r += 2 * blocks * 2**s          # two 3x3 convolutions per block, one pixel of that level each
    r += sum(2**s for s in (1, 2, 3))   # the 3x3 of each decoder sub-pixel convolution (runs at the coarser level)
    r += sum(2 ** (s - 1) for s in (1, 2, 3))  # PixelCrush 2x2 / stride 2: up to one pixel of the finer level
    for i in range(int(math.log2(config["upscale_ratio"]))):
        r += 3 * 2.0**-i                # head stage i: refiner block (2 convs) + sub-pixel conv at 2^i x resolution
    r += 2                              # bicubic skip: 4 taps
    return int(math.ceil(r / 8.0)) * 8 + 8


@torch.inference_mode()
def upscale_tiled(model, x: Tensor, tile: Tuple[int, int] = (512, 512), halo: Optional[int] = None) -> Tensor:
    """`model.upscale(x)` computed tile by tile.  `tile` = core size in low-resolution pixels (rounded up to multiples
    of 8); `halo` defaults to `receptive_field(config)`; a smaller halo is refused, because the result would no
    longer equal the untiled one."""
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("expected a [B, 3, H, W] tensor")
    cfg = model._cfg
    need = receptive_field(cfg)
    halo = need if halo is None else int(halo)
    if halo < need or halo % 8:
        raise ValueError(f"halo must be a multiple of 8 and at least the receptive field ({need} pixels), got {halo}")
    th, tw = (max(8, (int(t) + 7) // 8 * 8) for t in tile)
    B, _, H, W = x.shape
    r = cfg["upscale_ratio"]
    out = torch.empty((B, 3, H * r, W * r), dtype=x.dtype, device=x.device)
    for y0 in range(0, H, th):
        y1 = min(H, y0 + th)
        ya, yb = max(0, y0 - halo), min(H, y1 + halo)
        for x0 in range(0, W, tw):
            x1 = min(W, x0 + tw)
            xa, xb = max(0, x0 - halo), min(W, x1 + halo)
            sr = model.upscale(x[:, :, ya:yb, xa:xb].contiguous())
            out[:, :, y0 * r : y1 * r, x0 * r : x1 * r] = sr[:, :, (y0 - ya) * r : (y1 - ya) * r, (x0 - xa) * r : (x1 - xa) * r]
    return out
