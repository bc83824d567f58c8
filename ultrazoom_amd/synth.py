"""Deterministic synthetic weights and images.

No pretrained weights or images ship with the reference ("Empty data dirs"),
so tests, golden fixtures and the benchmark all draw from one integer-hash generator that gives
the same numbers on every machine and library version (it never touches a library RNG).
"""

from __future__ import annotations

import zlib
from math import sqrt
from typing import Dict, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised over a uint64 array."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(numel: int, seed: int, chunk: int = 1 << 24) -> np.ndarray:
    """`numel` float32 values in [0, 1), element i depends only on (seed, i)."""
    out = np.empty(numel, dtype=np.float32)
    base = _mix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    for start in range(0, numel, chunk):
        stop = min(numel, start + chunk)
        idx = np.arange(start, stop, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = _mix64(idx * np.uint64(0x9E3779B97F4A7C15) + base)
        # 24 high bits -> exactly representable float32 in [0, 1)
        out[start:stop] = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return out


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) << 20) ^ (seed * 0x51ED2701)


def synth_image(B: int, H: int, W: int, seed: int = 0, smooth: bool = True) -> torch.Tensor:
    """A (B, 3, H, W) float32 image batch in [0, 1].

    `smooth=True` mixes low-frequency structure with noise so that the image looks more like a
    photograph than like white noise (bicubic overshoot and the clamp both get exercised).
    """
    noise = hash_uniform(B * 3 * H * W, _name_seed("image", seed)).reshape(B, 3, H, W)
    if not smooth:
        return torch.from_numpy(noise)
    yy = np.arange(H, dtype=np.float32)[:, None] / max(H, 1)
    xx = np.arange(W, dtype=np.float32)[None, :] / max(W, 1)
    img = np.empty((B, 3, H, W), dtype=np.float32)
    for b in range(B):
        for c in range(3):
            f1, f2 = 3.0 + c + b, 5.0 - c + 0.5 * b
            base = 0.5 + 0.35 * np.sin(6.2831853 * (f1 * xx + 0.3 * c)) * np.cos(6.2831853 * f2 * yy)
            img[b, c] = base
    img = 0.75 * img + 0.25 * noise
    # hard edges: a bright and a dark rectangle so bicubic overshoots past [0, 1]
    img[:, :, H // 4 : H // 2, W // 4 : W // 2] = 1.0
    img[:, :, H // 2 : (3 * H) // 4, W // 2 : (3 * W) // 4] = 0.0
    return torch.from_numpy(np.clip(img, 0.0, 1.0).astype(np.float32))


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Hash-initialised float32 parameters for every (name, shape) in `shapes`.

    Conv weights are U(-b, b) with b = sqrt(3 / fan_in) (unit-gain variance-preserving), biases
    U(-0.1, 0.1) and the scalar mixing gates `alpha` U(-1, 1) so that sigmoid(alpha) != 0.5.
    """
    out: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        u = hash_uniform(n, _name_seed(name, seed))
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = sqrt(3.0 / fan_in)
            v = (2.0 * u - 1.0) * bound
        elif len(shape) == 1:
            v = (2.0 * u - 1.0) * 0.1
        else:
            v = 2.0 * u - 1.0
        out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out
