"""CPU-side checks of the C ABI: the library loads without a GPU, exports every declared symbol,
validates configurations like the reference, and describes the model's parameters correctly."""

import ctypes
import json
import re
from pathlib import Path

import pytest

from golden_util import GOLDEN, MODEL_CASES, GoldenCase
from oracle import mewzoom_oracle as oracle
from ultrazoom_amd import _ffi

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (REPO / "include" / "mewzoom_hip.h").read_text()
    declared = set(re.findall(r"\b(mz_[a-z0-9_]+)\s*\(", header))
    declared -= {"mz_handle", "mz_config", "mz_dtype", "mz_status"}
    assert len(declared) >= 15
    lib = ctypes.CDLL(str(_ffi.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mewzoom_hip.h but not exported"


def test_create_rejects_what_the_reference_rejects():
    trials = json.loads((GOLDEN / "validation.json").read_text())
    for name, t in trials.items():
        if t["raises"] is None:
            _ffi.Handle(t["kwargs"], _ffi.MZ_F32).close()
        else:
            with pytest.raises(_ffi.MewZoomHipError) as ei:
                _ffi.Handle(t["kwargs"], _ffi.MZ_F32)
            assert ei.value.code == _ffi.MZ_ERR_INVALID_ARGUMENT, name


@pytest.mark.parametrize("name", MODEL_CASES)
def test_weight_registry_matches_reference_state_dict(name):
    case = GoldenCase(name)
    for dt in (_ffi.MZ_F32, _ffi.MZ_BF16):
        h = _ffi.Handle(case.config, dt)
        infos = h.weight_infos()
        assert [k for k, _ in infos] == list(case.meta["shapes"])
        assert {k: list(v) for k, v in infos} == case.meta["shapes"]
        h.close()


def test_flops_and_workspace():
    case = GoldenCase("g7_cfg1_2x_c48")
    h = _ffi.Handle(case.config, _ffi.MZ_BF16)
    assert abs(h.flops_per_image(256, 256) - oracle.flops_per_image(case.config, 256, 256)) < 1.0
    assert abs(h.flops_per_image(540, 960) - oracle.flops_per_image(case.config, 540, 960)) < 1.0
    w1 = h.workspace_bytes(1, 256, 256)
    w4 = h.workspace_bytes(4, 256, 256, 4)
    w4_1 = h.workspace_bytes(4, 256, 256, 1)
    assert 0 < w1 == w4_1 < w4 <= 4 * w1 + 65536
    with pytest.raises(_ffi.MewZoomHipError):
        h.workspace_bytes(1, 4, 4)
    h.close()


def test_unknown_weight_and_missing_weights_fail_loudly():
    case = GoldenCase("g1_2x_c16")
    h = _ffi.Handle(case.config, _ffi.MZ_F32)
    with pytest.raises(_ffi.MewZoomHipError) as ei:
        h.weights_complete()
    assert "has not been set" in str(ei.value)
    h.close()


def test_weights_cannot_be_set_without_a_gpu():
    """No silent fallback: on a machine without an MI355X the library refuses instead of computing somewhere else."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less machines")
    case = GoldenCase("g1_2x_c16")
    h = _ffi.Handle(case.config, _ffi.MZ_F32)
    with pytest.raises(_ffi.MewZoomHipError) as ei:
        h.set_weight("stem.conv.bias", 0x1000, (16,), 0)
    assert ei.value.code in (-7, -6)  # MZ_ERR_NO_DEVICE / MZ_ERR_HIP
    with pytest.raises(_ffi.MewZoomHipError) as ei:
        h.set_weight("no.such.parameter", 0x1000, (16,), 0)
    assert ei.value.code == -2
    with pytest.raises(_ffi.MewZoomHipError) as ei:
        h.set_weight("stem.conv.bias", 0x1000, (17,), 0)
    assert ei.value.code == -3
    h.close()


@pytest.mark.parametrize(
    "geom",
    [
        # B, tiles_y, tiles_x, ntiles, gm, gn, blk4, th, tw
        (3, 135, 40, 2, 16, 2, 1, 8, 48),   # the headline's 1080 x 1920 level: 96 -> 192
        (3, 17, 5, 8, 4, 8, 1, 8, 48),      # 135 x 240, 1536 -> 768: more N tiles than a group is wide
        (3, 17, 6, 16, 2, 16, 0, 8, 40),    # the 8 x 40 geometry, row-major walk
        (2, 7, 3, 3, 5, 2, 1, 8, 48),       # partial groups in both directions, last block row of three tile rows
        (1, 1, 1, 1, 32, 1, 1, 12, 64),     # one tile
        (32, 45, 15, 1, 32, 1, 1, 12, 64),  # conv3t on BASELINE configs[1]'s level 1
    ],
)
def test_tile_list_lists_every_tile_once(geom):
    """The tile list the role-alternating kernels walk (mz_host.cpp: tile_list(), uploaded by Runner::tile_table()): every (image, tile row,
    tile column, N tile) exactly once, tile origins on the tile grid, and the N tiles of a group's pixel tile next to each other."""
    B, ty, tx, ntiles, gm, gn, blk4, th, tw = geom
    lib = ctypes.CDLL(str(_ffi.LIB_PATH))
    lib.mz_debug_tile_list.restype = ctypes.c_int
    total = B * ty * tx * ntiles
    buf = (ctypes.c_uint * (2 * total))()
    n = lib.mz_debug_tile_list(B, ty, tx, ntiles, gm, gn, blk4, th, tw, buf, total)
    assert n == total
    seen = set()
    for i in range(n):
        yx, bn = buf[2 * i], buf[2 * i + 1]
        y0, x0, b, nt = yx & 0xFFFF, yx >> 16, bn & 0xFFFF, bn >> 16
        assert y0 % th == 0 and x0 % tw == 0 and y0 // th < ty and x0 // tw < tx and b < B and nt < ntiles
        seen.add((b, y0, x0, nt))
    assert len(seen) == total
    if gn >= 2 and ntiles >= 2:  # consecutive entries share the pixel tile inside a group's row of N tiles
        same = sum(1 for i in range(n - 1) if buf[2 * i] == buf[2 * i + 2] and (buf[2 * i + 1] & 0xFFFF) == (buf[2 * i + 3] & 0xFFFF))
        assert same >= n // 3
    assert lib.mz_debug_tile_list(0, ty, tx, ntiles, gm, gn, blk4, th, tw, buf, total) < 0
