"""ctypes binding of libmewzoom_hip.so (include/mewzoom_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, an exception
is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or
``ultrazoom_amd/csrc/build.sh``).
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

LIB_NAME = "libmewzoom_hip.so"
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME

MZ_F32, MZ_BF16, MZ_F16 = 0, 1, 2

MZ_ERR_INVALID_ARGUMENT = -1


class MzConfig(Structure):
    _fields_ = [
        ("upscale_ratio", c_int32),
        ("primary_channels", c_int32),
        ("primary_layers", c_int32),
        ("secondary_channels", c_int32),
        ("secondary_layers", c_int32),
        ("tertiary_channels", c_int32),
        ("tertiary_layers", c_int32),
        ("quaternary_channels", c_int32),
        ("quaternary_layers", c_int32),
        ("hidden_ratio", c_int32),
        ("num_deg_features", c_int32),
    ]


class MewZoomHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmewzoom_hip error {code}: {message}")
        self.code = code


_lib = None


def _declare(lib) -> None:
    H = c_void_p
    lib.mz_create.argtypes = [POINTER(MzConfig), c_int, POINTER(H)]
    lib.mz_destroy.argtypes = [H]
    lib.mz_num_weights.argtypes = [H]
    lib.mz_weight_info.argtypes = [H, c_int, POINTER(c_char_p), POINTER(c_int64)]
    lib.mz_set_weight.argtypes = [H, c_char_p, c_void_p, POINTER(c_int64), c_int, c_void_p]
    lib.mz_weights_complete.argtypes = [H]
    lib.mz_workspace_bytes.argtypes = [H, c_int, c_int, c_int, c_int, POINTER(c_size_t)]
    lib.mz_forward.argtypes = [H, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]
    lib.mz_forward_u8.argtypes = [H, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]
    lib.mz_forward_u8.restype = c_int
    lib.mz_padded_channels.argtypes = [c_int]
    lib.mz_op_conv.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p] + [c_int] * 8 + [c_void_p]
    lib.mz_op_conv_film.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]
    lib.mz_op_conv_film.restype = c_int
    lib.mz_op_conv_mix.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p] + [c_int] * 5 + [c_void_p]
    lib.mz_op_conv_mix.restype = c_int
    lib.mz_op_stem.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.mz_op_final.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]
    lib.mz_last_error.restype = c_char_p
    lib.mz_version.restype = c_char_p
    lib.mz_flops_per_image.argtypes = [H, c_int, c_int]
    lib.mz_flops_per_image.restype = c_double
    lib.mz_profile_enable.argtypes = [H, c_int]
    lib.mz_profile_read.argtypes = [H] + [POINTER(c_double)] * 5
    lib.mz_profile_dump.argtypes = [H, c_char_p]
    lib.mz_profile_dump.restype = c_int
    for name in (
        "mz_create mz_destroy mz_num_weights mz_weight_info mz_set_weight mz_weights_complete mz_workspace_bytes "
        "mz_forward mz_padded_channels mz_op_conv mz_op_stem mz_op_final mz_profile_enable mz_profile_read"
    ).split():
        getattr(lib, name).restype = c_int


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("MEWZOOM_HIP_LIB", LIB_PATH))
        if not path.exists():
            raise ImportError(
                f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` from the repository root. There is no fallback path."
            )
        _lib = ctypes.CDLL(str(path))
        _declare(_lib)
    return _lib


def check(code: int) -> None:
    if code != 0:
        raise MewZoomHipError(code, lib().mz_last_error().decode())


def dtype_code(torch_dtype) -> int:
    import torch

    table = {torch.float32: MZ_F32, torch.bfloat16: MZ_BF16, torch.float16: MZ_F16}
    if torch_dtype not in table:
        raise TypeError(f"unsupported dtype {torch_dtype}; use float32, bfloat16 or float16")
    return table[torch_dtype]


def make_config(cfg: dict) -> MzConfig:
    return MzConfig(**{k: int(cfg[k]) for k, _ in MzConfig._fields_})


class Handle:
    """Owns one mz_handle."""

    def __init__(self, cfg: dict, dtype_code_: int):
        self._h = c_void_p()
        check(lib().mz_create(byref(make_config(cfg)), dtype_code_, byref(self._h)))
        self.dtype_code = dtype_code_

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value and _lib is not None:
            _lib.mz_destroy(self._h)
            self._h = c_void_p()

    __del__ = close

    # A handle owns device memory through a raw pointer: copying the Python object would free it twice.
    def __copy__(self):
        raise TypeError("an mz_handle cannot be copied; build a new Handle")

    def __deepcopy__(self, memo):
        raise TypeError("an mz_handle cannot be copied; build a new Handle")

    def __reduce__(self):
        raise TypeError("an mz_handle cannot be pickled; build a new Handle")

    @property
    def ptr(self):
        return self._h

    def weight_infos(self):
        n = lib().mz_num_weights(self._h)
        out = []
        for i in range(n):
            name = c_char_p()
            shape = (c_int64 * 4)()
            ndim = lib().mz_weight_info(self._h, i, byref(name), shape)
            if ndim < 0:
                check(ndim)
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim))))
        return out

    def set_weight(self, name: str, dev_ptr: int, shape, stream: int) -> None:
        arr = (c_int64 * 4)(*list(shape) + [0] * (4 - len(shape)))
        check(lib().mz_set_weight(self._h, name.encode(), c_void_p(dev_ptr), arr, len(shape), c_void_p(stream)))

    def weights_complete(self) -> None:
        check(lib().mz_weights_complete(self._h))

    def workspace_bytes(self, B: int, H: int, W: int, max_in_flight: int = 0) -> int:
        out = c_size_t()
        check(lib().mz_workspace_bytes(self._h, B, H, W, max_in_flight, byref(out)))
        return int(out.value)

    def forward(self, x_ptr, sr_ptr, qa_ptr, B, H, W, clamp, ws_ptr, ws_bytes, max_in_flight, stream) -> None:
        check(
            lib().mz_forward(
                self._h, c_void_p(x_ptr), c_void_p(sr_ptr), c_void_p(qa_ptr) if qa_ptr else None, B, H, W, int(clamp),
                c_void_p(ws_ptr), ws_bytes, max_in_flight, c_void_p(stream),
            )
        )

    def forward_u8(self, x_ptr, sr_ptr, qa_ptr, B, H, W, ws_ptr, ws_bytes, max_in_flight, stream) -> None:
        check(
            lib().mz_forward_u8(
                self._h, c_void_p(x_ptr), c_void_p(sr_ptr), c_void_p(qa_ptr) if qa_ptr else None, B, H, W,
                c_void_p(ws_ptr), ws_bytes, max_in_flight, c_void_p(stream),
            )
        )

    def flops_per_image(self, H: int, W: int) -> float:
        return float(lib().mz_flops_per_image(self._h, H, W))

    def profile_enable(self, on: bool) -> None:
        check(lib().mz_profile_enable(self._h, int(on)))

    def profile_dump(self, path: str) -> None:
        check(lib().mz_profile_dump(self._h, str(path).encode()))

    def profile_read(self) -> dict:
        vals = [c_double() for _ in range(5)]
        check(lib().mz_profile_read(self._h, *[byref(v) for v in vals]))
        keys = ("conv_ms", "conv_flops", "conv_launches", "other_ms", "conv_bytes")
        return {k: v.value for k, v in zip(keys, vals)}
