"""Helpers for the GPU parity tests: layout conversion to/from the library's NHWC padded tensors."""

from __future__ import annotations

import ctypes

import torch

from ultrazoom_amd import _ffi

DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# max-abs tolerance of ONE operator on O(1) data whose inputs were already rounded to the dtype:
# f32 = accumulation-order noise; 16-bit = one output rounding (2^-9 / 2^-11 relative) plus the same noise.
OP_TOL = {"f32": 2e-5, "bf16": 2.5e-2, "f16": 3e-3}


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


def to_nhwc(x: torch.Tensor, dtype, garbage_pad: bool = False) -> torch.Tensor:
    """[B,C,H,W] float CPU -> [B,H,W,pad16(C)] dtype on the GPU (pad channels zero)."""
    B, C, H, W = x.shape
    t = torch.zeros(B, H, W, pad16(C), dtype=dtype, device="cuda")
    t[..., :C] = x.permute(0, 2, 3, 1).to(dtype)
    return t.contiguous()


def from_nhwc(t: torch.Tensor, C: int) -> torch.Tensor:
    return t[..., :C].permute(0, 3, 1, 2).float().cpu()


def q(x: torch.Tensor, dtype) -> torch.Tensor:
    """Round a float32 CPU tensor to `dtype` and back: what the kernel actually sees."""
    return x.to(dtype).float()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def op_conv(dtype, kind, in0, in1, w, alpha, out, B, H, W, cin, cout, Hout=0, Wout=0, silu=0):
    wd = w.to("cuda", torch.float32).contiguous()
    _ffi.check(
        _ffi.lib().mz_op_conv(
            _ffi.dtype_code(dtype), kind, ctypes.c_void_p(in0.data_ptr()),
            ctypes.c_void_p(in1.data_ptr()) if in1 is not None else None, ctypes.c_void_p(wd.data_ptr()),
            ctypes.c_float(alpha), ctypes.c_void_p(out.data_ptr()), B, H, W, cin, cout, Hout, Wout, silu,
            ctypes.c_void_p(stream_ptr()),
        )
    )
    torch.cuda.synchronize()
