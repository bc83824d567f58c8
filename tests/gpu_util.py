"""Helpers for the GPU parity tests: layout conversion to/from the library's plane-major activation tensors."""

from __future__ import annotations

import ctypes

import torch

from ultrazoom_amd import _ffi

DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# Tolerance of ONE operator whose inputs were already rounded to the dtype and whose accumulation is fp32: the only
# legitimate error is accumulation-order noise (f32: 2e-5 absolute on O(1) data) plus, for the 16-bit types, ONE
# rounding of the output to the storage type: |got - want| <= 1 ulp_storage(|want|) + 1e-5, element by element.
F32_OP_TOL = 2e-5
MANTISSA_BITS = {"bf16": 7, "f16": 10}
MIN_EXPONENT = {"bf16": -126, "f16": -14}


def ulp_of(want: torch.Tensor, dt: str) -> torch.Tensor:
    """Spacing of the storage type `dt` at |want| (element-wise, float32)."""
    _, e = torch.frexp(want.abs().float())  # |want| = m * 2^e, m in [0.5, 1)  =>  floor(log2 |want|) = e - 1
    e = torch.where(want == 0, torch.full_like(e, MIN_EXPONENT[dt]), e - 1).clamp(min=MIN_EXPONENT[dt])
    return torch.ldexp(torch.ones_like(want, dtype=torch.float32), e - MANTISSA_BITS[dt])


def op_excess(got: torch.Tensor, want: torch.Tensor, dt: str) -> float:
    """max over elements of |got - want| / tolerance(element); <= 1 passes."""
    diff = (got.float() - want.float()).abs()
    if dt == "f32":
        return (diff / F32_OP_TOL).max().item()
    return (diff / (ulp_of(want, dt) + 1e-5)).max().item()


def op_excess_map(got: torch.Tensor, want: torch.Tensor, dt: str) -> torch.Tensor:
    """|got - want| / tolerance(element), element-wise."""
    diff = (got.float() - want.float()).abs()
    if dt == "f32":
        return diff / F32_OP_TOL
    return diff / (ulp_of(want, dt) + 1e-5)


def assert_op_close(got: torch.Tensor, want: torch.Tensor, dt: str, what: str = "") -> float:
    ex = op_excess(got, want, dt)
    worst = (got.float() - want.float()).abs().max().item()
    assert ex <= 1.0, f"{what} {dt}: |got - want| reaches {ex:.2f} x (1 ulp + 1e-5) (max-abs {worst:.3e})"
    return worst


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


def planes_per(dtype) -> int:
    """Channels per 16-byte plane."""
    return 16 // torch.empty((), dtype=dtype).element_size()


def alloc_act(B: int, C: int, H: int, W: int, dtype, fill: float = 7.0) -> torch.Tensor:
    """An activation tensor in the library's layout [B, P, H, W, channels-per-plane], pre-filled with garbage."""
    ppu = planes_per(dtype)
    return torch.full((B, pad16(C) // ppu, H, W, ppu), fill, dtype=dtype, device="cuda")


def to_act(x: torch.Tensor, dtype) -> torch.Tensor:
    """[B,C,H,W] float CPU -> plane-major [B, P, H, W, ppu] `dtype` on the GPU (pad channels zero)."""
    B, C, H, W = x.shape
    ppu = planes_per(dtype)
    t = torch.zeros(B, pad16(C), H, W, dtype=dtype, device="cuda")
    t[:, :C] = x.to(dtype)
    return t.reshape(B, pad16(C) // ppu, ppu, H, W).permute(0, 1, 3, 4, 2).contiguous()


def from_act(t: torch.Tensor, C: int) -> torch.Tensor:
    B, P, H, W, ppu = t.shape
    return t.permute(0, 1, 4, 2, 3).reshape(B, P * ppu, H, W)[:, :C].float().cpu()


def pad_part(t: torch.Tensor, C: int) -> torch.Tensor:
    """The pad channels (>= C) of an activation tensor; the kernels must write zeros there."""
    B, P, H, W, ppu = t.shape
    return t.permute(0, 1, 4, 2, 3).reshape(B, P * ppu, H, W)[:, C:]


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


def last_kernel() -> str:
    """Kernel family of the most recent convolution / mix launch of this thread (mz_debug_last_kernel)."""
    lib = _ffi.lib()
    lib.mz_debug_last_kernel.restype = ctypes.c_char_p
    return lib.mz_debug_last_kernel().decode()
