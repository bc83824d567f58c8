"""a17: the optional per-channel affine ("FiLM") epilogue of the 3x3 kernel.  NO REFERENCE: the snapshot has
no ControlModule, so this is checked against the build's own CPU statement of the operator (oracle.film_conv) -- parity unpinned."""

import ctypes

import pytest
import torch

from gpu_util import DTYPES, alloc_act, assert_op_close, from_act, pad16, pad_part, q, stream_ptr, to_act
from oracle import mewzoom_oracle as oracle
from ultrazoom_amd import _ffi
from ultrazoom_amd.synth import hash_uniform

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    n = 1
    for s in shape:
        n *= s
    return torch.from_numpy(((2.0 * hash_uniform(n, seed) - 1.0) * scale).reshape(shape))


def film(dtype, x, w, gamma, beta, B, H, W, cin, cout, silu):
    out = alloc_act(B, cout, H, W, dtype)
    xd = to_act(x, dtype)
    wd, gd, bd = w.cuda().float().contiguous(), gamma.cuda().float().contiguous(), beta.cuda().float().contiguous()
    rc = _ffi.lib().mz_op_conv_film(
        _ffi.dtype_code(dtype), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(wd.data_ptr()), ctypes.c_void_p(gd.data_ptr()),
        ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, H, W, cin, cout, silu, ctypes.c_void_p(stream_ptr()))
    torch.cuda.synchronize()
    return rc, out


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", [(2, 13, 37, 32, 48, 1), (1, 20, 70, 96, 96, 0), (3, 9, 33, 64, 40, 1), (2, 24, 50, 96, 192, 1)])
def test_film_epilogue(dt, case):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, silu = case
    x = q(rnd((B, cin, H, W), 41), dtype)
    w = q(rnd((cout, cin, 3, 3), 42, (3.0 / (9 * cin)) ** 0.5 * 1.7), dtype)
    gamma = 1.0 + 0.5 * rnd((B, cout), 43)
    beta = 0.3 * rnd((B, cout), 44)
    rc, out = film(dtype, x, w, gamma, beta, B, H, W, cin, cout, silu)
    _ffi.check(rc)
    want = oracle.film_conv(x, w, gamma.float(), beta.float(), bool(silu))
    assert_op_close(from_act(out, cout), want, dt, "film")
    if pad16(cout) > cout:
        assert pad_part(out, cout).abs().max().item() == 0.0, "pad channels must stay zero"


def test_film_epilogue_refuses_what_it_cannot_do():
    x = q(rnd((1, 16, 8, 8), 45), torch.float32)
    w = rnd((16, 16, 3, 3), 46)
    g, b = torch.ones(1, 16), torch.zeros(1, 16)
    rc, _ = film(torch.float32, x, w, g, b, 1, 8, 8, 16, 16, 0)   # fp32 has no such epilogue
    assert rc == _ffi.MZ_ERR_INVALID_ARGUMENT and b"FiLM" in _ffi.lib().mz_last_error()
    x48 = q(rnd((1, 48, 8, 8), 47), torch.bfloat16)                  # Cin = 48 does not run on the 16x16x32 kernel
    rc, _ = film(torch.bfloat16, x48, rnd((16, 48, 3, 3), 48), g, b, 1, 8, 8, 48, 16, 0)
    assert rc == _ffi.MZ_ERR_INVALID_ARGUMENT
