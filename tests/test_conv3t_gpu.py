"""conv3t_kernel: the role-alternating 3x3 kernel for ONE N tile of 33..48 output channels (three 16-channel fragments x twelve pixel
fragments per wave, 12 x 64 pixel tiles) -- conv2 of the level-1 block of the 48-channel models (BASELINE configs[0] / [1]; reference
model.py:746-748, 773-778) and, fused, conv2 + AdaptiveResidualMix (model.py:826-839).

Plain / SiLU: against the oracle (1 ulp of the storage type) and bit for bit against conv3s_kernel (MZ_NO_T=1): all 16x16x32 kernels
accumulate in the same order (chunk, tap, one 32-channel MFMA).  Fused: against the oracle within ulp(out) + ulp(z), and against
conv3s_kernel<.., FUSE>, whose gate sums its products in another order inside a K step (>= 98 % of the outputs bit-equal)."""

import ctypes

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DTYPES, alloc_act, assert_op_close, from_act, last_kernel, op_conv, pad_part, q, stream_ptr, to_act, ulp_of
from oracle import mewzoom_oracle as oracle
from ultrazoom_amd import _ffi
from ultrazoom_amd.synth import hash_uniform

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    n = 1
    for s in shape:
        n *= s
    return torch.from_numpy(((2.0 * hash_uniform(n, seed) - 1.0) * scale).reshape(shape))


def wrnd(shape, seed):
    fan_in = shape[1] * shape[2] * shape[3]
    return rnd(shape, seed, (3.0 / fan_in) ** 0.5 * 1.7)


def set_env(monkeypatch, env, wgs):
    for k in ("MZ_NO_T", "MZ_PERSIST_WGS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if wgs:
        monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))


T_CASES = [
    # B, H, W, cin, cout, silu, persistent workgroups
    (1, 12, 64, 96, 48, 0, 0),     # ONE tile: team X computes, team Y only loads; the final epilogue runs without a partner
    (1, 12, 128, 96, 48, 1, 8),    # two tiles in one workgroup: one per team
    (1, 24, 192, 96, 48, 1, 8),    # six tiles on eight workgroups
    (3, 40, 100, 96, 48, 1, 8),    # 24 tiles on 8 workgroups, ragged in both directions
    (2, 13, 37, 96, 48, 0, 0),     # a tile larger than the image
    (1, 70, 70, 192, 48, 1, 8),    # six chunks: one entry per step
    (1, 30, 130, 224, 48, 0, 8),   # seven chunks: a plain chunk behind the six epilogue chunks
    (1, 27, 200, 96, 40, 1, 8),    # Cout = 40: pad channels (zero weights) must come out as zeros
    (2, 25, 65, 96, 33, 0, 16),    # Cout = 33
    (1, 36, 64, 128, 48, 1, 8),    # four chunks: the host must NOT pick conv3t (conv3s takes it); both settings agree
    (1, 135, 240, 96, 48, 1, 0),   # 12 x 4 tiles, rows 135 = 11 tiles + 3 rows
    (2, 50, 190, 96, 48, 1, 24),   # 30 tiles on 24 workgroups (three per XCD): uneven shares of the tile list
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", T_CASES)
def test_conv3t_matches_oracle_and_conv3s(dt, case, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, silu, wgs = case
    x = q(rnd((B, cin, H, W), 51), dtype)
    w = q(wrnd((cout, cin, 3, 3), 52), dtype)
    xa = to_act(x, dtype)
    outs = {}
    for name, env in {"t": {}, "s": {"MZ_NO_T": "1"}}.items():
        set_env(monkeypatch, env, wgs)
        out = alloc_act(B, cout, H, W, dtype)
        op_conv(dtype, 0, xa, None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        outs[name] = out
        assert last_kernel() == ("conv3t" if name == "t" and cin // 32 in (3, 6, 7) else "conv3s"), last_kernel()
    want = F.conv2d(x, w, padding=1)
    if silu:
        want = F.silu(want)
    assert_op_close(from_act(outs["t"], cout), want, dt, "conv3t")
    assert torch.equal(outs["t"], outs["s"]), "conv3t and conv3s must agree bit for bit"
    if cout < 48:
        assert (pad_part(outs["t"], cout) == 0).all(), "pad channels must be written as zeros"


def op_conv_mix(dtype, hid, x, w2, wmix, alpha, out, B, H, W, cin, cout):
    w2d = w2.to("cuda", torch.float32).contiguous()
    wmd = wmix.to("cuda", torch.float32).contiguous()
    _ffi.check(_ffi.lib().mz_op_conv_mix(
        _ffi.dtype_code(dtype), ctypes.c_void_p(hid.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w2d.data_ptr()),
        ctypes.c_void_p(wmd.data_ptr()), ctypes.c_float(alpha), ctypes.c_void_p(out.data_ptr()), B, H, W, cin, cout,
        ctypes.c_void_p(stream_ptr())))
    torch.cuda.synchronize()


FUSE_CASES = [
    # B, H, W, cin, cout, persistent workgroups
    (1, 12, 64, 96, 48, 0),      # one tile: the final epilogue without a partner
    (1, 24, 192, 96, 48, 8),     # one tile per workgroup
    (3, 40, 100, 96, 48, 8),     # three tiles per workgroup, ragged edges: units under the partner's K loop (3 chunks: 0 + 2 + 2, 1 + 1 + 2, ..)
    (2, 13, 37, 96, 48, 0),
    (1, 70, 70, 192, 48, 8),     # hidden_ratio 4: six chunks, one unit per step
    (1, 30, 130, 224, 48, 8),    # seven chunks
    (1, 27, 200, 96, 40, 8),     # C = 40: pad channels in x, z and out
    (1, 135, 240, 96, 48, 0),    # many tiles per workgroup on the real device width
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", FUSE_CASES)
def test_conv3t_fused_mix(dt, case, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, wgs = case
    hid = q(rnd((B, cin, H, W), 61), dtype)
    x = q(rnd((B, cout, H, W), 62), dtype)
    w2 = q(wrnd((cout, cin, 3, 3), 63), dtype)
    wmix = q(rnd((cout, 2 * cout, 1, 1), 64, (3.0 / (2 * cout)) ** 0.5 * 1.7), dtype)
    alpha = 0.3
    ha, xa = to_act(hid, dtype), to_act(x, dtype)
    outs = {}
    for name, env in {"t": {}, "s": {"MZ_NO_T": "1"}}.items():
        set_env(monkeypatch, env, wgs)
        out = alloc_act(B, cout, H, W, dtype)
        op_conv_mix(dtype, ha, xa, w2, wmix, alpha, out, B, H, W, cin, cout)
        outs[name] = out
        assert last_kernel() == ("conv3t_fused" if name == "t" else "conv3s_fused"), last_kernel()
    # the kernel rounds z to the storage type before the gate GEMM and the blend (as the unfused path stores it)
    z = q(F.conv2d(hid, w2, padding=1), dtype)
    want = oracle.residual_mix(x, z, wmix, torch.tensor(alpha))
    got = from_act(outs["t"], cout)
    # one rounding of the output, plus one rounding step of z where its fp32 sum sits on a rounding boundary
    tol = ulp_of(want, dt) + ulp_of(z, dt) + 1e-5
    ex = ((got - want).abs() / tol).max().item()
    assert ex <= 1.0, f"fused conv2 + mix {dt}: {ex:.2f} x (ulp(out) + ulp(z) + 1e-5)"
    assert ((got - want).abs() > ulp_of(want, dt) + 1e-5).float().mean().item() < 0.02, "more than 2 % of the outputs are off by more than one ulp"
    b_ = from_act(outs["s"], cout)
    assert (got == b_).float().mean().item() > 0.98, "conv3t and conv3s fused kernels: more than 2 % of the outputs differ"
    assert ((got - b_).abs() / tol).max().item() <= 1.0, "conv3t and conv3s fused kernels differ by more than the tolerance"
    assert (pad_part(outs["t"], cout) == 0).all(), "pad channels must stay zero"


def _sweep_cases():
    import random
    rng = random.Random(20261005)
    cases = []
    for _ in range(20):
        H = rng.choice([11, 12, 13, 23, 24, 25, 37, 48, 61])
        W = rng.choice([63, 64, 65, 100, 127, 128, 129, 191, 200])
        cin = 32 * rng.choice([3, 3, 6, 8])
        cases.append((rng.choice([1, 2, 3]), H, W, cin, rng.choice([48, 48, 44]), rng.choice([0, 1]), rng.choice([0, 8, 16])))
    return cases


@pytest.mark.parametrize("case", _sweep_cases())
def test_conv3t_shape_sweep_equals_conv3s(case, monkeypatch):
    """Ragged heights / widths around the 12 x 64 tile and its 14 x 66 halo image (border and interior paths of the halo offsets,
    partial tiles, one-tile and many-tile workgroups), plain and fused, bf16: conv3t against conv3s."""
    dtype = DTYPES["bf16"]
    B, H, W, cin, cout, silu, wgs = case
    hid = q(rnd((B, cin, H, W), 71), dtype)
    w = q(wrnd((cout, cin, 3, 3), 72), dtype)
    ha = to_act(hid, dtype)
    outs = {}
    for name, env in {"t": {}, "s": {"MZ_NO_T": "1"}}.items():
        set_env(monkeypatch, env, wgs)
        out = alloc_act(B, cout, H, W, dtype)
        op_conv(dtype, 0, ha, None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        outs[name] = out
        assert last_kernel() == ("conv3t" if name == "t" else "conv3s"), last_kernel()
    assert torch.equal(outs["t"], outs["s"]), f"conv3t and conv3s differ on {case}"
    # ... and the fused variant on the same shape
    x = q(rnd((B, cout, H, W), 73), dtype)
    wmix = q(rnd((cout, 2 * cout, 1, 1), 74, (3.0 / (2 * cout)) ** 0.5 * 1.7), dtype)
    xa = to_act(x, dtype)
    fo = {}
    for name, env in {"t": {}, "s": {"MZ_NO_T": "1"}}.items():
        set_env(monkeypatch, env, wgs)
        out = alloc_act(B, cout, H, W, dtype)
        op_conv_mix(dtype, ha, xa, w, wmix, -0.4, out, B, H, W, cin, cout)
        fo[name] = from_act(out, cout)
        assert last_kernel() == ("conv3t_fused" if name == "t" else "conv3s_fused"), last_kernel()
    z = q(F.conv2d(hid, w, padding=1), dtype)
    tol = ulp_of(fo["s"], "bf16") + ulp_of(z, "bf16") + 1e-5
    assert ((fo["t"] - fo["s"]).abs() / tol).max().item() <= 1.0, f"fused conv3t and conv3s differ on {case}"
    assert (fo["t"] == fo["s"]).float().mean().item() > 0.97
