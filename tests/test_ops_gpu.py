"""Operator-level parity on a real MI355X: every kernel mz_forward launches, called through the C ABI
on its own, against the CPU oracle (torch CPU float32) on the same inputs.  Inputs and weights are
rounded to the compute dtype first, so what is measured is the kernel's arithmetic."""

import ctypes

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DTYPES, assert_op_close, alloc_act, from_act, op_conv, pad16, pad_part, q, stream_ptr, to_act
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


CONV_CASES = [
    # B, H, W, cin, cout, silu
    (1, 8, 32, 16, 32, 0),
    (2, 13, 37, 16, 48, 1),    # ragged tile edges, N=48 -> padded tile
    (1, 20, 70, 48, 96, 1),    # NT=3, 3 K-chunks
    (1, 9, 33, 24, 40, 0),     # channel counts that need padding to 16
    (1, 16, 40, 32, 128, 1),   # two N tiles
    (1, 5, 9, 160, 16, 0),     # long K, tiny image
    (3, 8, 8, 96, 192, 1),     # BASELINE cfg3 stage-1 channel shape, batch > 1
    (1, 17, 65, 192, 96, 0),
]


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(dt, case):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, silu = case
    x = q(rnd((B, cin, H, W), 1), dtype)
    w = q(wrnd((cout, cin, 3, 3), 2), dtype)
    out = alloc_act(B, cout, H, W, dtype)
    op_conv(dtype, 0, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
    want = F.conv2d(x, w, padding=1)
    if silu:
        want = F.silu(want)
    got = from_act(out, cout)
    assert_op_close(got, want, dt)
    if pad16(cout) > cout:
        assert pad_part(out, cout).abs().max().item() == 0.0, "pad channels must be written as zeros"


PERSIST_CASES = [
    # B, H, W, cin, cout, silu : enough tiles that 8 or 16 persistent workgroups each walk several of them
    (2, 45, 100, 32, 96, 1),    # 16x32 tiles, ragged edges, one N tile
    (1, 40, 200, 48, 192, 0),   # 8x64 tiles, two N tiles (weights change at tile boundaries)
    (3, 33, 65, 16, 288, 1),    # one K-stage per tile: every barrier is a tile boundary
    (1, 70, 70, 96, 40, 0),     # padded N, 6 K-stages
    (2, 33, 65, 64, 96, 1),     # 16x32 tiles on the 16x16x32 kernel (16-bit types), ragged in both directions
    (1, 20, 130, 144, 96, 0),   # Cin = 144: the last 32-channel chunk of the 16x16x32 kernel is half zero planes
    (1, 24, 200, 160, 192, 1),  # 5 chunks, two N tiles, right-edge tiles with pixel fragments outside the image
]


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", PERSIST_CASES)
def test_conv3x3_persistent(dt, case, monkeypatch):
    """The persistent kernels (one workgroup per CU walking many tiles, LDS rings turning across tile boundaries) must
    match the oracle and must not depend on how tiles are dealt to workgroups: 8 and 16 workgroups give the same bits.
    f32 additionally agrees bit for bit with the one-tile-per-workgroup kernel (same MFMA, same K order); the 16-bit
    types run the 16x16x32-MFMA kernel when persistent, whose K order differs from the 32x32x16 per-tile kernel."""
    dtype = DTYPES[dt]
    B, H, W, cin, cout, silu = case
    x = q(rnd((B, cin, H, W), 11), dtype)
    w = q(wrnd((cout, cin, 3, 3), 12), dtype)
    outs = []
    for env in ({"MZ_PERSIST_WGS": "8"}, {"MZ_PERSIST_WGS": "16"}, {"MZ_NO_PERSIST": "1"}, {"MZ_PERSIST_WGS": "8", "MZ_NO_S16": "1"}):
        for k in ("MZ_PERSIST_WGS", "MZ_NO_PERSIST", "MZ_NO_S16"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = alloc_act(B, cout, H, W, dtype)
        op_conv(dtype, 0, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        outs.append(out)
    want = F.conv2d(x, w, padding=1)
    if silu:
        want = F.silu(want)
    for o in outs:
        assert_op_close(from_act(o, cout), want, dt)
    assert torch.equal(outs[0], outs[1]), "the result must not depend on the number of persistent workgroups"
    assert torch.equal(outs[2], outs[3]), "persistent and per-tile 32x32 kernels must agree bit for bit"
    if dt == "f32":
        assert torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("dt", list(DTYPES))
def test_subpixel_conv_persistent(dt, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, Hout, Wout = 2, 40, 70, 32, 192, 81, 140
    cq = cout // 4
    x = q(rnd((B, cin, H, W), 13), dtype)
    w = q(wrnd((cout, cin, 3, 3), 14), dtype)
    monkeypatch.setenv("MZ_PERSIST_WGS", "8")
    out = alloc_act(B, cq, Hout, Wout, dtype)
    op_conv(dtype, 1, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, Hout, Wout)
    want = oracle.fit_to(oracle.subpixel_conv(x, w), (Hout, Wout))
    assert_op_close(from_act(out, cq), want, dt)


D2S_CASES = [
    # B, H, W, cin, cout(=4*cq), Hout, Wout
    (1, 7, 9, 32, 64, 14, 18),
    (2, 5, 6, 16, 96, 11, 13),    # cq = 24 -> padded to 32; target one larger: zero border
    (1, 9, 35, 64, 128, 19, 70),
    (1, 8, 32, 96, 384, 16, 64),  # head C -> 4C of cfg3
]


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", D2S_CASES)
def test_subpixel_conv(dt, case):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, Hout, Wout = case
    cq = cout // 4
    x = q(rnd((B, cin, H, W), 3), dtype)
    w = q(wrnd((cout, cin, 3, 3), 4), dtype)
    out = alloc_act(B, cq, Hout, Wout, dtype)
    op_conv(dtype, 1, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, Hout, Wout)
    want = oracle.fit_to(oracle.subpixel_conv(x, w), (Hout, Wout))
    got = from_act(out, cq)
    assert_op_close(got, want, dt)
    if pad16(cq) > cq:
        assert pad_part(out, cq).abs().max().item() == 0.0


CRUSH_CASES = [(2, 7, 9, 16, 32), (1, 16, 64, 48, 96), (1, 21, 19, 24, 40), (1, 34, 66, 96, 192), (1, 9, 40, 192, 384)]


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", CRUSH_CASES)
def test_pixel_crush(dt, case):
    dtype = DTYPES[dt]
    B, H, W, cin, cout = case
    x = q(rnd((B, cin, H, W), 5), dtype)
    w = q(wrnd((cout, cin, 2, 2), 6), dtype)
    out = alloc_act(B, cout, H // 2, W // 2, dtype)
    op_conv(dtype, 2, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout)
    want = F.conv2d(x, w, stride=2)
    assert_op_close(from_act(out, cout), want, dt)


MIX_CASES = [(2, 7, 9, 16), (1, 16, 17, 24), (1, 20, 33, 48), (2, 8, 40, 96), (1, 5, 13, 128), (1, 3, 50, 384),
             # C = k * 192: the 16-bit types run mix16_kernel (192-channel N tiles, x / z straight into MFMA operands)
             (1, 9, 40, 192), (3, 7, 23, 192), (2, 11, 29, 384), (1, 10, 27, 768)]


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", MIX_CASES)
def test_adaptive_residual_mix(dt, case):
    dtype = DTYPES[dt]
    B, H, W, c = case
    x = q(rnd((B, c, H, W), 7), dtype)
    z = q(rnd((B, c, H, W), 8), dtype)
    w = q(wrnd((c, 2 * c, 1, 1), 9), dtype)
    alpha = 0.37
    out = alloc_act(B, c, H, W, dtype)
    op_conv(dtype, 3, to_act(x, dtype), to_act(z, dtype), w, alpha, out, B, H, W, 2 * c, c)
    want = oracle.residual_mix(x, z, w, torch.tensor(alpha))
    assert_op_close(from_act(out, c), want, dt)
    if pad16(c) > c:
        assert pad_part(out, c).abs().max().item() == 0.0


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("c", [48, 96, 192, 384])
@pytest.mark.parametrize("alpha,wscale", [(-100.0, 1.0), (-100.0, 150.0), (100.0, 150.0), (0.37, 150.0)])
def test_adaptive_residual_mix_saturated(dt, c, alpha, wscale):
    """sigmoid(alpha) and the gate's sigmoid at the ends of their ranges (model.py:833-837).  alpha = -100 makes 1 / sigmoid(alpha) =
    1 + e^100 overflow float32; the blend folds that factor into the gate's reciprocal (blend_(), mz_device.h) and the host keeps it
    finite, so that the result is x as in the reference and never NaN -- also where the gate saturates (150 x weights: |beta| reaches a
    few hundred, e^-beta flushes to 0 or overflows).  Every mix kernel family (C = 48 / 96: general 1x1 kernel, 192: mix16b, 384: mix16)."""
    dtype = DTYPES[dt]
    B, H, W = 1, 9, 21
    x = q(rnd((B, c, H, W), 7), dtype)
    z = q(rnd((B, c, H, W), 8), dtype)
    w = q(wscale * wrnd((c, 2 * c, 1, 1), 9), dtype)
    out = alloc_act(B, c, H, W, dtype)
    op_conv(dtype, 3, to_act(x, dtype), to_act(z, dtype), w, alpha, out, B, H, W, 2 * c, c)
    got = from_act(out, c)
    assert torch.isfinite(got).all()
    want = oracle.residual_mix(x, z, w, torch.tensor(alpha))
    if alpha < -50:
        assert torch.equal(got, x)  # sigmoid(-100) = 4e-44: the mix returns its first input
    if dt == "f32" and wscale > 1:  # beta itself carries ~1e-4 of summation-order noise at this weight scale
        assert (got - want).abs().max().item() <= 2e-4
    else:
        assert_op_close(got, want, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("wgs", ["", "8"])
@pytest.mark.parametrize("case", [(1, 9, 40), (2, 33, 65), (3, 16, 16)])
def test_mix_c192_both_kernels(dt, case, wgs, monkeypatch):
    """C = 192 runs mix16b_kernel (persistent; gate rows in B-operand order: x, z and beta of a 16-byte entry in one lane, x and z
    read once); MZ_NO_MIX16B=1 keeps mix16_kernel (blend in accumulator layout).  Both against the oracle (model.py:826-839), and
    against each other: same K order inside every MFMA, same blend arithmetic => identical bits.  MZ_PERSIST_WGS=8: 64 waves, so
    that every wave walks several 32-pixel units (the loop that loads the next unit between the stores of the current one)."""
    dtype = DTYPES[dt]
    B, H, W = case
    if wgs:
        monkeypatch.setenv("MZ_PERSIST_WGS", wgs)
    c = 192
    x = q(rnd((B, c, H, W), 27), dtype)
    z = q(rnd((B, c, H, W), 28), dtype)
    w = q(wrnd((c, 2 * c, 1, 1), 29), dtype)
    alpha = -0.21
    want = oracle.residual_mix(x, z, w, torch.tensor(alpha))
    outs = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("MZ_NO_MIX16B", "1")
        else:
            monkeypatch.delenv("MZ_NO_MIX16B", raising=False)
        out = alloc_act(B, c, H, W, dtype)
        op_conv(dtype, 3, to_act(x, dtype), to_act(z, dtype), w, alpha, out, B, H, W, 2 * c, c)
        outs[knob] = from_act(out, c)
        assert_op_close(outs[knob], want, dt)
    assert torch.equal(outs["0"], outs["1"])


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", [(2, 9, 11, 16), (1, 33, 40, 48), (1, 8, 8, 24)])
def test_stem(dt, case):
    dtype = DTYPES[dt]
    B, H, W, c = case
    x = q(rnd((B, 3, H, W), 10).abs(), dtype)
    w = rnd((c, 3, 1, 1), 11)
    b = rnd((c,), 12, 0.1)
    out = alloc_act(B, c, H, W, dtype)
    xd, wd, bd = x.to("cuda", dtype).contiguous(), w.cuda(), b.cuda()
    _ffi.check(_ffi.lib().mz_op_stem(
        _ffi.dtype_code(dtype), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(wd.data_ptr()),
        ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, H, W, c, ctypes.c_void_p(stream_ptr())))
    torch.cuda.synchronize()
    want = F.conv2d(x, w, b)
    assert_op_close(from_act(out, c), want, dt)
    if pad16(c) > c:
        assert pad_part(out, c).abs().max().item() == 0.0


FINAL_CASES = [
    # B, H, W (conv grid), cin, R, clamp
    (1, 8, 32, 16, 2, 0),
    (2, 9, 11, 16, 2, 1),
    (1, 20, 36, 48, 4, 1),
    (1, 16, 24, 32, 8, 0),
    (1, 10, 70, 96, 4, 1),
]


@pytest.mark.parametrize("dt", list(DTYPES))
@pytest.mark.parametrize("case", FINAL_CASES)
def test_final_subpixel_bicubic_add_clamp(dt, case):
    dtype = DTYPES[dt]
    B, H, W, cin, R, clamp = case
    Hi, Wi = 2 * H // R, 2 * W // R
    assert Hi * R == 2 * H and Wi * R == 2 * W
    feat = q(rnd((B, cin, H, W), 13), dtype)
    img = q(rnd((B, 3, Hi, Wi), 14).abs(), dtype)
    w = q(wrnd((12, cin, 3, 3), 15) * 0.5, dtype)
    out = torch.full((B, 3, 2 * H, 2 * W), 7.0, dtype=dtype, device="cuda")
    fd, imd, wd = to_act(feat, dtype), img.to("cuda", dtype).contiguous(), w.cuda()
    _ffi.check(_ffi.lib().mz_op_final(
        _ffi.dtype_code(dtype), ctypes.c_void_p(fd.data_ptr()), ctypes.c_void_p(imd.data_ptr()),
        ctypes.c_void_p(wd.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, H, W, cin, R, clamp,
        ctypes.c_void_p(stream_ptr())))
    torch.cuda.synchronize()
    want = oracle.bicubic_upsample(img, R) + oracle.subpixel_conv(feat, w)
    if clamp:
        want = want.clamp(0, 1)
    assert_op_close(out.float().cpu(), want, dt)
