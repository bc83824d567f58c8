"""conv3r_kernel (8 x 48 / 8 x 40 pixel x 96-channel tiles with role-alternating waves: a tile's epilogue runs under the next tile's
K loop) against the oracle and against conv3s_kernel.  The 16x16x32 kernels accumulate every output element in the same order (chunk,
tap, 32-channel MFMA), so they must agree bit for bit.  conv3r is the default where it applies (96-channel N tiles, >= 3 chunks of 32
channels -- odd counts included --, no more padded pixels than the conv3s tiles; Cin = 48: its ragged two-chunk variant); MZ_NO_R=1
leaves conv3s."""

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DTYPES, alloc_act, assert_op_close, from_act, last_kernel, op_conv, op_excess, op_excess_map, pad_part, q, to_act
from oracle import mewzoom_oracle as oracle
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


R_CASES = [
    # B, H, W, cin, cout, silu, persistent workgroups
    (1, 8, 48, 96, 96, 0, 0),      # ONE tile: team X computes, team Y only loads; the final epilogue runs without a partner
    (1, 8, 96, 96, 96, 1, 8),      # two tiles in one workgroup: one per team
    (1, 16, 144, 96, 96, 1, 8),    # six tiles in one workgroup... (8 workgroups, 6 tiles: one each) -> see the next case
    (3, 40, 100, 96, 192, 1, 8),   # 45 pixel tiles x 2 N tiles on 8 workgroups: ~11 tiles per workgroup, weight switches, odd counts
    (2, 13, 37, 128, 96, 1, 0),    # ragged edges in both directions, four chunks
    (1, 70, 70, 192, 96, 0, 8),    # six chunks
    (1, 20, 130, 112, 96, 0, 8),   # Cin = 112: the last chunk has two planes only -> the host must NOT pick conv3r
    (2, 9, 250, 160, 288, 1, 16),  # five chunks, three N tiles
    (1, 135, 240, 192, 96, 1, 0),  # the level-4 geometry of cfg3 (5 tiles per row, 17 tile rows)
    (1, 24, 50, 96, 80, 1, 8),     # Cout = 80: the N tile's last plane pair does not exist (range-checked stores)
    (1, 32, 96, 224, 96, 1, 8),    # seven chunks
    # widths that 40 divides better than 48: the 8 x 40 tile (five pixel fragments per wave, fragment 2 straddles the wave's two rows)
    (1, 8, 40, 96, 96, 1, 0),      # ONE 8 x 40 tile
    (1, 16, 80, 96, 96, 0, 8),     # four tiles on eight workgroups
    (2, 67, 120, 384, 192, 1, 8),  # cfg2's level-4 geometry (67 x 120: 9 x 3 tiles), twelve chunks, two N tiles
    (3, 23, 117, 96, 96, 1, 8),    # ragged in both directions; three chunks: 15 entries as 3 + 3, 3 + 3, 3 + 0
    (1, 30, 200, 192, 288, 0, 16), # six chunks (1 + 2 entries per chunk over five of them), three N tiles
    (1, 9, 79, 160, 96, 1, 8),     # five chunks
    (2, 50, 190, 96, 192, 1, 24),  # 56 x 2 tiles on 24 workgroups (three per XCD): uneven shares of the tile list, 4 - 5 tiles per workgroup
    (1, 33, 97, 192, 96, 0, 40),   # 15 tiles on 40 workgroups: most workgroups get none, some XCDs two
]


def expected_r_kernel(H, W):
    """The host's choice (mz_host.cpp, Runner::conv3): conv3r_kernel in the tile geometry that pads fewer pixels (8 x 40 or 8 x 48), unless
    conv3s_kernel's 8 x 64 / 16 x 32 tiles pad fewer still."""
    up = lambda v, m: -(-v // m) * m
    pads = min(up(H, 8) * up(W, 64), up(H, 16) * up(W, 32))
    pad48, pad40 = up(H, 8) * up(W, 48), up(H, 8) * up(W, 40)
    geo1 = pad40 < pad48
    return ("conv3r_8x40" if geo1 else "conv3r") if (pad40 if geo1 else pad48) <= pads else "conv3s"


def run(dtype, kind, x_act, w, out_shape, args, env, monkeypatch, wgs):
    for k in ("MZ_NO_R", "MZ_PERSIST_WGS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if wgs:
        monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))
    out = alloc_act(*out_shape, dtype)
    op_conv(dtype, kind, x_act, None, w, 0.0, out, *args)
    return out


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", R_CASES)
def test_conv3r_matches_oracle_and_conv3s(dt, case, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, silu, wgs = case
    x = q(rnd((B, cin, H, W), 31), dtype)
    w = q(wrnd((cout, cin, 3, 3), 32), dtype)
    xa = to_act(x, dtype)
    outs = {}
    for name, env in {"r": {}, "s": {"MZ_NO_R": "1"}}.items():
        out = alloc_act(B, cout, H, W, dtype)
        for k in ("MZ_NO_R", "MZ_PERSIST_WGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if wgs:
            monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))
        op_conv(dtype, 0, xa, None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        outs[name] = out
        if name == "s":  # (Cin = 112 pads K by 14 %: the 32x32x16 kernel with exact 16-channel chunks takes it)
            assert last_kernel() == ("conv3s" if cin % 32 == 0 else "conv3p"), last_kernel()
        elif cin % 32 == 0:  # (Cin = 112: two planes in the last chunk -> the host keeps conv3r out)
            assert last_kernel() == expected_r_kernel(H, W), (last_kernel(), H, W)
    want = F.conv2d(x, w, padding=1)
    if silu:
        want = F.silu(want)
    assert_op_close(from_act(outs["r"], cout), want, dt, "conv3r")
    assert torch.equal(outs["r"], outs["s"]), "conv3r and conv3s must agree bit for bit"
    assert (pad_part(outs["r"], cout) == 0).all(), "pad channels must stay zero"


RAG_CASES = [
    # B, H, W, cout, persistent workgroups: Cin = 48 (two 32-channel chunks, the second with two real planes), conv1 + SiLU
    (1, 8, 48, 96, 0),       # one tile
    (1, 16, 144, 96, 8),     # six tiles on eight workgroups
    (3, 40, 100, 96, 8),     # 45 tiles on 8 workgroups: both teams, several tiles each, ragged edges
    (2, 13, 37, 192, 0),     # two N tiles, a tile larger than the image
    (1, 27, 200, 80, 16),    # Cout = 80: planes of the N tile that do not exist
    (2, 135, 240, 96, 0),    # a real level size on every CU
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", RAG_CASES)
def test_conv3r_ragged_cin48(dt, case, monkeypatch):
    """conv3r_kernel<.., RAG>: Cin = 48.  The pieces of the second chunk's missing planes are issued out of range (zeros into LDS), the
    weights are packed with K padded to 64.  Against the oracle (1 ulp), and bit for bit against conv3s_kernel with the same padded K
    (MZ_KPAD_PCT=34 lets it take Cin = 48): same chunk / tap / 32-channel MFMA order."""
    dtype = DTYPES[dt]
    B, H, W, cout, wgs = case
    cin = 48
    x = q(rnd((B, cin, H, W), 41), dtype)
    w = q(wrnd((cout, cin, 3, 3), 42), dtype)
    xa = to_act(x, dtype)
    outs = {}
    for name, env in {"r": {}, "s": {"MZ_NO_R2": "1", "MZ_KPAD_PCT": "34"}}.items():
        for k in ("MZ_NO_R", "MZ_NO_R2", "MZ_KPAD_PCT", "MZ_PERSIST_WGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if wgs:
            monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))
        out = alloc_act(B, cout, H, W, dtype)
        op_conv(dtype, 0, xa, None, w, 0.0, out, B, H, W, cin, cout, silu=1)
        outs[name] = out
        assert last_kernel() == ("conv3r_ragged" if name == "r" else "conv3s"), last_kernel()
    want = F.silu(F.conv2d(x, w, padding=1))
    assert_op_close(from_act(outs["r"], cout), want, dt, "conv3r ragged")
    assert torch.equal(outs["r"], outs["s"]), "conv3r (ragged Cin) and conv3s (K padded) must agree bit for bit"
    assert (pad_part(outs["r"], cout) == 0).all(), "pad channels must stay zero"
    # without the activation the host keeps the layer off the ragged variant (it exists for conv1 + SiLU)
    monkeypatch.delenv("MZ_NO_R2", raising=False); monkeypatch.delenv("MZ_KPAD_PCT", raising=False)
    out = alloc_act(B, cout, H, W, dtype)
    op_conv(dtype, 0, xa, None, w, 0.0, out, B, H, W, cin, cout, silu=0)
    assert last_kernel() != "conv3r_ragged"
    assert_op_close(from_act(out, cout), F.conv2d(x, w, padding=1), dt, "Cin = 48 without SiLU")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 40, 70, 96, 384, 81, 140), (1, 30, 100, 96, 192, 60, 200), (1, 16, 48, 128, 96, 33, 97),
                                   # 8 x 40 tiles (W = 120, 80, 37): the straddling fragment's sub-pixel store
                                   (2, 67, 120, 384, 768, 135, 240), (1, 16, 80, 96, 384, 32, 160), (1, 20, 37, 96, 192, 41, 75)])
def test_conv3r_subpixel(dt, shape, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, Hout, Wout = shape  # 96 -> 4 x 96 of the cfg3 head; 96 -> 4 x 48; 128 -> 4 x 24 (N tile spans all four sub-pixels)
    cq = cout // 4
    x = q(rnd((B, cin, H, W), 33), dtype)
    w = q(wrnd((cout, cin, 3, 3), 34), dtype)
    xa = to_act(x, dtype)
    outs = {}
    for name, env in {"r": {}, "s": {"MZ_NO_R": "1"}}.items():
        for k in ("MZ_NO_R",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("MZ_PERSIST_WGS", "8")
        out = alloc_act(B, cq, Hout, Wout, dtype)
        op_conv(dtype, 1, xa, None, w, 0.0, out, B, H, W, cin, cout, Hout, Wout)
        outs[name] = out
    want = oracle.fit_to(oracle.subpixel_conv(x, w), (Hout, Wout))
    assert_op_close(from_act(outs["r"], cq), want, dt, "conv3r d2s")
    assert torch.equal(outs["r"], outs["s"]), "conv3r and conv3s must agree bit for bit (sub-pixel store)"


# ---- fused conv2 + AdaptiveResidualMix (model.py:773-778, 826-839) on conv3r_kernel: one pixel fragment's gate GEMM + blend per chunk
#      under the partner's K loop; identical bits to conv3s_kernel<.., FUSE> ----
def op_conv_mix(dtype, hid, x, w2, wmix, alpha, out, B, H, W, cin, cout):
    import ctypes
    from ultrazoom_amd import _ffi
    from gpu_util import stream_ptr
    w2d = w2.to("cuda", torch.float32).contiguous()
    wmd = wmix.to("cuda", torch.float32).contiguous()
    _ffi.check(_ffi.lib().mz_op_conv_mix(
        _ffi.dtype_code(dtype), ctypes.c_void_p(hid.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w2d.data_ptr()),
        ctypes.c_void_p(wmd.data_ptr()), ctypes.c_float(alpha), ctypes.c_void_p(out.data_ptr()), B, H, W, cin, cout,
        ctypes.c_void_p(stream_ptr())))
    torch.cuda.synchronize()


FUSE_CASES = [
    # B, H, W, cin, cout, persistent workgroups
    (1, 8, 48, 192, 96, 0),      # one tile: the final epilogue without a partner
    (1, 16, 144, 192, 96, 8),    # one tile per workgroup
    (3, 40, 100, 192, 96, 8),    # ~6 tiles per workgroup, ragged edges
    (2, 13, 37, 192, 96, 0),
    (1, 70, 70, 224, 96, 8),     # seven chunks: a plain chunk behind the six epilogue chunks
    (1, 24, 50, 384, 96, 8),     # hidden_ratio 4: twelve chunks
    (1, 24, 50, 160, 80, 8),     # C = 80: five chunks -> the host keeps conv3s_kernel; both settings must still agree
    (1, 20, 60, 192, 88, 8),     # C = 88: pad channels in x, z and out
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", FUSE_CASES)
def test_conv3r_fused_mix(dt, case, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, wgs = case
    hid = q(rnd((B, cin, H, W), 41), dtype)
    x = q(rnd((B, cout, H, W), 42), dtype)
    w2 = q(wrnd((cout, cin, 3, 3), 43), dtype)
    wmix = q(rnd((cout, 2 * cout, 1, 1), 44, (3.0 / (2 * cout)) ** 0.5 * 1.7), dtype)
    alpha = 0.3
    ha, xa = to_act(hid, dtype), to_act(x, dtype)
    outs = {}
    for name, env in {"r": {}, "s": {"MZ_NO_R": "1"}}.items():
        for k in ("MZ_NO_R", "MZ_PERSIST_WGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if wgs:
            monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))
        out = alloc_act(B, cout, H, W, dtype)
        op_conv_mix(dtype, ha, xa, w2, wmix, alpha, out, B, H, W, cin, cout)
        outs[name] = out
    # the kernel rounds z to the storage type before the gate GEMM and the blend (as the unfused path stores it)
    z = q(F.conv2d(hid, w2, padding=1), dtype)
    want = oracle.residual_mix(x, z, wmix, torch.tensor(alpha))
    got = from_act(outs["r"], cout)
    # one rounding of the output, plus one rounding step of z where its fp32 sum sits on a rounding boundary (the blend passes
    # sigmoid(alpha) * beta <= 1 of it on): element-wise |got - want| <= ulp(want) + ulp(z) + 1e-5
    from gpu_util import ulp_of
    tol = ulp_of(want, dt) + ulp_of(z, dt) + 1e-5
    ex = ((got - want).abs() / tol).max().item()
    assert ex <= 1.0, f"fused conv2 + mix {dt}: {ex:.2f} x (ulp(out) + ulp(z) + 1e-5)"
    assert ((got - want).abs() > ulp_of(want, dt) + 1e-5).float().mean().item() < 0.02, "more than 2 % of the outputs are off by more than one ulp"
    # conv3s_kernel<.., FUSE> sums the same gate products in another order inside each 32-wide K step (its x half is packed in plane
    # order): equal up to the last bit of the output, and equal bit for bit nearly everywhere
    b_ = from_act(outs["s"], cout)
    assert (got == b_).float().mean().item() > 0.98, "conv3r and conv3s fused kernels: more than 2 % of the outputs differ"
    assert ((got - b_).abs() / tol).max().item() <= 1.0, "conv3r and conv3s fused kernels differ by more than the tolerance"
    assert (pad_part(outs["r"], cout) == 0).all(), "pad channels must stay zero"


# ---- seeded shape sweep: ragged heights / widths around the 8 x 48 tile and the 10 x 50 halo image (border and interior paths of the
#      halo offsets, partial tiles, one-tile and many-tile workgroups), Cin of 3 .. 8 whole chunks, bf16; conv3r against conv3s bit for
#      bit (the oracle comparison of the same kernels is in the cases above) ----
def _sweep_cases():
    import random
    rng = random.Random(20261004)
    cases = []
    for _ in range(24):
        H = rng.choice([8, 9, 15, 16, 17, 23, 31, 40, 57])
        W = rng.choice([47, 48, 49, 95, 96, 97, 100, 143, 145, 191])
        cin = 32 * rng.choice([3, 4, 5, 6, 8])
        cout = rng.choice([96, 96, 192])
        cases.append((rng.choice([1, 2, 3]), H, W, cin, cout, rng.choice([0, 1]), rng.choice([0, 8, 16])))
    return cases


@pytest.mark.parametrize("case", _sweep_cases())
def test_conv3r_shape_sweep_equals_conv3s(case, monkeypatch):
    dtype = DTYPES["bf16"]
    B, H, W, cin, cout, silu, wgs = case
    x = q(rnd((B, cin, H, W), 41), dtype)
    w = q(wrnd((cout, cin, 3, 3), 42), dtype)
    xa = to_act(x, dtype)
    outs = {}
    for name, env in {"r": {}, "s": {"MZ_NO_R": "1"}}.items():
        out = alloc_act(B, cout, H, W, dtype)
        for k in ("MZ_NO_R", "MZ_PERSIST_WGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if wgs:
            monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))
        op_conv(dtype, 0, xa, None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        outs[name] = out
    assert torch.equal(outs["r"], outs["s"]), f"conv3r and conv3s differ on {case}"
