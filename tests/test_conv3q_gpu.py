"""conv3q_kernel (one 512-register wave per SIMD, 8 x 48 pixel tiles, 96-channel N tiles) against the oracle and against
conv3s_kernel: both accumulate every output element in the same order (chunk, tap, 32-channel MFMA), so they must agree bit
for bit.  conv3q is the default where it applies (96-channel N tiles, even chunk counts, no more padded pixels than the
conv3s tiles); MZ_NO_Q=1 (read per mz_op_* call / at handle creation) keeps conv3s_kernel."""

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DTYPES, alloc_act, assert_op_close, from_act, op_conv, q, to_act
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


Q_CASES = [
    # B, H, W, cin, cout, silu, persistent workgroups   (the kernel takes even chunk counts: Cin = 64 k or 64 k - 16)
    (1, 8, 48, 64, 96, 0, 0),      # one tile, two chunks
    (1, 16, 96, 64, 96, 1, 0),     # four tiles
    (2, 13, 37, 128, 96, 1, 0),    # ragged edges in both directions, four chunks
    (3, 40, 100, 64, 192, 1, 8),   # two N tiles, many tiles per workgroup (8 workgroups): tile boundaries, weight switches
    (1, 70, 70, 192, 96, 0, 8),    # six chunks
    (1, 20, 130, 112, 96, 0, 8),   # Cin = 112: the last chunk is half zero planes
    (2, 9, 250, 48, 288, 1, 16),   # two chunks per tile, the second half zero planes; three N tiles
    (1, 135, 240, 192, 96, 1, 0),  # the level-4 geometry of cfg3 (5 tiles per row, 17 tile rows)
    (1, 24, 50, 96, 96, 1, 0),     # odd chunk count (3): the host keeps conv3s_kernel; both settings must still agree
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", Q_CASES)
def test_conv3q_matches_oracle_and_conv3s(dt, case, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, silu, wgs = case
    x = q(rnd((B, cin, H, W), 21), dtype)
    w = q(wrnd((cout, cin, 3, 3), 22), dtype)
    outs = {}
    for name, env in {"q": {"MZ_NO_R": "1"}, "s": {"MZ_NO_Q": "1", "MZ_NO_R": "1"}}.items():
        for k in ("MZ_NO_Q", "MZ_NO_R", "MZ_PERSIST_WGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if wgs:
            monkeypatch.setenv("MZ_PERSIST_WGS", str(wgs))
        out = alloc_act(B, cout, H, W, dtype)
        op_conv(dtype, 0, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        outs[name] = out
    want = F.conv2d(x, w, padding=1)
    if silu:
        want = F.silu(want)
    assert_op_close(from_act(outs["q"], cout), want, dt, "conv3q")
    assert torch.equal(outs["q"], outs["s"]), "conv3q and conv3s must agree bit for bit"


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_conv3q_subpixel(dt, monkeypatch):
    dtype = DTYPES[dt]
    B, H, W, cin, cout, Hout, Wout = 2, 40, 70, 64, 384, 81, 140  # 96 -> 4 x 96 of the cfg3 head, odd target size
    cq = cout // 4
    x = q(rnd((B, cin, H, W), 23), dtype)
    w = q(wrnd((cout, cin, 3, 3), 24), dtype)
    monkeypatch.delenv("MZ_NO_Q", raising=False)
    monkeypatch.setenv("MZ_NO_R", "1")
    monkeypatch.setenv("MZ_PERSIST_WGS", "8")
    out = alloc_act(B, cq, Hout, Wout, dtype)
    op_conv(dtype, 1, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, Hout, Wout)
    want = oracle.fit_to(oracle.subpixel_conv(x, w), (Hout, Wout))
    assert_op_close(from_act(out, cq), want, dt, "conv3q d2s")
