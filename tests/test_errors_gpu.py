"""Error behaviour on the GPU box: bad calls fail loudly with the library's message, and random operator shapes
(seeded fuzz) agree with the oracle."""

import ctypes

import pytest
import torch
import torch.nn.functional as F

from golden_util import GoldenCase
from gpu_util import DTYPES, assert_op_close, alloc_act, from_act, op_conv, q, to_act
from ultrazoom_amd import MewZoom, _ffi
from ultrazoom_amd.synth import hash_uniform

pytestmark = pytest.mark.gpu


def test_bad_calls_raise():
    case = GoldenCase("g1_2x_c16")
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda").eval()
    x = case.image().cuda()
    with pytest.raises(RuntimeError, match="should be the same"):
        m.upscale(x.half())  # dtype mismatch, as torch's conv2d would complain
    with pytest.raises(_ffi.MewZoomHipError, match="H, W >= 8"):
        m.upscale(x[:, :, :4, :4])
    with pytest.raises(AssertionError):
        m.upscale(x[:, :2])
    engine = m._get_engine(x)
    need = engine.handle.workspace_bytes(1, 32, 32)
    ws = torch.empty(need // 2, dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 3, 64, 64, device="cuda")
    with pytest.raises(_ffi.MewZoomHipError, match="workspace too small"):
        engine.handle.forward(x.data_ptr(), out.data_ptr(), 0, 1, 32, 32, True, ws.data_ptr(), ws.numel(), 0,
                              torch.cuda.current_stream().cuda_stream)


def _rnd(shape, seed, scale=1.0):
    n = 1
    for s in shape:
        n *= s
    return torch.from_numpy(((2.0 * hash_uniform(n, seed) - 1.0) * scale).reshape(shape))


@pytest.mark.parametrize("dt", list(DTYPES))
def test_conv3x3_fuzz(dt):
    """24 seeded random shapes (odd sizes, channel counts that need padding, several N tiles, both tile shapes)."""
    dtype = DTYPES[dt]
    u = hash_uniform(24 * 6, 4242).reshape(24, 6)
    worst = 0.0
    for i, row in enumerate(u):
        B = 1 + int(row[0] * 3)
        H = 1 + int(row[1] * 40)
        W = 1 + int(row[2] * 90)
        cin = 8 * (1 + int(row[3] * 12))
        cout = 8 * (1 + int(row[4] * 30))
        silu = int(row[5] * 2)
        x = q(_rnd((B, cin, H, W), 100 + i), dtype)
        w = q(_rnd((cout, cin, 3, 3), 200 + i, (3.0 / (9 * cin)) ** 0.5 * 1.7), dtype)
        out = alloc_act(B, cout, H, W, dtype)
        op_conv(dtype, 0, to_act(x, dtype), None, w, 0.0, out, B, H, W, cin, cout, silu=silu)
        want = F.conv2d(x, w, padding=1)
        if silu:
            want = F.silu(want)
        err = assert_op_close(from_act(out, cout), want, dt, f"case {i}: B={B} H={H} W={W} cin={cin} cout={cout} silu={silu}")
        worst = max(worst, err)
    print(f"conv3x3 fuzz {dt}: worst max-abs {worst:.3e}")
