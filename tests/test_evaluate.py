"""The PSNR / SSIM harness.  torchmetrics is absent here, so the restatement is checked against an
independent numpy/scipy computation of the same published definitions."""

import math

import numpy as np
import pytest
import torch
from scipy.ndimage import correlate1d

from ultrazoom_amd.evaluate import PSNR, SSIM, evaluate, ssim_per_image
from ultrazoom_amd.synth import synth_image


def test_psnr_is_global_mse_over_all_updates():
    a = synth_image(2, 16, 20, seed=1)
    b = (a + 0.1).clamp(0, 2)
    c = synth_image(1, 8, 8, seed=2)
    m = PSNR(1.0)
    m.update(b, a)
    m.update(c, c)  # a perfect image lowers the global MSE, it does not make the result infinite
    mse = float(((b - a).double() ** 2).sum()) / (a.numel() + c.numel())
    assert math.isclose(m.compute(), 10 * math.log10(1.0 / mse), rel_tol=1e-12)
    p = PSNR(1.0)
    p.update(c, c)
    assert p.compute() == float("inf")


def _ssim_numpy(p, t, data_range):
    x = np.arange(11) - 5.0
    g = np.exp(-x * x / (2 * 1.5**2))
    g /= g.sum()

    def blur(z):
        z = np.pad(z, 5, mode="reflect")
        return correlate1d(correlate1d(z, g, axis=0, mode="constant"), g, axis=1, mode="constant")[5:-5, 5:-5]

    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mp, mt = blur(p), blur(t)
    spp, stt, spt = blur(p * p) - mp * mp, blur(t * t) - mt * mt, blur(p * t) - mp * mt
    m = ((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp * mp + mt * mt + c1) * (spp + stt + c2))
    return m[5:-5, 5:-5].mean()


def test_ssim_matches_independent_restatement():
    t = synth_image(2, 40, 48, seed=3)
    p = (t + 0.05 * (synth_image(2, 40, 48, seed=4) - 0.5)).clamp(0, 1)
    got = ssim_per_image(p, t, data_range=1.0)
    for b in range(2):
        want = np.mean([_ssim_numpy(p[b, c].double().numpy(), t[b, c].double().numpy(), 1.0) for c in range(3)])
        assert math.isclose(float(got[b]), want, rel_tol=1e-9), (float(got[b]), want)
    assert torch.allclose(ssim_per_image(t, t), torch.ones(2, dtype=torch.float64))
    m = SSIM()
    m.update(p, t)
    assert 0.5 < m.compute() < 1.0


def test_evaluate_loop_with_a_stand_in_model():
    class Nearest:
        def upscale(self, x):
            return torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")

    hr = synth_image(3, 32, 32, seed=5)
    lr = hr[:, :, ::2, ::2]
    r = evaluate(Nearest(), [(lr[:2], hr[:2]), (lr[2:], hr[2:])])
    assert r["images"] == 3 and 5.0 < r["psnr"] < 40.0 and -1.0 <= r["ssim"] <= 1.0


@pytest.mark.gpu
def test_evaluate_on_the_hip_model_against_the_oracle():
    from golden_util import GoldenCase
    from oracle import mewzoom_oracle as oracle
    from ultrazoom_amd import MewZoom

    case = GoldenCase("g1_2x_c16")
    m = MewZoom(**case.config)
    m.load_state_dict(case.weights())
    m = m.to("cuda").eval()
    x = synth_image(2, 40, 56, seed=6)
    want = oracle.upscale(case.config, case.weights(), x)  # the "ground truth" of this check: the CPU oracle's output
    r = evaluate(m, [(x.cuda(), want.cuda())])
    assert r["images"] == 2 and r["psnr"] > 100.0 and r["ssim"] > 0.999999
