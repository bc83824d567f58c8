"""The PSNR / SSIM / VIF harness.  torchmetrics is absent here, so the restatement is checked against an
independent numpy/scipy computation of the same published definitions."""

import math

import numpy as np
import pytest
import torch
from scipy.ndimage import correlate1d

from ultrazoom_amd.evaluate import PSNR, SSIM, VIF, evaluate, ssim_per_image, vif_per_image
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


def _vif_numpy(p, t, sigma_n_sq=2.0):
    """Pixel-domain VIF of one channel, written independently with scipy (valid-mode Gaussian filtering by cropping)."""
    from scipy.ndimage import correlate

    num = den = 0.0
    for scale in range(4):
        n = 2 ** (4 - scale) + 1
        x = np.arange(n) - (n - 1) / 2.0
        k = np.exp(-(x[:, None] ** 2 + x[None, :] ** 2) / (2.0 * (n / 5.0) ** 2))
        k /= k.sum()
        h = n // 2
        valid = lambda z: correlate(z, k, mode="constant")[h:-h, h:-h]
        if scale > 0:
            t, p = valid(t)[::2, ::2], valid(p)[::2, ::2]
        mt, mp = valid(t), valid(p)
        stt = np.maximum(valid(t * t) - mt * mt, 0.0)
        spp = np.maximum(valid(p * p) - mp * mp, 0.0)
        stp = valid(t * p) - mt * mp
        g = stp / (stt + 1e-10)
        sv = spp - g * stp
        m = stt < 1e-10
        g[m] = 0.0; sv[m] = spp[m]; stt[m] = 0.0
        m = spp < 1e-10
        g[m] = 0.0; sv[m] = 0.0
        m = g < 0
        sv[m] = spp[m]; g[m] = 0.0
        sv = np.maximum(sv, 1e-10)
        num += np.log10(1.0 + g * g * stt / (sv + sigma_n_sq)).sum()
        den += np.log10(1.0 + stt / sigma_n_sq).sum()
    return num / den


def test_vif_matches_independent_restatement():
    t = synth_image(2, 64, 80, seed=7)
    p = (t + 0.08 * (synth_image(2, 64, 80, seed=8) - 0.5)).clamp(0, 1)
    got = vif_per_image(p, t)
    for b in range(2):
        want = np.mean([_vif_numpy(p[b, c].double().numpy(), t[b, c].double().numpy()) for c in range(3)])
        assert math.isclose(float(got[b]), want, rel_tol=1e-9), (float(got[b]), want)
    assert torch.allclose(vif_per_image(t, t), torch.ones(2, dtype=torch.float64), atol=1e-9)  # identical images: fidelity 1
    blurred = torch.nn.functional.avg_pool2d(t, 3, stride=1, padding=1)
    assert float(vif_per_image(blurred, t).max()) < 1.0   # lost detail = lost information
    m = VIF()
    m.update(p, t)
    assert math.isclose(m.compute(), float(got.mean()), rel_tol=1e-12)
    with pytest.raises(ValueError, match="41"):
        vif_per_image(t[:, :, :32, :32], t[:, :, :32, :32])


def test_evaluate_loop_with_a_stand_in_model():
    class Nearest:
        def upscale(self, x):
            return torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")

    hr = synth_image(3, 32, 32, seed=5)
    lr = hr[:, :, ::2, ::2]
    r = evaluate(Nearest(), [(lr[:2], hr[:2]), (lr[2:], hr[2:])])
    assert r["images"] == 3 and 5.0 < r["psnr"] < 40.0 and -1.0 <= r["ssim"] <= 1.0 and r["vif"] is None  # 32 x 32 < 41


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
    assert r["images"] == 2 and r["psnr"] > 100.0 and r["ssim"] > 0.999999 and abs(r["vif"] - 1.0) < 1e-3
