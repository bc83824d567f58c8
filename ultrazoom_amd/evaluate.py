"""PSNR / SSIM evaluation of `MewZoom.upscale` -- the validation loop of the reference's training scripts
(pretrain.py:209-211, 301-329; fine-tune.py:231-233) without its data loaders.

The reference takes both metrics from `torchmetrics` (absent from this image), with these settings, restated here:
  * `PeakSignalNoiseRatio(data_range=1.0)`: 10 log10(1 / MSE), the squared error and the element count accumulated
    over ALL updates before the division (one global MSE, not a mean of per-image PSNRs);
  * `StructuralSimilarityIndexMeasure()` defaults: 11x11 Gaussian window, sigma 1.5, k1 0.01, k2 0.03, the images
    reflect-padded by 5 before filtering and the border cropped again, per-image mean of the SSIM map, mean over
    images; `data_range=None` = max(range of preds, range of target) of the batch at hand.
  * `VisualInformationFidelity()` defaults (pretrain.py:211, 311-316): pixel-domain VIF (Sheikh & Bovik 2006) with
    `sigma_n_sq = 2.0`, four scales with Gaussian windows of 17, 9, 5, 3 taps (sigma = taps / 5), "valid" filtering, the
    coarser scales low-pass filtered and decimated by two; per channel, mean over channels, mean over images.  Images
    must be at least 41 pixels in both directions.
The arithmetic is plain torch and runs wherever the tensors live; it is the evaluation harness, not part of the
kernel path.  torchmetrics is absent from this image: the three metrics are checked against independent numpy / scipy
computations of the same published definitions ("parity unpinned" for the metrics themselves)."""

from __future__ import annotations

import math
from typing import Iterable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


class PSNR:
    def __init__(self, data_range: float = 1.0):
        self.data_range = float(data_range)
        self.reset()

    def reset(self) -> None:
        self.sq = 0.0
        self.n = 0

    def update(self, pred: Tensor, target: Tensor) -> None:
        d = pred.double() - target.double()
        self.sq += float((d * d).sum())
        self.n += d.numel()

    def compute(self) -> float:
        if self.n == 0:
            return float("nan")
        mse = self.sq / self.n
        return float("inf") if mse == 0.0 else 10.0 * math.log10(self.data_range**2 / mse)


def _gaussian_window(size: int, sigma: float, device, dtype) -> Tensor:
    x = torch.arange(size, device=device, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def ssim_per_image(pred: Tensor, target: Tensor, data_range: Optional[float] = None, size: int = 11, sigma: float = 1.5,
                   k1: float = 0.01, k2: float = 0.03) -> Tensor:
    """SSIM of each image of a [B, C, H, W] batch (Wang et al. 2004, Gaussian window)."""
    if pred.shape != target.shape or pred.dim() != 4:
        raise ValueError("expected two [B, C, H, W] tensors of one shape")
    p, t = pred.double(), target.double()
    if data_range is None:
        data_range = float(max(p.max() - p.min(), t.max() - t.min()))
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    pad = (size - 1) // 2
    C = p.shape[1]
    g = _gaussian_window(size, sigma, p.device, p.dtype)
    win = (g[:, None] * g[None, :]).expand(C, 1, size, size).contiguous()

    def blur(z: Tensor) -> Tensor:
        z = F.pad(z, (pad, pad, pad, pad), mode="reflect")
        return F.conv2d(z, win, groups=C)

    mu_p, mu_t = blur(p), blur(t)
    s_pp = blur(p * p) - mu_p * mu_p
    s_tt = blur(t * t) - mu_t * mu_t
    s_pt = blur(p * t) - mu_p * mu_t
    m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    m = m[..., pad:-pad, pad:-pad]  # the padded border is cropped again
    return m.reshape(m.shape[0], -1).mean(dim=1)


class SSIM:
    def __init__(self, data_range: Optional[float] = None):
        self.data_range = data_range
        self.reset()

    def reset(self) -> None:
        self.total = 0.0
        self.n = 0

    def update(self, pred: Tensor, target: Tensor) -> None:
        v = ssim_per_image(pred, target, self.data_range)
        self.total += float(v.sum())
        self.n += v.numel()

    def compute(self) -> float:
        return self.total / self.n if self.n else float("nan")


def _gaussian_kernel2d(size: int, sigma: float, device, dtype) -> Tensor:
    x = torch.arange(size, device=device, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(x[:, None] ** 2 + x[None, :] ** 2) / (2.0 * sigma * sigma))
    return g / g.sum()


def vif_per_image(pred: Tensor, target: Tensor, sigma_n_sq: float = 2.0) -> Tensor:
    """Pixel-domain visual information fidelity of each image of a [B, C, H, W] batch (mean over channels)."""
    if pred.shape != target.shape or pred.dim() != 4:
        raise ValueError("expected two [B, C, H, W] tensors of one shape")
    if pred.shape[-1] < 41 or pred.shape[-2] < 41:
        raise ValueError(f"VIF needs images of at least 41 x 41 pixels, got {tuple(pred.shape[-2:])}")
    eps = 1e-10
    B, C = pred.shape[:2]
    p = pred.double().reshape(B * C, 1, *pred.shape[2:])
    t = target.double().reshape(B * C, 1, *target.shape[2:])
    num = torch.zeros(B * C, dtype=torch.float64, device=p.device)
    den = torch.zeros(B * C, dtype=torch.float64, device=p.device)
    for scale in range(4):
        n = 2 ** (4 - scale) + 1
        k = _gaussian_kernel2d(n, n / 5.0, p.device, p.dtype)[None, None]
        if scale > 0:
            t = F.conv2d(t, k)[:, :, ::2, ::2]
            p = F.conv2d(p, k)[:, :, ::2, ::2]
        mu_t, mu_p = F.conv2d(t, k), F.conv2d(p, k)
        s_tt = (F.conv2d(t * t, k) - mu_t * mu_t).clamp(min=0.0)
        s_pp = (F.conv2d(p * p, k) - mu_p * mu_p).clamp(min=0.0)
        s_tp = F.conv2d(t * p, k) - mu_t * mu_p
        g = s_tp / (s_tt + eps)
        s_v = s_pp - g * s_tp
        m = s_tt < eps
        g = torch.where(m, torch.zeros_like(g), g)
        s_v = torch.where(m, s_pp, s_v)
        s_tt = torch.where(m, torch.zeros_like(s_tt), s_tt)
        m = s_pp < eps
        g = torch.where(m, torch.zeros_like(g), g)
        s_v = torch.where(m, torch.zeros_like(s_v), s_v)
        m = g < 0
        s_v = torch.where(m, s_pp, s_v)
        g = torch.where(m, torch.zeros_like(g), g)
        s_v = s_v.clamp(min=eps)
        num = num + torch.log10(1.0 + g * g * s_tt / (s_v + sigma_n_sq)).sum(dim=(1, 2, 3))
        den = den + torch.log10(1.0 + s_tt / sigma_n_sq).sum(dim=(1, 2, 3))
    return (num / den).reshape(B, C).mean(dim=1)


class VIF:
    def __init__(self, sigma_n_sq: float = 2.0):
        self.sigma_n_sq = float(sigma_n_sq)
        self.reset()

    def reset(self) -> None:
        self.total = 0.0
        self.n = 0

    def update(self, pred: Tensor, target: Tensor) -> None:
        v = vif_per_image(pred, target, self.sigma_n_sq)
        self.total += float(v.sum())
        self.n += v.numel()

    def compute(self) -> float:
        return self.total / self.n if self.n else float("nan")


@torch.inference_mode()
def evaluate(model, pairs: Iterable[Tuple[Tensor, Tensor]]) -> dict:
    """`pairs` yields (low-resolution input, high-resolution target) batches already on the model's device/dtype;
    returns {"psnr": ..., "ssim": ..., "vif": ..., "images": n} as the reference's test loop accumulates them (pretrain.py:301-329;
    "vif" is None when the images are smaller than the 41 x 41 pixels the metric needs)."""
    psnr, ssim, vif = PSNR(1.0), SSIM(), VIF()
    n = 0
    for x, y in pairs:
        sr = model.upscale(x)
        psnr.update(sr, y)
        ssim.update(sr, y)
        if min(y.shape[-2:]) >= 41:
            vif.update(sr, y)
        n += x.shape[0]
    return {"psnr": psnr.compute(), "ssim": ssim.compute(), "vif": vif.compute() if vif.n else None, "images": n}
