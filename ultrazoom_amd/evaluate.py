"""PSNR / SSIM evaluation of `MewZoom.upscale` -- the validation loop of the reference's training scripts
(pretrain.py:209-211, 301-329; fine-tune.py:231-233) without its data loaders.

The reference takes both metrics from `torchmetrics` (absent from this image), with these settings, restated here:
  * `PeakSignalNoiseRatio(data_range=1.0)`: 10 log10(1 / MSE), the squared error and the element count accumulated
    over ALL updates before the division (one global MSE, not a mean of per-image PSNRs);
  * `StructuralSimilarityIndexMeasure()` defaults: 11x11 Gaussian window, sigma 1.5, k1 0.01, k2 0.03, the images
    reflect-padded by 5 before filtering and the border cropped again, per-image mean of the SSIM map, mean over
    images; `data_range=None` = max(range of preds, range of target) of the batch at hand.
The arithmetic is plain torch and runs wherever the tensors live; it is the evaluation harness, not part of the
kernel path.  VIF (pretrain.py:211) is not restated."""

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


@torch.inference_mode()
def evaluate(model, pairs: Iterable[Tuple[Tensor, Tensor]]) -> dict:
    """`pairs` yields (low-resolution input, high-resolution target) batches already on the model's device/dtype;
    returns {"psnr": ..., "ssim": ..., "images": n} exactly as the reference's test loop accumulates them."""
    psnr, ssim = PSNR(1.0), SSIM()
    n = 0
    for x, y in pairs:
        sr = model.upscale(x)
        psnr.update(sr, y)
        ssim.update(sr, y)
        n += x.shape[0]
    return {"psnr": psnr.compute(), "ssim": ssim.compute(), "images": n}
