"""Shared helpers for the golden-vector tests (CPU and GPU sides)."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from ultrazoom_amd.synth import synth_image, synth_state_dict

GOLDEN = Path(__file__).resolve().parent / "golden"

MODEL_CASES = sorted(p.stem for p in GOLDEN.glob("g*.npz") if p.stem not in ("g6_ops", "g10_checkpoint"))


class GoldenCase:
    def __init__(self, name: str):
        self.name = name
        self.data = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
        self.meta = json.loads(str(self.data["meta"]))
        self.config = self.meta["config"]
        self.B, self.H, self.W = self.meta["input"]

    def weights(self):
        shapes = {k: tuple(v) for k, v in self.meta["shapes"].items()}
        return synth_state_dict(shapes, self.meta["weight_seed"])

    def image(self) -> torch.Tensor:
        return synth_image(self.B, self.H, self.W, self.meta["image_seed"])

    @property
    def sampled(self) -> bool:
        return "sr_idx" in self.data.files

    def compare_sr(self, sr: torch.Tensor, up: torch.Tensor | None = None):
        """max-abs error of `sr` (and optionally the clamped `up`) against the reference's output."""
        sr = sr.detach().float().cpu()
        errs = {}
        if self.sampled:
            idx = torch.from_numpy(self.data["sr_idx"])
            errs["sr"] = (sr.reshape(-1)[idx] - torch.from_numpy(self.data["sr_samples"])).abs().max().item()
            errs["chan_mean"] = (
                (sr.double().mean(dim=(0, 2, 3)) - torch.from_numpy(self.data["sr_chan_mean"])).abs().max().item()
            )
            if up is not None:
                up = up.detach().float().cpu()
                errs["up"] = (up.reshape(-1)[idx] - torch.from_numpy(self.data["up_samples"])).abs().max().item()
        else:
            errs["sr"] = (sr - torch.from_numpy(self.data["sr"])).abs().max().item()
            if up is not None:
                up = up.detach().float().cpu()
                errs["up"] = (up - torch.from_numpy(self.data["up"])).abs().max().item()
        return errs


def _mse_up(self, up: torch.Tensor) -> float:
    """MSE of a clamped result against the reference's fp32 `upscale` output (over the stored samples for sampled cases)."""
    up = up.detach().float().cpu()
    if self.sampled:
        idx = torch.from_numpy(self.data["sr_idx"])
        return (up.reshape(-1)[idx].double() - torch.from_numpy(self.data["up_samples"]).double()).pow(2).mean().item()
    return (up.double() - torch.from_numpy(self.data["up"]).double()).pow(2).mean().item()


GoldenCase.mse_up = _mse_up


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = (a.double() - b.double()).pow(2).mean().item()
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)
