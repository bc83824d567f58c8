"""Repository rules that keep the parity claims honest."""

import re
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def test_product_package_never_touches_the_oracle_or_the_reference():
    for path in (REPO / "ultrazoom_amd").rglob("*"):
        if path.suffix not in {".py", ".cpp", ".hip", ".h", ".sh"}:
            continue
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{path} imports the oracle"
        assert "/root/reference" not in text, f"{path} reads the reference at run time"
        if path.name == "evaluate.py":
            # the PSNR / SSIM harness measures images that were ALREADY upscaled (its Gaussian window is a torch
            # convolution); it is not on the upscale path and must not carry any of the model's arithmetic
            assert "def upscale" not in text and "def forward" not in text
            continue
        assert "torch.nn.functional" not in text and "F.conv2d" not in text, f"{path} has a PyTorch compute path"


def test_gpu_side_files_do_not_read_the_reference():
    for rel in ("bench.py", "__graft_entry__.py"):
        p = REPO / rel
        if p.exists():
            assert "/root/reference" not in p.read_text()
    for p in (REPO / "tests").glob("test_*.py"):
        if p.name == "test_layout.py":
            continue
        assert "/root/reference" not in p.read_text(), p
