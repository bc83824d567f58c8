"""MI355X-native implementation of the MewZoom upscale path (andrewdalpino/UltraZoom v0.3.0).

    from ultrazoom_amd import MewZoom          # drop-in for ultrazoom.model.MewZoom
    model = MewZoom.from_pretrained(...).to("cuda").eval()
    y = model.upscale(x)

All arithmetic runs in hand-written gfx950 kernels behind the C ABI in include/mewzoom_hip.h.
"""

from .model import MewZoom, bake_state_dict  # noqa: F401

__all__ = ["MewZoom", "bake_state_dict"]
