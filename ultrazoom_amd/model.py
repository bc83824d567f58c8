"""Drop-in ``MewZoom`` whose forward pass runs on the hand-written gfx950 kernels.

Mirrors the Python interface of the reference model (src/ultrazoom/model.py:43-192 of
andrewdalpino/UltraZoom v0.3.0): same constructor keyword arguments, same ``state_dict`` key names
and shapes (so ``PyTorchModelHubMixin.from_pretrained`` / ``load_state_dict`` work unchanged), same
``forward`` / ``upscale`` / ``predict_degredation`` methods, same ``AssertionError``s.

The module tree below only HOLDS parameters; it contains no PyTorch arithmetic.  All compute goes
through the C ABI of ``libmewzoom_hip.so``.  There is no CPU or eager-PyTorch fallback: calling the
model with a non-GPU tensor raises.
"""

from __future__ import annotations

import math
from math import ceil, floor, log2
from typing import Dict, Iterable, Optional, Tuple

import torch
from torch import Tensor, nn
from huggingface_hub import PyTorchModelHubMixin

from . import _ffi


# --------------------------------------------------------------------------------------------
# parameter containers (names match the reference module tree)
# --------------------------------------------------------------------------------------------
class _ConvParams(nn.Module):
    """Holds ``weight`` (and optionally ``bias``) of a Conv2d; never called."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool = False):
        super().__init__()
        assert cin > 0, "Input channels must be greater than 0."
        assert cout > 0, "Output channels must be greater than 0."
        w = torch.empty(cout, cin, k, k)
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(w, -bound, bound)  # same distribution as torch's Conv2d default
        self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))


class _Holder(nn.Module):
    """A module with a single ``conv`` child, e.g. ``stem.conv`` / ``downsample1.conv``."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool = False):
        super().__init__()
        self.conv = _ConvParams(cin, cout, k, bias)


class _MixParams(nn.Module):  # AdaptiveResidualMix, model.py:795-839
    def __init__(self, c: int):
        super().__init__()
        self.conv = _ConvParams(2 * c, c, 1)
        self.alpha = nn.Parameter(torch.tensor(0.0))


class _BottleneckParams(nn.Module):  # InvertedBottleneck, model.py:731-778
    def __init__(self, c: int, hidden_ratio: int):
        super().__init__()
        assert c > 0, "Number of channels must be greater than 0."
        assert hidden_ratio in {1, 2, 4}, "Hidden ratio must be either 1, 2, or 4."
        self.conv1 = _ConvParams(c, hidden_ratio * c, 3)
        self.conv2 = _ConvParams(hidden_ratio * c, c, 3)


class _BlockParams(nn.Module):  # EncoderBlock / DecoderBlock, model.py:487-511
    def __init__(self, c: int, hidden_ratio: int):
        super().__init__()
        self.convnet = _BottleneckParams(c, hidden_ratio)
        self.skip = _MixParams(c)


def _stage(c: int, n: int, hr: int) -> nn.ModuleList:
    return nn.ModuleList([_BlockParams(c, hr) for _ in range(n)])


class _EncoderParams(nn.Module):  # model.py:326-392
    def __init__(self, ch, layers, hr, num_deg_features):
        super().__init__()
        names = ("primary", "secondary", "tertiary", "quaternary")
        for n, l in zip(names, layers):
            assert l > 0, f"Number of {n} layers must be greater than 0."
        assert num_deg_features > 0, "Number of quality assessor features must be greater than 0."
        for i in range(4):
            setattr(self, f"stage{i + 1}", _stage(ch[i], layers[i], hr))
        for i in range(3):
            setattr(self, f"downsample{i + 1}", _Holder(ch[i], ch[i + 1], 2))
        self.qa_head = _Holder(ch[3], num_deg_features, 3, bias=True)


class _DecoderParams(nn.Module):  # model.py:514-575; stage1 is the coarsest level
    def __init__(self, ch, layers, hr):
        super().__init__()
        for i in range(4):
            setattr(self, f"stage{i + 1}", _stage(ch[3 - i], layers[3 - i], hr))
        for i in range(3):
            setattr(self, f"upsample{i + 1}", _Holder(ch[3 - i], 4 * ch[2 - i], 3))
        for i in range(3):
            setattr(self, f"skip{i + 1}", _MixParams(ch[2 - i]))


class _UNetParams(nn.Module):  # model.py:245-300
    def __init__(self, ch, layers, hr, num_deg_features):
        super().__init__()
        names = ("primary", "secondary", "tertiary", "quaternary")
        for n, l in zip(names, layers):
            assert l > 1, f"Number of {n} layers must be greater than 1."
        self.encoder = _EncoderParams(ch, [ceil(l / 2) for l in layers], hr, num_deg_features)
        self.decoder = _DecoderParams(ch, [floor(l / 2) for l in layers], hr)


class _SR2XParams(nn.Module):  # SR2XBlock, model.py:975-983
    def __init__(self, c: int, hr: int, cout: int):
        super().__init__()
        self.refiner = _BlockParams(c, hr)
        self.upscale = _Holder(c, 4 * cout, 3)


class _HeadParams(nn.Module):  # SuperResolver, model.py:933-954
    def __init__(self, c: int, hr: int, ratio: int):
        super().__init__()
        assert ratio in {2, 4, 8}, "Upscale ratio must be either 2, 4, or 8."
        n = int(log2(ratio))
        self.layers = nn.ModuleList([_SR2XParams(c, hr, c) for _ in range(n - 1)] + [_SR2XParams(c, hr, 3)])


# --------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------
class MewZoom(nn.Module, PyTorchModelHubMixin):
    """Image super-resolution U-Net with adaptive residual connections, computed on MI355X.

    Constructor arguments are the reference's (model.py:51-64).
    """

    AVAILABLE_UPSCALE_RATIOS = {2, 4, 8}

    def __init__(
        self,
        upscale_ratio: int,
        primary_channels: int,
        primary_layers: int,
        secondary_channels: int,
        secondary_layers: int,
        tertiary_channels: int,
        tertiary_layers: int,
        quaternary_channels: int,
        quaternary_layers: int,
        hidden_ratio: int,
        num_deg_features: int,
    ):
        super().__init__()

        assert (
            upscale_ratio in self.AVAILABLE_UPSCALE_RATIOS
        ), f"Upscale ratio must be one of {self.AVAILABLE_UPSCALE_RATIOS}, but got {upscale_ratio}."
        assert 3 < primary_channels, "Output channels must be greater than input channels."

        self._cfg = dict(
            upscale_ratio=upscale_ratio,
            primary_channels=primary_channels,
            primary_layers=primary_layers,
            secondary_channels=secondary_channels,
            secondary_layers=secondary_layers,
            tertiary_channels=tertiary_channels,
            tertiary_layers=tertiary_layers,
            quaternary_channels=quaternary_channels,
            quaternary_layers=quaternary_layers,
            hidden_ratio=hidden_ratio,
            num_deg_features=num_deg_features,
        )
        ch = (primary_channels, secondary_channels, tertiary_channels, quaternary_channels)
        layers = (primary_layers, secondary_layers, tertiary_layers, quaternary_layers)

        self.stem = _Holder(3, primary_channels, 1, bias=True)
        self.unet = _UNetParams(ch, layers, hidden_ratio, num_deg_features)
        self.head = _HeadParams(primary_channels, hidden_ratio, upscale_ratio)
        self.upscale_ratio = upscale_ratio

        # how many images of a batch are in flight at once inside the library (0 = its default)
        self.max_images_in_flight = 0
        self._engine: Optional[_Engine] = None
        # checkpoint-recipe compatibility (add_weight_norms / add_lora_adapters, below)
        self._expects_weight_norm = False
        self._lora_alphas: list = []

    # ---- bookkeeping identical to the reference -------------------------------------------
    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    @property
    def num_trainable_params(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def freeze_parameters(self) -> None:
        for p in self.parameters():
            p.requires_grad = False

    def initialize_weights(self) -> None:
        """Kaiming-uniform initialisation of every convolution weight (what model.py:104-109 intends; the reference's own
        method stops with an AttributeError at model.py:413)."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("conv.weight") or ".convnet.conv" in name:
                    nn.init.kaiming_uniform_(p)
        self.refresh_weights()

    def enable_activation_checkpointing(self) -> None:
        """Accepted for interface compatibility (model.py:141-147): a training-memory option; this class records no
        autograd graph, so there is nothing to recompute."""

    # ---- the reference's checkpoint recipe (test_compare.py:32-45, validate.py:57-67) ---------------------
    # `model.add_weight_norms()` [-> `model.add_lora_adapters(rank, alpha)`] -> `model.load_state_dict(raw)` ->
    # `model.remove_parameterizations()` runs unchanged against this class: the add_* calls record which training-time
    # keys to expect, `load_state_dict` bakes them (w = g * v / ||v||, + alpha * A @ B) into the plain `conv.weight`
    # tensors this model holds, and `remove_parameterizations` has nothing left to do.  Unlike the reference, the module
    # tree never holds parametrised weights (it is inference-only), so `state_dict()` always has the baked layout.
    def add_weight_norms(self) -> None:  # model.py:117-122
        self._expects_weight_norm = True

    def add_lora_adapters(self, rank: int, alpha: float) -> None:  # model.py:124-129
        assert rank > 0, "Rank must be greater than 0."
        assert alpha > 0.0, "Alpha must be greater than 0."
        self._lora_alphas.append(float(alpha))

    def remove_parameterizations(self) -> None:  # model.py:131-139
        self._expects_weight_norm = False
        self._lora_alphas = []

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """`nn.Module.load_state_dict`, also accepting the training-time layouts announced by `add_weight_norms` /
        `add_lora_adapters` (weight-norm `parametrizations.weight.original0/1`, LoRA `lora_a/lora_b`) and `_orig_mod.`
        prefixes, which are baked on the way in."""
        if any(".parametrizations." in k or k.startswith("_orig_mod.") for k in state_dict):
            has_lora = any(k.endswith((".lora_a", ".lora_b")) for k in state_dict)
            if has_lora and not self._lora_alphas:
                raise RuntimeError(
                    "Error(s) in loading state_dict for MewZoom: the checkpoint holds LoRA adapter tensors (lora_a / lora_b) "
                    "but add_lora_adapters(rank, alpha) was not called: alpha is not stored in a checkpoint."
                )
            state_dict = bake_state_dict(state_dict, self._lora_alphas or None)
        result = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self.refresh_weights()
        return result

    # ---- engine management ------------------------------------------------------------------
    def refresh_weights(self) -> None:
        """Drops the packed copy of the weights inside the library; the next call re-packs from the parameters.
        Needed only after writes the version counter cannot see (`p.data.copy_()` / `p.data.mul_()`, as EMA loops and some
        optimisers do); ordinary in-place updates, `load_state_dict` and `.to()` are detected automatically."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def _weights_signature(self) -> Tuple:
        def version(p):
            try:
                return p._version
            except RuntimeError:  # inference tensors (created under torch.inference_mode) are not version-tracked
                return -1
        return tuple((p.data_ptr(), version(p), p.dtype, p.device) for p in self.parameters())

    # the engine owns device memory through a C handle: copies / pickles of the module start without one (it is rebuilt
    # lazily by _get_engine), so copy.deepcopy(model), pickle and torch.save(model) keep working after a forward
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        return state

    def _get_engine(self, x: Tensor) -> "_Engine":
        if not x.is_cuda:
            raise RuntimeError(
                "ultrazoom_amd.MewZoom computes on an MI355X only: move the model and the input to a 'cuda' device. "
                "There is no CPU path."
            )
        p0 = next(self.parameters())
        if p0.device != x.device:
            raise RuntimeError(f"model is on {p0.device} but the input is on {x.device}")
        if x.dtype != p0.dtype:
            raise RuntimeError(f"Input type ({x.dtype}) and weight type ({p0.dtype}) should be the same")
        sig = self._weights_signature()
        if self._engine is None or self._engine.signature != sig:
            if self._engine is not None:
                self._engine.close()
            self._engine = _Engine(self._cfg, self.state_dict(), p0.dtype, x.device, sig)
        return self._engine

    def _run(self, x: Tensor, clamp: bool, want_qa: bool):
        assert x.dim() == 4 and x.shape[1] == 3, "expected a (B, 3, H, W) tensor"
        engine = self._get_engine(x)
        return engine.run(x.contiguous(), clamp, want_qa, self.max_images_in_flight)

    # ---- the reference's public methods -----------------------------------------------------
    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        """Returns ``(s + z, z_qa)``: the un-clamped super-resolved image and the degradation features
        (model.py:149-164).  Inference only: no autograd graph is recorded."""
        sr, qa = self._run(x, clamp=False, want_qa=True)
        return sr, qa.to(x.dtype)

    @torch.inference_mode()
    def upscale(self, x: Tensor) -> Tensor:
        """``clamp(forward(x)[0], 0, 1)`` (model.py:166-179); the quality head is skipped."""
        sr, _ = self._run(x, clamp=True, want_qa=False)
        return sr

    @torch.inference_mode()
    def predict_degredation(self, x: Tensor) -> Tensor:  # (sic) model.py:181-192
        _, qa = self._run(x, clamp=False, want_qa=True)
        return qa.to(x.dtype)

    @torch.inference_mode()
    def upscale_uint8(self, x: Tensor) -> Tensor:
        """uint8 in, uint8 out: ``save_image``-style rounding of ``upscale(x / 255)``.

        What every caller of the reference does around ``upscale`` (``ToDtype(float32, scale=True)`` before,
        ``save_image`` after; README.md:72-83, test_compare.py:53-57,89), with both conversions fused into the first
        and last kernels."""
        assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.uint8, "expected a (B, 3, H, W) uint8 tensor"
        if not x.is_cuda:
            raise RuntimeError("ultrazoom_amd.MewZoom computes on an MI355X only: move the input to a 'cuda' device.")
        p0 = next(self.parameters())
        engine = self._get_engine(torch.empty(0, dtype=p0.dtype, device=x.device))
        return engine.run_u8(x.contiguous(), self.max_images_in_flight)

    # ---- checkpoint ingestion (test_compare.py:32-45 of the reference) -------------------------
    def load_training_checkpoint(self, state_dict: Dict[str, Tensor], lora_alpha: Optional[float] = None) -> None:
        """Loads a raw training checkpoint: strips ``_orig_mod.`` prefixes left by torch.compile, bakes weight-norm
        parametrisations (w = g * v / ||v||) and merges LoRA adapters (needs ``lora_alpha``) into plain
        ``conv.weight`` tensors -- what the reference does with ``add_weight_norms`` / ``add_lora_adapters`` ->
        ``load_state_dict`` -> ``remove_parameterizations`` (test_compare.py:32-45)."""
        nn.Module.load_state_dict(self, bake_state_dict(state_dict, lora_alpha))
        self.refresh_weights()


def bake_state_dict(state_dict: Dict[str, Tensor], lora_alpha=None) -> Dict[str, Tensor]:
    """Turns a training-time state_dict into the baked layout this model (and the HF export) uses: what the
    reference's `remove_parameterizations()` leaves behind (model.py:131-139, test_compare.py:32-45).

    * `_orig_mod.` prefixes of torch.compile'd modules are stripped;
    * weight norm (`add_weight_norms`, model.py:117-122): `w = g * v / ||v||`, norm over all dims but 0;
    * LoRA adapters (`add_lora_adapters`, model.py:124-129; `ChannelLoRA.forward`, model.py:1361-1390):
      `w += alpha * (A @ B).permute(2, 3, 0, 1)` with A `[kh, kw, out, rank]`, B `[kh, kw, rank, in]`.  `alpha` is a
      Python attribute of the adapter, not a tensor, so it is not in the checkpoint: pass it as `lora_alpha` (one float,
      or one per `add_lora_adapters` call in registration order).
    Parametrizations apply in registration order (weight norm can only be the first one)."""
    sd = {k.replace("_orig_mod.", ""): v for k, v in state_dict.items()}
    marker = ".parametrizations.weight."
    out: Dict[str, Tensor] = {}
    groups: Dict[str, Dict[str, Tensor]] = {}
    for k, v in sd.items():
        if marker in k:
            base, leaf = k.split(marker, 1)
            groups.setdefault(base, {})[leaf] = v
        else:
            out[k] = v
    for base, leaves in groups.items():
        if "original" in leaves:
            w = leaves["original"].clone()
        elif "original0" in leaves and "original1" in leaves:
            g, vv = leaves["original0"], leaves["original1"]
            norm = vv.flatten(1).norm(dim=1).reshape(-1, *([1] * (vv.dim() - 1)))
            w = g * vv / norm
        else:
            raise KeyError(f"{base}: parametrized weight without its original tensor(s): {sorted(leaves)}")
        adapters = sorted({int(leaf.split(".")[0]) for leaf in leaves if leaf.endswith((".lora_a", ".lora_b"))})
        unknown = [leaf for leaf in leaves if not leaf.startswith("original") and not leaf.endswith((".lora_a", ".lora_b"))]
        if unknown:
            raise KeyError(f"{base}: unknown parametrization tensors {unknown}")
        for n, i in enumerate(adapters):
            if lora_alpha is None:
                raise ValueError("the checkpoint holds LoRA adapters: pass lora_alpha (the `alpha` given to add_lora_adapters)")
            if isinstance(lora_alpha, (list, tuple)):
                if n >= len(lora_alpha):
                    raise ValueError(f"{base}: {len(adapters)} LoRA adapters in the checkpoint but only {len(lora_alpha)} alphas given")
                alpha = float(lora_alpha[n])
            else:
                alpha = float(lora_alpha)
            a, b = leaves[f"{i}.lora_a"], leaves[f"{i}.lora_b"]
            w = w + alpha * (a.to(w.dtype) @ b.to(w.dtype)).permute(2, 3, 0, 1)
        out[base + ".weight"] = w
    return out


class _Engine:
    """One mz_handle plus its packed weights and cached workspaces for a (dtype, device)."""

    def __init__(self, config: dict, state_dict: Dict[str, Tensor], dtype, device, signature):
        self.signature = signature
        self.dtype = dtype
        self.device = device
        self.handle = _ffi.Handle(config, _ffi.dtype_code(dtype))
        self.config = config
        self._workspace: Optional[Tensor] = None
        # the workspace is reused by every call: a call on another stream than the previous one waits for it
        self._last_stream = None
        self._last_done: Optional[torch.cuda.Event] = None
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device)
            names = dict(self.handle.weight_infos())
            missing = [k for k in names if k not in state_dict]
            if missing:
                raise KeyError(f"state_dict lacks {missing[:3]} ...")
            for name, shape in names.items():
                t = state_dict[name].detach()
                if tuple(t.shape) != shape:
                    raise ValueError(f"{name}: shape {tuple(t.shape)} != expected {shape}")
                t32 = t.to(device=device, dtype=torch.float32).contiguous()
                self.handle.set_weight(name, t32.data_ptr(), shape, stream.cuda_stream)
                # keep t32 alive until the packing kernel has consumed it
                t32.record_stream(stream)
            self.handle.weights_complete()
            stream.synchronize()

    def close(self) -> None:
        self.handle.close()
        self._workspace = None

    def _workspace_for(self, need: int, stream) -> Tensor:
        if self._last_done is not None and self._last_stream != stream:
            stream.wait_event(self._last_done)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace

    def _mark_done(self, stream) -> None:
        if self._last_done is None:
            self._last_done = torch.cuda.Event()
        self._last_done.record(stream)
        self._last_stream = stream

    def run(self, x: Tensor, clamp: bool, want_qa: bool, max_in_flight: int):
        B, _, H, W = x.shape
        r = self.config["upscale_ratio"]
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            ws = self._workspace_for(self.handle.workspace_bytes(B, H, W, max_in_flight), stream)
            sr = torch.empty((B, 3, H * r, W * r), dtype=self.dtype, device=self.device)
            qa = torch.empty((B, self.config["num_deg_features"]), dtype=torch.float32, device=self.device) if want_qa else None
            self.handle.forward(
                x.data_ptr(), sr.data_ptr(), qa.data_ptr() if want_qa else 0, B, H, W, clamp,
                ws.data_ptr(), ws.numel(), max_in_flight, stream.cuda_stream,
            )
            self._mark_done(stream)
        return sr, qa

    def run_u8(self, x: Tensor, max_in_flight: int) -> Tensor:
        B, _, H, W = x.shape
        r = self.config["upscale_ratio"]
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            ws = self._workspace_for(self.handle.workspace_bytes(B, H, W, max_in_flight), stream)
            sr = torch.empty((B, 3, H * r, W * r), dtype=torch.uint8, device=self.device)
            self.handle.forward_u8(x.data_ptr(), sr.data_ptr(), 0, B, H, W, ws.data_ptr(), ws.numel(), max_in_flight,
                                   stream.cuda_stream)
            self._mark_done(stream)
        return sr
