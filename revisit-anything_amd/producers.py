"""Feature producer on PyTorch(-ROCm) host code: the DINOv2 patch-token block the hot
path consumes, ``/{image}/ift_dino`` float32 ``[1, D, h, w]``.

The reference builds it with ``DinoV2ExtractFeatures("dinov2_vitg14", 31, 'value', norm_descs=False)``
(func_vpr.py:530-532, utilities.py:219-288: a forward hook on ``blocks[31].attn.qkv`` whose output's last third is the
VALUE facet, CLS token dropped) inside ``getAnyLocFt`` / ``process_single_DINO`` (func_vpr.py:489-506, 549-562):
resize to the configured size, ``ToTensor`` + ImageNet normalisation, centre-crop to multiples of 14, extract,
reshape ``[1, h, w, D] -> [1, D, h, w]``.

``torch.hub`` is unreachable here, so the backbone is the ``transformers`` implementation of the same architecture
(``Dinov2Model``; the converted checkpoints ``facebook/dinov2-*`` hold the same weights).  It keeps q, k, v as separate
linears, so the value facet is simply the output of ``encoder.layer[L].attention.attention.value`` -- identical to the
last third of a fused qkv projection (tests/test_producers.py shows the identity).  With weights on disk
(``from_pretrained(local_dir)``) this reproduces the reference's tokens up to backbone numerics -- unpinned here: the
image has neither the weights nor the hub code; with ``from_config`` (random initialisation) it serves
throughput-only end-to-end runs (BASELINE configs[2])."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

# hidden size, layers, heads of the four published DINOv2 backbones (patch 14)
_ARCH = {"small": (384, 12, 6), "base": (768, 12, 12), "large": (1024, 24, 16), "giant": (1536, 40, 24)}


class DinoV2ValueFacet:
    """``extractor(img [B, 3, H, W] normalised, H and W multiples of 14) -> [B, (H/14)*(W/14), D]``: the value facet
    of transformer layer ``layer`` without the CLS token, not normalised (norm_descs=False, func_vpr.py:532)."""

    def __init__(self, model, layer: int = 31, device: Union[str, torch.device] = "cpu"):
        self.device = torch.device(device)
        self.model = model.eval().to(self.device)
        n_layers = len(self.model.encoder.layer)
        if not 0 <= layer < n_layers:
            raise ValueError(f"layer {layer} outside the model's {n_layers} layers")
        self.layer = layer
        self._out: Optional[torch.Tensor] = None
        self._handle = self.model.encoder.layer[layer].attention.attention.value.register_forward_hook(self._hook)
        self.n_prefix = 1 + int(getattr(self.model.config, "num_register_tokens", 0) or 0)   # CLS (+ registers)

    def _hook(self, module, inputs, output):
        self._out = output

    @classmethod
    def from_config(cls, size: str = "giant", layer: int = 31, device="cpu", **overrides) -> "DinoV2ValueFacet":
        """Random-initialised backbone of a published size (or any ``Dinov2Config`` fields via ``overrides``)."""
        from transformers import Dinov2Config, Dinov2Model

        hidden, layers, heads = _ARCH[size]
        kw = dict(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, patch_size=14, image_size=518,
                  mlp_ratio=4, use_swiglu_ffn=(size == "giant"))
        kw.update(overrides)
        return cls(Dinov2Model(Dinov2Config(**kw)), layer=layer, device=device)

    @classmethod
    def from_pretrained(cls, path: str, layer: int = 31, device="cpu") -> "DinoV2ValueFacet":
        """Weights staged on disk (a ``facebook/dinov2-*`` snapshot directory; there is no network here)."""
        from transformers import Dinov2Model

        return cls(Dinov2Model.from_pretrained(path, local_files_only=True), layer=layer, device=device)

    @torch.no_grad()
    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        if img.ndim != 4 or img.shape[2] % 14 or img.shape[3] % 14:
            raise ValueError(f"expected [B, 3, H, W] with H, W multiples of 14, got {tuple(img.shape)}")
        self._out = None
        self.model(pixel_values=img.to(self.device))
        assert self._out is not None, "no data from the hook"
        res = self._out[:, self.n_prefix:, :]
        self._out = None
        return res

    def close(self):
        self._handle.remove()


def center_crop_to_patches(img: torch.Tensor, patch: int = 14) -> torch.Tensor:
    """``tvf.CenterCrop((h // 14 * 14, w // 14 * 14))`` (func_vpr.py:497-500), same rounding of the offsets."""
    h, w = img.shape[-2:]
    hn, wn = (h // patch) * patch, (w // patch) * patch
    top, left = int(round((h - hn) / 2.0)), int(round((w - wn) / 2.0))
    return img[..., top:top + hn, left:left + wn]


def image_to_tokens(img_rgb: np.ndarray, extractor, cfg: Optional[dict] = None, normalize: bool = True) -> torch.Tensor:
    """``process_single_DINO`` + ``getAnyLocFt(..., upsample=False)`` (func_vpr.py:549-562, 489-506):
    ``img_rgb`` uint8 ``[H, W, 3]`` (already RGB) -> float32 ``[1, D, h, w]`` on the extractor's device,
    L2-normalised over the channel axis exactly as ``process_single_DINO`` returns it (func_vpr.py:561) -- i.e. what the
    reference writes to ``/{key}/ift_dino``.  ``normalize=False`` gives ``getAnyLocFt``'s raw value-facet map.
    ``cfg['resize']`` / ``desired_width`` / ``desired_height`` as in place_rec_global_config.py; the resize is bilinear
    with pixel-centre alignment like ``cv2.resize``'s default (cv2 rounds the result to uint8, so does this)."""
    a = np.asarray(img_rgb)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError(f"expected a uint8 [H, W, 3] RGB image, got {a.dtype} {a.shape}")
    x = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float()          # [3, H, W], 0..255
    if cfg and cfg.get("resize"):
        x = torch.nn.functional.interpolate(x[None], size=(int(cfg["desired_height"]), int(cfg["desired_width"])),
                                            mode="bilinear", align_corners=False, antialias=False)[0]
        x = x.round().clamp_(0, 255)
    x = x / 255.0                                                                   # ToTensor
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    x = (x - mean) / std
    x = center_crop_to_patches(x)[None]
    hr, wr = x.shape[2] // 14, x.shape[3] // 14
    feat = extractor(x)                                                             # [1, hr*wr, D]
    feat = feat.reshape(1, hr, wr, -1).permute(0, 3, 1, 2).contiguous().float()     # [1, D, hr, wr]
    return torch.nn.functional.normalize(feat, dim=1) if normalize else feat
