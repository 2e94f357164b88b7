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
throughput-only end-to-end runs (BASELINE configs[2]).

The second half of the row, the SAM automatic mask generator at half resolution, is ``SamAutoMasks`` further down."""
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


def process_single_DINO(cfg: dict, img_bgr: np.ndarray, extractor) -> tuple:
    """func_vpr.py:549-562: BGR -> RGB, resize when cfg['resize'], value-facet tokens L2-normalised over channels;
    returns (the processed RGB image, float32 ``[1, D, h, w]``)."""
    img = np.ascontiguousarray(np.asarray(img_bgr)[:, :, ::-1])
    img_p = resize_like_cv2(img, cfg["desired_width"], cfg["desired_height"]) if cfg.get("resize") else img
    return img_p, image_to_tokens(img_p, extractor, None, normalize=True)


def dino_given_image(extractor, img_bgr: np.ndarray, cfg: dict) -> torch.Tensor:
    """func_vpr.py:626-644 with the image already decoded: crop rows ``rmin:`` (the caller-side crop every reference
    loader applies before process_single_DINO -- also process_dino_ft_to_h5, func_vpr.py:647-662), resize to the
    configured DINO resolution, tokens on the CPU like the reference returns them."""
    im = np.asarray(img_bgr)[int(cfg.get("rmin", 0)):, :, :]
    cfg_dino = {"desired_width": cfg["desired_width"], "desired_height": cfg["desired_height"], "resize": True}
    return process_single_DINO(cfg_dino, im, extractor)[1].detach().cpu()


# =====================================================================================================================
# SAM automatic masks (second half)
# =====================================================================================================================
# The reference produces ``/{image}/masks/{j}/segmentation`` with ``SamAutomaticMaskGenerator(sam_vit_h)`` at its
# defaults (func_vpr.py:510-516) on the image resized to HALF the DINO resolution (place_rec_SAM_DINO.py:51-63,
# func_vpr.py:564-590).  ``segment_anything`` needs torchvision (absent here), so the generator below restates its
# published algorithm (sam/segment_anything/automatic_mask_generator.py:137-330, utils/amg.py) on the ``transformers``
# implementation of the same network (``SamModel``; the converted ``facebook/sam-vit-*`` checkpoints hold the same
# weights):  point grid -> 3 masks per point -> predicted-IoU filter -> stability-score filter -> threshold -> boxes ->
# box NMS (by predicted IoU) -> records.  crop_n_layers = 0 (the reference's default): one crop = the whole image.
# With weights staged on disk it reproduces the reference's masks up to backbone numerics (unpinned here: no weights);
# random-initialised it serves throughput-only end-to-end runs (BASELINE configs[2]).
SAM_PIXEL_MEAN = (123.675, 116.28, 103.53)
SAM_PIXEL_STD = (58.395, 57.12, 57.375)


def build_point_grid(n_per_side: int) -> np.ndarray:
    """The prompt grid of SamAutomaticMaskGenerator (utils/amg.py:179-186): the n x n cell centres of the unit square as
    (x, y) rows, x running fastest.  Pinned bit for bit against the vendored function (tests/golden/producers.npz)."""
    centres = np.linspace(0.5 / n_per_side, 1.0 - 0.5 / n_per_side, n_per_side)
    gx, gy = np.meshgrid(centres, centres)             # 'xy' indexing: gx varies along the last axis
    return np.column_stack([gx.ravel(), gy.ravel()])


def stability_score(mask_logits: torch.Tensor, mask_threshold: float, offset: float) -> torch.Tensor:
    """utils/amg.py calculate_stability_score: IoU between the mask binarised at threshold + offset and at
    threshold - offset (the first is contained in the second, so it is a ratio of areas)."""
    inter = (mask_logits > (mask_threshold + offset)).flatten(-2).sum(-1, dtype=torch.int32)
    union = (mask_logits > (mask_threshold - offset)).flatten(-2).sum(-1, dtype=torch.int32)
    return inter / union


def mask_boxes(masks: torch.Tensor) -> torch.Tensor:
    """Tight XYXY boxes (inclusive pixel coordinates) of bool masks [..., H, W]; an empty mask gives [0, 0, 0, 0] -- the
    output of utils/amg.py:303-338 (batched_mask_to_box), pinned against it in tests/golden/producers.npz.  Found from the
    first / last occupied row and column (argmax of the occupancy profile and of its mirror image)."""
    lead = masks.shape[:-2]
    if masks.numel() == 0:
        return torch.zeros((*lead, 4), dtype=torch.int64, device=masks.device)
    h, w = masks.shape[-2:]
    rows = masks.any(dim=-1).to(torch.uint8)            # [..., H]: which rows hold a pixel
    cols = masks.any(dim=-2).to(torch.uint8)            # [..., W]
    top = rows.argmax(dim=-1)
    bottom = (h - 1) - rows.flip(-1).argmax(dim=-1)
    left = cols.argmax(dim=-1)
    right = (w - 1) - cols.flip(-1).argmax(dim=-1)
    box = torch.stack([left, top, right, bottom], dim=-1)
    occupied = rows.amax(dim=-1).to(torch.bool)
    return box * occupied.unsqueeze(-1)


def box_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS with torchvision.ops.nms semantics: indices of the kept boxes in order of decreasing score; a box is
    dropped when its IoU with an already kept, higher-scoring box exceeds ``iou_threshold``."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    b = boxes.float()
    order = torch.argsort(scores, descending=True, stable=True)
    b = b[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.maximum(b[:, None, :2], b[None, :, :2])
    rb = torch.minimum(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / (area[:, None] + area[None, :] - inter)
    over = (iou > iou_threshold).cpu().numpy()
    keep = np.ones(n, dtype=bool)
    for i in range(n):
        if keep[i]:
            sup = over[i].copy()
            sup[:i + 1] = False
            keep[sup] = False
    return order[torch.from_numpy(np.nonzero(keep)[0]).to(order.device)]


class SamAutoMasks:
    """``generate(img_rgb uint8 [H, W, 3]) -> list of SAM records`` (keys as automatic_mask_generator.py:150-168:
    segmentation bool [H, W], area, bbox XYWH, predicted_iou, point_coords, stability_score, crop_box), ordered by
    decreasing predicted IoU (the order batched_nms returns)."""

    def __init__(self, model, points_per_side: int = 32, points_per_batch: int = 64, pred_iou_thresh: float = 0.88,
                 stability_score_thresh: float = 0.95, stability_score_offset: float = 1.0, box_nms_thresh: float = 0.7,
                 min_mask_region_area: int = 0, mask_threshold: float = 0.0, device: Union[str, torch.device] = "cpu"):
        self.device = torch.device(device)
        self.model = model.eval().to(self.device)
        self.img_size = int(self.model.config.vision_config.image_size)
        self.points_per_batch = int(points_per_batch)
        self.pred_iou_thresh = float(pred_iou_thresh)
        self.stability_score_thresh = float(stability_score_thresh)
        self.stability_score_offset = float(stability_score_offset)
        self.box_nms_thresh = float(box_nms_thresh)
        self.min_mask_region_area = int(min_mask_region_area)
        self.mask_threshold = float(mask_threshold)
        self.point_grid = build_point_grid(points_per_side)

    @classmethod
    def from_config(cls, size: str = "huge", device="cpu", **kw) -> "SamAutoMasks":
        """Random-initialised SAM of a published size (vit_b / vit_l / vit_h geometry)."""
        from transformers import SamConfig, SamModel

        arch = {"base": (768, 12, 12, (2, 5, 8, 11)), "large": (1024, 24, 16, (5, 11, 17, 23)),
                "huge": (1280, 32, 16, (7, 15, 23, 31))}[size]
        vis = dict(hidden_size=arch[0], num_hidden_layers=arch[1], num_attention_heads=arch[2], global_attn_indexes=list(arch[3]))
        return cls(SamModel(SamConfig(vision_config=vis)), device=device, **kw)

    @classmethod
    def from_pretrained(cls, path: str, device="cpu", **kw) -> "SamAutoMasks":
        from transformers import SamModel

        return cls(SamModel.from_pretrained(path, local_files_only=True), device=device, **kw)

    # ---- predictor.set_image: ResizeLongestSide + normalise + pad (predictor.py, utils/transforms.py, modeling/sam.py) ----
    def preprocess(self, img_rgb: np.ndarray):
        from PIL import Image

        h, w = img_rgb.shape[:2]
        scale = self.img_size * 1.0 / max(h, w)
        nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)
        im = np.array(Image.fromarray(img_rgb).resize((nw, nh), Image.BILINEAR))        # torchvision's resize of a PIL image
        x = torch.from_numpy(np.ascontiguousarray(im)).permute(2, 0, 1).float()
        x = (x - torch.tensor(SAM_PIXEL_MEAN).view(3, 1, 1)) / torch.tensor(SAM_PIXEL_STD).view(3, 1, 1)
        x = torch.nn.functional.pad(x, (0, self.img_size - nw, 0, self.img_size - nh))
        return x[None].to(self.device), (nh, nw)

    def _upscale(self, low_res: torch.Tensor, in_hw, out_hw) -> torch.Tensor:
        """Sam.postprocess_masks: low-res logits -> padded model frame -> un-pad -> original size (bilinear both times)."""
        m = torch.nn.functional.interpolate(low_res, (self.img_size, self.img_size), mode="bilinear", align_corners=False)
        m = m[..., :in_hw[0], :in_hw[1]]
        return torch.nn.functional.interpolate(m, out_hw, mode="bilinear", align_corners=False)

    # ---- the network half: everything that needs weights -------------------------------------------------------------
    def _embed(self, img_rgb: np.ndarray):
        """predictor.set_image: the image embedding and the (height, width) of the resized image inside the model frame."""
        x, in_hw = self.preprocess(img_rgb)
        return self.model.get_image_embeddings(x), in_hw

    def _predict(self, state, pts_img: np.ndarray, out_hw):
        """predictor.predict_torch(points[:, None, :], ones, multimask_output=True, return_logits=True) for one batch of
        prompt points given in IMAGE coordinates: mask logits at the image's resolution [nb, 3, H, W] and the predicted
        IoUs [nb, 3].  (ResizeLongestSide.apply_coords scales the points into the model frame.)"""
        emb, (nh, nw) = state
        H, W = out_hw
        pm = torch.as_tensor(pts_img * np.array([[nw / W, nh / H]]), dtype=torch.float32, device=self.device)
        nb = pm.shape[0]
        out = self.model(image_embeddings=emb, input_points=pm.view(1, nb, 1, 2),
                         input_labels=torch.ones((1, nb, 1), dtype=torch.int, device=self.device), multimask_output=True)
        return self._upscale(out.pred_masks[0], (nh, nw), (H, W)), out.iou_scores[0]

    # ---- the flow of SamAutomaticMaskGenerator.generate (automatic_mask_generator.py:137-330) ---------------------------
    # Pinned: tools/make_golden.py runs the vendored class body (generate -> _generate_masks -> _process_crop -> _process_batch)
    # on a stub predictor with seeded logits / IoUs; tests/test_producers.py feeds the same logits through _predict and
    # requires the same records in the same order (tests/golden/sam_generate.npz).
    @torch.no_grad()
    def generate(self, img_rgb: np.ndarray):
        a = np.asarray(img_rgb)
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError(f"expected a uint8 [H, W, 3] RGB image, got {a.dtype} {a.shape}")
        H, W = a.shape[:2]
        state = self._embed(a)
        pts_img = self.point_grid * np.array([[W, H]], dtype=np.float64)                 # (x, y) in the image
        masks_l, ious_l, stab_l, pts_l = [], [], [], []
        for b0 in range(0, len(pts_img), self.points_per_batch):                         # _process_batch
            pb = pts_img[b0:b0 + self.points_per_batch]
            logits, iou = self._predict(state, pb, (H, W))
            per_point = logits.shape[1]
            logits, iou = logits.flatten(0, 1), iou.flatten(0, 1)                          # [nb * 3, H, W], [nb * 3]
            pts = torch.as_tensor(pb, dtype=torch.float64).repeat_interleave(per_point, dim=0)
            if self.pred_iou_thresh > 0.0:
                keep = iou > self.pred_iou_thresh
                logits, iou, pts = logits[keep], iou[keep], pts[keep.cpu()]
            st = stability_score(logits, self.mask_threshold, self.stability_score_offset)
            if self.stability_score_thresh > 0.0:
                keep = st >= self.stability_score_thresh
                logits, iou, pts, st = logits[keep], iou[keep], pts[keep.cpu()], st[keep]
            masks_l.append(logits > self.mask_threshold)
            ious_l.append(iou), stab_l.append(st), pts_l.append(pts)
            # (is_box_near_crop_edge keeps every box here: with crop_n_layers = 0 the crop IS the image, and a box on the
            #  image's own border is exempt -- automatic_mask_generator.py:271-274, utils/amg.py:80-90)
        masks = torch.cat(masks_l)
        ious, stab, pts = torch.cat(ious_l), torch.cat(stab_l), torch.cat(pts_l)
        boxes = mask_boxes(masks)
        keep = box_nms(boxes, ious, self.box_nms_thresh)                                   # _process_crop: NMS by predicted IoU
        kc = keep.cpu()
        masks, ious, stab, pts, boxes = masks[keep].cpu().numpy(), ious[keep].cpu(), stab[keep].cpu(), pts[kc], boxes[keep].cpu()
        recs = []
        for j in range(masks.shape[0]):
            area = int(masks[j].sum())
            if area > self.min_mask_region_area:
                x0, y0, x1, y1 = (int(v) for v in boxes[j])
                recs.append({"segmentation": masks[j], "area": area, "bbox": [x0, y0, x1 - x0, y1 - y0],
                             "predicted_iou": float(ious[j]), "point_coords": [[float(pts[j, 0]), float(pts[j, 1])]],
                             "stability_score": float(stab[j]), "crop_box": [0, 0, W, H]})
        return recs


def resize_like_cv2(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """``cv2.resize(img, (width, height))`` (bilinear, pixel centres, no antialiasing), uint8 in and out."""
    x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float()[None]
    y = torch.nn.functional.interpolate(x, size=(int(height), int(width)), mode="bilinear", align_corners=False, antialias=False)
    return y[0].round().clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()


def process_single_SAM(cfg: dict, img_bgr: np.ndarray, generator: SamAutoMasks):
    """func_vpr.py:537-547: BGR -> RGB, resize to (desired_width, desired_height) when cfg['resize'], generate."""
    img = np.ascontiguousarray(np.asarray(img_bgr)[:, :, ::-1])
    img_p = resize_like_cv2(img, cfg["desired_width"], cfg["desired_height"]) if cfg.get("resize") else img
    return img_p, generator.generate(img_p)


def masks_given_image(generator: SamAutoMasks, img_bgr: np.ndarray, cfg: dict, mask_full_resolution: bool = False):
    """func_vpr.py:564-590 with the image already decoded (cv2.imread is the caller's): crop rows ``rmin:``, run SAM at
    HALF the configured resolution unless ``mask_full_resolution``; returns (list of bool masks, SAM records)."""
    im = np.asarray(img_bgr)[int(cfg.get("rmin", 0)):, :, :]
    f = 1.0 if mask_full_resolution else 0.5
    cfg_sam = {"desired_width": int(f * cfg["desired_width"]), "desired_height": int(f * cfg["desired_height"]), "resize": True}
    _, masks = process_single_SAM(cfg_sam, im, generator)
    return [m["segmentation"] for m in masks], masks
