"""The reference's experiment loop (place_rec_main.py:244-373) over stored inputs, as batch calls.

``place_rec_main.py`` walks the reference images, then the query images, one at a time:
``preload_masks -> getIdxSingleFast -> nbrMasksAGGFastSingle -> seg_vlad_gpu_single -> [PCA per 100 images]``
(``:244-276`` / ``:309-341``), concatenates the segment descriptors, builds ``segRange`` / ``imInds`` (``:287-288,
351-352``) and hands everything to ``recall_segloc`` (``:373``).  Here the same bookkeeping feeds
``SegVLADPipeline.describe`` with whole batches of images (tokens and masks go to the device once per batch); the
inputs come from an open ``h5py.File`` or from ``store.FeatureStore`` -- anything with the reference's nesting.

The describing function is a parameter (``describe(tokens[B,D,N], masks[S_tot,Hm,Wm], seg_offsets) -> [S_tot, P]``),
so the host logic -- batching, ragged segment counts, the imInds / segRange maps and their quirks -- is testable on a
CPU with a checker in its place (tests/test_driver.py)."""
from __future__ import annotations

import re
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .func_vpr import getIdxSingleFast, preload_masks


def natural_sorted(names: Sequence[str]) -> List[str]:
    """natsorted() of the reference (place_rec_main.py sorts the image lists with it)."""
    return sorted(names, key=lambda s: [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", str(s))])


def load_image_inputs(dino_in, masks_in, key) -> Tuple[np.ndarray, np.ndarray]:
    """One image: tokens ``[D, N]`` float32 (the ``[1, D, h, w]`` block flattened as func_vpr.py:1083 does) and masks
    ``[S, Hm, Wm]`` uint8 in natural key order (func_vpr.py:757-759)."""
    t = np.asarray(dino_in[key]["ift_dino"][()], dtype=np.float32)
    if t.ndim != 4 or t.shape[0] != 1:
        raise ValueError(f"{key}: ift_dino must be [1, D, h, w], got {t.shape}")
    tokens = t.reshape(t.shape[1], t.shape[2] * t.shape[3])
    segs = preload_masks(masks_in, key)
    if len(segs):
        masks = np.ascontiguousarray(np.stack([np.asarray(m) for m in segs]).astype(np.uint8))
    else:
        masks = np.zeros((0, 1, 1), np.uint8)
    return tokens, masks


def describe_split(dino_in, masks_in, image_keys: Sequence[str], describe: Callable, batch_size: int = 100,
                   min_area: int = 400) -> Tuple[np.ndarray, np.ndarray, List[np.ndarray]]:
    """Segment descriptors of a whole split.

    Returns ``(desc [S_tot, P], imInds [S_tot], segRange)`` exactly as the reference builds them:
    ``imInds[j]`` = index of the image segment j belongs to (place_rec_main.py:250-252), ``segRange[i]`` = the rows of
    image i for ``i <= imInds[-1]`` (``:287-288``: trailing images WITHOUT segments get no entry, as in the reference).
    ``batch_size`` mirrors the reference's PCA batch (``:214, 263``); images in one batch must share the token and
    mask geometry."""
    descs: List[np.ndarray] = []
    im_inds: List[np.ndarray] = []
    for b0 in range(0, len(image_keys), batch_size):
        keys = image_keys[b0:b0 + batch_size]
        toks, msks, offs = [], [], [0]
        for j, key in enumerate(keys):
            t, m = load_image_inputs(dino_in, masks_in, key)
            ii, _, seg = getIdxSingleFast(b0 + j, list(m), minArea=min_area)      # minArea is ignored, as in the reference
            im_inds.append(np.asarray(ii, dtype=np.int64))
            toks.append(t)
            msks.append(m)
            offs.append(offs[-1] + len(seg))
        shapes = {t.shape for t in toks}
        if len(shapes) != 1:
            raise ValueError(f"images {keys[0]}..{keys[-1]}: token blocks of different shapes {sorted(shapes)} in one batch")
        with_masks = [m for m in msks if m.shape[0]]
        mshapes = {m.shape[1:] for m in with_masks}
        if len(mshapes) > 1:
            raise ValueError(f"images {keys[0]}..{keys[-1]}: masks of different sizes {sorted(mshapes)} in one batch")
        masks = np.concatenate(with_masks) if with_masks else np.zeros((0, 1, 1), np.uint8)
        d = describe(np.stack(toks), masks, np.asarray(offs, dtype=np.int32))
        descs.append(np.asarray(d.cpu() if hasattr(d, "cpu") else d))
    im = np.concatenate(im_inds) if im_inds else np.zeros(0, np.int64)
    desc = np.concatenate(descs) if descs else np.zeros((0, 0), np.float32)
    seg_range = [np.where(im == i)[0] for i in range(int(im[-1]) + 1)] if len(im) else []
    return desc, im, seg_range


def run_segloc(dino_r, masks_r, keys_r, dino_q, masks_q, keys_q, gt, pipeline, batch_size: int = 100, n_top: int = 5,
               k_search: int = 200, k_vote: int = 50, save_results: Optional[dict] = None):
    """Reference split -> index, query split -> ranked reference images -> recall@1..n_top, on the device pipeline
    (``pipeline``: a ``SegVLADPipeline`` whose engine has the vocabulary and, if used, the PCA model set).
    The chain is ``recall_segloc``'s (place_rec_main.py:44-96): normalised descriptors, exact search ``k_search``,
    keep ``k_vote``, ``2 - d^2``, similarity-weighted image vote, ``calc_recall``.

    ``save_results`` = ``{"workdir", "dataset_name", "experiment_name", "experiment_config", "domain"}`` writes what the
    reference pickles under its ``--save_results`` switch, under the reference's file names: the reference descriptors
    ``segFtVLAD1`` (place_rec_main.py:292-305), the query descriptors ``segFtVLAD2`` (``:357-370``) -- torch CPU tensors, the
    rows as they are handed to ``recall_segloc`` -- and ``{'sims', 'matches'}``, the ``k_search``-deep search output
    (``:61-75``)."""
    from .pipeline import recall_at
    from . import store

    def describe(tokens, masks, offs):
        import torch

        dev = pipeline.eng.device
        return pipeline.describe(torch.from_numpy(tokens).to(dev), torch.from_numpy(masks).to(dev), offs, l2norm=True)

    d1, im1, _ = describe_split(dino_r, masks_r, keys_r, describe, batch_size)
    d2, im2, seg_range2 = describe_split(dino_q, masks_q, keys_q, describe, batch_size)
    if len(seg_range2) != len(keys_q):
        raise ValueError("the last query image(s) have no segments: the reference's segRange2 would be too short for gt")
    pipeline.index_reset()
    pipeline.index_add(d1, im1.astype(np.int32))
    q_off = np.concatenate([[0], np.cumsum([len(r) for r in seg_range2])]).astype(np.int32)
    pred, _, matches, sims = pipeline.retrieve(d2, q_off, k_search=k_search, k_vote=k_vote, n_top=n_top)
    pred = pred.cpu().numpy() if hasattr(pred, "cpu") else np.asarray(pred)
    if save_results is not None:
        full_d2, full_idx = pipeline.last_search
        store.save_experiment_pickles(save_results["workdir"], save_results["dataset_name"], save_results["experiment_name"],
                                      save_results["experiment_config"], save_results["domain"], segFtVLAD1=d1, segFtVLAD2=d2,
                                      sims=full_d2, matches=full_idx)
    return recall_at(pred, gt, n_top), pred, matches, sims
