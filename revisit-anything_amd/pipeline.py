"""Batched SegVLAD pipeline on one GPU: the reference's per-image driver loop
(place_rec_main.py:244-276 / 309-341: masks -> imInds -> adjacency -> seg-VLAD -> [PCA per batch] -> concat)
restated as batch calls over device-resident inputs, followed by the ``recall_segloc`` chain
(place_rec_main.py:44-96).  Host work is limited to what the reference also does on the host
(Qhull Delaunay via scipy) plus launch bookkeeping."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import SEGVLAD_ERR_LIMIT, SegVLADDegenerateError, SegVLADError
from .engine import SegVLADEngine
from .func_vpr import adjacency_from_centroids

_POOL: Optional[ThreadPoolExecutor] = None


def _pool(workers: int) -> ThreadPoolExecutor:
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=workers)
    return _POOL


def adjacency_batch(centroids: np.ndarray, seg_offsets: Sequence[int], order: int, workers: int = 8) -> np.ndarray:
    """Concatenated per-image [S_b,S_b] uint8 adjacency blocks from the centroids [S_tot,2]
    (func_vpr.py:1315-1345 per image; Qhull releases the GIL, so images run on a thread pool)."""
    B = len(seg_offsets) - 1

    def one(b):
        c = centroids[seg_offsets[b]:seg_offsets[b + 1]]
        return adjacency_from_centroids(c, order).numpy().astype(np.uint8).reshape(-1)

    if B == 0:
        return np.zeros(0, np.uint8)
    parts = list(_pool(workers).map(one, range(B))) if B > 1 else [one(0)]
    return np.concatenate(parts)


class SegVLADPipeline:
    def __init__(self, engine: SegVLADEngine, H: int, W: int, patch: int = 14, order: int = 3, use_pca: bool = True,
                 adj_workers: int = 8, host_adjacency: bool = False, fuse_pca: bool = True, check_empty: bool = True):
        self.eng = engine
        self.H, self.W, self.patch = H, W, patch
        self.order = order
        self.use_pca = use_pca
        self.adj_workers = adj_workers
        self.host_adjacency = host_adjacency
        self.fuse_pca = fuse_pca
        # an empty mask has no centroid: the reference fails there (func_vpr.py:1314 -> Delaunay on NaN); with
        # check_empty the device adjacency reports it (one 4-byte read-back per batch) and describe() raises ValueError
        self.check_empty = check_empty
        self.N = (H // patch) * (W // patch)
        self._eager_flags = 0   # > 0: read the adjacency flags BEFORE describing (set after a batch that needed a Qhull patch)

    # ---- a2..a9: images -> (normalised) segment descriptors ------------------------------------------
    def describe(self, tokens: torch.Tensor, masks: torch.Tensor, seg_offsets: np.ndarray, adj=None,
                 l2norm: bool = True) -> torch.Tensor:
        """tokens [B,D,N] fp32 (device), masks [S_tot,Hm,Wm] uint8 (device), seg_offsets [B+1] (host).
        adj: precomputed concatenated adjacency (uint8) or None to derive it (order>0) from the masks.
        Returns [S_tot, P] (PCA'd, row-normalised when l2norm) or [S_tot, K*D]."""
        eng = self.eng
        lazy = None   # (device flags, centroids) of a device adjacency whose per-image flags have not been looked at yet
        if (self.order and adj is None and not self.host_adjacency and hasattr(eng, "describe_begin")
                and isinstance(tokens, torch.Tensor) and tokens.is_cuda and isinstance(masks, torch.Tensor) and masks.is_cuda
                and (self.fuse_pca or not self.use_pca)):
            # THREE calls (segvlad_describe_begin / _flags / _end): the mask branch (incidence + centroids -> adjacency) runs on the
            # context's side stream beside the assignment pass; its per-image flags and centroids reach the host while that pass
            # keeps the device busy, so an empty mask is reported, and the images with a non-generic centroid configuration get
            # Qhull's adjacency (the reference's own library, ~0.2 ms of host work per image), WITHOUT an idle device -- rounds 3-4
            # read the flags between two launches (a synchronisation in the middle of the stage) or described the batch twice.
            try:
                h = eng.describe_begin(masks, tokens, seg_offsets, self.H, self.W, self.patch, self.order, pca=self.use_pca)
            except SegVLADError as e:
                # SEGVLAD_ERR_LIMIT = "the split call cannot, the separate entry points can": more segments in an image than the
                # in-LDS Delaunay holds, an empty batch, a PCA model without the fp16x3 form (pca_arith=fp32 / K*D % 32 != 0:
                # images_pca then projects the stored descriptor with the plain GEMM) -> the paths below
                if e.code != SEGVLAD_ERR_LIMIT:
                    raise
                h = None
            if h is not None:
                imgs, blocks = None, None
                if self.check_empty:
                    flags, cent = eng.describe_flags(h)
                    if (flags & 1).any():
                        eng.describe_cancel(h)
                        raise ValueError(f"{int((flags & 1).sum())} image(s) with an empty mask: centroid undefined")
                    imgs = np.nonzero(flags & 2)[0]
                    self.n_flag_checks = getattr(self, "n_flag_checks", 0) + 1
                    self.n_flagged_images = getattr(self, "n_flagged_images", 0) + int(len(imgs))
                    so = np.asarray(seg_offsets)
                    try:
                        blocks = [adjacency_from_centroids(cent[so[b]:so[b + 1]], self.order).numpy().astype(np.uint8) for b in imgs]
                    except Exception:
                        eng.describe_cancel(h)
                        raise
                return eng.describe_end(h, imgs, blocks, l2norm=l2norm)["out"]
        if self.order and adj is None:
            bits, cent = eng.incidence_centroids(masks, self.H, self.W, self.patch)   # one pass over the mask bytes
            if self.host_adjacency:   # scipy/Qhull on the host, exactly the reference's library (slow: ~0.4 ms/image)
                adj = adjacency_batch(cent.cpu().numpy(), seg_offsets, self.order, self.adj_workers)
            else:                     # device kernel: no host round trip (check_empty: one flag byte per image comes back)
                try:
                    if self.check_empty and hasattr(eng, "adjacency_flagged"):
                        # The flags are READ after the kernels that consume the adjacency have been enqueued (below): reading
                        # them here is a host synchronisation in the middle of the describe stage, with an idle device behind
                        # it.  A flagged image is patched afterwards and the batch described AGAIN -- twice the work, so a
                        # batch that needed it switches the next 16 batches back to reading the flags first (data whose
                        # centroids are non-generic again and again: axis-aligned synthetic masks; real SAM masks are not).
                        adj, flags_dev = eng.adjacency_flagged(cent, seg_offsets, self.order, device_flags=True)
                        if self._eager_flags > 0:
                            self._eager_flags -= 1
                            adj = self._check_flags(flags_dev, adj, cent, seg_offsets)
                        else:
                            lazy = (flags_dev, cent)
                    else:
                        adj = eng.adjacency(cent, seg_offsets, self.order, check_empty=self.check_empty)
                except SegVLADError as e:
                    # typed, not message-matched: a documented implementation limit (SEGVLAD_ERR_LIMIT) or the
                    # degenerate-configuration report are recoverable; anything else is not
                    if not isinstance(e, SegVLADDegenerateError) and e.code != SEGVLAD_ERR_LIMIT:
                        raise
                    # an image with more segments (~620) than the in-LDS Delaunay holds (or a report from an engine
                    # without per-image flags): the reference's own Qhull path for this batch
                    adj = adjacency_batch(cent.cpu().numpy(), seg_offsets, self.order, self.adj_workers)
        else:
            bits = eng.incidence(masks, self.H, self.W, self.patch)
            if not self.order:
                adj = None
        out = self._describe_with(tokens, bits, seg_offsets, adj, l2norm)
        if lazy is not None:
            patched = self._check_flags(lazy[0], adj, lazy[1], seg_offsets)
            if patched is not adj:
                self._eager_flags = 16
                out = self._describe_with(tokens, bits, seg_offsets, patched, l2norm)
        return out

    def _check_flags(self, flags_dev, adj, cent, seg_offsets):
        """Reads the per-image flags of a device adjacency (a host synchronisation): raises on an empty mask; images with a
        non-generic centroid configuration (duplicate / co-circular centroids: the triangulation is Qhull's tie-breaking) get
        the reference's own Qhull path.  Returns ``adj`` itself when nothing was flagged, else the patched adjacency."""
        flags = flags_dev.cpu().numpy()
        if (flags & 1).any():
            raise ValueError(f"{int((flags & 1).sum())} image(s) with an empty mask: centroid undefined")
        bad = np.nonzero(flags & 2)[0]
        self.n_flag_checks = getattr(self, "n_flag_checks", 0) + 1
        self.n_flagged_images = getattr(self, "n_flagged_images", 0) + int(len(bad))
        if len(bad):
            return self._patch_with_qhull(adj.clone(), cent, np.asarray(seg_offsets), bad)
        return adj

    def _describe_with(self, tokens, bits, seg_offsets, adj, l2norm):
        eng = self.eng
        if self.use_pca and self.fuse_pca:   # aggregation feeds the projection GEMM directly (segvlad_images_pca)
            return eng.seg_vlad_pca(tokens, bits, seg_offsets, adj, l2norm=l2norm)["out"]
        desc = eng.seg_vlad(tokens, bits, seg_offsets, adj)["out"]
        if self.use_pca:
            desc = eng.pca_apply(desc, l2norm=l2norm)
        return desc

    def _patch_with_qhull(self, adj: torch.Tensor, cent: torch.Tensor, so: np.ndarray, images) -> torch.Tensor:
        """Overwrite the [S_b, S_b] blocks of the listed images in the concatenated device adjacency with Qhull's."""
        sizes = (so[1:] - so[:-1]).astype(np.int64)
        adj_off = np.concatenate([[0], np.cumsum(sizes * sizes)])
        lo, hi = int(so[min(images)]), int(so[max(images) + 1])
        ch = cent[lo:hi].cpu().numpy()
        for b in images:
            blk = adjacency_from_centroids(ch[so[b] - lo:so[b + 1] - lo], self.order).numpy().astype(np.uint8).reshape(-1)
            adj[int(adj_off[b]):int(adj_off[b + 1])] = torch.from_numpy(blk).to(adj.device)
        return adj

    # ---- a10: index ------------------------------------------------------------------------------------
    def index_reset(self):
        self.eng.db_reset()

    def index_add(self, desc: torch.Tensor, img_of_seg):
        self.eng.db_add(desc, img_of_seg)

    # ---- a10..a12: descriptors -> ranked reference images ----------------------------------------------
    def retrieve(self, qdesc: torch.Tensor, qseg_offsets: np.ndarray, k_search: int = 200, k_vote: int = 50, n_top: int = 5,
                 mode: int = _lib.VOTE_WT_BORDA_IM, want_scores: bool = False, vote_depth_only: bool = False):
        """search k_search (place_rec_main.py:56) -> keep k_vote and 2-d^2 (:78-81) -> vote (:84).  vote_depth_only: search
        only as deep as the vote reads (the 200-wide lists are only pickled under save_results, :61-75): the same k_vote columns
        -- an exact search's first columns do not depend on its depth -- for a quarter of the refinement."""
        self.eng.hint_query_groups(qseg_offsets)
        d2, idx = self.eng.search(qdesc, k_vote if vote_depth_only else k_search)
        self.last_search = (d2, idx)   # the full-depth lists: what `save_results` pickles (place_rec_main.py:61-75)
        sims, m = self.eng.sims_from_d2(d2, idx, k_vote)
        pred, sc = self.eng.vote(m, sims, qseg_offsets, n_top=n_top, mode=mode, want_scores=want_scores)
        return pred, sc, m, sims


def recall_at(preds: np.ndarray, gt: List[Sequence[int]], n: int) -> List[float]:
    """calc_recall (func_vpr.py:396-422) without the print: first correct rank, queries with empty GT skipped."""
    rec = np.zeros(n)
    num = 0
    for i, g in enumerate(gt):
        if len(g) == 0:
            continue
        num += 1
        for j in range(min(n, preds.shape[1])):
            if preds[i, j] in g:
                rec[j] += 1
                break
    return (np.cumsum(rec) / max(num, 1)).tolist()
