"""CPU ORACLE for the SegVLAD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``revisit_anything_amd``) never does; it fails loudly when the HIP
library is missing.

This is a NumPy restatement of the reference algorithm (AnyLoc/Revisit-Anything, snapshot
2025-04-04).  Every function cites the reference ``file:line`` it follows.  Parity status:

* PINNED against the reference's own function bodies (AST-extracted and executed in the build
  container by ``tools/make_golden.py``; outputs committed under ``tests/golden/``):
  pixel->token map, incidence, adjacency, labels, segment-VLAD, vote (both modes), recall,
  ``recall_segloc`` chain.  ``tests/test_oracle_golden.py`` checks this module against them.
* UNPINNED third-party arithmetic (no source under the reference tree, no reference tests):
  faiss 1.7.3 ``IndexFlatL2`` (restated from its documented semantics: exact squared L2, ascending,
  fp32) and scikit-learn 1.3.2 ``PCA.transform`` (restated as ``(X-mean) @ components.T /
  sqrt(explained_variance)``; the golden fixture pins it against the sklearn 1.7.2 installed in the
  build container).  scipy's Qhull Delaunay is *called* (same library as the reference), not restated.

Arithmetic types follow the reference: fp32 for token normalisation / assignment / residuals,
fp64 for aggregation and norms, fp32 for kNN distances and vote weights, fp64 vote accumulation.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-12  # torch.nn.functional.normalize default eps (func_vpr.py:1085,1145,1202,1205)


# --------------------------------------------------------------------------------------------
# a1  pixel -> token map                                   place_rec_main.py:187-194
# --------------------------------------------------------------------------------------------
def pixel_to_token_index(H: int, W: int, patch: int = 14):
    """``idx[i,j] = (clip(i//p,0,dh-1), clip(j//p,0,dw-1))``; ``ind = ravel_multi_index``.

    Returns (idx_matrix [H,W,2] int32, ind_matrix [H*W] int64)   (place_rec_main.py:187-193).
    """
    dh, dw = H // patch, W // patch
    ii = np.clip(np.arange(H) // patch, 0, dh - 1)
    jj = np.clip(np.arange(W) // patch, 0, dw - 1)
    idx = np.empty((H, W, 2), dtype=np.int32)
    idx[:, :, 0] = ii[:, None]
    idx[:, :, 1] = jj[None, :]
    ind = (ii[:, None] * dw + jj[None, :]).reshape(-1).astype(np.int64)
    return idx, ind


def nearest_src_index(out_size: int, in_size: int) -> np.ndarray:
    """Source index of torch's legacy ``mode='nearest'`` interpolate (func_vpr.py:1089):
    ``src = min(floor(dst * float32(in/out)), in-1)`` with the scale held in fp32."""
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


# --------------------------------------------------------------------------------------------
# a5 (first half)  mask -> token incidence                 func_vpr.py:1088-1092
# --------------------------------------------------------------------------------------------
def incidence(masks, H: int, W: int, patch: int = 14) -> np.ndarray:
    """``inc[s,t]`` = any true pixel of the nearest-upsampled mask ``s`` falls in token cell ``t``.

    masks: [S,Hm,Wm] bool.  Returns bool [S, dh*dw]   (func_vpr.py:1088-1092).
    """
    masks = np.asarray(masks).astype(bool)
    S, Hm, Wm = masks.shape
    dh, dw = H // patch, W // patch
    ri = nearest_src_index(H, Hm)
    ci = nearest_src_index(W, Wm)
    ti = np.clip(np.arange(H) // patch, 0, dh - 1)
    tj = np.clip(np.arange(W) // patch, 0, dw - 1)
    # row stage: for each token row ty, OR of the source rows that feed it
    rows = np.zeros((S, dh, Wm), dtype=bool)
    for i in range(H):
        rows[:, ti[i], :] |= masks[:, ri[i], :]
    inc = np.zeros((S, dh, dw), dtype=bool)
    for j in range(W):
        inc[:, :, tj[j]] |= rows[:, :, ci[j]]
    return inc.reshape(S, dh * dw)


def pack_bits_u64(b: np.ndarray) -> np.ndarray:
    """Pack a bool [R, N] matrix into little-endian u64 words [R, ceil(N/64)] (bit t%64 of word t//64)."""
    b = np.asarray(b, dtype=bool)
    R, N = b.shape
    nw = (N + 63) // 64
    pad = np.zeros((R, nw * 64), dtype=bool)
    pad[:, :N] = b
    by = np.packbits(pad.reshape(R, nw, 8, 8), axis=-1, bitorder="little")  # [R,nw,8,1]
    return by.reshape(R, nw, 8).copy().view("<u8").reshape(R, nw)


def unpack_bits_u64(w: np.ndarray, N: int) -> np.ndarray:
    w = np.ascontiguousarray(w, dtype="<u8")
    R, nw = w.shape
    by = w.view(np.uint8).reshape(R, nw * 8)
    return np.unpackbits(by, axis=1, bitorder="little")[:, :N].astype(bool)


# --------------------------------------------------------------------------------------------
# a3  segment -> image bookkeeping                         func_vpr.py:762-786
# --------------------------------------------------------------------------------------------
def get_idx_single_fast(img_idx, masks_seg, minArea=400, returnMask=True):
    """``minArea`` is ignored by the reference (func_vpr.py:779 commented out)."""
    n = len(masks_seg)
    segmask = list(masks_seg) if returnMask else []
    return np.array([img_idx] * n), list(range(n)), segmask


# --------------------------------------------------------------------------------------------
# a4  Delaunay neighbourhood adjacency                     func_vpr.py:1309-1347, 1241-1245
# --------------------------------------------------------------------------------------------
def mask_centroids(masks_seg) -> np.ndarray:
    """Mean of the non-zero (row, col) of each mask, reversed to (x, y)   (func_vpr.py:1314)."""
    return np.array([np.array(np.nonzero(m)).mean(1)[::-1] for m in masks_seg])


def adjacency_from_centroids(cords: np.ndarray, order: int = 1) -> np.ndarray:
    """bool [S,S]; S<=3: every row = e0(+e1) (func_vpr.py:1339-1344); else Delaunay neighbours
    + self loop, raised to ``order`` by repeated matmul and thresholded (func_vpr.py:1317-1337)."""
    S = len(cords)
    A = np.zeros((S, S), dtype=np.float32)
    if S > 3:
        from scipy.spatial import Delaunay

        tri = Delaunay(cords)
        indptr, indices = tri.vertex_neighbor_vertices
        for v in range(S):
            A[v, v] = 1
            A[v, indices[indptr[v]:indptr[v + 1]]] = 1
        P = A.copy()
        for _ in range(order - 1):
            P = P @ A
        return P.astype(bool)
    nbr = [0, 1] if S > 1 else [0]
    for v in range(S):
        A[v, nbr] = 1
    return A.astype(bool)


def nbr_masks_agg_fast_single(masks_seg, order: int = 1) -> np.ndarray:
    return adjacency_from_centroids(mask_centroids(masks_seg), order)


# --------------------------------------------------------------------------------------------
# a5/a6/a7  segment VLAD                                   func_vpr.py:1065-1210
# --------------------------------------------------------------------------------------------
def normalize_tokens_f32(tokens_dn: np.ndarray) -> np.ndarray:
    """``F.normalize(dino_desc, dim=1)`` in fp32 (func_vpr.py:1085).  tokens_dn: [D,N] -> x^ [N,D] fp32."""
    t = np.asarray(tokens_dn, dtype=np.float32)
    nrm = np.sqrt((t.astype(np.float64) ** 2).sum(0)).astype(np.float32)
    xn = t / np.maximum(nrm, np.float32(EPS))[None, :]
    return np.ascontiguousarray(xn.T)


def assign_labels(xn: np.ndarray, c_centers: np.ndarray):
    """``argmax(x^ @ normalize(C).T)`` (func_vpr.py:1145-1146): cosine arg-max against the
    L2-normalised centres, first max on ties.  Scores are evaluated in fp64 from the fp32 operands
    (the reference's fp32 GEMM differs from this only by summation-order rounding); the top-2 gap is
    returned so tests can exclude near-ties.  Returns (labels int64 [N], gap float64 [N])."""
    C = np.asarray(c_centers, dtype=np.float32)
    cn = C / np.maximum(np.sqrt((C.astype(np.float64) ** 2).sum(1)).astype(np.float32), np.float32(EPS))[:, None]
    sc = xn.astype(np.float64) @ cn.astype(np.float64).T
    labels = np.argmax(sc, axis=1)
    if sc.shape[1] > 1:
        part = np.partition(sc, -2, axis=1)
        gap = part[:, -1] - part[:, -2]
    else:
        gap = np.full(sc.shape[0], np.inf)
    return labels.astype(np.int64), gap


def vlad_matmuls_per_cluster(num_c, masks, res, labels, adj=None, return_block_norms=False):
    """func_vpr.py:1181-1210 in fp64: per cluster ``inc' = (adj @ inc[:,T]) > 0``;
    ``V[:,k,:] = inc' @ res[T]``; intra-normalise; flatten cluster-major; L2-normalise rows."""
    masks = np.asarray(masks, dtype=np.float64)
    res = np.asarray(res, dtype=np.float64)
    S = masks.shape[0]
    D = res.shape[1]
    if adj is None:
        adj = np.eye(S)
    adj = np.asarray(adj, dtype=np.float64)
    V = np.zeros((S, num_c, D))
    bn = np.zeros((S, num_c))
    for k in range(num_c):
        T = np.where(labels == k)[0]
        agg = ((adj @ masks[:, T]) != 0).astype(np.float64)
        v = agg @ res[T, :]
        n = np.sqrt((v * v).sum(1))
        bn[:, k] = n
        V[:, k, :] = v / np.maximum(n, EPS)[:, None]
    flat = V.reshape(S, num_c * D)
    n = np.sqrt((flat * flat).sum(1))
    out = flat / np.maximum(n, EPS)[:, None]
    if return_block_norms:
        return out, bn
    return out


def seg_vlad(tokens_dn, inc, c_centers, adj=None, return_aux=False):
    """``seg_vlad_gpu_single_img`` + ``vlad_single`` (func_vpr.py:1103-1179) given the incidence.

    tokens_dn [D,N] fp32 as stored (D-major), inc bool [S,N], c_centers [K,D] fp32, adj bool [S,S]|None.
    Returns fp64 [S, K*D] (and aux dict with labels, gap, block norms when asked).
    NB: K comes from ``c_centers`` (the reference hard-codes 32, func_vpr.py:1142)."""
    C = np.asarray(c_centers, dtype=np.float32)
    K = C.shape[0]
    xn = normalize_tokens_f32(tokens_dn)
    labels, gap = assign_labels(xn, C)
    res = xn - C[labels]  # fp32 subtract against the UN-normalised centre (func_vpr.py:1151)
    out, bn = vlad_matmuls_per_cluster(K, inc, res.astype(np.float64), labels, adj, return_block_norms=True)
    if return_aux:
        return out, {"labels": labels, "gap": gap, "block_norms": bn}
    return out


def seg_vlad_from_masks(tokens_dn, masks, c_centers, H, W, adj=None, patch=14):
    return seg_vlad(tokens_dn, incidence(masks, H, W, patch), c_centers, adj)


# --------------------------------------------------------------------------------------------
# a8  PCA transform (sklearn PCA.transform restated)       func_vpr.py:1419-1443
# --------------------------------------------------------------------------------------------
def pca_transform(X, mean, components, explained_variance, whiten=True):
    Y = (np.asarray(X, dtype=np.float64) - np.asarray(mean, dtype=np.float64)) @ np.asarray(components, dtype=np.float64).T
    if whiten:
        Y = Y / np.sqrt(np.asarray(explained_variance, dtype=np.float64))
    return Y


# --------------------------------------------------------------------------------------------
# a9  normalizeFeat                                        func_vpr.py:1673-1676
# --------------------------------------------------------------------------------------------
def normalize_feat(rfts):
    r = np.array(rfts).reshape([len(rfts), -1])
    r = r / np.linalg.norm(r, axis=1)[:, None]
    return r


# --------------------------------------------------------------------------------------------
# a10  exact kNN (faiss.IndexFlatL2 semantics restated)    place_rec_main.py:53-60
# --------------------------------------------------------------------------------------------
def l2_matrix(R, Q, rows_block=0):
    """[nq, nr] squared L2 distances: inputs coerced to fp32 (as faiss does), evaluated in fp64, rounded to fp32.
    rows_block > 0 bounds the fp64 copy of R (large databases)."""
    R32 = np.ascontiguousarray(R, dtype=np.float32)
    q = np.ascontiguousarray(Q, dtype=np.float32).astype(np.float64)
    q2 = (q * q).sum(1)[:, None]
    nr = R32.shape[0]
    out = np.empty((q.shape[0], nr), dtype=np.float32)
    step = rows_block if rows_block > 0 else max(nr, 1)
    for a in range(0, nr, step):
        Rd = R32[a:a + step].astype(np.float64)
        out[:, a:a + step] = (q2 + (Rd * Rd).sum(1)[None, :] - 2.0 * (q @ Rd.T)).astype(np.float32)
    np.maximum(out, 0.0, out=out, where=out < 0)    # faiss: `if (dis < 0) dis = 0` (see knn_l2)
    return out


def topk_from_d2(d2, k):
    """Ascending top-k of each row of a distance matrix, ties -> lower index (a stable argsort's first k columns),
    found by partition + sort of the tie-complete candidate set; rows with fewer than k columns pad with (inf, -1)."""
    nq, nr = d2.shape
    D2 = np.full((nq, k), np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    kk = min(k, nr)
    if kk == 0:
        return D2, I
    kth = np.partition(d2, kk - 1, axis=1)[:, kk - 1]
    for r in range(nq):
        cand = np.nonzero(d2[r] <= kth[r])[0]                       # ascending ids, includes every boundary tie
        order = cand[np.argsort(d2[r, cand], kind="stable")][:kk]
        D2[r, :kk] = d2[r, order]
        I[r, :kk] = order
    return D2, I


def knn_l2(R, Q, k, block=2048):
    """Exact squared-L2 top-k, ascending, ties -> lower index.  Inputs are coerced to fp32 (as
    faiss does), distances evaluated in fp64 and rounded to fp32, negative values set to zero BEFORE the
    selection (faiss 1.7.3 utils/distances.cpp, exhaustive_L2sqr_blas: `if (dis < 0) dis = 0`; its
    small-batch path sums squared differences and cannot go negative either).  Rows beyond n_r are (inf, -1)."""
    R32 = np.ascontiguousarray(R, dtype=np.float32)
    Q32 = np.ascontiguousarray(Q, dtype=np.float32)
    nr, nq = R32.shape[0], Q32.shape[0]
    Rd = R32.astype(np.float64)
    rn = (Rd * Rd).sum(1)
    D2 = np.full((nq, k), np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    kk = min(k, nr)
    for s in range(0, nq, block):
        q = Q32[s:s + block].astype(np.float64)
        d2 = (q * q).sum(1)[:, None] + rn[None, :] - 2.0 * (q @ Rd.T)
        d2 = d2.astype(np.float32)
        np.maximum(d2, 0.0, out=d2, where=d2 < 0)
        order = np.argsort(d2, axis=1, kind="stable")[:, :kk]
        D2[s:s + block, :kk] = np.take_along_axis(d2, order, axis=1)
        I[s:s + block, :kk] = order
    return D2, I


def merge_topk(d2_parts, idx_parts, k):
    """Merge per-shard ascending top-k lists (global ids) into the global top-k: by distance, ties
    by lower global id.  (No reference counterpart: the reference is single-process.)"""
    d2 = np.concatenate(d2_parts, axis=1)
    idx = np.concatenate(idx_parts, axis=1)
    big = np.where(idx < 0, np.iinfo(np.int64).max, idx)
    order = np.lexsort((big, d2), axis=1)[:, :k]
    return np.take_along_axis(d2, order, 1), np.take_along_axis(idx, order, 1)


# --------------------------------------------------------------------------------------------
# a12  image vote                                          func_vpr.py:61-77, 118-125, 207-224
# --------------------------------------------------------------------------------------------
def weighted_borda_count(*ranked_lists_with_scores):
    scores = {}
    for ranked_list in ranked_lists_with_scores:
        for index, score in ranked_list:
            if index in scores:
                scores[index] += score
            else:
                scores[index] = score
    return sorted(scores.keys(), key=lambda index: scores[index], reverse=True), scores


def get_matches_wt_borda_im(matches, n_query, sims, segRangeQuery, imIndsRef, n=1, return_scores=False):
    """method="max_seg_topk_wt_borda_Im" (func_vpr.py:207-224): min-max normalise with the GLOBAL
    extrema (fp32), accumulate per reference image id in rank-major order (python float = fp64),
    stable descending sort, first n."""
    sims = np.asarray(sims)
    smax = np.max(sims)
    smin = np.min(sims)
    preds, all_scores = [], []
    for i in range(n_query):
        mp = matches[segRangeQuery[i]].T
        sp = sims[segRangeQuery[i]].T
        sp = (sp - smin) / (smax - smin)
        pair = [list(zip(imIndsRef[mp[k]].tolist(), sp[k].tolist())) for k in range(len(sp))]
        ranked, sc = weighted_borda_count(*pair)
        preds.append(ranked[:n])
        all_scores.append([sc[r] for r in ranked[:n]])
    if return_scores:
        return preds, all_scores
    return preds


def get_matches_max_seg_topk(matches, n_query, segRangeQuery, imIndsRef, n=1):
    """method="max_seg_topk" (func_vpr.py:118-125): bincount of image ids of all matches.
    Returns (preds via the reference's argsort rule, counts list) -- the reference's tie order is
    implementation-defined; compare counts, or ids only when the maximum is unique."""
    preds, counts = [], []
    for i in range(n_query):
        mp = matches[segRangeQuery[i]].flatten()
        bc = np.bincount(imIndsRef[mp])
        segIdx = np.where(bc > 0)[0]
        pred = segIdx[np.flip(np.argsort(bc[segIdx])[-n:])]
        preds.append(pred)
        counts.append(bc)
    return preds, counts


# --------------------------------------------------------------------------------------------
# a13  recall                                              func_vpr.py:396-422
# --------------------------------------------------------------------------------------------
def calc_recall(pred, gt, n):
    recall = [0] * n
    num_eval = 0
    for i in range(len(gt)):
        if len(gt[i]) == 0:
            continue
        num_eval += 1
        for j in range(len(pred[i])):
            if n == 1:
                if pred[i] in gt[i]:
                    recall[j] += 1
                    break
            else:
                if pred[i][j] in gt[i]:
                    recall[j] += 1
                    break
    return (np.cumsum(recall) / float(num_eval)).tolist()


# --------------------------------------------------------------------------------------------
# recall_segloc chain                                      place_rec_main.py:44-96
# --------------------------------------------------------------------------------------------
def recall_segloc(segFtVLAD1, segFtVLAD2, gt, segRange2, imInds1, pca=True, k_search=200, k_vote=50, n=5):
    """(normalise if pca) -> add -> search 200 -> keep 50 -> 2-d^2 -> wt_borda_Im n=5 -> recall@1..5."""
    R = np.asarray(segFtVLAD1)
    Q = np.asarray(segFtVLAD2)
    if pca:
        R, Q = normalize_feat(R), normalize_feat(Q)
    d2, idx = knn_l2(R, Q, k_search)
    sims = (2 - d2[:, :k_vote]).astype(np.float32)
    m50 = idx[:, :k_vote]
    preds = get_matches_wt_borda_im(m50, len(gt), sims, segRange2, np.asarray(imInds1), n=n)
    return calc_recall(preds, gt, n), preds, m50, sims


def gt_17places(n_query: int, loc_rad: int = 15):
    """gt.py:60-64."""
    return [list(np.arange(i - loc_rad, i + loc_rad + 1)) for i in range(n_query)]


# --------------------------------------------------------------------------------------------
# f5  AnyLoc global VLAD                                   func_vpr.py:886-946 (vlad, segment=False), utilities.py:819-890
# --------------------------------------------------------------------------------------------
def global_vlad(tokens_dn, c_centers):
    """VLAD.generate over ALL tokens of an image: the segment-VLAD of one all-token segment (float64; the reference
    returns float32)."""
    N = np.asarray(tokens_dn).shape[1]
    return seg_vlad(tokens_dn, np.ones((1, N), dtype=bool), c_centers, None)[0]


def get_recall_anyloc(db, q, gt, k=5):
    """func_vpr.py:834-884: exact top-k (fp64 brute force instead of the KDTree: same neighbours), recall@1..k in percent."""
    d2, ids = knn_l2(db, q, k)
    recall = np.zeros(k)
    n_eval = 0
    for i in range(len(q)):
        if len(gt[i]) == 0:
            continue
        n_eval += 1
        for j in range(k):
            if ids[i, j] in gt[i]:
                recall[j] += 1
                break
    return np.cumsum(recall) / float(n_eval) * 100, ids


# --------------------------------------------------------------------------------------------------
# f4: vocabulary k-means (vlad_c_centers_pt_gen.py:86-158 -> utilities.py:749-791 VLAD.fit ->
# fast_pytorch_kmeans.KMeans(mode='cosine')).  fast-pytorch-kmeans 0.2.0.1 is a third-party dependency with no source in
# the reference tree: its published Lloyd iteration is restated here -- PARITY UNPINNED for this function; the cosine
# assignment it uses is the pinned ``assign_labels`` above.
# --------------------------------------------------------------------------------------------------
def kmeans_cosine_step(X_unit: np.ndarray, C: np.ndarray):
    """One Lloyd half-step on unit rows ``X_unit [n, D]`` with centres ``C [K, D]``: cosine arg-max labels (first maximum),
    per-cluster sums of the assigned rows (fp64) and counts.  Returns (labels, gap, sums, counts)."""
    X32 = np.ascontiguousarray(X_unit, dtype=np.float32)
    labels, gap = assign_labels(X32, np.asarray(C, dtype=np.float32))
    K = C.shape[0]
    sums = np.zeros((K, X32.shape[1]), dtype=np.float64)
    np.add.at(sums, labels, X32.astype(np.float64))
    return labels, gap, sums, np.bincount(labels, minlength=K)
