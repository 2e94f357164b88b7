"""Seeded synthetic inputs for the SegVLAD hot path.

NumPy only; used by ``bench.py``, ``tools/make_golden.py`` and the tests so that every party sees
bit-identical inputs from a seed.  Nothing here computes any part of the hot path.
"""
from __future__ import annotations

import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def _l2n(x, axis=-1):
    return x / np.linalg.norm(x, axis=axis, keepdims=True)


def make_vocab(K: int, D: int, seed: int = 1000) -> np.ndarray:
    """``normalize(g) * u``, u ~ U(0.35, 1.0): row norms like the shipped vocabularies (0.31-0.99)."""
    r = _rng(seed)
    g = r.standard_normal((K, D))
    u = r.uniform(0.35, 1.0, size=(K, 1))
    return (_l2n(g) * u).astype(np.float32)


def make_tokens(C: np.ndarray, N: int, seed: int, noise: float = 0.05, adversarial: bool = False) -> np.ndarray:
    """Token block as stored by the reference: fp32 ``[D][N]`` (N contiguous), unit-norm columns.

    ``x_t = normalize(C[z_t] + noise * g_t)`` (balanced clusters) or ``normalize(g_t)`` (near-tied
    assignments, for the tie audit)."""
    r = _rng(seed)
    K, D = C.shape
    g = r.standard_normal((N, D))
    if adversarial:
        x = g
    else:
        z = r.integers(0, K, size=N)
        x = C[z].astype(np.float64) + noise * g
    x = _l2n(x).astype(np.float32)
    return np.ascontiguousarray(x.T)


def make_masks(S: int, Hm: int, Wm: int, seed: int, hmin=8, hmax=60, wmin=8, wmax=80) -> np.ndarray:
    """S axis-aligned rectangles on an Hm x Wm grid, bool ``[S,Hm,Wm]``."""
    r = _rng(seed)
    m = np.zeros((S, Hm, Wm), dtype=bool)
    for s in range(S):
        h = int(r.integers(min(hmin, Hm), min(hmax, Hm) + 1))
        w = int(r.integers(min(wmin, Wm), min(wmax, Wm) + 1))
        y0 = int(r.integers(0, Hm - h + 1))
        x0 = int(r.integers(0, Wm - w + 1))
        m[s, y0:y0 + h, x0:x0 + w] = True
    return m


def make_blob_masks(S: int, Hm: int, Wm: int, seed: int) -> np.ndarray:
    """Irregular (non-rectangular) masks: thresholded random ellipses with holes; exercises ragged
    token coverage.  bool ``[S,Hm,Wm]``, every mask non-empty."""
    r = _rng(seed)
    yy, xx = np.mgrid[0:Hm, 0:Wm]
    m = np.zeros((S, Hm, Wm), dtype=bool)
    for s in range(S):
        cy, cx = r.uniform(0, Hm), r.uniform(0, Wm)
        ry, rx = r.uniform(3, Hm / 3), r.uniform(3, Wm / 3)
        e = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        e &= r.random((Hm, Wm)) > 0.15
        if not e.any():
            e[int(min(max(cy, 0), Hm - 1)), int(min(max(cx, 0), Wm - 1))] = True
        m[s] = e
    return m


def make_planted_db(n_img: int, S: int, d: int, seed: int = 3000, group: int = 4, sigma_r: float = 0.05,
                    dtype=np.float32):
    """Planted near-duplicate reference descriptors: images in groups of ``group`` share
    unit-norm base rows; ``r[i,j] = normalize(b[i//group, j] + sigma_r * g / sqrt(d))``.
    Returns (R [n_img*S, d], img_of_seg int32 [n_img*S])."""
    r = _rng(seed)
    ng = (n_img + group - 1) // group
    base = _l2n(r.standard_normal((ng, S, d)))
    R = np.empty((n_img, S, d), dtype=dtype)
    for i in range(n_img):
        R[i] = _l2n(base[i // group] + sigma_r * r.standard_normal((S, d)) / np.sqrt(d)).astype(dtype)
    img = np.repeat(np.arange(n_img, dtype=np.int32), S)
    return R.reshape(n_img * S, d), img


def make_planted_queries(R: np.ndarray, n_img: int, S: int, n_query: int, seed: int = 4000, sigma_q: float = 4.0):
    """Query image = random reference image tau with per-row noise sigma_q; GT = {tau}.
    Returns (Q [n_query*S, d], tau int64 [n_query], seg_offsets int32 [n_query+1])."""
    r = _rng(seed)
    d = R.shape[1]
    tau = r.integers(0, n_img, size=n_query)
    Q = np.empty((n_query, S, d), dtype=R.dtype)
    R3 = R.reshape(n_img, S, d)
    for i, t in enumerate(tau):
        Q[i] = _l2n(R3[t].astype(np.float64) + sigma_q * r.standard_normal((S, d)) / np.sqrt(d)).astype(R.dtype)
    off = (np.arange(n_query + 1) * S).astype(np.int32)
    return Q.reshape(n_query * S, d), tau.astype(np.int64), off


def make_pca_model(KD: int, P: int, seed: int = 5000):
    """A synthetic affine PCA model (mean, components [P,KD], explained_variance [P]) in fp32.
    Parity for the PCA path is defined *given* a model, so any full-rank map works."""
    r = _rng(seed)
    mean = (r.standard_normal(KD) * (0.2 / np.sqrt(KD))).astype(np.float32)
    comps = (r.standard_normal((P, KD)) / np.sqrt(KD)).astype(np.float32)
    var = (np.geomspace(1e-3, 1e-6, P)).astype(np.float32)
    return mean, comps, var
