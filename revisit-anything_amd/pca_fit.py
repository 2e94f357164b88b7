"""PCA fit for the segment descriptors: the model ``segvlad_pca_set`` consumes.

The reference fits ``sklearn.decomposition.PCA(n_components=1024, whiten=True, svd_solver="arpack")`` on at most
50 000 segment descriptors sampled image by image (place_rec_pca.py:330-342, 380-411) and pickles the model.  For
K*D = 49 152 .. 98 304 columns that is the largest CPU cost of the whole reference pipeline.  Here the fit is a block
subspace iteration with a Rayleigh-Ritz step, written against two products only,

    X_c V   ([n, KD] x [KD, p])      and      X_c^T U   ([KD, n] x [n, p]),      X_c = X - 1 mean^T,

so that the data never has to be centred or copied: the centring enters as rank-one corrections.  Both products are
NT GEMMs of exactly the shape ``segvlad_pca_apply`` runs on the matrix cores (``EnginePcaBackend`` below drives them
through the C-ABI; ``NumpyBackend`` is the CPU form used by the tests).  The result follows sklearn's conventions:
``components_`` rows orthonormal, ``explained_variance_ = sigma^2 / (n - 1)``, signs fixed by ``svd_flip`` (largest
|u| positive).  The reference's own fit is not reproducible (random sampling, place_rec_pca.py:389), so the acceptance
test is subspace / spectrum agreement with sklearn's exact solver and invariance of the whitened distances
(tests/test_pca_fit.py), not element-wise equality."""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import numpy as np


# ---- the reference's sampling rule (place_rec_pca.py:330-334, 385-398) ----------------------------------------------
def sample_segments(batches: Iterable[np.ndarray], num_segments_total: int, max_segments: int = 50000,
                    rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """Rows for the fit: from every image's descriptor block ``int(S_img * ratio)`` rows picked at random,
    ``ratio = min(1, max_segments / num_segments_total)``, stopping once ``max_segments`` are accumulated."""
    rng = rng or np.random.default_rng()
    ratio = min(1.0, max_segments / max(int(num_segments_total), 1))
    out: List[np.ndarray] = []
    acc = 0
    for gd in batches:
        gd = np.asarray(gd, dtype=np.float32)          # the reference converts to float32 "to keep RAM in check" (:382)
        k = int(gd.shape[0] * ratio)
        if k > 0:
            out.append(gd[rng.permutation(gd.shape[0])[:k]])
            acc += k
        if acc >= max_segments:
            break
    return np.concatenate(out) if out else np.zeros((0, 0), np.float32)


# ---- product backends -----------------------------------------------------------------------------------------------
class NumpyBackend:
    """X V and X^T U on the host (float64 accumulation)."""

    def __init__(self, X: np.ndarray):
        self.X = np.asarray(X)
        self.n, self.kd = self.X.shape

    def col_mean(self) -> np.ndarray:
        return self.X.mean(axis=0, dtype=np.float64)

    def xv(self, V: np.ndarray) -> np.ndarray:            # [n, p]
        return self.X.astype(np.float64, copy=False) @ V

    def xtu(self, U: np.ndarray) -> np.ndarray:           # [KD, p]
        return self.X.T.astype(np.float64, copy=False) @ U


class EnginePcaBackend:
    """The same two products on the device through the C-ABI's projection GEMM: ``X V = pca_apply(X)`` with the model
    ``comps = V^T``; ``X^T U = pca_apply(U^T)^T`` with ``comps = X^T`` (rows = descriptor columns).  X stays resident
    in HBM ([n, KD] fp32; 50 000 x 98 304 = 19.7 GB).  GPU-tested against sklearn's exact solver
    (tests/test_gpu_frows.py); the CPU tests cover the algorithm through NumpyBackend."""

    def __init__(self, engine, X):
        import torch

        self.eng = engine
        self.X = X if isinstance(X, torch.Tensor) else torch.as_tensor(np.asarray(X, dtype=np.float32))
        self.X = self.X.to(engine.device, dtype=torch.float32).contiguous()
        self.n, self.kd = self.X.shape
        self._Xt = None

    def col_mean(self) -> np.ndarray:
        return self.X.mean(dim=0, dtype=self.X.dtype).double().cpu().numpy()

    def xv(self, V: np.ndarray) -> np.ndarray:
        import torch

        comps = torch.as_tensor(np.ascontiguousarray(V.T, dtype=np.float32)).to(self.eng.device)
        self.eng.pca_set(None, comps, None, whiten=False)
        return self.eng.pca_apply(self.X, l2norm=False).double().cpu().numpy()

    def xtu(self, U: np.ndarray) -> np.ndarray:
        import torch

        if self._Xt is None:
            self._Xt = self.X.t().contiguous()            # [KD, n]: the "components" of the transposed product
        self.eng.pca_set(None, self._Xt, None, whiten=False)
        ut = torch.as_tensor(np.ascontiguousarray(U.T, dtype=np.float32)).to(self.eng.device)   # [p, n]
        return self.eng.pca_apply(ut, l2norm=False).double().cpu().numpy().T


# ---- the fit ---------------------------------------------------------------------------------------------------------
def fit_pca(X=None, n_components: int = 1024, *, backend=None, n_oversamples: int = 32, n_iter: int = 8,
            seed: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(mean [KD], components [P, KD], explained_variance [P]) of the rows of X (or of ``backend``'s matrix)."""
    be = backend if backend is not None else NumpyBackend(X)
    n, kd = be.n, be.kd
    p = int(n_components)
    if not 0 < p <= min(n - 1, kd):
        raise ValueError(f"n_components={p} must be in 1..min(n-1, KD) = {min(n - 1, kd)}")
    q = min(p + int(n_oversamples), min(n, kd))
    mean = be.col_mean()

    def xc_v(V):       # X_c V = X V - 1 (mean^T V)
        return be.xv(V) - (mean @ V)[None, :]

    def xct_u(U):      # X_c^T U = X^T U - mean (1^T U)
        return be.xtu(U) - np.outer(mean, U.sum(axis=0))

    rng = np.random.Generator(np.random.PCG64(seed))
    V = np.linalg.qr(rng.standard_normal((kd, q)))[0]
    for _ in range(max(int(n_iter), 1)):           # subspace iteration on X_c^T X_c, re-orthonormalised every half step
        U = np.linalg.qr(xc_v(V))[0]
        V = np.linalg.qr(xct_u(U))[0]
    B = xc_v(V)                                     # [n, q]: X_c restricted to the subspace
    Ub, s, Wt = np.linalg.svd(B, full_matrices=False)            # Rayleigh-Ritz: X_c ~ Ub diag(s) (V Wt^T)^T
    comps = (V @ Wt.T).T[:p]                        # [p, KD], orthonormal rows
    Ub, s = Ub[:, :p], s[:p]
    # sklearn's svd_flip (u-based): the entry of largest magnitude in each left singular vector is positive
    signs = np.sign(Ub[np.abs(Ub).argmax(axis=0), np.arange(p)])
    signs[signs == 0] = 1.0
    comps *= signs[:, None]
    var = (s ** 2) / (n - 1)
    return mean.astype(np.float32), comps.astype(np.float32), var.astype(np.float32)
