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


# ---- the fit, resident on the device (the reference's scale: 50 000 x 49 152 .. 98 304 -> 1024) ------------------------------
def _chol_qr2_cols(Y, shift_tries: int = 3):
    """Orthonormalise the COLUMNS of a tall device matrix Y [m, q] (fp32) by CholeskyQR2: G = Y^T Y in fp64 on the device,
    its q x q Cholesky factor on the host (q ~ 1000: milliseconds), Y <- Y R^-1 on the device; twice (the second pass removes
    the kappa^2 loss of the first).  Returns fp32 [m, q].  A Gram matrix that is not positive definite in fp64 (numerically
    rank-deficient block) is regularised by a relative shift and the pass repeated."""
    import torch

    for _ in range(2):
        Yd = Y.double()
        G = (Yd.t() @ Yd).cpu().numpy()
        scale = float(np.trace(G)) / G.shape[0]
        R = None
        for t in range(shift_tries + 1):
            try:
                R = np.linalg.cholesky(G + (0.0 if t == 0 else scale * 10.0 ** (-14 + 2 * t)) * np.eye(G.shape[0])).T   # G = R^T R
                break
            except np.linalg.LinAlgError:
                continue
        if R is None:   # hopeless block: Householder QR on the host (never seen on descriptor data)
            return torch.as_tensor(np.linalg.qr(Yd.cpu().numpy())[0].astype(np.float32)).to(Y.device)
        Rinv = np.linalg.solve(R, np.eye(R.shape[0]))
        Y = (Yd @ torch.as_tensor(Rinv).to(Y.device)).float()
        del Yd
    return Y


def fit_pca_device(engine, X, n_components: int = 1024, *, n_oversamples: int = 32, n_iter: int = 8, seed: int = 0,
                   timings: Optional[dict] = None, stats: Optional[dict] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``fit_pca`` with everything but two q x q factorisations on the device: X [n, KD] fp32 stays in HBM (50 000 x 98 304 =
    19.7 GB, + its transpose), the two tall products of every half step run on the C-ABI's projection GEMM
    (segvlad_pca_apply: three fp16 MFMA products of a two-term split, fp32-class) in TWO contexts -- one keeps ``X^T`` as
    its "components" for the whole fit, the other receives the current basis -- and the re-orthonormalisation is
    CholeskyQR2 (Gram matrices in fp64 on the device, q x q Cholesky on the host).  The Rayleigh-Ritz step diagonalises the
    q x q Gram matrix of X_c V instead of taking an SVD of the [n, q] block.  Same conventions / return value as fit_pca.
    (Host QR of the [98 304, 1056] and [50 000, 1056] blocks was > 90 % of the round-2 fit.)

    ``engine`` only names the device: both products run in contexts of the fit's own, so the caller's engine -- its PCA model
    in particular -- is left exactly as it was (round 3 left the last iteration basis in it).  ``stats`` (optional dict)
    receives ``total_variance``: the sum of the columns' sample variances, which sklearn's own fit derives
    ``explained_variance_ratio_`` and ``noise_variance_`` from (to_sklearn_pca)."""
    import time

    import torch

    from .engine import SegVLADEngine

    t_start = time.perf_counter()
    dev = engine.device
    X = X if isinstance(X, torch.Tensor) else torch.as_tensor(np.asarray(X, dtype=np.float32))
    X = X.to(dev, dtype=torch.float32).contiguous()
    n, kd = int(X.shape[0]), int(X.shape[1])
    p = int(n_components)
    if not 0 < p <= min(n - 1, kd):
        raise ValueError(f"n_components={p} must be in 1..min(n-1, KD) = {min(n - 1, kd)}")
    q = min(p + int(n_oversamples), min(n, kd))
    q = (q + 3) & ~3 if ((q + 3) & ~3) <= min(n, kd) else q
    mean64 = X.mean(dim=0, dtype=torch.float64)                                     # [KD]
    # the transposed product's inner dimension is n: padded with zero columns to a multiple of 32 so that it stays on the
    # split fp16 GEMM (api.hip: KD % 32), which changes no sum
    n_pad = (n + 31) // 32 * 32
    Xt = torch.zeros((kd, n_pad), dtype=torch.float32, device=dev)
    Xt[:, :n] = X.t()
    eng_t = SegVLADEngine(dev)                                                      # comps = X^T, set once
    eng_t.pca_set(None, Xt, None, whiten=False)
    del Xt
    eng_v = SegVLADEngine(dev)                                                      # comps = the current basis, set every half step
    if stats is not None:   # sum of the columns' sample variances, in fp64, a block of rows at a time
        tv = torch.zeros((), dtype=torch.float64, device=dev)
        for r0 in range(0, n, 2048):
            blk = X[r0:r0 + 2048].double() - mean64[None, :]
            tv += (blk * blk).sum()
            del blk
        stats["total_variance"] = float(tv.item()) / max(n - 1, 1)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    Vt = _chol_qr2_cols(torch.randn(kd, q, device=dev, generator=g)).t().contiguous()          # [q, KD], orthonormal rows

    def xc_v(Vt_):       # X_c V = X V - 1 (mean^T V): [n, q]
        eng_v.pca_set(None, Vt_, None, whiten=False)
        Y = eng_v.pca_apply(X, l2norm=False)
        return Y - (Vt_.double() @ mean64).float()[None, :]

    def xct_u_t(U):      # (X_c^T U)^T = U^T X - (U^T 1) mean^T: [q, KD]
        Ut = torch.zeros((q, n_pad), dtype=torch.float32, device=dev)
        Ut[:, :n] = U.t()
        Z = eng_t.pca_apply(Ut, l2norm=False)
        return Z - torch.outer(U.double().sum(dim=0), mean64).float()

    t_setup = time.perf_counter()
    for _ in range(max(int(n_iter), 1)):
        U = _chol_qr2_cols(xc_v(Vt))                                               # [n, q]
        Vt = _chol_qr2_cols(xct_u_t(U).t().contiguous()).t().contiguous()          # [q, KD]
    t_iter = time.perf_counter()
    B = xc_v(Vt)                                                                   # [n, q] = X_c V
    Bd = B.double()
    M = (Bd.t() @ Bd).cpu().numpy()                                                # q x q, = W diag(s^2) W^T
    w, Wm = np.linalg.eigh(M)
    order = np.argsort(w)[::-1][:p]
    s2 = np.maximum(w[order], 0.0)
    Wp = torch.as_tensor(np.ascontiguousarray(Wm[:, order])).to(dev)               # [q, p]
    comps = (Wp.t() @ Vt.double())                                                 # [p, KD] = (V W)^T, orthonormal rows
    # sklearn's svd_flip (u-based): u_j = B w_j / s_j; the entry of largest magnitude of every u_j is made positive
    Uj = Bd @ Wp                                                                   # [n, p] (= u_j s_j: the sign is that of u_j)
    rows = Uj.abs().argmax(dim=0)
    signs = torch.sign(Uj[rows, torch.arange(p, device=dev)])
    signs[signs == 0] = 1.0
    comps = (comps * signs[:, None]).float().cpu().numpy()
    var = (s2 / (n - 1)).astype(np.float32)
    eng_t.close()
    eng_v.close()
    if timings is not None:
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        timings.update(setup_s=t_setup - t_start, iterations_s=t_iter - t_setup, rayleigh_ritz_s=t_end - t_iter,
                       total_s=t_end - t_start, n=n, kd=kd, p=p, q=q, n_iter=int(n_iter))
    return mean64.float().cpu().numpy(), comps, var


def to_sklearn_pca(mean, components, explained_variance, n_samples: int, whiten: bool = True, total_variance: Optional[float] = None):
    """A ``sklearn.decomposition.PCA`` carrying the fitted model -- what the reference pickles (place_rec_pca.py:403-411) and
    what its ``apply_pca_transform_from_pkl`` (func_vpr.py:1419-1443) unpickles and calls ``.transform`` on.
    ``total_variance`` (fit_pca_device's ``stats``): the data's total variance, from which ``explained_variance_ratio_`` and
    ``noise_variance_`` get the values sklearn's own fit gives them (sklearn/decomposition/_pca.py: ratio = variance / total,
    noise = (total - retained) / (min(n_samples, n_features) - n_components)); without it the two attributes are left as
    NaN rather than filled with numbers a reader could mistake for sklearn's (``.transform`` needs neither)."""
    from sklearn.decomposition import PCA

    comps = np.asarray(components, dtype=np.float32)
    var = np.asarray(explained_variance, dtype=np.float32)
    m = PCA(n_components=int(comps.shape[0]), whiten=bool(whiten), svd_solver="arpack")
    m.mean_ = np.asarray(mean, dtype=np.float32)
    m.components_ = comps
    m.explained_variance_ = var
    m.singular_values_ = np.sqrt(var.astype(np.float64) * max(int(n_samples) - 1, 1)).astype(np.float32)
    m.n_components_ = int(comps.shape[0])
    m.n_features_in_ = int(comps.shape[1])
    m.n_samples_ = int(n_samples)
    if total_variance is not None and total_variance > 0:
        m.explained_variance_ratio_ = (var.astype(np.float64) / float(total_variance)).astype(np.float32)
        rest = min(int(n_samples), int(comps.shape[1])) - int(comps.shape[0])
        m.noise_variance_ = max(float(total_variance) - float(var.astype(np.float64).sum()), 0.0) / rest if rest > 0 else 0.0
    else:
        m.explained_variance_ratio_ = np.full(var.shape, np.nan, dtype=np.float32)
        m.noise_variance_ = float("nan")
    return m


def fit_from_store(dino_in, masks_in, image_keys, pipeline, out_pkl: Optional[str] = None, n_components: int = 1024,
                   max_segments: int = 50000, num_segments_total: Optional[int] = None, batch_size: int = 100,
                   n_iter: int = 8, seed: int = 0, rng: Optional[np.random.Generator] = None, timings: Optional[dict] = None):
    """The reference's PCA-fitting run (place_rec_pca.py:320-411) as one call: count the reference split's segments, walk its
    images, describe each batch WITHOUT PCA on the device (raw K*D segment descriptors), keep ``int(S_img * ratio)`` randomly
    chosen rows of every image (ratio = min(1, max_segments / total); stop at ``max_segments``), fit on the device, and --
    with ``out_pkl`` -- pickle a ``sklearn.decomposition.PCA`` that ``apply_pca_transform_from_pkl`` (here or in the
    reference) loads.  ``pipeline``: a SegVLADPipeline with use_pca=False whose engine holds the vocabulary.
    Returns (mean, components, explained_variance)."""
    import pickle

    import torch

    from .driver import load_image_inputs
    from .func_vpr import getIdxSingleFast

    if getattr(pipeline, "use_pca", False):
        raise ValueError("fit_from_store needs a pipeline with use_pca=False (the fit sees the raw K*D descriptors)")
    rng = rng or np.random.default_rng()
    dev = pipeline.eng.device
    if num_segments_total is None:   # countNumMasksInDataset (func_vpr.py)
        num_segments_total = sum(len(getIdxSingleFast(i, list(load_image_inputs(dino_in, masks_in, k)[1]))[2])
                                 for i, k in enumerate(image_keys))
    ratio = min(1.0, max_segments / max(int(num_segments_total), 1))
    kept, acc = [], 0
    for b0 in range(0, len(image_keys), batch_size):
        keys = image_keys[b0:b0 + batch_size]
        toks, msks, offs = [], [], [0]
        for key in keys:
            t, m = load_image_inputs(dino_in, masks_in, key)
            toks.append(t)
            msks.append(m)
            offs.append(offs[-1] + m.shape[0])
        masks = np.concatenate([m for m in msks if m.shape[0]]) if any(m.shape[0] for m in msks) else np.zeros((0, 1, 1), np.uint8)
        gd = pipeline.describe(torch.from_numpy(np.stack(toks)).to(dev), torch.from_numpy(masks).to(dev),
                               np.asarray(offs, dtype=np.int32), l2norm=False)                    # [S_batch, K*D] on the device
        for j in range(len(keys)):                                                                  # per image, as :385-398
            s_img = offs[j + 1] - offs[j]
            k = int(s_img * ratio)
            if k > 0:
                pick = torch.as_tensor(rng.permutation(s_img)[:k] + offs[j], device=dev)
                kept.append(gd[pick])
                acc += k
            if acc >= max_segments:
                break
        if acc >= max_segments:
            break
    if not kept:
        raise ValueError("no segments sampled")
    X = torch.cat(kept)
    del kept
    st = {}
    mean, comps, var = fit_pca_device(pipeline.eng, X, n_components=n_components, n_iter=n_iter, seed=seed, timings=timings, stats=st)
    if out_pkl:
        with open(out_pkl, "wb") as f:
            pickle.dump(to_sklearn_pca(mean, comps, var, n_samples=int(X.shape[0]), whiten=True, total_variance=st.get("total_variance")), f)
    return mean, comps, var
