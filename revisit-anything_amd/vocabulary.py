"""Vocabulary (cluster centres) for the VLAD stage.

The reference builds ``c_centers.pt`` with ``VLAD(num_c).fit(tokens)`` (vlad_c_centers_pt_gen.py:86-158,
utilities.py:749-791): the L2-normalised patch tokens of the (sampled) reference images go through
``fast_pytorch_kmeans.KMeans(K, mode='cosine')`` -- Lloyd iterations with cosine assignment, centres = plain means of
the assigned unit vectors (NOT re-normalised: their norms, 0.3-1.0, matter because the VLAD residual is taken against
the raw centre, func_vpr.py:1151), random initial points, ``max_iter=100``, ``tol=1e-4`` on the squared centre shift.
fast-pytorch-kmeans 0.2.0.1 is a third-party dependency without source in the reference tree: this restates its
published algorithm (parity unpinned).

On the device a half-step is ``segvlad_kmeans_step`` (round 6): the assignment kernel of the VLAD stage + a centroid-update
kernel of its own (``csrc/kmeans_kernels.hip``: per-cluster sums of the normalised tokens in a fixed order, no atomics on the
sums, images reduced in fp64).  Rounds 3-5 needed no new kernel: for one pseudo-segment covering all tokens of an image the
VLAD block of cluster k is ``V_k = sum_{t in k} (x_t - C_k) = S_k - n_k C_k`` with ``S_k`` the sum of the assigned unit tokens, so

    mean_k = S_k / n_k = C_k + V_k / n_k                      (new centre = old centre + mean residual)

and ``V_k`` is recovered from the normalised output of ``segvlad_images`` and its block norms -- kept as
``DeviceBackend.step_from_vlad``, a cross-check: exact algebra, but a cluster whose residual sum is tiny against ``n_k C_k``
(tokens tight around their centre: the converged state) gets its sum back through a cancellation.  ``NumpyBackend`` is the plain
CPU form; tests/test_vocabulary.py checks the identity."""
from __future__ import annotations

import random
from typing import List, Optional, Sequence, Tuple

import numpy as np


# ---- which tokens go into the fit (vlad_c_centers_pt_gen.py:84-113) --------------------------------------------------
def choose_images(keys: Sequence[str], sample_threshold: int = 2000, sample_percentage: float = 0.3) -> Tuple[List[str], bool]:
    """More than ``sample_threshold`` images: ``random.seed(42); random.sample(keys, 30 %)`` and every 2nd token in both
    directions; otherwise everything.  Returns (keys to process, subsample flag)."""
    keys = list(keys)
    if len(keys) > sample_threshold:
        random.seed(42)
        return random.sample(keys, k=int(len(keys) * sample_percentage)), True
    return keys, False


def tokens_for_fit(ift_dino: np.ndarray, subsample: bool) -> np.ndarray:
    """``[1, D, h, w]`` -> unit rows ``[n, D]`` (``[:, :, ::2, ::2]`` when subsampling, :107-113)."""
    a = np.asarray(ift_dino, dtype=np.float32)
    if subsample:
        a = a[:, :, ::2, ::2]
    x = a.reshape(a.shape[1], -1).T
    return x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)


# ---- backends: one Lloyd half-step = (labels, per-cluster sums, counts) ---------------------------------------------
class NumpyBackend:
    def __init__(self, X: np.ndarray):
        X = np.asarray(X, dtype=np.float64)
        self.X = X / np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-12)
        self.n, self.d = self.X.shape

    def init_points(self, idx: np.ndarray) -> np.ndarray:
        return self.X[idx].copy()

    def step(self, C: np.ndarray):
        Cn = C / np.maximum(np.linalg.norm(C, axis=1, keepdims=True), 1e-12)
        labels = (self.X @ Cn.T).argmax(axis=1)                 # first maximum, as torch.max / the assign kernel
        K = C.shape[0]
        sums = np.zeros((K, self.d))
        np.add.at(sums, labels, self.X)
        counts = np.bincount(labels, minlength=K)
        return labels, sums, counts


class DeviceBackend:
    """The same half-step from the segment-VLAD kernels (one all-token pseudo-segment per image).  ``tokens``:
    ``[B, D, N]`` fp32 on the engine's device, as the reference stores them.  GPU-tested against NumpyBackend.step
    (labels / counts bit for bit, sums to 1e-6 per token: tests/test_gpu_frows.py)."""

    def __init__(self, engine, tokens, batch: int = 64):
        import torch

        self.eng, self.batch = engine, batch
        self.tok = tokens if isinstance(tokens, torch.Tensor) else torch.as_tensor(np.asarray(tokens, dtype=np.float32))
        self.tok = self.tok.to(engine.device, dtype=torch.float32).contiguous()
        self.B, self.d, self.N = self.tok.shape
        self.n = self.B * self.N
        nw = (self.N + 63) // 64
        ones = np.zeros(nw, np.uint64)
        for t in range(self.N):
            ones[t >> 6] |= np.uint64(1) << np.uint64(t & 63)
        self.bits_row = torch.from_numpy(ones.view(np.int64))

    def init_points(self, idx: np.ndarray) -> np.ndarray:
        import torch

        b, t = np.divmod(np.asarray(idx), self.N)
        x = self.tok[torch.as_tensor(b), :, torch.as_tensor(t)].double()
        return torch.nn.functional.normalize(x, dim=1).cpu().numpy()

    def step(self, C: np.ndarray):
        """One half-step through ``segvlad_kmeans_step`` (round 6: the assignment kernel + a centroid-update kernel of its own --
        per-cluster sums of the normalised tokens in a fixed order, images reduced in fp64; csrc/kmeans_kernels.hip)."""
        import torch

        K = C.shape[0]
        self.eng.set_vocab(np.ascontiguousarray(C, dtype=np.float32))
        sums = torch.zeros(K, self.d, dtype=torch.float64, device=self.eng.device)
        counts = torch.zeros(K, dtype=torch.int64, device=self.eng.device)
        labels = []
        for b0 in range(0, self.B, self.batch):
            nb = min(self.batch, self.B - b0)
            labels.append(self.eng.kmeans_step(self.tok[b0:b0 + nb], sums, counts, want_labels=True).reshape(-1).long())
        return torch.cat(labels).cpu().numpy(), sums.cpu().numpy(), counts.cpu().numpy()

    def step_from_vlad(self, C: np.ndarray):
        """The round-3 form of the same half-step, kept as a cross-check: the sums recovered from the NORMALISED output of the
        segment-VLAD kernels (one all-token pseudo-segment per image), S_k = out_k * ||V_k|| * sqrt(#blocks) + n_k C_k.  Exact
        algebra, but a cluster whose residual sum is tiny against n_k C_k gets its sum back through a cancellation."""
        import torch

        K = C.shape[0]
        self.eng.set_vocab(np.ascontiguousarray(C, dtype=np.float32))
        sums = torch.zeros(K, self.d, dtype=torch.float64, device=self.eng.device)
        counts = torch.zeros(K, dtype=torch.int64, device=self.eng.device)
        labels = []
        Cd = torch.as_tensor(np.asarray(C, dtype=np.float64)).to(self.eng.device)
        for b0 in range(0, self.B, self.batch):
            nb = min(self.batch, self.B - b0)
            bits = self.bits_row.repeat(nb, 1).to(self.eng.device)
            r = self.eng.seg_vlad(self.tok[b0:b0 + nb], bits, np.arange(nb + 1, dtype=np.int32), None, want_labels=True,
                                  want_block_norms=True)
            lab = r["labels"].long()                                             # [nb, N]
            bn = r["block_norms"].double()                                       # [nb, K]  ||V_k|| before intra-norm
            out = r["out"].double().view(nb, K, self.d)                          # V_k / ||V_k|| / sqrt(#non-empty)
            nonempty = torch.zeros(nb, K, dtype=torch.float64, device=lab.device)
            nonempty.scatter_(1, lab, 1.0)
            g = nonempty.sum(1, keepdim=True).clamp_min(1.0).sqrt()              # sqrt(#non-empty blocks)
            V = out * (g * bn).unsqueeze(-1)                                     # raw residual sums V_k
            n_bk = torch.zeros(nb, K, dtype=torch.float64, device=lab.device).scatter_add_(1, lab, torch.ones_like(lab, dtype=torch.float64))
            sums += (V + n_bk.unsqueeze(-1) * Cd.unsqueeze(0)).sum(0)            # S_k = V_k + n_k C_k
            counts += n_bk.sum(0).long()
            labels.append(lab.reshape(-1))
        return torch.cat(labels).cpu().numpy(), sums.cpu().numpy(), counts.cpu().numpy()


# ---- Lloyd iterations (fast_pytorch_kmeans.KMeans.fit, cosine mode) ---------------------------------------------------
def cosine_kmeans(X=None, num_clusters: int = 32, *, backend=None, max_iter: int = 100, tol: float = 1e-4,
                  seed: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, int]:
    """(centres [K, D] float32 -- means of the assigned unit vectors, labels [n], iterations run)."""
    be = backend if backend is not None else NumpyBackend(X)
    K = int(num_clusters)
    if not 0 < K <= be.n:
        raise ValueError(f"num_clusters={K} must be in 1..{be.n}")
    rng = np.random.Generator(np.random.PCG64(seed)) if seed is not None else np.random.default_rng()
    C = be.init_points(rng.choice(be.n, size=K, replace=False)).astype(np.float64)      # random initial points
    labels = None
    it = 0
    for it in range(1, max_iter + 1):
        labels, sums, counts = be.step(C)
        C_new = C.copy()
        hit = counts > 0
        C_new[hit] = sums[hit] / counts[hit, None]           # clusters that lost all points keep their centre
        shift = float(((C_new - C) ** 2).sum())
        C = C_new
        if shift <= tol:
            break
    return C.astype(np.float32), labels, it
