"""Vocabulary k-means: the Lloyd iteration of the reference's VLAD.fit (cosine assignment,
centres = plain means of unit vectors), and the identity that lets the segment-VLAD kernels do its half-step on the
device: new centre = old centre + mean residual, with the residual sums recovered from the normalised VLAD output and
its block norms."""
import random

import numpy as np
import pytest

from revisit_anything_amd import synth, vocabulary as vc


def planted(K, D, n_per, seed, noise=0.15):
    rng = np.random.Generator(np.random.PCG64(seed))
    C = rng.standard_normal((K, D))
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    z = np.repeat(np.arange(K), n_per)
    X = C[z] + noise * rng.standard_normal((K * n_per, D)) / np.sqrt(D)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    perm = rng.permutation(len(z))
    return X[perm].astype(np.float32), z[perm], C


def test_recovers_planted_clusters_and_keeps_raw_means():
    X, z, C_true = planted(K=6, D=24, n_per=120, seed=1)
    best = None
    for seed in range(4):            # random initial points, as the reference: take the best of a few starts
        C, labels, it = vc.cosine_kmeans(X, 6, seed=seed)
        purity = sum(np.bincount(z[labels == k]).max() for k in range(6) if (labels == k).any()) / len(z)
        best = max(best or (0,), (purity, seed, it))
        if purity == 1.0:
            break
    assert best[0] == 1.0 and best[2] < 100
    # centres are the plain means of the assigned unit vectors: norm < 1 (utilities.py keeps them un-normalised)
    for k in range(6):
        assert np.allclose(C[k], X[labels == k].astype(np.float64).mean(0), atol=1e-6)
    norms = np.linalg.norm(C, axis=1)
    assert (norms < 1.0).all() and (norms > 0.9).all()


def test_half_step_identity_from_the_vlad_output():
    """What DeviceBackend.step computes from segvlad_images' outputs equals the plain sums: S_k = V_k + n_k C_k, with
    V_k = out_k * sqrt(#non-empty blocks) * ||V_k||.  The VLAD side is the CPU oracle here (test infrastructure)."""
    from oracle import segvlad_oracle as O

    K, D, N = 8, 32, 150
    C = synth.make_vocab(K, D, seed=3).astype(np.float64)
    tok = synth.make_tokens(C.astype(np.float32), N, seed=4, noise=0.4)          # [D, N]
    out, aux = O.seg_vlad(tok, np.ones((1, N), bool), C.astype(np.float32), None, return_aux=True)
    labels, bn = aux["labels"], aux["block_norms"][0]
    g = np.sqrt(len(np.unique(labels)))
    V = out.reshape(K, D) * (g * bn)[:, None]
    n_k = np.bincount(labels, minlength=K)
    S_from_vlad = V + n_k[:, None] * C.astype(np.float32).astype(np.float64)
    be = vc.NumpyBackend(tok.T)
    lab2, sums, counts = be.step(C.astype(np.float32).astype(np.float64))
    assert np.array_equal(lab2, labels) and np.array_equal(counts, n_k)
    assert np.allclose(S_from_vlad, sums, atol=1e-6)
    hit = n_k > 0
    assert np.allclose(C[hit] + V[hit] / n_k[hit, None], sums[hit] / n_k[hit, None], atol=1e-6)


def test_empty_clusters_keep_their_centre_and_arguments_are_checked():
    X, _, _ = planted(K=2, D=8, n_per=30, seed=5)
    C, labels, it = vc.cosine_kmeans(X, 5, seed=0, max_iter=50)
    assert C.shape == (5, 8) and np.isfinite(C).all() and set(labels) <= set(range(5))
    with pytest.raises(ValueError):
        vc.cosine_kmeans(X, 0)
    with pytest.raises(ValueError):
        vc.cosine_kmeans(X, len(X) + 1)


def test_image_and_token_sampling_rules():
    """vlad_c_centers_pt_gen.py:84-113: > 2000 images -> random.seed(42) 30 % sample and every 2nd token per axis."""
    keys = [f"im{i}.png" for i in range(2500)]
    chosen, sub = vc.choose_images(keys)
    random.seed(42)
    assert sub and chosen == random.sample(keys, k=750)
    few, sub2 = vc.choose_images(keys[:100])
    assert not sub2 and few == keys[:100]
    blk = np.arange(1 * 4 * 6 * 8, dtype=np.float32).reshape(1, 4, 6, 8) + 1
    x = vc.tokens_for_fit(blk, subsample=True)
    assert x.shape == (3 * 4, 4) and np.allclose(np.linalg.norm(x, axis=1), 1.0)
    assert np.allclose(x[1] * np.linalg.norm(blk[0, :, 0, 2]), blk[0, :, 0, 2])       # token (0, 2) is the second kept one
    assert vc.tokens_for_fit(blk, subsample=False).shape == (48, 4)
