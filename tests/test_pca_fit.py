"""PCA fit: the subspace iteration against sklearn's exact solver -- the reference's fit
(place_rec_pca.py:339-342, svd_solver="arpack") is itself not reproducible (random sampling), so the acceptance is
spectrum / subspace agreement and invariance of the whitened distances, which is what the retrieval consumes."""
import numpy as np
import pytest
from sklearn.decomposition import PCA

from revisit_anything_amd import pca_fit as pf


def descriptor_like(n, kd, rank, seed, decay=0.92):
    rng = np.random.Generator(np.random.PCG64(seed))
    basis = np.linalg.qr(rng.standard_normal((kd, rank)))[0]
    scales = decay ** np.arange(rank)
    X = (rng.standard_normal((n, rank)) * scales) @ basis.T + 0.002 * rng.standard_normal((n, kd))
    X += 0.05 * rng.standard_normal(kd)                  # a non-zero mean: the centring must matter
    return X.astype(np.float32)


@pytest.mark.parametrize("n,kd,p", [(400, 300, 24), (250, 600, 40)])      # tall and wide (n < KD, the real case)
def test_fit_matches_sklearn_exact_solver(n, kd, p):
    X = descriptor_like(n, kd, rank=60, seed=n)
    mean, comps, var = pf.fit_pca(X, p, seed=1)
    ref = PCA(n_components=p, whiten=True, svd_solver="full").fit(X)
    assert mean.shape == (kd,) and comps.shape == (p, kd) and var.shape == (p,)
    assert np.allclose(mean, ref.mean_, atol=1e-5)
    assert np.allclose(comps @ comps.T, np.eye(p), atol=1e-5)                       # orthonormal rows
    assert np.allclose(var, ref.explained_variance_, rtol=2e-3)
    # same subspace: all principal angles ~ 0
    sv = np.linalg.svd(comps.astype(np.float64) @ ref.components_.T.astype(np.float64), compute_uv=False)
    assert sv.min() > 1 - 1e-4
    # what retrieval consumes: whitened coordinates agree up to the sign of a component; with svd_flip even the signs
    y = ((X - mean) @ comps.T) / np.sqrt(var)
    y_ref = ref.transform(X)
    top = slice(0, p // 2)                                                         # well-separated components
    assert np.allclose(np.abs(y[:, top]), np.abs(y_ref[:, top]), atol=2e-2 * np.abs(y_ref).max())
    # signs: svd_flip on the LEFT vectors (sklearn 1.3.2, the reference's pin; newer sklearn decides on the components):
    # in every score column the entry of largest magnitude is positive
    scores = (X - mean) @ comps.T
    assert (scores[np.abs(scores).argmax(axis=0), np.arange(p)] > 0).all()
    d = ((y[:40, None, :] - y[None, :40, :]) ** 2).sum(-1)
    d_ref = ((y_ref[:40, None, :] - y_ref[None, :40, :]) ** 2).sum(-1)
    assert np.allclose(d, d_ref, rtol=5e-3, atol=1e-6 * d_ref.max())


def test_fit_feeds_the_projection_surface(tmp_path):
    """fit -> save -> load -> the arithmetic of apply_pca_transform_from_pkl (func_vpr.py:1434-1443)."""
    from revisit_anything_amd import store as st

    X = descriptor_like(300, 200, rank=30, seed=5)
    mean, comps, var = pf.fit_pca(X, 16)
    st.save_pca(str(tmp_path / "pca.npz"), mean, comps, var, whiten=True)
    m2, c2, v2, w2 = st.load_pca(str(tmp_path / "pca.npz"))
    y = ((X[:5] - m2) @ c2.T) / np.sqrt(v2)
    assert w2 and y.shape == (5, 16) and np.isfinite(y).all()
    assert abs(y.std() - 1.0) < 0.5                       # whitened coordinates have unit-order variance


def test_sampling_rule_of_the_reference():
    """place_rec_pca.py:330-334, 385-398: int(S_img * min(1, max/total)) rows per image, stop at max."""
    rng = np.random.Generator(np.random.PCG64(7))
    blocks = [np.full((s, 3), i, np.float32) for i, s in enumerate([50, 10, 0, 33, 80, 80, 80])]
    total = sum(b.shape[0] for b in blocks)
    got = pf.sample_segments(blocks, total, max_segments=100, rng=rng)
    ratio = 100 / total
    per_img = [int(b.shape[0] * ratio) for b in blocks]
    assert got.shape[0] == sum(per_img) and got.shape[0] <= 100
    assert [int((got[:, 0] == i).sum()) for i in range(len(blocks))] == per_img
    everything = pf.sample_segments(blocks, total, max_segments=10 ** 6, rng=rng)
    assert everything.shape[0] == total
    early = pf.sample_segments(blocks, 60, max_segments=60, rng=rng)      # under-counted total: stops once 60 are in
    assert early.shape[0] == 60 and set(early[:, 0]) <= {0.0, 1.0}


def test_bad_arguments():
    X = descriptor_like(50, 40, rank=10, seed=2)
    with pytest.raises(ValueError):
        pf.fit_pca(X, 0)
    with pytest.raises(ValueError):
        pf.fit_pca(X, 45)
