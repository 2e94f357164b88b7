"""SURVEY section 8f rows ON THE DEVICE: the PCA fit's device backend against sklearn's exact solver,
the vocabulary k-means half-step from the segment-VLAD kernels against its NumPy form, and the experiment loop
(driver.run_segloc over a store.FeatureStore) against the oracle's recall_segloc chain."""
import os

import numpy as np
import pytest
from conftest import engine_scope

pytestmark = pytest.mark.gpu


@pytest.fixture(scope=engine_scope)
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a ROCm device (no CPU fallback exists)"
    from revisit_anything_amd.engine import SegVLADEngine

    e = SegVLADEngine(0)
    yield e
    e.close()


def O():
    from oracle import segvlad_oracle

    return segvlad_oracle


def synth():
    from revisit_anything_amd import synth as s

    return s


# ------------------------------------------------------------------------------------------------
# f2: PCA fit (place_rec_pca.py:339-342) -- device products through segvlad_pca_apply
# ------------------------------------------------------------------------------------------------
def test_pca_fit_device_backend_matches_sklearn_exact_solver(eng):
    from sklearn.decomposition import PCA

    from revisit_anything_amd import pca_fit

    rng = np.random.Generator(np.random.PCG64(21))
    n, kd, p = 640, 1024, 16
    spectrum = np.geomspace(3.0, 0.2, 24)
    X = (rng.standard_normal((n, 24)) * spectrum) @ np.linalg.qr(rng.standard_normal((kd, 24)))[0].T
    X = (X + 0.01 * rng.standard_normal((n, kd)) + 0.3 * rng.standard_normal(kd)).astype(np.float32)   # + a non-zero mean
    ref = PCA(n_components=p, whiten=True, svd_solver="full").fit(X.astype(np.float64))
    be = pca_fit.EnginePcaBackend(eng, X)
    mean, comps, var = pca_fit.fit_pca(n_components=p, backend=be, n_iter=6, seed=3)
    mean_h, comps_h, var_h = pca_fit.fit_pca(X, n_components=p, n_iter=6, seed=3)             # NumpyBackend, same seed
    assert np.abs(mean - ref.mean_).max() < 1e-5
    assert np.allclose(var, ref.explained_variance_, rtol=2e-4)                                  # spectrum
    # subspace: principal angles between the two row spaces (orthonormal rows): singular values of A B^T all ~ 1
    sv = np.linalg.svd(comps.astype(np.float64) @ ref.components_.T, compute_uv=False)
    assert (1 - sv).max() < 1e-5
    # same signs / vectors where the spectrum separates them (all 16 do here)
    assert np.abs(np.abs((comps * ref.components_).sum(1)) - 1).max() < 1e-4
    # (signs: svd_flip's convention changed between the reference's sklearn 1.3.2 (u-based) and the one installed here;
    #  a whitened DISTANCE does not depend on them)
    # device products == host products (fp32-class split GEMM vs fp64): the two fits agree closely
    assert np.abs(comps - comps_h).max() < 5e-4 and np.allclose(var, var_h, rtol=1e-4)
    # downstream invariant: whitened distances agree with sklearn's transform
    Xt = X[:50].astype(np.float64)
    y_ref = ref.transform(Xt)
    y = O().pca_transform(Xt, mean, comps, var, True)
    d_ref = ((y_ref[:, None] - y_ref[None]) ** 2).sum(-1)
    d = ((y[:, None] - y[None]) ** 2).sum(-1)
    # (2e-3: two different MODELS are compared here -- six subspace iterations against sklearn's exact solver: eigenvalues
    #  to 2e-4, components to 1e-4, and whitening divides by sqrt(lambda) -- not two evaluations of one model)
    assert np.abs(d - d_ref).max() < 2e-3 * d_ref.max()
    # and the fitted model, loaded into the engine, projects like sklearn's (again model against model) ...
    eng.pca_set(mean, comps, var, whiten=True)
    yd = eng.pca_apply(X[:50]).cpu().numpy()
    assert np.abs(np.abs(yd) - np.abs(y_ref)).max() < 2e-3 * np.abs(y_ref).max()
    # ... while the device's APPLY of that model is held to the fp64 evaluation of the SAME model at the tolerance of every
    # other projection test (5e-5 relative)
    assert np.abs(yd - y).max() < 5e-5 * np.abs(y).max()


def test_pca_fit_on_the_device_at_a_reduced_real_shape_and_the_pickle_flow(eng, tmp_path):
    """fit_pca_device (CholeskyQR2 + Gram Rayleigh-Ritz on the device, no host QR) at descriptor width: n = 1024 rows of
    K*D = 49 152 columns -> 64 components, against sklearn's exact solver (the full 50 000 x 49 152 / 98 304 -> 1024 fit is
    timed by tools/probe_pca_fit.py: profiles/r03_pca_fit.json); then the place_rec_pca.py flow: the pickle
    written by to_sklearn_pca is a real sklearn PCA whose transform equals the engine's projection of the same model, and
    apply_pca_transform_from_pkl loads it."""
    import pickle

    import torch
    from sklearn.decomposition import PCA

    from revisit_anything_amd import func_vpr, pca_fit

    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    n, kd, p, r = 1024, 49152, 64, 96        # (sklearn's exact SVD of this block is ~15 s of host time; the device fit < 1 s)
    spec = torch.logspace(0, -1.5, r, device=dev)
    X = (torch.randn(n, r, device=dev, generator=g) * spec) @ (torch.randn(r, kd, device=dev, generator=g) / kd ** 0.5)
    X = X + 0.003 * torch.randn(n, kd, device=dev, generator=g) + 0.02 * torch.randn(kd, device=dev, generator=g) / kd ** 0.5
    tm, st = {}, {}
    # the caller's engine keeps ITS model through the fit (round 3 left the last iteration basis in it: ADVICE r03)
    marker = torch.eye(8, 64, device=dev)
    eng.pca_set(None, marker, None, whiten=False)
    probe = torch.arange(64, device=dev, dtype=torch.float32)[None, :].contiguous()
    before = eng.pca_apply(probe, l2norm=False).clone()
    mean, comps, var = pca_fit.fit_pca_device(eng, X, n_components=p, n_iter=6, seed=3, timings=tm, stats=st)
    assert torch.equal(eng.pca_apply(probe, l2norm=False), before)
    Xh = X.cpu().numpy()
    ref = PCA(n_components=p, whiten=True, svd_solver="full").fit(Xh.astype(np.float64))
    # the attributes sklearn's own fit derives from the total variance
    mdl = pca_fit.to_sklearn_pca(mean, comps, var, n_samples=n, whiten=True, total_variance=st["total_variance"])
    assert np.allclose(mdl.explained_variance_ratio_, ref.explained_variance_ratio_, rtol=1e-3)
    assert np.isclose(mdl.noise_variance_, ref.noise_variance_, rtol=1e-3)
    assert np.isnan(pca_fit.to_sklearn_pca(mean, comps, var, n_samples=n).noise_variance_)   # unknown total: not a made-up number
    assert np.abs(mean - ref.mean_).max() < 1e-6
    assert np.allclose(var, ref.explained_variance_, rtol=5e-4)
    sv = np.linalg.svd(comps.astype(np.float64) @ ref.components_.T, compute_uv=False)
    assert (1 - sv).max() < 1e-5                                                  # the same 128-dimensional subspace
    assert np.abs(np.abs((comps.astype(np.float64) * ref.components_).sum(1)) - 1).max() < 1e-3   # and the same vectors
    assert tm["q"] % 4 == 0 and tm["total_s"] < 60
    # n not a multiple of 32 (the transposed product pads its inner dimension): checked against the exact spectrum / row
    # space from the n x n Gram matrix of the centred rows (fp64 on the host)
    n2 = 999
    mean2, comps2, var2 = pca_fit.fit_pca_device(eng, X[:n2], n_components=16, n_iter=6, seed=3)
    Xc = Xh[:n2].astype(np.float64) - Xh[:n2].astype(np.float64).mean(0)
    w, Uu = np.linalg.eigh(Xc @ Xc.T)
    w, Uu = w[::-1][:16], Uu[:, ::-1][:, :16]
    assert np.allclose(var2, w / (n2 - 1), rtol=5e-4)
    exact_rows = (Xc.T @ Uu / np.sqrt(w)).T                                          # right singular vectors
    assert (1 - np.linalg.svd(comps2.astype(np.float64) @ exact_rows.T, compute_uv=False)).max() < 1e-5
    # the pickle of the reference's flow
    path = str(tmp_path / "pca_model.pkl")
    with open(path, "wb") as f:
        pickle.dump(pca_fit.to_sklearn_pca(mean, comps, var, n_samples=n, whiten=True), f)
    with open(path, "rb") as f:
        model = pickle.load(f)
    y_sk = model.transform(Xh[:40].astype(np.float64))                            # what the reference's loader computes
    y_ref = ref.transform(Xh[:40].astype(np.float64))
    assert np.abs(np.abs(y_sk) - np.abs(y_ref)).max() < 5e-3 * np.abs(y_ref).max()
    y_dev = func_vpr.apply_pca_transform_from_pkl(torch.from_numpy(Xh[:40]), path).numpy()
    # one model, two evaluations (the device's fp32-class split GEMM against sklearn's fp64 transform of the same pickle)
    assert np.abs(y_dev - y_sk).max() < 5e-5 * np.abs(y_sk).max()


# ------------------------------------------------------------------------------------------------
# f4: vocabulary k-means (utilities.py:766-787) -- Lloyd half-step from the segment-VLAD kernels
# ------------------------------------------------------------------------------------------------
def test_vocabulary_device_half_step_equals_numpy(eng):
    from revisit_anything_amd import vocabulary as vq

    K, D, B, N = 8, 64, 6, 300
    C0 = synth().make_vocab(K, D, seed=31)
    toks = np.stack([synth().make_tokens(C0, N, seed=3100 + b, noise=0.25) for b in range(B)])     # [B, D, N]
    X = np.concatenate([t.T for t in toks])                                                          # [B*N, D] unit rows
    nb, db = vq.NumpyBackend(X), vq.DeviceBackend(eng, toks, batch=4)
    C = nb.init_points(np.arange(0, B * N, B * N // K)[:K])
    assert np.abs(db.init_points(np.arange(0, B * N, B * N // K)[:K]) - C).max() < 1e-6
    for _ in range(2):
        ln, sn, cn = nb.step(C)
        ld, sd, cd = db.step(C)
        assert np.array_equal(ld, ln) and np.array_equal(cd, cn)                  # labels / counts bit for bit
        assert np.abs(sd - sn).max() < 1e-6 * max(1.0, float(cn.max()))             # sums: <= 1e-6 per accumulated token
        C = np.where(cn[:, None] > 0, sn / np.maximum(cn, 1)[:, None], C)
    cen_h, lab_h, it_h = vq.cosine_kmeans(X, K, seed=5, max_iter=25)
    cen_d, lab_d, it_d = vq.cosine_kmeans(num_clusters=K, backend=db, seed=5, max_iter=25)
    assert abs(it_h - it_d) <= 1 and (lab_h == lab_d).mean() > 0.999     # a token on a cell boundary may flip at 1e-6
    assert np.abs(cen_h - cen_d).max() < 1e-3
    # the centres are means of unit vectors (norm < 1), NOT re-normalised (the VLAD residual uses the raw centre)
    assert np.all(np.linalg.norm(cen_d, axis=1) < 1.0)


def test_vocabulary_half_step_at_the_reference_shape_against_the_oracle(eng):
    """K = 32 clusters of D = 1536 (the shipped vocabularies' shape: cache/vocabulary/dinov2_vitg14/l31_value_c32) over
    48 images x 1530 tokens = 73 440 unit rows: the device half-step (segment-VLAD kernels, one all-token pseudo-segment
    per image) against the ORACLE's pinned cosine assignment + fp64 sums -- labels identical wherever the oracle's top-2
    gap exceeds 1e-6, counts consistent, per-cluster sums within 1e-6 per accumulated token."""
    from revisit_anything_amd import vocabulary as vq

    K, D, B, N = 32, 1536, 48, 1530
    C0 = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocab_indoor_k32_d1536.npy"))   # a shipped vocabulary
    toks = np.stack([synth().make_tokens(C0, N, seed=3300 + b, noise=0.08) for b in range(B)])     # [B, D, N]
    X = np.concatenate([t.T for t in toks])
    db = vq.DeviceBackend(eng, toks, batch=16)
    rng = np.random.Generator(np.random.PCG64(8))
    C = db.init_points(rng.choice(B * N, size=K, replace=False))                                     # random initial points
    for _ in range(2):
        ld, sd, cd = db.step(C)
        lo, gap, so, co = O().kmeans_cosine_step(X, C.astype(np.float32))
        clear = gap > 1e-6
        assert clear.mean() > 0.999 and np.array_equal(ld[clear], lo[clear])
        assert cd.sum() == B * N and np.abs(cd - co).sum() <= 2 * int((~clear).sum())
        same = ld == lo
        if not same.all():                       # move the few near-tie rows to the device's side before comparing sums
            so = so.copy()
            for t in np.nonzero(~same)[0]:
                so[lo[t]] -= X[t].astype(np.float64)
                so[ld[t]] += X[t].astype(np.float64)
        assert np.abs(sd - so).max() < 1e-6 * max(1.0, float(cd.max()))
        C = np.where(cd[:, None] > 0, sd / np.maximum(cd, 1)[:, None], C)
    assert np.all(np.linalg.norm(C, axis=1) < 1.0 + 1e-6)


def test_kmeans_step_kernel_with_a_converged_and_an_empty_cluster(eng):
    """segvlad_kmeans_step (round 6, csrc/kmeans_kernels.hip) where the round-3 recovery of the sums from
    normalised VLAD descriptors is at its weakest: a cluster whose tokens sit ON their centre (||V_k|| << n_k ||C_k||: the
    converged state), a cluster that gets no token at all, ragged last column chunk (D = 200 is not a multiple of the kernel's
    chunk), accumulation over two calls, K = 70 > 64 (the narrow assignment kernel).  Against NumPy in fp64: labels and counts bit
    for bit, sums to 1e-6 RELATIVE per cluster; the old form stays within its 1e-6 per token."""
    import torch

    from revisit_anything_amd import vocabulary as vq

    K, D, B, N = 70, 200, 5, 333
    rng = np.random.Generator(np.random.PCG64(77))
    C = rng.standard_normal((K, D))
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    C *= rng.uniform(0.4, 1.0, (K, 1))                                   # raw centres: means of unit vectors have norm < 1
    z = rng.integers(0, K - 1, (B, N))                                   # cluster K-1 never drawn ...
    Cn = C / np.linalg.norm(C, axis=1, keepdims=True)
    X = Cn[z] + 0.3 * rng.standard_normal((B, N, D)) / np.sqrt(D)
    tight = z == 3
    X[tight] = Cn[3] + 1e-5 * rng.standard_normal((int(tight.sum()), D)) / np.sqrt(D)    # ... and cluster 3 sits on its centre
    X *= rng.uniform(0.5, 3.0, (B, N, 1))                                # un-normalised tokens: the kernel normalises them itself
    toks = np.ascontiguousarray(X.transpose(0, 2, 1)).astype(np.float32)                 # [B, D, N]
    db = vq.DeviceBackend(eng, toks, batch=2)
    nb = vq.NumpyBackend(toks.transpose(0, 2, 1).reshape(B * N, D))
    ln, sn, cn = nb.step(C.astype(np.float32).astype(np.float64))
    ld, sd, cd = db.step(C)
    assert np.array_equal(ld, ln) and np.array_equal(cd, cn) and cd[K - 1] == 0 and cd[3] == int(tight.sum()) > 10
    for k in range(K):
        assert np.abs(sd[k] - sn[k]).max() <= 1e-6 * max(np.abs(sn[k]).max(), 1e-30) * max(1, int(cn[k]) ** 0.5), k
    assert np.all(sd[K - 1] == 0)
    lv, sv, cv = db.step_from_vlad(C)
    assert np.array_equal(lv, ln) and np.array_equal(cv, cn) and np.abs(sv - sn).max() < 1e-6 * max(1.0, float(cn.max()))
    # the entry accumulates: a second pass over the same tokens doubles both outputs
    sums = torch.from_numpy(sd).to(eng.device)
    counts = torch.from_numpy(cd).to(eng.device)
    eng.kmeans_step(torch.from_numpy(toks).to(eng.device), sums, counts)
    assert np.array_equal(counts.cpu().numpy(), 2 * cn)
    assert np.abs(sums.cpu().numpy() - 2 * sn).max() < 4e-6 * max(1.0, float(cn.max()) ** 0.5)


# ------------------------------------------------------------------------------------------------
# f1: the experiment loop over stored inputs (place_rec_main.py:244-373)
# ------------------------------------------------------------------------------------------------
def test_fit_from_store_is_the_reference_pca_run_in_one_call(eng, tmp_path):
    """place_rec_pca.py:320-411 as one call: walk a reference split in the on-disk layout, describe without PCA, keep
    int(S_img * ratio) random rows per image up to max_segments, fit on the device, pickle a sklearn PCA -- which
    recall_segloc's loader (apply_pca_transform_from_pkl) then applies.  Checked against an fp64 fit of exactly the
    sampled rows (the sampling is seeded here; the reference's is not)."""
    import pickle

    import torch

    from revisit_anything_amd import func_vpr, pca_fit, store as st
    from revisit_anything_amd.pipeline import SegVLADPipeline

    K, D, H, W = 8, 64, 112, 140
    N = (H // 14) * (W // 14)
    C = synth().make_vocab(K, D, seed=41)
    droot, mroot = str(tmp_path / "dino"), str(tmp_path / "masks")
    keys, n_seg = [], 0
    for i in range(30):
        key = f"img_{i}.jpg"
        S = 5 + i % 4
        st.write_dino(droot, key, synth().make_tokens(C, N, seed=7000 + i, noise=0.3).reshape(1, D, H // 14, W // 14))
        st.write_masks(mroot, key, synth().make_masks(S, H // 2, W // 2, seed=8000 + i, hmin=6, hmax=30, wmin=6, wmax=40)[:S])
        keys.append(key)
        n_seg += S
    eng.set_vocab(C)
    pipe = SegVLADPipeline(eng, H, W, 14, order=2, use_pca=False)
    path = str(tmp_path / "pca.pkl")
    max_seg = 120                                                # < 192 segments in the split: ratio 0.62, stops early
    mean, comps, var = pca_fit.fit_from_store(st.FeatureStore(droot, "dino"), st.FeatureStore(mroot, "masks"), keys, pipe, out_pkl=path,
                                              n_components=12, max_segments=max_seg, batch_size=7, n_iter=6, seed=2,
                                              rng=np.random.default_rng(5))
    # the same sampling rule on the host (sample_segments) over the same descriptors and the same random stream
    blocks = []
    for i, key in enumerate(keys):
        t = np.asarray(st.FeatureStore(droot, "dino")[key]["ift_dino"][()]).reshape(D, N)
        m = np.stack(func_vpr.preload_masks(st.FeatureStore(mroot, "masks"), key)).astype(np.uint8)
        blocks.append(pipe.describe(torch.from_numpy(t[None]).to(eng.device), torch.from_numpy(m).to(eng.device),
                                    np.array([0, len(m)], np.int32), l2norm=False).cpu().numpy())
    Xs = pca_fit.sample_segments(blocks, n_seg, max_segments=max_seg, rng=np.random.default_rng(5))
    ratio = max_seg / n_seg
    assert Xs.shape[0] == sum(int((5 + i % 4) * ratio) for i in range(30)) < max_seg     # int() truncates per image (:384)
    mean_h, comps_h, var_h = pca_fit.fit_pca(Xs, n_components=12, n_iter=6, seed=2)
    assert np.abs(mean - mean_h).max() < 1e-6 and np.allclose(var, var_h, rtol=1e-3)
    assert (1 - np.linalg.svd(comps.astype(np.float64) @ comps_h.astype(np.float64).T, compute_uv=False)).max() < 1e-5
    model = pickle.load(open(path, "rb"))
    assert type(model).__name__ == "PCA" and model.whiten and model.n_components_ == 12 and model.n_features_in_ == K * D
    y_dev = func_vpr.apply_pca_transform_from_pkl(torch.from_numpy(Xs[:20]), path).numpy()
    assert np.abs(y_dev - model.transform(Xs[:20].astype(np.float64))).max() < 5e-5 * np.abs(y_dev).max()   # one model, two evaluations
    with pytest.raises(ValueError):
        pca_fit.fit_from_store(None, None, [], SegVLADPipeline(eng, H, W, 14, order=2, use_pca=True))


def test_driver_run_segloc_over_a_feature_store_equals_oracle(eng, tmp_path):
    from revisit_anything_amd import driver, store as st
    from revisit_anything_amd.pipeline import SegVLADPipeline

    K, D, H, W = 8, 64, 112, 140
    N = (H // 14) * (W // 14)
    C = synth().make_vocab(K, D, seed=41)
    rng = np.random.Generator(np.random.PCG64(42))
    n_ref, n_q = 12, 6

    def make(split, n, noise_seed):
        droot, mroot = str(tmp_path / f"{split}_dino"), str(tmp_path / f"{split}_masks")
        keys, toks, masks = [], [], []
        for i in range(n):
            key = f"img_{i}.jpg"
            base = i if split == "ref" else int(tau[i])
            t = synth().make_tokens(C, N, seed=5000 + base, noise=0.3)
            if split == "q":                                   # a query = its reference image, perturbed
                t = t + 0.05 * np.random.Generator(np.random.PCG64(noise_seed + i)).standard_normal(t.shape).astype(np.float32)
            S = int(rng.integers(4, 8))
            m = synth().make_masks(S, H // 2, W // 2, seed=6000 + base, hmin=6, hmax=30, wmin=6, wmax=40)[:S]
            st.write_dino(droot, key, t.reshape(1, D, H // 14, W // 14))
            st.write_masks(mroot, key, m)
            keys.append(key), toks.append(t), masks.append(m)
        return st.FeatureStore(droot, "dino"), st.FeatureStore(mroot, "masks"), keys, toks, masks

    tau = rng.permutation(n_ref)[:n_q]
    dr, mr, kr, tr, msr = make("ref", n_ref, 0)
    dq, mq, kq, tq, msq = make("q", n_q, 777)
    gt = [[int(t)] for t in tau]
    eng.set_vocab(C)
    pipe = SegVLADPipeline(eng, H, W, 14, order=2, use_pca=False)
    from revisit_anything_amd.place_rec import default_experiment
    save = {"workdir": str(tmp_path / "work"), "dataset_name": "17places", "experiment_name": "exp1_test",
            "experiment_config": default_experiment(order=2, pca=False), "domain": "indoor"}
    rec, pred, matches, sims = driver.run_segloc(dr, mr, kr, dq, mq, kq, gt, pipe, batch_size=5, n_top=3, k_search=20, k_vote=10,
                                                 save_results=save)
    # the three pickles of the reference's --save_results run, under ITS file names (place_rec_main.py:62-75, 292-305, 357-370)
    import pickle
    folder = tmp_path / "work" / "results" / "global" / "exp1_test"
    names = sorted(f.name for f in folder.iterdir())
    assert names == [f"17places_{kind}_domain_indoor___results_SegLoc_VLAD_o2.pkl" for kind in ("matches_sims", "segFtVLAD1", "segFtVLAD2")]
    with open(folder / names[0], "rb") as f:
        ms = pickle.load(f)
    with open(folder / names[1], "rb") as f:
        ft1 = pickle.load(f)
    with open(folder / names[2], "rb") as f:
        ft2 = pickle.load(f)
    import torch
    assert isinstance(ft1, torch.Tensor) and isinstance(ft2, torch.Tensor) and ft1.device.type == "cpu"
    assert set(ms) == {"sims", "matches"} and ms["sims"].shape == (ft2.shape[0], 20) and ms["matches"].dtype == np.int64
    assert np.array_equal(ms["matches"][:, :10], matches.cpu().numpy())                    # the vote read their first 10 columns
    assert np.array_equal((2 - ms["sims"][:, :10]).astype(np.float32), sims.cpu().numpy())  # 'sims' holds d^2, as in the reference

    # oracle: the same chain in fp64 (seg-VLAD per image with order-2 neighbourhoods -> exact kNN -> vote -> recall)
    def odesc(toks, masks):
        out, im = [], []
        for i, (t, m) in enumerate(zip(toks, masks)):
            adj = O().nbr_masks_agg_fast_single([x for x in m], 2)
            out.append(O().seg_vlad_from_masks(t, m, C, H, W, adj))
            im += [i] * len(m)
        return np.concatenate(out), np.array(im, np.int64)

    R, imr = odesc(tr, msr)
    Q, imq = odesc(tq, msq)
    assert np.abs(ft1.numpy() - R).max() < 2e-6 and np.abs(ft2.numpy() - Q).max() < 2e-6   # the pickled rows ARE the descriptors
    d2, idx = O().knn_l2(R.astype(np.float32), Q.astype(np.float32), 20)
    osims = (2 - d2[:, :10]).astype(np.float32)
    seg_range = [np.where(imq == i)[0] for i in range(n_q)]
    opred = O().get_matches_wt_borda_im(idx[:, :10], n_q, osims, seg_range, imr, n=3)
    assert np.abs(sims.cpu().numpy() - osims).max() < 1e-4
    assert np.array_equal(matches.cpu().numpy()[:, 0], idx[:, 0])
    assert [int(p[0]) for p in pred] == [int(p[0]) for p in opred]
    assert np.allclose(rec, O().calc_recall([list(p) for p in opred], gt, 3))
    assert rec[0] >= 0.8                                        # the perturbed queries find their reference image
