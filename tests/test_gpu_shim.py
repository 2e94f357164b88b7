"""The drop-in surface ON THE DEVICE: every callable a maintainer would import from
revisit_anything_amd.func_vpr / revisit_anything_amd.place_rec in place of the reference's func_vpr.py /
place_rec_main.py is executed on the GPU with the reference's own argument conventions and compared with the
reference-generated fixtures (tests/golden, produced by tools/make_golden.py from the reference's function bodies).
Every name of tests/golden/signatures.json is called here."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fv():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a ROCm device (no CPU fallback exists)"
    from revisit_anything_amd import func_vpr

    return func_vpr


def O():
    from oracle import segvlad_oracle

    return segvlad_oracle


def synth():
    from revisit_anything_amd import synth as s

    return s


def _tiny():
    z = np.load(os.path.join(G, "vlad_tiny.npz"))
    D, K, H, W, S = (int(z[k]) for k in "DKHWS")
    C = synth().make_vocab(K, D, seed=1001)
    tok = synth().make_tokens(C, (H // 14) * (W // 14), seed=2001, noise=0.3)
    return z, D, K, H, W, S, C, tok


class _H5Like(dict):
    """desc_path_in[img_key]['ift_dino'][()] -> ndarray, as an h5py file would give (func_vpr.py:1082)."""


# ------------------------------------------------------------------------------------------------
# a4  nbrMasksAGGFastSingle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [1, 2, 3, 4, 6, 12, 50])
def test_nbrMasksAGGFastSingle_golden(fv, S):
    import torch

    z = np.load(os.path.join(G, "adjacency_cases.npz"))
    m = np.unpackbits(z[f"S{S}_masks"], axis=1)[:, :60 * 80].reshape(S, 60, 80).astype(bool)
    for order in (1, 2, 3):
        a = fv.nbrMasksAGGFastSingle([x for x in m], order)
        assert isinstance(a, torch.Tensor) and a.dtype == torch.bool and tuple(a.shape) == (S, S)
        assert np.array_equal(a.numpy(), z[f"S{S}_o{order}"])
    if S == 4:
        bad = [x.copy() for x in m]
        bad[2][:] = False
        with pytest.raises(ValueError):      # empty mask: the reference fails in np.mean / Delaunay
            fv.nbrMasksAGGFastSingle(bad, 1)


# ------------------------------------------------------------------------------------------------
# a5-a7  seg_vlad_gpu_single(_img), vlad_single, vlad_matmuls_per_cluster
# ------------------------------------------------------------------------------------------------
def test_seg_vlad_gpu_single_img_and_h5_twin_golden(fv):
    import torch

    z, D, K, H, W, S, C, tok = _tiny()
    dh, dw = H // 14, W // 14
    dino = torch.from_numpy(tok.reshape(1, D, dh, dw).copy())
    cfg = {"desired_height": H, "desired_width": W, "rmin": 0}
    cc = torch.from_numpy(C)
    segMask = [m for m in z["masks"]]
    h5 = _H5Like({"img7": {"ift_dino": tok.reshape(1, D, dh, dw).copy()}})
    for order in (0, 1, 3):
        adj = None if order == 0 else torch.from_numpy(z[f"adj_o{order}"])
        out = fv.seg_vlad_gpu_single_img(None, None, dino, "img7", segMask, cc, cfg, desc_dim=D, adj_mat=adj)
        # return convention of the reference: float64 CPU tensor [S, K*D] (func_vpr.py:1100,1172)
        assert isinstance(out, torch.Tensor) and out.dtype == torch.float64 and out.device.type == "cpu"
        assert tuple(out.shape) == (S, K * D)
        ref = z[f"vlad_o{order}"]
        assert np.abs(out.numpy() - ref).max() < 2e-6            # fp32 device vs the reference's fp64, unit rows
        out2 = fv.seg_vlad_gpu_single(None, None, h5, "img7", segMask, cc, cfg, desc_dim=D, adj_mat=adj)
        assert torch.equal(out, out2)                             # the H5 twin: identical body after the read
    # no segments: an empty descriptor block, not an error
    e = fv.seg_vlad_gpu_single_img(None, None, dino, "img7", [], cc, cfg, desc_dim=D)
    assert tuple(e.shape) == (0, K * D)


def test_vocabulary_cache_never_aliases(fv):
    """a second vocabulary of the same shape must be uploaded even if it reuses the first one's storage
    address; an in-place update must be seen too."""
    import torch

    z, D, K, H, W, S, C, tok = _tiny()
    dino = torch.from_numpy(tok.reshape(1, D, H // 14, W // 14).copy())
    cfg = {"desired_height": H, "desired_width": W}
    segMask = [m for m in z["masks"]]
    c1 = torch.from_numpy(C.copy())
    a = fv.seg_vlad_gpu_single_img(None, None, dino, "k", segMask, c1, cfg, desc_dim=D)
    C2 = synth().make_vocab(K, D, seed=4242)
    del c1
    c2 = torch.from_numpy(C2.copy())                      # may land on c1's freed address
    b = fv.seg_vlad_gpu_single_img(None, None, dino, "k", segMask, c2, cfg, desc_dim=D)
    inc = O().incidence(z["masks"], H, W)
    assert np.abs(b.numpy() - O().seg_vlad(tok, inc, C2, None)).max() < 2e-6
    assert np.abs(a.numpy() - O().seg_vlad(tok, inc, C, None)).max() < 2e-6
    c2.copy_(torch.from_numpy(C))                         # in-place change of the SAME tensor object
    c = fv.seg_vlad_gpu_single_img(None, None, dino, "k", segMask, c2, cfg, desc_dim=D)
    assert np.abs(c.numpy() - a.numpy()).max() < 1e-12


def test_vlad_single_golden(fv):
    import torch

    z, D, K, H, W, S, C, tok = _tiny()
    xn = torch.from_numpy(O().normalize_tokens_f32(tok))            # [N, D] L2-normalised, as vlad_single receives them
    masks = torch.from_numpy(z["inc"])                              # bool [S, N]
    for order in (0, 3):
        adj = None if order == 0 else torch.from_numpy(z[f"adj_o{order}"])
        out, secs = fv.vlad_single(xn, torch.from_numpy(C), None, masks, adj_mat=adj)
        assert out.dtype == torch.float64 and isinstance(secs, float) and secs >= 0
        assert np.abs(out.cpu().numpy() - z[f"vlad_o{order}"]).max() < 2e-6


def test_vlad_matmuls_per_cluster_golden(fv):
    """The K-parametric entry (segvlad_cluster_aggregate) with the reference's argument types: masks / adjacency as
    float64 0/1 matrices, residuals float64, labels int64 (func_vpr.py:1173-1176 casts)."""
    import torch

    z = np.load(os.path.join(G, "vlad_k64.npz"))
    D, K, S, N = (int(z[k]) for k in ("D", "K", "S", "N"))
    C = synth().make_vocab(K, D, seed=1003)
    tok = synth().make_tokens(C, N, seed=2004, noise=0.2)
    xn = O().normalize_tokens_f32(tok)
    lab = z["labels"]
    res = torch.from_numpy((xn - C[lab]).astype(np.float64))
    out, secs = fv.vlad_matmuls_per_cluster(K, torch.from_numpy(z["inc"]).double(), res, torch.from_numpy(lab),
                                            adjMat=torch.from_numpy(z["adj"]).double())
    assert out.dtype == torch.float64 and tuple(out.shape) == (S, K * D) and isinstance(secs, float)
    assert np.abs(out.cpu().numpy() - z["vlad"]).max() < 2e-6
    out0, _ = fv.vlad_matmuls_per_cluster(K, torch.from_numpy(z["inc"]).double(), res, torch.from_numpy(lab))
    assert np.abs(out0.cpu().numpy() - O().vlad_matmuls_per_cluster(K, z["inc"], xn - C[lab], lab, None)).max() < 2e-6
    assert np.all(out0.cpu().numpy()[3] == 0)            # a token-less segment stays all-zero without the union


# ------------------------------------------------------------------------------------------------
# a8/a9  apply_pca_transform_from_pkl(_numpy), normalizeFeat
# ------------------------------------------------------------------------------------------------
def _sk_pca(mean, comps, var, whiten=True):
    from sklearn.decomposition import PCA

    p = PCA(n_components=comps.shape[0], whiten=whiten)
    p.mean_, p.components_, p.explained_variance_ = mean, comps, var
    return p


def test_apply_pca_transform_from_pkl_golden_and_cache(fv, tmp_path):
    import torch

    z = np.load(os.path.join(G, "pca_small.npz"))
    path = str(tmp_path / "fitted_pca_model.pkl")
    with open(path, "wb") as f:
        pickle.dump(_sk_pca(z["mean"], z["components"], z["explained_variance"]), f)
    y = fv.apply_pca_transform_from_pkl(torch.from_numpy(z["X"]), path)
    assert isinstance(y, torch.Tensor) and y.device.type == "cpu" and y.dtype == torch.float64
    assert np.abs(y.numpy() - z["Y"]).max() < 5e-5 * np.abs(z["Y"]).max()
    yn = fv.apply_pca_transform_from_pkl_numpy(z["X"], path)
    assert isinstance(yn, np.ndarray) and np.array_equal(yn, y.numpy())
    # somebody else re-programs the engine's PCA -> the next call must reload the pickle
    mean2, comps2, var2 = synth().make_pca_model(64, 8, seed=77)
    fv.engine().pca_set(mean2, comps2, var2, whiten=True)
    y_again = fv.apply_pca_transform_from_pkl(torch.from_numpy(z["X"]), path)
    assert torch.equal(y_again, y)
    # ... and a re-fitted model saved under the SAME name (the place_rec_pca flow) must be picked up
    with open(path, "wb") as f:
        pickle.dump(_sk_pca(mean2.astype(np.float64), comps2.astype(np.float64), var2.astype(np.float64)), f)
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 1_000_000))
    y2 = fv.apply_pca_transform_from_pkl(torch.from_numpy(z["X"]), path)
    ref2 = O().pca_transform(z["X"].astype(np.float32), mean2, comps2, var2, True)
    assert np.abs(y2.numpy() - ref2).max() < 5e-5 * np.abs(ref2).max()
    # un-whitened model
    with open(path, "wb") as f:
        pickle.dump(_sk_pca(z["mean"], z["components"], z["explained_variance"], whiten=False), f)
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 2_000_000))
    y3 = fv.apply_pca_transform_from_pkl(torch.from_numpy(z["X"]), path)
    ref3 = O().pca_transform(z["X"].astype(np.float32), z["mean"], z["components"], None, False)
    assert np.abs(y3.numpy() - ref3).max() < 5e-5 * np.abs(ref3).max()


def test_normalizeFeat_is_a_copy(fv):
    rng = np.random.Generator(np.random.PCG64(9))
    x = rng.standard_normal((37, 1024))
    keep = x.copy()
    y = fv.normalizeFeat(x)
    assert np.array_equal(x, keep) and y.shape == x.shape and y.dtype == np.float64
    assert np.abs(y - O().normalize_feat(x)).max() < 1e-6
    y3 = fv.normalizeFeat(x.reshape(37, 32, 32))          # reshape([n, -1]) of the reference
    assert y3.shape == (37, 1024) and np.abs(y3 - y).max() == 0


# ------------------------------------------------------------------------------------------------
# a12  get_matches (both device methods, non-contiguous segRangeQuery), weighted_borda_count
# ------------------------------------------------------------------------------------------------
def test_get_matches_golden_contiguous_and_permuted(fv):
    z = np.load(os.path.join(G, "vote_cases.npz"))
    off = z["off"]
    n_q = len(off) - 1
    segRange = [np.arange(off[i], off[i + 1]) for i in range(n_q)]
    gt = [[0]] * n_q
    for n in (1, 5):
        p = fv.get_matches(z["matches"], gt, z["sims"], segRange, z["imInds"], n=n, method="max_seg_topk_wt_borda_Im")
        assert len(p) == n_q and all(isinstance(x, list) for x in p)
        assert all(isinstance(v, np.int64) for x in p for v in x)
        got = np.array([list(x) + [-1] * (n - len(x)) for x in p], dtype=np.int64)
        assert np.array_equal(got, z[f"wt_n{n}"])
    p = fv.get_matches(z["matches"], gt, z["sims"], segRange, z["imInds"], n=5, method="max_seg_topk")
    ref, counts = O().get_matches_max_seg_topk(z["matches"], n_q, segRange, z["imInds"], n=5)
    for i in range(n_q):          # counts exact; order = (count desc, image id asc) where the reference's argsort is unique
        bc = counts[i]
        assert sorted(bc[np.asarray(p[i])].tolist(), reverse=True) == np.sort(bc)[::-1][:len(p[i])].tolist()
    # rows of the query images interleaved (a non-contiguous segRangeQuery): same predictions
    perm = np.random.Generator(np.random.PCG64(3)).permutation(int(off[-1]))
    inv = np.argsort(perm)
    m2, s2 = z["matches"][perm], z["sims"][perm]
    # the reference walks each query's rows in segRangeQuery order; keep each image's row ORDER (first appearance
    # decides ties): rows of image i, in their original order, now live at positions inv[off[i]..off[i+1])
    segRange2 = [inv[off[i]:off[i + 1]] for i in range(n_q)]
    p2 = fv.get_matches(m2, gt, s2, segRange2, z["imInds"], n=5, method="max_seg_topk_wt_borda_Im")
    got2 = np.array([list(x) + [-1] * (5 - len(x)) for x in p2], dtype=np.int64)
    assert np.array_equal(got2, z["wt_n5"])
    # hand-made tie case: equal weights -> first appearance (rank-major, then segment)
    p = fv.get_matches(z["tie_matches"], [[0]], z["tie_sims"], [np.arange(2)], z["tie_imInds"], n=4,
                       method="max_seg_topk_wt_borda_Im")
    assert [int(v) for v in p[0]] == z["tie_pred"].tolist()
    with pytest.raises(NotImplementedError):   # a branch that calls an undefined helper in the reference (func_vpr.py:128)
        fv.get_matches(z["matches"], gt, z["sims"], segRange, z["imInds"], n=1, method="max_seg_topk_borda")


def test_weighted_borda_count_host_helper(fv):
    out = fv.weighted_borda_count([(3, 1.0), (1, 0.5)], [(1, 0.75), (2, 0.25)])
    assert out == [1, 3, 2]


# ------------------------------------------------------------------------------------------------
# recall_segloc + IndexFlatL2 (place_rec_main.py:44-96)
# ------------------------------------------------------------------------------------------------
def _e2e_inputs():
    z = np.load(os.path.join(G, "e2e_small.npz"))
    n_img, S, d, n_q = (int(z[k]) for k in ("n_img", "S", "d", "n_q"))
    R, img = synth().make_planted_db(n_img, S, d, seed=3000)
    Q, tau, off = synth().make_planted_queries(R, n_img, S, n_q, seed=4000, sigma_q=3.0)
    sr = np.random.Generator(np.random.PCG64(1)).uniform(0.5, 2.0, size=(R.shape[0], 1))
    sq = np.random.Generator(np.random.PCG64(2)).uniform(0.5, 2.0, size=(Q.shape[0], 1))
    gt = [[int(t)] for t in tau]
    gt[5] = []
    segRange2 = [np.arange(off[i], off[i + 1]) for i in range(n_q)]
    return z, R.astype(np.float64) * sr, Q.astype(np.float64) * sq, gt, segRange2, img.astype(np.int64)


def test_recall_segloc_golden(fv, tmp_path, capsys):
    import torch

    from revisit_anything_amd import place_rec

    z, R, Q, gt, segRange2, imInds1 = _e2e_inputs()
    cfg = {"pca": True, "results_pkl_suffix": "_res.pkl"}
    rec = place_rec.recall_segloc(str(tmp_path), "synthetic", cfg, "e2e", torch.from_numpy(R), torch.from_numpy(Q), gt,
                                  segRange2, imInds1, False, "indoor", save_results=True)
    assert np.allclose(rec, z["recalls"])                                    # Recall@1..5 of the reference
    out = capsys.readouterr().out
    assert "POSITIVES/TOTAL segVLAD for this dataset" in out and "Max Seg Logs" in out
    pkl = tmp_path / "results" / "global" / "e2e" / "synthetic_matches_sims_domain_indoor___res.pkl"
    saved = pickle.load(open(pkl, "rb"))                                    # the results pickle of place_rec_main.py:70-75
    assert saved["sims"].shape == (Q.shape[0], 200) and saved["matches"].shape == (Q.shape[0], 200)
    assert saved["sims"].dtype == np.float32 and saved["matches"].dtype == np.int64
    assert np.abs((2 - saved["sims"][:, :50]) - z["sims_50"]).max() < 1e-4   # north_star: cosine scores within 1e-4
    # every id mismatch against the fixture is a near-tie: the row we rank there is as close as the fixture's
    Rn = O().normalize_feat(R).astype(np.float32).astype(np.float64)
    Qn = O().normalize_feat(Q).astype(np.float32).astype(np.float64)
    qq, rr = np.nonzero(saved["matches"][:, :50] != z["matches_50"])
    assert len(qq) < 0.01 * z["matches_50"].size
    if len(qq):
        mine = ((Qn[qq] - Rn[saved["matches"][qq, rr]]) ** 2).sum(1)
        theirs = ((Qn[qq] - Rn[z["matches_50"][qq, rr]]) ** 2).sum(1)
        assert np.abs(mine - theirs).max() < 1e-5
    # pca=False branch: raw rows, no normalisation
    rec2 = place_rec.recall_segloc(str(tmp_path), "synthetic", {"pca": False, "results_pkl_suffix": "_raw.pkl"}, "e2e",
                                   torch.from_numpy(O().normalize_feat(R)), torch.from_numpy(O().normalize_feat(Q)), gt,
                                   segRange2, imInds1, True, "indoor", save_results=False)
    assert np.allclose(rec2, z["recalls"])


def test_two_IndexFlatL2_alive_at_once(fv):
    """a second index must not wipe the first (each owns its context)."""
    from revisit_anything_amd import place_rec

    rng = np.random.Generator(np.random.PCG64(5))
    A = rng.standard_normal((500, 64)).astype(np.float32)
    B = rng.standard_normal((300, 32)).astype(np.float32)
    ia = place_rec.IndexFlatL2(64)
    ia.add(A[:200])
    ib = place_rec.IndexFlatL2(32)
    ib.add(B)
    ia.add(A[200:])
    assert ia.ntotal == 500 and ib.ntotal == 300
    Da, Ia = ia.search(A[:7], 4)
    Db, Ib = ib.search(B[:7], 4)
    assert Da.dtype == np.float32 and Ia.dtype == np.int64 and Da.shape == (7, 4)
    ra, rb = O().knn_l2(A, A[:7], 4), O().knn_l2(B, B[:7], 4)
    assert np.array_equal(Ia, ra[1]) and np.abs(Da - ra[0]).max() < 1e-4
    assert np.array_equal(Ib, rb[1]) and np.abs(Db - rb[0]).max() < 1e-4
    with pytest.raises(ValueError):
        ia.add(B)


# ------------------------------------------------------------------------------------------------
# host-only helpers of the surface, executed in the same (GPU) process for completeness
# ------------------------------------------------------------------------------------------------
def test_host_helpers_preload_getIdx_calc_recall(fv, capsys):
    masks = {"im0": {"masks": {str(j): {"segmentation": np.full((2, 2), j)} for j in (10, 2, 1)}}}
    got = fv.preload_masks(masks, "im0")
    assert [int(m[0, 0]) for m in got] == [1, 2, 10]                         # natural key order
    imInds, regInds, seg = fv.getIdxSingleFast(7, got, minArea=10 ** 9)       # minArea ignored, like the reference
    assert imInds.tolist() == [7, 7, 7] and regInds == [0, 1, 2] and len(seg) == 3
    z = np.load(os.path.join(G, "recall_cases.npz"))
    gts = [[int(v) for v in row if v >= 0] for row in z["gt"]]
    rec = fv.calc_recall([list(p) for p in z["preds"]], gts, 5)
    assert np.allclose(rec, z["recalls"])
    assert "POSITIVES/TOTAL segVLAD for this dataset" in capsys.readouterr().out


# ------------------------------------------------------------------------------------------------
# f5  AnyLoc global-VLAD baseline: aggFt(..., 'vlad', vlad) + get_recall (place_rec_main.py:379-389)
# ------------------------------------------------------------------------------------------------
def test_anyloc_aggFt_and_get_recall_golden(fv, tmp_path, capsys):
    import types

    import torch

    from revisit_anything_amd import store as st

    z = np.load(os.path.join(G, "anyloc_cases.npz"))
    voc = np.load(os.path.join(G, "vocab_indoor_k32_d1536.npy"))
    toks = [synth().make_tokens(voc, 34 * 45, seed=s, noise=0.2) for s in (2010, 2011)]
    h5 = {f"img_{j}.jpg": {"ift_dino": t.reshape(1, 1536, 34, 45)} for j, t in enumerate(toks)}
    vlad = types.SimpleNamespace(c_centers=torch.from_numpy(voc), num_clusters=32, desc_dim=1536)      # the reference's VLAD object
    out = fv.aggFt(h5, None, None, {"desired_height": 480, "desired_width": 640}, "vlad", vlad, upsample=True)
    assert isinstance(out, list) and len(out) == 2 and out[0].shape == (32 * 1536,) and out[0].dtype == np.float32
    Gm = np.random.Generator(np.random.PCG64(780)).standard_normal((32 * 1536, 8))
    for j in range(2):
        assert np.abs(out[j][::37] - z[f"vlad{j}_sub"]).max() < 1e-6
        assert np.abs(out[j].astype(np.float64) @ Gm - z[f"vlad{j}_proj"]).max() < 2e-4
        assert np.abs(out[j] - O().global_vlad(toks[j], voc)).max() < 1e-6
    # a store directory works like the h5 file, keys in natural order
    for j, t in enumerate(toks):
        st.write_dino(str(tmp_path / "d"), f"img_{j}.jpg", t.reshape(1, 1536, 34, 45))
    out2 = fv.aggFt(str(tmp_path / "d"), None, None, {}, "vlad", torch.from_numpy(voc))
    assert all(np.array_equal(a, b) for a, b in zip(out, out2))
    with pytest.raises(NotImplementedError):
        fv.aggFt(h5, None, None, {}, "avg", vlad)
    gt = [[int(g)] if g >= 0 else [] for g in z["gt"]]
    rec, matches = fv.get_recall(fv.normalizeFeat(z["db"]), fv.normalizeFeat(z["q"]), gt, k=5)
    assert np.allclose(rec, z["recall"])
    assert np.array_equal(np.stack([m["img_id_r"] for m in matches]), z["ids"])
    assert "POSITIVES/TOTAL AnyLoc for this dataset" in capsys.readouterr().out
    rec2, per_q, _ = fv.get_recall(z["db"], z["q"], gt, analysis=True, k=5)
    assert np.allclose(rec2, z["recall"]) and sum(per_q) == round(z["recall"][-1] / 100 * sum(1 for g in gt if g))
