"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the committed golden
vectors.  Bit-exact for integer/bit/index outputs; fp32-vs-fp64 tolerances are written at each
assert.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest
from conftest import engine_scope

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


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


def cos_rows(a, b):
    num = (a * b).sum(1)
    den = np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1)
    out = np.ones_like(num)
    nz = den > 0
    out[nz] = num[nz] / den[nz]
    return out


def cat_adj(adjs):
    return np.concatenate([np.asarray(a, dtype=np.uint8).reshape(-1) for a in adjs]) if adjs else None


# ------------------------------------------------------------------------------------------------
def test_native_library_is_loaded(eng):
    from revisit_anything_amd import _lib

    assert os.path.exists(_lib.lib_path())
    with open("/proc/self/maps") as f:
        assert "libsegvlad_hip.so" in f.read()


@pytest.mark.parametrize("name", ["same", "x2", "x2clip", "nonint", "down"])
def test_incidence_golden(eng, name):
    z = np.load(os.path.join(G, "incidence_cases.npz"))
    S, Hm, Wm, H, W = (int(v) for v in z[f"{name}_shape"])
    m = np.unpackbits(z[f"{name}_masks"], axis=1)[:, :Hm * Wm].reshape(S, Hm, Wm).astype(np.uint8)
    bits = eng.incidence(m, H, W).cpu().numpy().view(np.uint64)
    N = (H // 14) * (W // 14)
    assert np.array_equal(O().unpack_bits_u64(bits, N), z[f"{name}_inc"])  # bit-exact


def test_incidence_ref_geometry_and_device_input(eng):
    import torch

    masks = synth().make_blob_masks(50, 240, 320, seed=11)
    want = O().incidence(masks, 480, 640)
    md = torch.from_numpy(masks.astype(np.uint8)).to(eng.device)
    bits = eng.incidence(md, 480, 640).cpu().numpy().view(np.uint64)
    assert np.array_equal(O().unpack_bits_u64(bits, 34 * 45), want)
    # pad bits above N must be zero
    assert (bits[:, -1] >> np.uint64(1530 - 23 * 64)).max() == 0


def test_mask_centroids_bit_exact(eng):
    masks = synth().make_blob_masks(20, 60, 80, seed=3)
    c = eng.mask_centroids(masks).cpu().numpy()
    assert np.array_equal(c, O().mask_centroids([m for m in masks]))  # exact integer sums, one fp64 division


# ------------------------------------------------------------------------------------------------
def run_vlad(eng, tokens_list, inc_list, adj_list, C, **kw):
    eng.set_vocab(C)
    B = len(tokens_list)
    toks = np.stack(tokens_list)
    N = toks.shape[2]
    offs = np.concatenate([[0], np.cumsum([i.shape[0] for i in inc_list])]).astype(np.int32)
    bits = np.concatenate([O().pack_bits_u64(i) for i in inc_list]) if offs[-1] else np.zeros((0, (N + 63) // 64), np.uint64)
    adj = None
    if adj_list is not None:
        adj = cat_adj([a if a is not None else np.eye(i.shape[0], dtype=bool) for a, i in zip(adj_list, inc_list)])
    r = eng.seg_vlad(toks, bits.view(np.int64), offs, adj, **kw)
    return {k: v.cpu().numpy() for k, v in r.items()}, offs


def test_vlad_tiny_golden(eng):
    z = np.load(os.path.join(G, "vlad_tiny.npz"))
    D, K, H, W, S = (int(z[k]) for k in "DKHWS")
    C = synth().make_vocab(K, D, seed=1001)
    tok = synth().make_tokens(C, (H // 14) * (W // 14), seed=2001, noise=0.3)
    inc = z["inc"]
    for order in (0, 1, 3):
        adj = None if order == 0 else [z[f"adj_o{order}"]]
        r, _ = run_vlad(eng, [tok], [inc], adj, C, want_labels=True, want_gap=True)
        assert np.array_equal(r["labels"][0], z["labels"].astype(np.uint8))
        ref = z[f"vlad_o{order}"]
        assert np.abs(r["out"] - ref).max() < 2e-6          # fp32 device vs fp64 reference, unit-norm rows
        assert (1 - cos_rows(r["out"].astype(np.float64), ref)).max() < 1e-6


def test_vlad_ref_shape_golden(eng):
    z = np.load(os.path.join(G, "vlad_ref_shape.npz"))
    voc = np.load(os.path.join(G, "vocab_indoor_k32_d1536.npy"))
    tok = synth().make_tokens(voc, 34 * 45, seed=2002)
    inc = np.unpackbits(z["inc"], axis=1)[:, :1530].astype(bool)
    r, _ = run_vlad(eng, [tok], [inc], [z["adj"]], voc, want_labels=True, want_block_norms=True)
    assert np.array_equal(r["labels"][0], z["labels"])
    out = r["out"].astype(np.float64)
    assert np.abs(out[:, ::61] - z["sub"]).max() < 1e-6
    assert np.abs(out[:, :256] - z["head"]).max() < 1e-6
    assert np.abs(out[:, -256:] - z["tail"]).max() < 1e-6
    Gm = np.random.Generator(np.random.PCG64(777)).standard_normal((32 * 1536, 16))
    assert np.abs(out @ Gm - z["proj"]).max() < 1e-4
    # full-tensor check against the oracle (cosine 1-1e-6 per SURVEY App. D 1)
    ref, aux = O().seg_vlad(tok, inc, voc, z["adj"], return_aux=True)
    assert (1 - cos_rows(out, ref)).max() < 1e-6
    assert np.abs(out - ref).max() < 1e-6
    assert np.allclose(r["block_norms"], aux["block_norms"], rtol=1e-5, atol=1e-6)


def test_vlad_k64_golden(eng):
    z = np.load(os.path.join(G, "vlad_k64.npz"))
    D, K, S, N = (int(z[k]) for k in ("D", "K", "S", "N"))
    C = synth().make_vocab(K, D, seed=1003)
    tok = synth().make_tokens(C, N, seed=2004, noise=0.2)
    r, _ = run_vlad(eng, [tok], [z["inc"]], [z["adj"]], C, want_labels=True)  # S=66 > 64: two segment chunks
    assert np.array_equal(r["labels"][0], z["labels"].astype(np.uint8))
    assert np.abs(r["out"] - z["vlad"]).max() < 2e-6


def test_vlad_adversarial_tie_audit(eng):
    """Isotropic tokens: near-tied assignments.  Labels must equal the oracle's wherever the oracle's
    fp64 top-2 gap exceeds fp32 rounding; the device's own gap output must flag the rest."""
    voc = np.load(os.path.join(G, "vocab_indoor_k32_d1536.npy"))
    tok = synth().make_tokens(voc, 34 * 45, seed=2003, adversarial=True)
    masks = synth().make_masks(50, 240, 320, seed=2102)
    inc = O().incidence(masks, 480, 640)
    r, _ = run_vlad(eng, [tok], [inc], None, voc, want_labels=True, want_gap=True)
    labels, gap = O().assign_labels(O().normalize_tokens_f32(tok), voc)
    ok = gap > 1e-6
    assert np.array_equal(r["labels"][0][ok], labels[ok].astype(np.uint8))
    assert np.allclose(r["gap"][0][ok], gap[ok], atol=2e-6)
    if np.array_equal(r["labels"][0], labels.astype(np.uint8)):
        ref = O().seg_vlad(tok, inc, voc, None)
        assert (1 - cos_rows(r["out"].astype(np.float64), ref)).max() < 1e-6


def test_vlad_ragged_batch_and_edge_cases(eng):
    """Batch of images with different segment counts (0, 1, 3, 40, 70), empty segments, empty clusters."""
    D, K, N = 64, 32, 15 * 20
    C = synth().make_vocab(K, D, seed=77)
    rng = np.random.Generator(np.random.PCG64(78))
    toks, incs, adjs = [], [], []
    for b, S in enumerate([0, 1, 3, 40, 70]):
        toks.append(synth().make_tokens(C[:5] if b == 2 else C, N, seed=900 + b, noise=0.3))  # b==2: only 5 clusters used
        inc = rng.random((S, N)) < 0.2
        if S > 2:
            inc[1] = False                        # a segment covering no token
        incs.append(inc)
        a = (rng.random((S, S)) < 0.1) | np.eye(S, dtype=bool)
        adjs.append(a)
    toks[2] = synth().make_tokens(C, N, seed=902, noise=0.3)
    r, offs = run_vlad(eng, toks, incs, adjs, C, want_labels=True)
    for b in range(5):
        ref, aux = O().seg_vlad(toks[b], incs[b], C, adjs[b], return_aux=True)
        got = r["out"][offs[b]:offs[b + 1]]
        assert np.array_equal(r["labels"][b], aux["labels"].astype(np.uint8))
        assert got.shape == ref.shape
        if ref.size:
            assert np.abs(got - ref).max() < 2e-6
    # identity adjacency == adj=None
    r2, _ = run_vlad(eng, toks, incs, None, C)
    r3, _ = run_vlad(eng, toks, incs, [np.eye(i.shape[0], dtype=bool) for i in incs], C)
    assert np.array_equal(r2["out"], r3["out"])
    # zero rows for token-less segments without a neighbourhood
    assert np.all(r2["out"][offs[3] + 1] == 0)


def test_vlad_odd_token_count_and_d768(eng):
    voc = np.load(os.path.join(G, "vocab_nv_k32_d768.npy"))
    N = 37 * 37  # odd: exercises the unaligned 8-byte token loads
    tok = synth().make_tokens(voc, N, seed=31, noise=0.1)
    rng = np.random.Generator(np.random.PCG64(32))
    inc = rng.random((9, N)) < 0.3
    r, _ = run_vlad(eng, [tok], [inc], None, voc, want_labels=True)
    ref, aux = O().seg_vlad(tok, inc, voc, None, return_aux=True)
    assert np.array_equal(r["labels"][0], aux["labels"].astype(np.uint8))
    assert np.abs(r["out"] - ref).max() < 1e-6


def test_vlad_deterministic(eng):
    voc = np.load(os.path.join(G, "vocab_indoor_k32_d1536.npy"))
    tok = synth().make_tokens(voc, 300, seed=5)
    inc = np.random.Generator(np.random.PCG64(6)).random((50, 300)) < 0.3
    a, _ = run_vlad(eng, [tok, tok], [inc, inc], None, voc)
    b, _ = run_vlad(eng, [tok, tok], [inc, inc], None, voc)
    assert np.array_equal(a["out"], b["out"])            # run twice, compare bits
    assert np.array_equal(a["out"][:50], a["out"][50:])  # batch position does not matter


# ------------------------------------------------------------------------------------------------
def test_pca_golden_and_normalize(eng):
    z = np.load(os.path.join(G, "pca_small.npz"))
    eng.pca_set(z["mean"], z["components"], z["explained_variance"], whiten=True)
    Y = eng.pca_apply(z["X"].astype(np.float32)).cpu().numpy()
    assert np.abs(Y - z["Y"]).max() < 5e-5 * np.abs(z["Y"]).max()
    Yn = eng.pca_apply(z["X"].astype(np.float32), l2norm=True).cpu().numpy()
    assert np.abs(Yn - O().normalize_feat(z["Y"])).max() < 1e-5


def test_pca_large_shape(eng):
    mean, comps, var = synth().make_pca_model(2048 * 3, 200, seed=5)   # P not a multiple of the tile
    X = np.random.Generator(np.random.PCG64(6)).standard_normal((333, 2048 * 3)).astype(np.float32) / 78.0
    eng.pca_set(mean, comps, var, whiten=True)
    Y = eng.pca_apply(X).cpu().numpy()
    ref = O().pca_transform(X, mean, comps, var, True)
    assert np.abs(Y - ref).max() < 2e-4 * np.abs(ref).max()


def test_knn_exact_against_oracle(eng):
    rng = np.random.Generator(np.random.PCG64(100))
    R = rng.standard_normal((3000, 256)).astype(np.float32)
    R /= np.linalg.norm(R, axis=1, keepdims=True)
    Q = rng.standard_normal((257, 256)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    eng.db_reset()
    eng.db_add(R[:1000])
    eng.db_add(R[1000:])   # incremental add like index.add called twice
    d2, idx = eng.search(Q, 200)
    d2, idx = d2.cpu().numpy(), idx.cpu().numpy()
    rd2, ridx = O().knn_l2(R, Q, 200)
    assert np.abs(d2 - rd2).max() < 1e-5                # squared L2 of unit vectors, fp32 GEMM
    assert np.all(np.diff(d2, axis=1) >= 0)             # ascending
    # ids equal wherever neighbouring reference distances are separated by more than fp32 noise
    sep = np.minimum(np.diff(rd2, axis=1, prepend=-1), np.diff(rd2, axis=1, append=9)) > 1e-5
    assert np.array_equal(idx[sep], ridx[sep])
    assert (idx == ridx).mean() > 0.995
    # the set of the 200 nearest is the same except at the k-boundary
    assert np.mean([len(set(a) & set(b)) for a, b in zip(idx, ridx)]) > 199.5


def test_knn_ties_and_small_db(eng):
    R = np.zeros((5, 8), np.float32)
    R[:, 0] = [1, 1, 2, 1, 3]      # rows 0,1,3 identical -> ties resolved by lower id
    Q = np.zeros((2, 8), np.float32)
    Q[0, 0] = 1
    Q[1, 0] = 2.9
    eng.db_reset()
    eng.db_add(R)
    d2, idx = eng.search(Q, 7)     # k > n: padded with (+inf, -1) like faiss
    d2, idx = d2.cpu().numpy(), idx.cpu().numpy()
    assert idx[0].tolist() == [0, 1, 3, 2, 4, -1, -1]
    assert np.allclose(d2[0][:5], [0, 0, 0, 1, 4]) and np.isinf(d2[0][5:]).all()
    assert idx[1].tolist()[:2] == [4, 2]
    d2b, idxb = eng.search(Q, 2)   # boundary tie: exactly 2 of the 3 equal rows, lowest ids
    assert idxb.cpu().numpy()[0].tolist() == [0, 1]


def test_merge_topk_equals_single_shard(eng):
    rng = np.random.Generator(np.random.PCG64(101))
    R = rng.standard_normal((2000, 64)).astype(np.float32)
    Q = rng.standard_normal((40, 64)).astype(np.float32)
    eng.db_reset()
    eng.db_add(R)
    d2, idx = (t.cpu().numpy() for t in eng.search(Q, 50))
    parts = [(0, 700), (700, 1400), (1400, 2000)]
    dp, ip = [], []
    for a, b in parts:
        eng.db_reset()
        eng.db_add(R[a:b])
        d, i = eng.search(Q, 50)
        dp.append(d.cpu().numpy())
        ip.append(i.cpu().numpy() + a)
    dm, im = eng.merge_topk(np.concatenate(dp, 1), np.concatenate(ip, 1), 3, 50)
    assert np.array_equal(im.cpu().numpy(), idx)          # merged ids == single-shard ids, bit for bit
    assert np.array_equal(dm.cpu().numpy(), d2)
    om, oi = O().merge_topk(dp, ip, 50)
    assert np.array_equal(oi, idx)


def test_vote_golden_bit_exact(eng):
    z = np.load(os.path.join(G, "vote_cases.npz"))
    off = z["off"].astype(np.int32)
    segRange = [np.arange(off[i], off[i + 1]) for i in range(len(off) - 1)]
    imInds = z["imInds"].astype(np.int32)
    for n in (1, 5):
        pred, sc = eng.vote(z["matches"], z["sims"], off, n_top=n, img_of_seg=imInds)
        assert np.array_equal(pred.cpu().numpy(), z[f"wt_n{n}"])          # identical image ids
    _, rs = O().get_matches_wt_borda_im(z["matches"], len(segRange), z["sims"], segRange, z["imInds"], n=5, return_scores=True)
    got = sc.cpu().numpy()
    for i, row in enumerate(rs):
        assert np.array_equal(got[i][:len(row)], np.array(row))            # fp64 scores bit-identical
    # integer mode: counts exact; ids compared where the oracle's n-th count is unique
    from revisit_anything_amd._lib import VOTE_COUNT
    pred, sc = eng.vote(z["matches"], None, off, n_top=5, mode=VOTE_COUNT, img_of_seg=imInds)
    pred, sc = pred.cpu().numpy(), sc.cpu().numpy()
    _, counts = O().get_matches_max_seg_topk(z["matches"], len(segRange), segRange, z["imInds"], n=5)
    for i, bc in enumerate(counts):
        top = np.sort(bc)[::-1][:5]
        assert np.array_equal(sc[i][:len(top[top > 0])], top[top > 0].astype(np.float64))
        for j in range(5):
            if pred[i, j] >= 0:
                assert bc[pred[i, j]] == sc[i, j]
        # documented rule: (count desc, image id asc)
        order = np.lexsort((np.arange(len(bc)), -bc))[:5]
        order = order[bc[order] > 0]
        assert pred[i][:len(order)].tolist() == order.tolist()
    # hand-made tie case: first appearance decides
    pred, _ = eng.vote(z["tie_matches"], z["tie_sims"], np.array([0, 2], np.int32), n_top=4, img_of_seg=z["tie_imInds"].astype(np.int32))
    assert pred.cpu().numpy()[0].tolist() == z["tie_pred"].tolist()


def test_vote_long_runs_exact_and_inexact_sums_are_bit_identical(eng):
    """The vote kernel adds an image's fp64 weights with LDS atomics when every weight of the run is 0 or in [2^-17, 1]
    (all partial sums are then exact: the order cannot matter) and sequentially, in the reference's visiting order,
    otherwise.  Three query images whose matches concentrate on a few reference images (runs of hundreds of entries):
    (a) ordinary similarities, (b) the same with a handful of weights pushed below 2^-17 (the sums are no longer exact:
    the sequential path must be taken, and only it reproduces the reference's rounding), (c) k = 3 short lists.  fp64
    scores bit-identical to the oracle's python-float sums, ids identical."""
    rng = np.random.Generator(np.random.PCG64(77))
    S, k, n_ref_img = 40, 50, 60
    imInds = np.repeat(np.arange(n_ref_img), 20).astype(np.int32)              # 1200 reference segments
    off = np.array([0, S, 2 * S, 2 * S + 3], np.int32)
    nseg = int(off[-1])
    matches = rng.integers(0, len(imInds), size=(nseg, k))
    hot = rng.random((nseg, k)) < 0.8
    matches = np.where(hot, rng.integers(0, 40, size=(nseg, k)), matches).astype(np.int64)   # 80 % on images 0 and 1
    sims = rng.random((nseg, k)).astype(np.float32)
    sims_tiny = sims.copy()
    lo = float(sims.min())
    flat = sims_tiny.reshape(-1)
    pick = rng.choice(flat.size, size=12, replace=False)
    flat[pick] = np.float32(lo) + np.float32(1e-6) * rng.random(12).astype(np.float32)       # weights ~1e-6 < 2^-17
    segRange = [np.arange(off[i], off[i + 1]) for i in range(len(off) - 1)]
    for sm in (sims, sims_tiny):
        pred, sc = eng.vote(matches, sm, off, n_top=5, img_of_seg=imInds)
        opred, rs = O().get_matches_wt_borda_im(matches, len(segRange), sm, segRange, imInds, n=5, return_scores=True)
        got = sc.cpu().numpy()
        for i, row in enumerate(rs):
            assert np.array_equal(got[i][:len(row)], np.array(row)), i                         # bit-identical fp64 sums
            assert pred.cpu().numpy()[i][:len(row)].tolist() == [int(p) for p in opred[i][:len(row)]]
    w = (sims_tiny - sims_tiny.min()) / (sims_tiny.max() - sims_tiny.min())
    assert ((w > 0) & (w < 2.0 ** -17)).sum() >= 8                                             # the case really is exercised


def test_e2e_small_golden(eng):
    """recall_segloc chain on the device: normalise -> add -> search 200 -> keep 50 -> 2-d2 -> vote -> recall."""
    z = np.load(os.path.join(G, "e2e_small.npz"))
    n_img, S, d, n_q = (int(z[k]) for k in ("n_img", "S", "d", "n_q"))
    R, img = synth().make_planted_db(n_img, S, d, seed=3000)
    Q, tau, off = synth().make_planted_queries(R, n_img, S, n_q, seed=4000, sigma_q=3.0)
    sr = np.random.Generator(np.random.PCG64(1)).uniform(0.5, 2.0, size=(R.shape[0], 1))
    sq = np.random.Generator(np.random.PCG64(2)).uniform(0.5, 2.0, size=(Q.shape[0], 1))
    Rn = eng.normalize_rows((R.astype(np.float64) * sr).astype(np.float32))
    Qn = eng.normalize_rows((Q.astype(np.float64) * sq).astype(np.float32))
    eng.db_reset()
    eng.db_add(Rn, img)
    d2, idx = eng.search(Qn, 200)
    sims, m50 = eng.sims_from_d2(d2, idx, 50)
    assert np.abs(sims.cpu().numpy() - z["sims_50"]).max() < 1e-4           # "2 - d^2" within 1e-4 (north_star)
    # ids: EVERY mismatch against the reference-generated fixture must be a near-tie (the row ranked there is as
    # close as the fixture's to 1e-5 in exact arithmetic) -- not a rate
    got = m50.cpu().numpy()
    qq, rr = np.nonzero(got != z["matches_50"])
    assert len(qq) < 0.01 * got.size
    if len(qq):
        Rn64 = Rn.cpu().numpy().astype(np.float64)
        Qn64 = Qn.cpu().numpy().astype(np.float64)
        mine = ((Qn64[qq] - Rn64[got[qq, rr]]) ** 2).sum(1)
        theirs = ((Qn64[qq] - Rn64[z["matches_50"][qq, rr]]) ** 2).sum(1)
        assert np.abs(mine - theirs).max() < 1e-5
    pred, _ = eng.vote(m50, sims, off, n_top=5)
    pred = pred.cpu().numpy()
    assert np.array_equal(pred[:, 0], z["preds"][:, 0])                        # identical top-1 image ids
    # ranks 2..5: identical, or a score near-tie in the oracle's own fp64 scores
    segRange = [np.arange(off[i], off[i + 1]) for i in range(n_q)]
    _, oscores = O().get_matches_wt_borda_im(z["matches_50"], n_q, z["sims_50"], segRange, img.astype(np.int64), n=5,
                                             return_scores=True)
    for i in range(n_q):
        for j in range(5):
            a, b = int(pred[i, j]), int(z["preds"][i, j])
            if a != b:   # only legitimate at a score near-tie between neighbouring ranks of the oracle's own list
                sc = oscores[i]
                near = [abs(sc[j] - sc[t]) for t in (j - 1, j + 1) if 0 <= t < len(sc)] if j < len(sc) else []
                assert near and min(near) < 1e-5, (i, j, a, b, sc)
    gt = [[int(t)] for t in tau]
    gt[5] = []
    rec = O().calc_recall([list(p[p >= 0]) for p in pred], gt, 5)
    assert np.allclose(rec, z["recalls"])


# ------------------------------------------------------------------------------------------------
# large-database search path: sampled thresholds + filtered GEMM epilogue (exactness must not depend on it)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [48, 96, 64])   # 48: fp32 filter levels; 96: bf16x3 filter; 64: fp16 filter (+ exact refinement)
def test_knn_leveled_filter_path_is_exact(eng, d):
    rng = np.random.Generator(np.random.PCG64(200))
    n, nq, k = 70001, 300, 50                 # > 32768 rows: one filter level (stride 16)
    R = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    R[5000] = R[123]                            # exact duplicates: ties must resolve to the lower id
    R[60000] = R[123]
    Q[0] = R[123] + 1e-3
    eng.db_reset()
    eng.db_add(R)
    d2, idx = (t.cpu().numpy() for t in eng.search(Q, k))
    rd2, ridx = O().knn_l2(R, Q, k)
    assert np.all(np.diff(d2, axis=1) >= 0)
    assert np.abs(d2 - rd2).max() < 2e-4 * rd2.max()
    assert idx[0][:3].tolist() == [123, 5000, 60000]
    sep = np.minimum(np.diff(rd2, axis=1, prepend=-1), np.diff(rd2, axis=1, append=1e9)) > 1e-3
    assert np.array_equal(idx[sep], ridx[sep])
    assert (idx == ridx).mean() > 0.99
    # the same rows through the plain matrix path (database split below the level threshold) give identical bits
    parts = [(0, 30000), (30000, 60000), (60000, n)]
    dp, ip = [], []
    for a, b in parts:
        eng.db_reset()
        eng.db_add(R[a:b])
        dd, ii = eng.search(Q, k)
        dp.append(dd.cpu().numpy())
        ip.append(np.where(ii.cpu().numpy() >= 0, ii.cpu().numpy() + a, -1))
    dm, im = O().merge_topk(dp, ip, k)
    assert np.array_equal(im, idx) and np.array_equal(dm, d2)


@pytest.mark.parametrize("n,nq", [(3000, 64), (70001, 300), (70001, 40)])   # matrix path; leveled batch; single-image pass
def test_knn_exact_duplicates_are_at_distance_zero_like_faiss(eng, n, nq):
    """faiss's IndexFlatL2 sets a negative ||q||^2 + ||r||^2 - 2 q.r to zero before its heap sees it (utils/distances.cpp,
    exhaustive_L2sqr_blas), so a query that IS a reference row has d2 = 0 and `2 - d2` (place_rec_main.py:78-81) never exceeds 2
    (VERDICT r05 missing #5: the unclamped form returned -1e-7).  Unit rows of d = 1024, every query a copy of a reference row,
    some rows present two or three times: d2 >= 0 everywhere, the copies first and in id order, sims <= 2, values equal to the
    oracle's (which clamps like faiss)."""
    import torch
    rng = np.random.Generator(np.random.PCG64(606))
    d, k = 1024, 20
    R = rng.standard_normal((n, d)).astype(np.float32)
    R /= np.linalg.norm(R, axis=1, keepdims=True)
    src = rng.choice(n - 10, nq, replace=False)
    R[n - 1] = R[src[0]]
    R[n - 2] = R[src[0]]
    R[n - 3] = R[src[1]]
    Q = R[src].copy()
    eng.db_reset()
    eng.db_add(R)
    d2, idx = eng.search(Q, k)
    sims, _ = eng.sims_from_d2(d2, idx, 10)
    d2, idx, sims = d2.cpu().numpy(), idx.cpu().numpy(), sims.cpu().numpy()
    assert (d2 >= 0).all() and (sims <= 2.0).all()
    # (||q||^2 + ||r||^2 - 2 q.r of a row with itself is a few ulp of 2 either side of zero in fp32 -- in faiss too; the negative
    #  ones are the ones the clamp catches)
    assert (d2[:, 0] < 5e-6).all() and (d2[:, 0] == 0).any() and (sims[:, 0] > 2.0 - 5e-6).all()
    assert idx[0, :3].tolist() == sorted([int(src[0]), n - 2, n - 1]) and (d2[0, :3] == d2[0, 0]).all()
    assert idx[1, :2].tolist() == sorted([int(src[1]), n - 3]) and (d2[1, :2] == d2[1, 0]).all()
    assert np.array_equal(idx[2:, 0], src[2:])
    rd2, ridx = O().knn_l2(R, Q, k)
    assert (rd2 >= 0).all() and np.abs(d2 - rd2).max() < 1e-5
    eng.db_reset()


def test_knn_raw_descriptor_width_refines_without_lds_query_cache(eng):
    """Raw K*D-wide rows (BASELINE configs[1] searches 98 304-d descriptors without PCA): the query row no longer fits
    the refinement kernel's LDS cache.  Filtered search over the whole database == matrix-path search over three
    parts merged, bit for bit.  Rows live on a 32-d subspace so that the fp16 filter margin stays selective."""
    import torch
    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    n, nq, k, d, lat = 40000, 200, 50, 40960, 32
    B = torch.linalg.qr(torch.randn(d, lat, device=dev, generator=g))[0].T.contiguous()      # [lat, d], orthonormal rows
    R = torch.nn.functional.normalize(torch.randn(n, lat, device=dev, generator=g), dim=1) @ B
    Q = torch.nn.functional.normalize(torch.randn(nq, lat, device=dev, generator=g), dim=1) @ B
    eng.db_reset()
    eng.db_add(R)
    d2, idx = (t.cpu().numpy() for t in eng.search(Q, k))
    dp, ip = [], []
    for a, b in [(0, 15000), (15000, 30000), (30000, n)]:
        eng.db_reset()
        eng.db_add(R[a:b].contiguous())
        dd, ii = eng.search(Q, k)
        dp.append(dd.cpu().numpy())
        ip.append(ii.cpu().numpy() + a)
    eng.db_reset()
    dm, im = O().merge_topk(dp, ip, k)
    assert np.array_equal(im, idx) and np.array_equal(dm, d2)
    # and against float64 distances in the latent space (B has orthonormal rows)
    Rl, Ql = (R @ B.T).double().cpu().numpy(), (Q @ B.T).double().cpu().numpy()
    rd2, ridx = O().knn_l2(Rl, Ql, k)
    assert np.abs(d2 - rd2).max() < 1e-4
    assert (idx == ridx).mean() > 0.98


def test_knn_dense_block_of_neighbours_in_consecutive_rows(eng):
    """Spatially coherent database (the 50 segments of a place sit in consecutive rows): 64 similar queries whose 200
    nearest rows are the SAME 200 consecutive rows.  One wave's 64 x 128 block of the fp16 filter then holds far more hits
    than its LDS list (2048): the block must take the direct path and the search must stay exact, without a fall-back."""
    rng = np.random.Generator(np.random.PCG64(205))
    n, nq, k, d = 70001, 64, 200, 64
    R = rng.standard_normal((n, d)).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32) * 3.0
    R[5000:5260] = b + 0.05 * rng.standard_normal((260, d)).astype(np.float32)
    Q = (b + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    eng.db_reset()
    eng.db_add(R)
    eng.set_profiling(True)
    eng.profile_reset()
    d2, idx = (t.cpu().numpy() for t in eng.search(Q, k))
    launches = eng.stage_ms("knn_gemm")[1] + eng.stage_ms("knn_level0")[1]
    eng.set_profiling(False)
    st = eng.search_stats()
    rd2, ridx = O().knn_l2(R, Q, k)
    # level-0 matrix GEMM + the filter levels (+ the rigorous redo's, if the low-rank thresholds did not verify on this
    # place-coherent layout): never the exact distance-matrix fall-back
    assert st["n_fallback"] == 0 and launches == 1 + st["levels"] + (2 if st["n_redo"] else 0), (launches, st)
    assert np.all((idx >= 5000) & (idx < 5260))
    # ||q||^2 ~ 600 and d2 ~ 0.3: the fp32 form q2 + r2 - 2 q.r cancels ~11 bits (as faiss' does)
    tol = 4e-6 * float((Q * Q).sum(1).max() + (R * R).sum(1).max())
    assert np.abs(d2 - rd2).max() < tol
    sep = np.minimum(np.diff(rd2, axis=1, prepend=-1), np.diff(rd2, axis=1, append=1e9)) > 2 * tol
    assert np.array_equal(idx[sep], ridx[sep])
    assert np.all(np.diff(d2, axis=1) >= 0)
    # the same rows through the plain matrix path (database split below the level threshold) give identical bits
    dp, ip = [], []
    for a, b_ in [(0, 30000), (30000, 60000), (60000, n)]:
        eng.db_reset()
        eng.db_add(R[a:b_])
        dd, ii = eng.search(Q, k)
        dp.append(dd.cpu().numpy())
        ip.append(np.where(ii.cpu().numpy() >= 0, ii.cpu().numpy() + a, -1))
    dm, im = O().merge_topk(dp, ip, k)
    assert np.array_equal(im, idx) and np.array_equal(dm, d2)


@pytest.mark.parametrize("d", [16, 32, 64])   # fp32 / bf16x3 / fp16 filter (the last also overflows its per-wave hit lists)
def test_knn_filter_overflow_falls_back_to_exact(eng, d):
    """Adversarial layout: every sampled row (id % 16 == 0) is far away, all other rows are near -> the sampled
    threshold admits everything, the candidate lists overflow, and the search must still be exact."""
    rng = np.random.Generator(np.random.PCG64(201))
    n, nq, k = 40000, 20, 10
    R = (rng.standard_normal((n, d)) * 0.01).astype(np.float32)
    R[::16] += 100.0
    Q = (rng.standard_normal((nq, d)) * 0.01).astype(np.float32)
    eng.db_reset()
    eng.db_add(R)
    d2, idx = (t.cpu().numpy() for t in eng.search(Q, k))
    assert eng.search_stats()["n_fallback"] == nq          # every query overflowed -> every query was redone exactly
    rd2, ridx = O().knn_l2(R, Q, k)
    assert np.abs(d2 - rd2).max() < 1e-6
    assert (idx == ridx).mean() > 0.9 and np.all(idx % 16 != 0)


# ------------------------------------------------------------------------------------------------
# device adjacency (empty-circle test) against Qhull (the reference's library) and the golden cases
# ------------------------------------------------------------------------------------------------
def test_adjacency_device_matches_golden_masks(eng):
    z = np.load(os.path.join(G, "adjacency_cases.npz"))
    for S in (1, 2, 3, 4, 6, 12, 50):
        m = np.unpackbits(z[f"S{S}_masks"], axis=1)[:, :60 * 80].reshape(S, 60, 80).astype(np.uint8)
        cent = eng.mask_centroids(m)
        for order in (1, 2, 3):
            a = eng.adjacency(cent, np.array([0, S], np.int32), order).cpu().numpy().reshape(S, S)
            assert np.array_equal(a.astype(bool), z[f"S{S}_o{order}"]), (S, order)


def test_adjacency_device_matches_qhull_on_random_centroids(eng):
    rng = np.random.Generator(np.random.PCG64(300))
    sizes = [0, 1, 2, 3, 4, 5, 7, 20, 50, 64, 65, 150]
    cents = [rng.uniform(0, 320, size=(S, 2)) * [1.0, 0.75] for S in sizes]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cat = np.concatenate(cents)
    for order in (1, 2, 3):
        a = eng.adjacency(cat, offs, order).cpu().numpy()
        pos = 0
        for S, c in zip(sizes, cents):
            want = O().adjacency_from_centroids(c, order)
            got = a[pos:pos + S * S].reshape(S, S).astype(bool)
            assert np.array_equal(got, want), (S, order)
            pos += S * S
        assert pos == len(a)


def test_adjacency_empty_mask_raises(eng):
    m = np.zeros((5, 20, 20), np.uint8)
    m[:4, 3:6, 3:9] = 1
    m[1, 10:12, 1:3] = 1
    cent = eng.mask_centroids(m)
    with pytest.raises(ValueError):
        eng.adjacency(cent, np.array([0, 5], np.int32), 2, check_empty=True)


def test_adjacency_flags_non_generic_centroids_and_the_pipeline_uses_qhull(eng):
    """Exactly co-circular centroids (a lattice rectangle with an empty circle) and duplicate centroids have no unique
    Delaunay triangulation: the device kernel keeps both diagonals / links the duplicates, Qhull picks one diagonal /
    drops the duplicate.  The kernel reports such images (bits 16.. of the counter); with check_empty the engine raises
    and SegVLADPipeline recomputes the batch with the reference's Qhull path."""
    import torch
    from revisit_anything_amd._lib import SegVLADDegenerateError, SegVLADError  # noqa: F401
    from revisit_anything_amd.pipeline import SegVLADPipeline

    generic = np.array([[3.0, 4.0], [40.5, 7.25], [21.0, 33.0], [8.0, 29.5], [30.0, 18.0]])
    rect = np.array([[10.0, 10.0], [30.0, 10.0], [30.0, 22.0], [10.0, 22.0], [55.0, 60.0]])    # 4 co-circular + 1 far away
    dup = np.array([[5.0, 5.0], [25.0, 6.0], [25.0, 6.0], [9.0, 31.0], [40.0, 28.0]])
    offs = np.array([0, 5], np.int32)
    eng.adjacency(generic, offs, 1, check_empty=True)                                            # generic: no complaint
    for c in (rect, dup):
        with pytest.raises(SegVLADDegenerateError):
            eng.adjacency(c, offs, 1, check_empty=True)
        eng.adjacency(c, offs, 1)                                                                # unchecked: still computes
    a = eng.adjacency(rect, offs, 1).cpu().numpy().reshape(5, 5).astype(bool)
    assert a[0, 2] and a[1, 3]                                                                   # both diagonals kept
    q = O().adjacency_from_centroids(rect, 1)
    assert q[0, 2] != q[1, 3]                                                                    # Qhull keeps exactly one
    # pipeline: masks whose centroids form the rectangle -> descriptors equal the host-Qhull pipeline's bit for bit
    K, D, H, W = 8, 64, 112, 140
    C = synth().make_vocab(K, D, seed=2)
    eng.set_vocab(C)
    m = np.zeros((5, 56, 70), np.uint8)
    for j, (x, y) in enumerate([(10, 10), (30, 10), (30, 22), (10, 22), (55, 44)]):
        m[j, y - 2:y + 3, x - 2:x + 3] = 1                                                       # 5 x 5 squares: centroid = centre
    toks = torch.from_numpy(synth().make_tokens(C, 80, seed=77, noise=0.2)[None]).to(eng.device)
    masks = torch.from_numpy(m).to(eng.device)
    d_dev = SegVLADPipeline(eng, H, W, order=1, use_pca=False).describe(toks, masks, offs).cpu().numpy()
    d_host = SegVLADPipeline(eng, H, W, order=1, use_pca=False, host_adjacency=True).describe(toks, masks, offs).cpu().numpy()
    assert np.array_equal(d_dev, d_host)
    # per-image flags (segvlad_adjacency_flagged): in a batch of [generic, rect, generic, dup] exactly images 1 and 3 are
    # reported, and the pipeline patches exactly their blocks with Qhull's answer (the others keep the device kernel's)
    cat = np.concatenate([generic, rect, generic + 1.5, dup])
    offs4 = np.array([0, 5, 10, 15, 20], np.int32)
    adj, flags = eng.adjacency_flagged(cat, offs4, 1)
    assert flags.tolist() == [0, 2, 0, 2]
    pipe = SegVLADPipeline(eng, H, W, order=1, use_pca=False)
    patched = pipe._patch_with_qhull(adj.clone(), torch.from_numpy(cat).to(eng.device), offs4, [1, 3]).cpu().numpy()
    for b_, c_ in enumerate((generic, rect, generic + 1.5, dup)):
        want = O().adjacency_from_centroids(c_, 1).astype(np.uint8).reshape(-1)
        assert np.array_equal(patched[25 * b_:25 * b_ + 25], want), b_
    assert np.array_equal(adj.cpu().numpy()[:25], patched[:25])                                  # untouched
    # a quadruple that is co-circular only up to rounding (the rectangle, rotated by an irrational angle) is reported too
    th = 0.7312
    rot = rect @ np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]).T + np.array([100.0, 50.0])
    assert eng.adjacency_flagged(rot, offs, 1)[1].tolist() == [2]
    # an empty mask sets bit 0 of ITS image only
    cent_nan = np.concatenate([generic, generic + 2.0])
    cent_nan[7] = np.nan
    assert eng.adjacency_flagged(cent_nan, np.array([0, 5, 10], np.int32), 1)[1].tolist() == [0, 1]


def test_pipeline_device_adjacency_equals_host_qhull(eng):
    import torch
    from revisit_anything_amd.pipeline import SegVLADPipeline

    K, D, H, W, S, B = 32, 64, 112, 140, 12, 3
    C = synth().make_vocab(K, D, seed=1)
    eng.set_vocab(C)
    toks = torch.from_numpy(np.stack([synth().make_tokens(C, 80, seed=10 + b, noise=0.2) for b in range(B)])).to(eng.device)
    masks = torch.from_numpy(np.concatenate([synth().make_blob_masks(S, 56, 70, seed=20 + b) for b in range(B)]).astype(np.uint8)).to(eng.device)
    offs = (np.arange(B + 1) * S).astype(np.int32)
    d_dev = SegVLADPipeline(eng, H, W, order=3, use_pca=False, host_adjacency=False).describe(toks, masks, offs).cpu().numpy()
    d_host = SegVLADPipeline(eng, H, W, order=3, use_pca=False, host_adjacency=True).describe(toks, masks, offs).cpu().numpy()
    assert np.array_equal(d_dev, d_host)


@pytest.mark.parametrize("small_plan", [0, 1])
@pytest.mark.parametrize("n", [600000, 250000])
def test_knn_bf16_path_two_levels_unit_vectors(eng, n, small_plan):
    """Two filter levels (strides 256, 16, 1); unit-norm 128-d rows like the PCA'd descriptors.  600 000 rows: the
    sample of 2343 rows admits 8.5 % of the next level; 250 000 rows (the 1 M-row database on 4 GPUs): the level plan's
    smallest sample for k = 200, 976 rows, admits 20.5 % -- just below the filter's per-wave list capacity.
    The fp16 and the bf16x3 filters (+ fp32 refinement) must reproduce the all-fp32 path bit for bit.
    small_plan = 1 (the default for <= 128 queries): ONE filter level behind a 2048..4096-row sample (strides 256 / 64),
    lists ranked by the workgroup-per-list select, refinement lists shared by workgroups -- same bits again."""
    import torch

    g = torch.Generator(device=eng.device)
    g.manual_seed(7)
    d, nq, k = 128, 64, 200
    R = torch.nn.functional.normalize(torch.randn(n, d, device=eng.device, generator=g), dim=1)
    Q = torch.nn.functional.normalize(R[torch.randint(0, n, (nq,), device=eng.device, generator=g)] +
                                      0.3 * torch.randn(nq, d, device=eng.device, generator=g) / d ** 0.5, dim=1)
    eng.db_reset()
    eng.db_add(R)
    eng.set_option("small_plan", small_plan)
    try:
        d2, idx = eng.search(Q, k)                 # default: fp16 single-product filter
        st = eng.search_stats()
        assert st["filter"] == "f16" and st["levels"] == (1 if small_plan else 2)
        assert st["n_fallback"] == 0
        eng.set_option("knn_filter", "fp32")       # same levels, fp32 filter GEMM
        d2f, idxf = eng.search(Q, k)
        assert eng.search_stats()["filter"] == "fp32"
        eng.set_option("knn_filter", "bf16x3")
        d2b, idxb = eng.search(Q, k)
        assert eng.search_stats()["filter"] == "bf16x3"
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("small_plan", 1)
    assert torch.equal(idx, idxf) and torch.equal(d2, d2f)
    assert torch.equal(idxb, idxf) and torch.equal(d2b, d2f)
    # and against the oracle on a slice of the queries
    Rh = R.cpu().numpy()
    rd2, ridx = O().knn_l2(Rh, Q[:8].cpu().numpy(), k)
    assert np.abs(d2[:8].cpu().numpy() - rd2).max() < 1e-5
    assert (idx[:8].cpu().numpy() == ridx).mean() > 0.98


def test_knn_f16_filter_unnormalised_scales(eng):
    """The fp16 filter rescales queries and rows by powers of two: rows of very different magnitude (1e-3 .. 1e3),
    incremental adds that force a rescale, and zero rows must not change the exact result."""
    rng = np.random.Generator(np.random.PCG64(202))
    n, d, nq, k = 50000, 64, 40, 20
    R = rng.standard_normal((n, d)).astype(np.float32) * 1e-3
    R[n // 2:] *= 1e3
    R[7] = 0
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q[:10] *= 1e-3
    eng.db_reset()
    eng.db_add(R[:40000] * np.float32(1.0))
    d2a, _ = eng.search(Q[:10], k)              # builds the fp16 image with the small scale
    eng.db_add(R[40000:])                        # larger magnitudes arrive: the image must be rebuilt
    d2, idx = (t.cpu().numpy() for t in eng.search(Q, k))
    rd2, ridx = O().knn_l2(R, Q, k)
    assert np.allclose(d2, rd2, rtol=2e-5, atol=1e-9)
    sep = np.minimum(np.diff(rd2, axis=1, prepend=-1), np.diff(rd2, axis=1, append=1e30)) > 1e-4 * rd2
    assert np.array_equal(idx[sep], ridx[sep])


def test_incidence_centroids_fused_equals_separate(eng):
    for (S, Hm, Wm, H, W, seed) in [(50, 240, 320, 480, 640, 1), (9, 64, 96, 126, 150, 2), (5, 208, 272, 100, 130, 3)]:
        masks = synth().make_blob_masks(S, Hm, Wm, seed=seed).astype(np.uint8) * np.uint8(255)   # any non-zero byte counts
        bits, cent = eng.incidence_centroids(masks, H, W)
        N = (H // 14) * (W // 14)
        assert np.array_equal(O().unpack_bits_u64(bits.cpu().numpy().view(np.uint64), N), O().incidence(masks != 0, H, W))
        assert np.array_equal(cent.cpu().numpy(), O().mask_centroids([m for m in masks]))
    # width not a multiple of 16 -> the per-token fallback kernel, same results
    masks = synth().make_blob_masks(6, 50, 70, seed=4).astype(np.uint8)
    bits, cent = eng.incidence_centroids(masks, 126, 150)
    assert np.array_equal(O().unpack_bits_u64(bits.cpu().numpy().view(np.uint64), 90), O().incidence(masks != 0, 126, 150))
    assert np.array_equal(cent.cpu().numpy(), O().mask_centroids([m for m in masks]))


def test_pca_f16_split_path_is_fp32_class(eng):
    """The projection runs as three fp16 MFMA products of a two-term split.  It must be at least as close to the
    fp64 oracle as the all-fp32 MFMA GEMM (option pca_arith=fp32) on descriptor-like data with a wide dynamic range."""
    rng = np.random.Generator(np.random.PCG64(400))
    KD, P, n = 64 * 256, 96, 300
    mean, comps, var = synth().make_pca_model(KD, P, seed=6)
    X = rng.standard_normal((n, KD)).astype(np.float32)
    X[:, ::7] *= 1e-4                                   # tiny components next to large ones
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    eng.pca_set(mean, comps, var, whiten=True)
    ref = O().pca_transform(X, mean, comps, var, True)
    y16 = eng.pca_apply(X).cpu().numpy()
    try:
        eng.set_option("pca_arith", "fp32")
        y32 = eng.pca_apply(X).cpu().numpy()
    finally:
        eng.set_option("pca_arith", "auto")
    e16 = np.abs(y16 - ref).max() / np.abs(ref).max()
    e32 = np.abs(y32 - ref).max() / np.abs(ref).max()
    assert e16 < 2e-5 and e16 < 4 * e32 + 1e-6, (e16, e32)
    cos = (y16 * ref).sum(1) / (np.linalg.norm(y16, axis=1) * np.linalg.norm(ref, axis=1))
    assert (1 - cos).max() < 5e-7   # fp32 outputs: the cosine itself is only resolved to ~1e-7


def test_seg_vlad_more_segments_than_tokens(eng):
    """S > N (70 segments over 48 tokens): the per-image scratch that first holds the token order and then the
    per-segment block counts must be sized by the larger of the two."""
    D, K, N = 64, 8, 48
    C = synth().make_vocab(K, D, seed=291)
    rng = np.random.Generator(np.random.PCG64(292))
    toks, incs, adjs = [], [], []
    for b, S in enumerate([70, 3]):
        toks.append(synth().make_tokens(C, N, seed=2950 + b, noise=0.3))
        incs.append(rng.random((S, N)) < 0.2)
        adjs.append(np.eye(S, dtype=bool) | (rng.random((S, S)) < 0.1))
    eng.set_vocab(C)
    offs = np.concatenate([[0], np.cumsum([i.shape[0] for i in incs])]).astype(np.int32)
    bits = np.concatenate([O().pack_bits_u64(i) for i in incs]).view(np.int64)
    out = eng.seg_vlad(np.stack(toks), bits, offs, cat_adj(adjs))["out"].cpu().numpy()
    ref = np.concatenate([O().seg_vlad(toks[b], incs[b], C, adjs[b]) for b in range(2)])
    assert np.abs(out - ref).max() < 2e-6


def test_images_pca_project_form_big_clusters_and_partial_chunk(eng):
    """The "project then aggregate" form of segvlad_images_pca where its token kernel leaves the common case: a cluster
    holding >= 256 tokens of one image (lists read from global memory), D = 96 (a partial 128-column chunk), S > 64
    (two segment chunks) and an image without segments -- against the fp64 oracle and the "planes" form."""
    D, K, N, P = 96, 8, 40 * 18, 24
    C = synth().make_vocab(K, D, seed=191)
    rng = np.random.Generator(np.random.PCG64(192))
    toks, incs, adjs = [], [], []
    for b, S in enumerate([70, 0, 9]):
        toks.append(synth().make_tokens(C[:2] if b == 0 else C, N, seed=1950 + b, noise=0.3))   # b == 0: ~360 tokens per cluster
        incs.append(rng.random((S, N)) < 0.1)
        adjs.append(np.eye(S, dtype=bool) | (rng.random((S, S)) < 0.05))
    mean, comps, var = synth().make_pca_model(K * D, P, seed=19)
    eng.set_vocab(C)
    eng.pca_set(mean, comps, var, whiten=True)
    offs = np.concatenate([[0], np.cumsum([i.shape[0] for i in incs])]).astype(np.int32)
    bits = np.concatenate([O().pack_bits_u64(i) for i in incs]).view(np.int64)
    adj = cat_adj(adjs)
    tk = np.stack(toks)
    lab = eng.seg_vlad(tk, bits, offs, adj, want_labels=True)["labels"].cpu().numpy()
    assert np.bincount(lab[0], minlength=K).max() >= 256
    ref = np.concatenate([O().seg_vlad(toks[b], incs[b], C, adjs[b]) for b in range(3) if incs[b].shape[0]])
    ref = O().pca_transform(ref, mean, comps, var, True)
    ys = {}
    for path in ("project", "planes"):
        eng.set_option("pca_path", path)
        try:
            ys[path] = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=False)["out"].cpu().numpy()
        finally:
            eng.set_option("pca_path", "auto")
        assert np.abs(ys[path] - ref).max() <= 3e-5 * np.abs(ref).max()
    assert np.abs(ys["project"] - ys["planes"]).max() <= 1e-5 * np.abs(ref).max()
    # the token kernels' counted waits (DMA queue + plane stores in one vmcnt stream) against the same kernels waiting for
    # everything at every step (development switch debug_search = 7): bit-identical -- with the Gram kernel taking the tasks
    # of <= 32 tokens (the default) and with the block-sum kernel taking every task (tnk_gram = 0); the two agree to fp32 noise
    eng.set_option("pca_path", "project")
    try:
        for gram in ("1", "0"):
            eng.set_option("tnk_gram", gram)
            y_cnt = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=False)["out"].cpu().numpy()
            eng.set_option("debug_search", "7")
            y_safe = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=False)["out"].cpu().numpy()
            eng.set_option("debug_search", "0")
            assert np.array_equal(y_safe, y_cnt), gram
            if gram == "1":
                assert np.array_equal(y_cnt, ys["project"])
            else:
                assert np.abs(y_cnt - ref).max() <= 3e-5 * np.abs(ref).max()
                assert np.abs(y_cnt - ys["project"]).max() <= 2e-6 * np.abs(ref).max()
    finally:
        eng.set_option("debug_search", "0")
        eng.set_option("tnk_gram", "1")
        eng.set_option("pca_path", "auto")


def test_block_norms_from_the_gram_matrix_equal_the_block_sums(eng):
    """Round 4: tasks (image, cluster) of <= 32 tokens take their block norms from the Gram matrix of their residuals on the
    16-bit matrix pipe (gram_norms_kernel: ||sum_t m_t r_t||^2 = m^T (R R^T) m, three split products); larger tasks and
    tnk_gram = 0 keep the fp32 block sums.  Bench proportions scaled down (24 tokens per cluster on average, so both kernels
    get tasks: some clusters of an image hold > 32 tokens), 50 and 70 segments (one and two segment chunks), an empty
    cluster, heavy cancellation (segments covering tokens spread around the centre): both forms against the fp64 oracle, and
    against each other far inside that tolerance."""
    D, K, N, P = 128, 16, 16 * 24, 48
    C = synth().make_vocab(K, D, seed=491)
    rng = np.random.Generator(np.random.PCG64(492))
    toks, incs, adjs = [], [], []
    for b, S in enumerate([50, 70, 50, 3]):
        cents = C[:K - 1] if b == 2 else C          # image 2: cluster K - 1 stays empty
        toks.append(synth().make_tokens(cents, N, seed=4950 + b, noise=0.5 if b in (1, 3) else 0.05))
        incs.append(rng.random((S, N)) < (0.3 if b != 1 else 0.05))
        adjs.append(np.eye(S, dtype=bool) | (rng.random((S, S)) < 0.08))
    mean, comps, var = synth().make_pca_model(K * D, P, seed=49)
    eng.set_vocab(C)
    eng.pca_set(mean, comps, var, whiten=True)
    offs = np.concatenate([[0], np.cumsum([i.shape[0] for i in incs])]).astype(np.int32)
    bits = np.concatenate([O().pack_bits_u64(i) for i in incs]).view(np.int64)
    adj = cat_adj(adjs)
    tk = np.stack(toks)
    lab = eng.seg_vlad(tk, bits, offs, adj, want_labels=True)["labels"].cpu().numpy()
    cnt = np.stack([np.bincount(lab[b], minlength=K) for b in range(4)])
    assert (cnt > 32).any() and (cnt <= 32).sum() > cnt.size // 2 and (cnt[2] == 0).any()
    ref = np.concatenate([O().seg_vlad(toks[b], incs[b], C, adjs[b]) for b in range(4)])
    ref = O().pca_transform(ref, mean, comps, var, True)
    ys = {}
    eng.set_option("pca_path", "project")
    try:
        for gram in ("1", "0"):
            eng.set_option("tnk_gram", gram)
            ys[gram] = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=False)["out"].cpu().numpy()
            assert np.abs(ys[gram] - ref).max() <= 3e-5 * np.abs(ref).max(), gram
    finally:
        eng.set_option("tnk_gram", "1")
        eng.set_option("pca_path", "auto")
    assert np.abs(ys["1"] - ys["0"]).max() <= 2e-6 * np.abs(ref).max()
    assert not np.array_equal(ys["1"], ys["0"])   # (the Gram kernel did run)
    # ... and the P-space aggregation's tile sums on the 16-bit pipe (0 / 1 coverage x a two-term split of the projected
    # residuals, the weights applied to the tile sums) against the fp32-MFMA form (pj_f16 = 0)
    eng.set_option("pca_path", "project")
    try:
        eng.set_option("pj_f16", "0")
        y32 = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=False)["out"].cpu().numpy()
    finally:
        eng.set_option("pj_f16", "1")
        eng.set_option("pca_path", "auto")
    assert np.abs(y32 - ref).max() <= 3e-5 * np.abs(ref).max()
    assert np.abs(y32 - ys["1"]).max() <= 2e-6 * np.abs(ref).max()
    assert not np.array_equal(y32, ys["1"])


def test_gram_block_norms_with_cancelling_segments_at_d1536(eng):
    """||sum_t m_t r_t||^2 as the quadratic form m^T G m carries an error relative to the covered DIAGONAL of the
    Gram matrix, not to its own value -- harmless while a segment's residuals add up like a random walk, not when they nearly
    cancel.  Here they do, by construction: the tokens of a cluster come in antipodal pairs around the centre (x = cos(t) u +-
    sin(t) v with ||C_k|| = cos(t): residuals +- sin(t) v, the second partner tilted by 10 %), so a segment that covers whole
    pairs has ||sum||^2 = 0.5 % of the diagonal sum, 32 tokens per task.  The Gram kernel flags such tasks and the fp32
    block-sum kernel recomputes them: against the fp64 oracle at D = 1536 at the tolerance of the benign case, and the two
    forms agree as closely as there.  (Measured with the guard off: the Gram form's errors add up like a random walk over the
    n^2 entries -- ~n 2^-22 of the diagonal, 2.5e-5 of the norm at this 0.5 % ratio -- so even here the unguarded result
    stays inside the tolerance; the guard keeps it there for ratios a test does not construct.)"""
    D, K, P, pairs = 1536, 8, 64, 16
    rng = np.random.Generator(np.random.PCG64(881))
    C = synth().make_vocab(K, D, seed=880)
    toks = np.empty((2 * pairs * K, D), np.float32)
    for k in range(K):
        nc = np.linalg.norm(C[k])
        u = C[k] / nc
        sin_t = np.sqrt(max(1.0 - nc * nc, 0.0))
        for j in range(pairs):
            v = rng.standard_normal(D)
            v -= (v @ u) * u
            v /= np.linalg.norm(v)
            w = rng.standard_normal(D)
            w -= (w @ u) * u
            v2 = v + 0.1 * w / np.linalg.norm(w)
            v2 -= (v2 @ u) * u
            v2 /= np.linalg.norm(v2)
            toks[(k * pairs + j) * 2] = nc * u + sin_t * v
            toks[(k * pairs + j) * 2 + 1] = nc * u - sin_t * v2
    N = toks.shape[0]
    tok_dn = np.ascontiguousarray(toks.T)                              # [D, N] as stored
    S = 12
    inc = np.zeros((S, N), bool)
    for s in range(S):                                                 # whole pairs only: every covered task cancels
        pick = rng.random(pairs * K) < (0.9 if s < 4 else 0.4)
        inc[s] = np.repeat(pick, 2)
    inc[5, 1] = not inc[5, 1]                                          # ... and one segment that splits a pair (no cancellation there)
    adj = np.eye(S, dtype=bool)
    mean, comps, var = synth().make_pca_model(K * D, P, seed=882)
    eng.set_vocab(C)
    eng.pca_set(mean, comps, var, whiten=True)
    offs = np.array([0, S], np.int32)
    bits = O().pack_bits_u64(inc).view(np.int64)
    desc, aux = O().seg_vlad(tok_dn, inc, C, adj, return_aux=True)
    bn = aux["block_norms"]
    lab = eng.seg_vlad(tok_dn[None], bits, offs, cat_adj([adj]), want_labels=True)["labels"].cpu().numpy()[0]
    assert np.array_equal(lab, np.repeat(np.arange(K), 2 * pairs))    # every token sits with its centre
    diag = np.array([[((np.linalg.norm(toks[t] - C[k]) ** 2) if inc[s, t] and lab[t] == k else 0.0) for t in range(N)] for s in range(S) for k in range(K)]).sum(1).reshape(S, K)
    ratio = (bn ** 2) / np.maximum(diag, 1e-30)
    assert (ratio[:4] < 0.02).all() and ratio[5].max() > 0.02         # the construction cancels where it should
    ref = O().pca_transform(desc, mean, comps, var, True)
    eng.set_option("pca_path", "project")
    try:
        ys = {}
        for gram in ("1", "0"):
            eng.set_option("tnk_gram", gram)
            ys[gram] = eng.seg_vlad_pca(tok_dn[None], bits, offs, cat_adj([adj]), l2norm=False)["out"].cpu().numpy()
            assert np.abs(ys[gram] - ref).max() <= 3e-5 * np.abs(ref).max(), (gram, np.abs(ys[gram] - ref).max() / np.abs(ref).max())
        assert np.abs(ys["1"] - ys["0"]).max() <= 2e-6 * np.abs(ref).max()
    finally:
        eng.set_option("tnk_gram", "1")
        eng.set_option("pca_path", "auto")


def test_project_form_with_large_norm_centres_keeps_its_fp16_scales_in_range(eng):
    """the 16-bit P-space sums scale the projected residuals by a power of two from the bound
    |z| <= sqrt(D) max|W| (1 + max ||C_k||) -- valid because the kernels normalise every token themselves and set_vocab keeps
    max ||C_k|| current.  Centres far from the unit sphere (norms up to ~40: nothing k-means on unit tokens would produce, but
    segvlad_set_vocab takes any) must move the scales, not overflow them: finite results, the oracle's tolerance."""
    D, K, N, P, S = 128, 16, 16 * 20, 48, 24
    rng = np.random.Generator(np.random.PCG64(771))
    C = (synth().make_vocab(K, D, seed=770) * rng.uniform(5.0, 40.0, size=(K, 1))).astype(np.float32)
    Cn = C / np.linalg.norm(C, axis=1, keepdims=True)
    tok = synth().make_tokens(Cn.astype(np.float32), N, seed=772, noise=0.3)
    inc = rng.random((S, N)) < 0.3
    adj = np.eye(S, dtype=bool) | (rng.random((S, S)) < 0.1)
    mean, comps, var = synth().make_pca_model(K * D, P, seed=773)
    eng.set_vocab(C)
    eng.pca_set(mean, comps, var, whiten=True)
    offs = np.array([0, S], np.int32)
    bits = O().pack_bits_u64(inc).view(np.int64)
    ref = O().pca_transform(O().seg_vlad(tok, inc, C, adj), mean, comps, var, True)
    eng.set_option("pca_path", "project")
    try:
        for pj in ("1", "0"):
            eng.set_option("pj_f16", pj)
            y = eng.seg_vlad_pca(tok[None], bits, offs, cat_adj([adj]), l2norm=False)["out"].cpu().numpy()
            assert np.isfinite(y).all()
            assert np.abs(y - ref).max() <= 5e-5 * np.abs(ref).max(), (pj, np.abs(y - ref).max() / np.abs(ref).max())
    finally:
        eng.set_option("pj_f16", "1")
        eng.set_option("pca_path", "auto")


# ------------------------------------------------------------------------------------------------
# fused segment-VLAD -> PCA (segvlad_images_pca): the aggregation kernel emits the projection GEMM's fp16 planes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P", [64, 40])   # 64: fused split-GEMM path; 40: shapes it does not take -> unfused inside the call
def test_images_pca_fused_equals_two_calls_and_oracle(eng, P):
    D, K, N = 128, 32, 15 * 20
    C = synth().make_vocab(K, D, seed=91)
    rng = np.random.Generator(np.random.PCG64(92))
    toks, incs, adjs = [], [], []
    for b, S in enumerate([5, 0, 33, 70]):
        toks.append(synth().make_tokens(C[:7] if b == 2 else C, N, seed=950 + b, noise=0.3))   # b==2: empty clusters
        inc = rng.random((S, N)) < 0.15
        if S > 2:
            inc[1] = False                      # token-less segment: its descriptor row is 0, its projection -mean.W
        incs.append(inc)
        adjs.append(np.eye(S, dtype=bool) | (rng.random((S, S)) < 0.05))
    mean, comps, var = synth().make_pca_model(K * D, P, seed=9)
    eng.set_vocab(C)
    eng.pca_set(mean, comps, var, whiten=True)
    offs = np.concatenate([[0], np.cumsum([i.shape[0] for i in incs])]).astype(np.int32)
    bits = np.concatenate([O().pack_bits_u64(i) for i in incs]).view(np.int64)
    adj = cat_adj(adjs)
    tk = np.stack(toks)
    for l2 in (True, False):
        fused = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=l2, want_desc=True, want_labels=True)
        desc = eng.seg_vlad(tk, bits, offs, adj, want_labels=True)
        two = eng.pca_apply(desc["out"], l2norm=l2).cpu().numpy()
        y = fused["out"].cpu().numpy()
        assert np.array_equal(fused["desc"].cpu().numpy(), desc["out"].cpu().numpy())
        assert np.array_equal(fused["labels"].cpu().numpy(), desc["labels"].cpu().numpy())
        assert np.abs(y - two).max() <= 2e-5 * np.abs(two).max()
        ref = np.concatenate([O().seg_vlad(toks[b], incs[b], C, adjs[b]) for b in range(4) if incs[b].shape[0]])
        ref = O().pca_transform(ref, mean, comps, var, True)
        if l2:
            ref = O().normalize_feat(ref)
        assert np.abs(y - ref).max() <= 3e-5 * np.abs(ref).max()
        # without the descriptor output: four small images hold too few tokens per cluster for the "project" form to pay
        # (auto keeps the descriptor planes: bit-identical with or without the descriptor output) ...
        y2 = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=l2)["out"].cpu().numpy()
        assert np.array_equal(y, y2)
        eng.set_option("pca_path", "planes")
        try:
            y3 = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=l2)["out"].cpu().numpy()
        finally:
            eng.set_option("pca_path", "auto")
        assert np.array_equal(y, y3)
        # ... and the "project then aggregate" form on request (tokens' residuals projected with their cluster's slice of
        # the components, segments aggregated in the P-d space: ragged S incl. 0 and > 64, empty clusters, a token-less
        # segment) -- the same fp32-class result in a different summation order
        eng.set_option("pca_path", "project")
        try:
            y4 = eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=l2)["out"].cpu().numpy()
        finally:
            eng.set_option("pca_path", "auto")
        assert np.abs(y4 - ref).max() <= 3e-5 * np.abs(ref).max()
        assert np.abs(y4 - y).max() <= 1e-5 * np.abs(ref).max()
