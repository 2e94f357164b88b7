"""GPU parity for the two regimes VERDICT r02 found untested (run with `-m gpu` on an MI355X):

* BASELINE configs[1] in its literal form -- `pca = False` in recall_segloc (place_rec_main.py:49-60): raw
  K*D = 64*1536 = 98 304-d segment descriptors, 1000 reference images x 50 segments, 200 query images, search 200.
  The rows come out of segvlad_images itself (real VLAD-kernel output, not a low-rank toy), the search runs the deep-row
  fp16 filter (blocked accumulation) + the coalesced exact refinement and must equal the fp64 oracle and, bit for bit,
  the fp32 filter.
* a 17places-like TEMPORALLY REDUNDANT database (gt.py:60-64: ground truth = +-15 frames, i.e. every reference segment
  has ~30 near-duplicates): 1 M x 1024 rows in groups of 31 near-duplicate "frames" of 200 scene types; exactness on a
  query subset, the list occupancies, and no query on the distance-matrix path.
* the second refinement tier and the matrix-path fallback, each forced for exactly one query.
"""
import os
import sys
import time

import numpy as np
import pytest
from conftest import engine_scope

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def _near_tie_check(dmat, dd, ii, rd2, ridx, tol):
    """ids identical wherever the oracle's neighbouring distances are further apart than `tol`; every other mismatch is a
    near-tie (the device's row at that rank is, by the oracle's own distances, within tol of the oracle's row)."""
    assert np.abs(dd - rd2).max() < tol
    clear = np.minimum(np.diff(rd2, axis=1, prepend=-1.0), np.diff(rd2, axis=1, append=10.0)) > tol
    assert np.array_equal(ii[clear], ridx[clear])
    qq, rr = np.nonzero(ii != ridx)
    if len(qq):
        assert np.abs(dmat[qq, ii[qq, rr]] - rd2[qq, rr]).max() < tol
    return float(clear.mean()), int(len(qq))


def test_config2_raw_98304d_search_equals_oracle_and_fp32_filter(eng, capsys):
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from revisit_anything_amd import synth
    from revisit_anything_amd.pipeline import SegVLADPipeline

    dev = eng.device
    K, D, S, H, W = 64, 1536, 50, 480, 640
    N, Hm, Wm = (H // 14) * (W // 14), H // 2, W // 2
    n_ref, n_q, k = 1000, 200, 200
    C_np = synth.make_vocab(K, D, seed=1000)
    eng.set_vocab(C_np)
    pipe = SegVLADPipeline(eng, H, W, 14, order=3, use_pca=False)
    fac = bench.ImageFactory(dev, torch.from_numpy(C_np).to(dev), N, S, Hm, Wm, 0.12, 4)
    rows = torch.empty(n_ref * S, K * D, device=dev)
    bb = 100
    tok = torch.empty(bb, D, N, device=dev)
    msk = torch.empty(bb * S, Hm, Wm, dtype=torch.uint8, device=dev)
    offs = (np.arange(bb + 1) * S).astype(np.int32)
    for b0 in range(0, n_ref, bb):
        for j in range(bb):
            tok[j], msk[j * S:(j + 1) * S] = fac.reference(b0 + j)
        rows[b0 * S:(b0 + bb) * S] = pipe.describe(tok, msk, offs)
    tau = np.random.Generator(np.random.PCG64(4000)).integers(0, n_ref, size=n_q)
    Q = torch.empty(n_q * S, K * D, device=dev)
    for b0 in range(0, n_q, bb):
        for j in range(bb):
            tok[j], msk[j * S:(j + 1) * S] = fac.query(int(tau[b0 + j]), b0 + j)
        Q[b0 * S:(b0 + bb) * S] = pipe.describe(tok, msk, offs)
    del tok, msk
    eng.db_reset()
    eng.db_add(rows)
    eng.set_option("search_stats", 1)
    try:
        eng.search(Q, k)                                        # warm-up (plane conversion, scratch sized for the batch)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        d2, idx = eng.search(Q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        stage = {s_: round(eng.stage_ms(s_)[0], 1) for s_ in ("knn_level0", "knn_gemm", "knn_select")}
        eng.set_profiling(False)
        st = eng.search_stats()
        # the deep-row fp16 filter ran, nobody needed the distance-matrix path, and the refine band stayed a band
        assert st["filter"] == "f16" and st["levels"] >= 1, st
        assert st["n_fallback"] == 0, st
        assert st["refine_sum"] / st["n_queries"] < 2.0 * k, st
        sel = np.arange(0, n_q * S, 13)[:512]
        Qs = Q[torch.from_numpy(sel).to(dev)].contiguous()
        eng.set_option("knn_filter", "fp32")
        d2f, idxf = eng.search(Qs, k)
        assert eng.search_stats()["filter"] == "fp32"
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("search_stats", 0)
    selt = torch.from_numpy(sel).to(dev)
    assert torch.equal(idx[selt], idxf) and torch.equal(d2[selt], d2f)      # bit-identical to the all-fp32 filter
    # fp64 oracle over ALL 50 000 rows for 48 queries (the 48 first of the 512: query segments of many images)
    osel = sel[:48]
    Rh = rows.cpu().numpy()
    Qh = Q[torch.from_numpy(osel).to(dev)].cpu().numpy()
    dmat = O().l2_matrix(Rh, Qh, rows_block=2048)
    rd2, ridx = O().topk_from_d2(dmat, k)
    dd, ii = d2.cpu().numpy()[osel], idx.cpu().numpy()[osel]
    # fp32 fma chain over 98 304 terms of unit vectors against fp64: ~sqrt(d) ulps (north_star tolerance: 1e-4)
    clear_frac, n_mis = _near_tie_check(dmat, dd, ii, rd2, ridx, 1e-4)
    # the right place: a query segment's nearest row is a segment of its own reference image's sibling group
    top_img = idx.cpu().numpy()[:, 0] // S
    q_img = np.repeat(tau, S)
    assert np.mean(top_img // 4 == q_img // 4) > 0.9
    with capsys.disabled():
        print(f"\n[config2] 10000 x 50000 x 98304 raw search: {ms:.1f} ms, stages {stage}, stats {st}, oracle ids clear {clear_frac:.3f}, "
              f"near-tie mismatches {n_mis}", file=sys.stderr)
    eng.db_reset()


def _redundant_db(dev, n_rows, d, group, n_types, sigma, seed):
    """rows = frames of places: place = normalize(type + u), row = normalize(place + sigma * noise); `group` consecutive
    'frames' (rows here: one segment per frame keeps the structure and the row order of a video: the near-duplicates of
    a row sit next to it) share a place."""
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_places = (n_rows + group - 1) // group
    types = torch.nn.functional.normalize(torch.randn(n_types, d, device=dev, generator=g), dim=1)
    t_of_p = torch.randint(0, n_types, (n_places,), device=dev, generator=g)
    R = torch.empty(n_rows, d, device=dev)
    places = torch.empty(n_places, d, device=dev)
    for p0 in range(0, n_places, 4096):
        p1 = min(n_places, p0 + 4096)
        u = torch.nn.functional.normalize(torch.randn(p1 - p0, d, device=dev, generator=g), dim=1)
        places[p0:p1] = torch.nn.functional.normalize(types[t_of_p[p0:p1]] + u, dim=1)
        r0, r1 = p0 * group, min(n_rows, p1 * group)
        pr = places[p0:p1].repeat_interleave(group, dim=0)[: r1 - r0]
        nz = torch.nn.functional.normalize(torch.randn(r1 - r0, d, device=dev, generator=g), dim=1)
        R[r0:r1] = torch.nn.functional.normalize(pr + sigma * nz, dim=1)
    return R, places


def test_redundant_db_31_near_duplicates_per_segment_1m_rows(eng, capsys):
    import torch

    dev = eng.device
    n, d, group, k, nq = 1_000_000, 1024, 31, 200, 10_000
    R, places = _redundant_db(dev, n, d, group, 200, 0.35, seed=31)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    qp = torch.randint(0, places.shape[0] - 1, (nq,), device=dev, generator=g)
    Q = torch.nn.functional.normalize(places[qp] + 0.35 * torch.nn.functional.normalize(torch.randn(nq, d, device=dev, generator=g), dim=1), dim=1)
    eng.db_reset()
    eng.db_add(R)
    eng.set_option("search_stats", 1)
    try:
        eng.search(Q[:256], k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d2, idx = eng.search(Q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        st = eng.search_stats()
        t0 = time.perf_counter()
        d2s, idxs = eng.search(Q[:50], k)                      # one query image per pass (the streaming regime)
        torch.cuda.synchronize()
        ms50 = (time.perf_counter() - t0) * 1e3
        st50 = eng.search_stats()
        sel = torch.arange(0, nq, 19, device=dev)[:512]
        eng.set_option("knn_filter", "fp32")
        d2f, idxf = eng.search(Q[sel].contiguous(), k)
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("search_stats", 0)
    assert st["filter"] == "f16" and st["n_fallback"] == 0, st       # nobody on the distance-matrix path
    assert st["n_redo"] <= nq // 100, st                                # the low-rank thresholds still verify (<= 1 % redone)
    assert torch.equal(idx[sel], idxf) and torch.equal(d2[sel], d2f)   # bit-identical to the fp32 filter
    assert torch.equal(idxs, idx[:50]) and torch.equal(d2s, d2[:50])   # and to itself at another batch size
    # every query's 31 near-duplicate frames are its 31 nearest rows
    top = idx[:, :group].cpu().numpy() // group
    assert np.mean(top == qp.cpu().numpy()[:, None]) > 0.999
    # fp64 oracle on 64 queries over all 1 M rows
    osel = sel[:64]
    dmat = O().l2_matrix(R.cpu().numpy(), Q[osel].cpu().numpy(), rows_block=100000)
    rd2, ridx = O().topk_from_d2(dmat, k)
    clear_frac, n_mis = _near_tie_check(dmat, d2[osel].cpu().numpy(), idx[osel].cpu().numpy(), rd2, ridx, 1e-5)
    with capsys.disabled():
        print(f"\n[redundant_db] 10000 x 1M x 1024, groups of {group}: {ms:.1f} ms ({nq / 50 / ms * 1e3:.0f} images/s), stats {st}; "
              f"50-query pass {ms50:.2f} ms, stats {st50}; oracle ids clear {clear_frac:.3f}, near-tie mismatches {n_mis}", file=sys.stderr)
    eng.db_reset()


def test_refine_band_overflow_takes_the_second_tier_and_list_overflow_the_matrix_path(eng):
    """300 queries against 200 k rows.  Query 137 has 600 exact duplicates of itself in the database: its refine band
    (600 rows at distance 0) exceeds the 512-entry first-tier list, so it is refined straight from its candidate list
    (second tier) -- NOT on the distance-matrix path.  Query 201 has 9000 duplicates: its candidate list itself overflows
    (cap 8192), so it -- alone -- is redone on the exact matrix path.  Everything equals the oracle (ties -> lower id)."""
    import torch

    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    n, d, nq, k = 200000, 256, 300, 50
    # the two duplicated points sit (mostly) on axes nobody else uses: every OTHER query sees them at distance^2 = 2 +- 0.03,
    # beyond its thresholds (its k-th neighbour among random unit vectors of 254 dimensions is at ~1.6), so that exactly one
    # query meets each tie; the 0.2 share of the common subspace keeps the stars' own distances to the other rows spread out
    # (an exactly orthogonal star is equidistant from all 200 000 rows: every threshold level would tie)
    R = torch.randn(n, d, device=dev, generator=g)
    R[:, d - 2:] = 0.0
    R = torch.nn.functional.normalize(R, dim=1)
    star = 0.2 * torch.nn.functional.normalize(torch.randn(2, d, device=dev, generator=g) * (torch.arange(d, device=dev) < d - 2), dim=1)
    star[0, d - 2] = 1.0
    star[1, d - 1] = 1.0
    star = torch.nn.functional.normalize(star, dim=1)
    dup_a = torch.arange(0, 600, device=dev) * 331 + 17          # 600 scattered rows
    dup_b = torch.arange(0, 9000, device=dev) * 22 + 5           # 9000 scattered rows
    dup_b = dup_b[~torch.isin(dup_b, dup_a)]                     # (the few that coincide with dup_a stay with dup_a)
    R[dup_a] = star[0]
    R[dup_b] = star[1]
    special = torch.cat([dup_a, dup_b])
    src = torch.randint(0, n - 1, (nq,), device=dev, generator=g)
    for _ in range(4):                                            # never a duplicate row: only queries 137 / 201 see the ties
        src = torch.where(torch.isin(src, special), src + 1, src)
    assert not bool(torch.isin(src, special).any())
    Q = R[src] + (1.0 / d ** 0.5) * torch.randn(nq, d, device=dev, generator=g)
    Q[:, d - 2:] = 0.0
    Q = torch.nn.functional.normalize(Q, dim=1)
    Q[137] = star[0]
    Q[201] = star[1]
    eng.db_reset()
    eng.db_add(R)
    eng.set_profiling(True)
    eng.profile_reset()
    d2, idx = eng.search(Q, k)
    st = eng.search_stats()
    ms, rows_redone = eng.stage_ms("knn_fallback")
    eng.set_profiling(False)
    assert st["levels"] >= 1 and st["filter"] == "f16"
    assert st["n_refine2"] == 1 and st["n_fallback"] == 1 and rows_redone == 1, (st, rows_redone)
    rd2, ridx = O().topk_from_d2(O().l2_matrix(R.cpu().numpy(), Q.cpu().numpy()), k)
    dd, ii = d2.cpu().numpy(), idx.cpu().numpy()
    assert np.abs(dd - rd2).max() < 1e-5
    assert np.array_equal(ii[137], np.sort(dup_a.cpu().numpy())[:k])    # 600-way tie: the k lowest ids
    assert np.array_equal(ii[201], np.sort(dup_b.cpu().numpy())[:k])    # 9000-way tie
    assert np.abs(dd[[137, 201]]).max() < 1e-6
    clear = np.minimum(np.diff(rd2, axis=1, prepend=-1.0), np.diff(rd2, axis=1, append=10.0)) > 1e-5
    clear[[137, 201]] = False
    assert np.array_equal(ii[clear], ridx[clear])
    # rigorous thresholds (no low-rank guesses): the same two rows take the same two paths, same bits
    eng.set_option("knn_heuristic", 0)
    try:
        d2r, idxr = eng.search(Q, k)
        st2 = eng.search_stats()
    finally:
        eng.set_option("knn_heuristic", 1)
    assert st2["n_refine2"] == 1 and st2["n_fallback"] == 1 and st2["n_redo"] == 0, st2
    assert torch.equal(idxr, idx) and torch.equal(d2r, d2)
    # with neither overflow nothing is charged to either path
    eng.set_profiling(True)
    eng.profile_reset()
    eng.search(Q[:100], k)
    st3 = eng.search_stats()
    assert st3["n_fallback"] == 0 and st3["n_refine2"] == 0
    with pytest.raises(Exception):
        eng.stage_ms("knn_fallback")
    eng.set_profiling(False)
    eng.db_reset()


def test_second_tier_runs_per_chunk_of_queries(eng):
    """More queries than one chunk of the search (16 384): a query of the FIRST chunk whose refine band overflows must be
    refined from its candidate list before the second chunk overwrites the lists."""
    import torch

    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    n, d, nq, k = 120000, 128, 20000, 20
    R = torch.randn(n, d, device=dev, generator=g)
    R[:, d - 1] = 0.0
    R = torch.nn.functional.normalize(R, dim=1)
    star = 0.2 * torch.nn.functional.normalize(torch.randn(1, d, device=dev, generator=g) * (torch.arange(d, device=dev) < d - 1), dim=1)
    star[0, d - 1] = 1.0
    star = torch.nn.functional.normalize(star, dim=1)
    dup = torch.arange(0, 700, device=dev) * 151 + 9
    R[dup] = star[0]
    src = torch.randint(0, n - 1, (nq,), device=dev, generator=g)
    for _ in range(4):
        src = torch.where(torch.isin(src, dup), src + 1, src)
    Q = R[src] + (1.0 / d ** 0.5) * torch.randn(nq, d, device=dev, generator=g)
    Q[:, d - 1] = 0.0
    Q = torch.nn.functional.normalize(Q, dim=1)
    Q[77] = star[0]                      # chunk 0
    Q[19000] = star[0]                   # chunk 1
    eng.db_reset()
    eng.db_add(R)
    d2, idx = eng.search(Q, k)
    st = eng.search_stats()
    assert st["n_refine2"] == 2 and st["n_fallback"] == 0, st
    want = np.sort(dup.cpu().numpy())[:k]
    assert np.array_equal(idx[77].cpu().numpy(), want) and np.array_equal(idx[19000].cpu().numpy(), want)
    assert float(d2[[77, 19000]].abs().max()) < 1e-6
    sel = torch.tensor([0, 1, 76, 78, 16383, 16384, 18999, 19999], device=dev)
    rd2, ridx = O().knn_l2(R.cpu().numpy(), Q[sel].cpu().numpy(), k)
    assert np.abs(d2[sel].cpu().numpy() - rd2).max() < 1e-5
    clear = np.minimum(np.diff(rd2, axis=1, prepend=-1.0), np.diff(rd2, axis=1, append=10.0)) > 1e-5
    assert np.array_equal(idx[sel].cpu().numpy()[clear], ridx[clear])
    eng.db_reset()


@pytest.mark.parametrize("d", [256, 1024])
def test_single_image_pass_overflow_paths_on_a_fresh_context(d):
    """The same two overflows in ONE query image's pass (40 query rows: the single-image plan -- workgroup selects, and at
    d = 1024 the shared-list refinement), on a context that has never run anything else (grow-only scratch sized by an earlier,
    larger call would hide an under-allocation): query 7's 600 duplicates take the second refinement tier, query 23's 9000
    overflow the candidate list -> redone with the rigorous thresholds -> matrix path, alone; everything equals the oracle."""
    import torch

    from revisit_anything_amd.engine import SegVLADEngine

    eng = SegVLADEngine(0)
    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(13)
    n, nq, k = 200000, 40, 50
    R = torch.randn(n, d, device=dev, generator=g)
    R[:, d - 2:] = 0.0
    R = torch.nn.functional.normalize(R, dim=1)
    star = 0.2 * torch.nn.functional.normalize(torch.randn(2, d, device=dev, generator=g) * (torch.arange(d, device=dev) < d - 2), dim=1)
    star[0, d - 2] = 1.0
    star[1, d - 1] = 1.0
    star = torch.nn.functional.normalize(star, dim=1)
    dup_a = torch.arange(0, 600, device=dev) * 331 + 17
    dup_b = torch.arange(0, 9000, device=dev) * 22 + 5
    dup_b = dup_b[~torch.isin(dup_b, dup_a)]
    R[dup_a] = star[0]
    R[dup_b] = star[1]
    special = torch.cat([dup_a, dup_b])
    src = torch.randint(0, n - 1, (nq,), device=dev, generator=g)
    for _ in range(4):
        src = torch.where(torch.isin(src, special), src + 1, src)
    Q = R[src] + (1.0 / d ** 0.5) * torch.randn(nq, d, device=dev, generator=g)
    Q[:, d - 2:] = 0.0
    Q = torch.nn.functional.normalize(Q, dim=1)
    Q[7] = star[0]
    Q[23] = star[1]
    eng.db_add(R)
    d2, idx = eng.search(Q, k)
    st = eng.search_stats()
    assert st["levels"] == 1 and st["filter"] == "f16", st
    # (round 6: the pass finishes its flagged rows on the device -- small_tail_kernel -- so the overflowing list is counted as a
    #  row redone exactly, n_redo, not as a matrix-path fallback of the host's)
    assert st["n_refine2"] >= 1 and st["n_fallback"] + st["n_redo"] == 1, st
    rd2, ridx = O().topk_from_d2(O().l2_matrix(R.cpu().numpy(), Q.cpu().numpy()), k)
    dd, ii = d2.cpu().numpy(), idx.cpu().numpy()
    assert np.abs(dd - rd2).max() < 1e-5
    assert np.array_equal(ii[7], np.sort(dup_a.cpu().numpy())[:k])
    assert np.array_equal(ii[23], np.sort(dup_b.cpu().numpy())[:k])
    clear = np.minimum(np.diff(rd2, axis=1, prepend=-1.0), np.diff(rd2, axis=1, append=10.0)) > 1e-5
    clear[[7, 23]] = False
    assert np.array_equal(ii[clear], ridx[clear])
    # a second pass on the same context (scratch now sized, tickets back at zero): same bits
    d2b, idxb = eng.search(Q, k)
    assert torch.equal(idxb, idx) and torch.equal(d2b, d2)
    eng.close()
