"""kNN ids WITHOUT a near-tie allowance: at 1 M x 1024 the device's (distance, id) lists must equal, bit for bit
and ties included, the lists a host emulation of the device's documented fp32 arithmetic produces (tests/fp32_emu.py: the
sequential fma chain, the row_sumsq lane order, sv_d2 with faiss's clamp) -- on 64 queries of the three kinds (planted, random,
near-copies), for the batch search, the single-image pass and the fp32 filter.

The emulation runs on the CONTESTED rows only: per query every row whose distance (a plain fp32 GEMM on the device: a tool of the
test, not the product) lies within 1e-3 of the k-th smallest -- 100 x the worst fp32 deviation, so no row outside the set can be in
anybody's top k -- typically 200-400 rows.  The fp64 oracle stays beside it: every emulated distance within 2e-6 of it."""
import numpy as np
import pytest
import torch

import fp32_emu as E

pytestmark = pytest.mark.gpu


def _data():
    dev = torch.device("cuda:0")
    n_img, S, d, group = 20000, 50, 1024, 4
    g = torch.Generator(device=dev)
    g.manual_seed(3000)
    R = torch.empty(n_img * S, d, device=dev)
    for g0 in range(0, n_img // group, 500):
        ng = min(500, n_img // group - g0)
        base = torch.nn.functional.normalize(torch.randn(ng, 1, S, d, device=dev, generator=g), dim=3)
        blk = base + (0.05 / d ** 0.5) * torch.randn(ng, group, S, d, device=dev, generator=g)
        R[g0 * group * S:(g0 + ng) * group * S] = torch.nn.functional.normalize(blk, dim=3).reshape(-1, d)
    gq = torch.Generator(device=dev)
    gq.manual_seed(4100)
    tau = torch.randint(0, n_img * S, (40,), device=dev, generator=gq)
    q_pl = torch.nn.functional.normalize(R[tau] + (4.0 / d ** 0.5) * torch.randn(40, d, device=dev, generator=gq), dim=1)
    q_rand = torch.nn.functional.normalize(torch.randn(12, d, device=dev, generator=gq), dim=1)
    src = torch.randint(0, n_img * S, (12,), device=dev, generator=gq)
    q_dup = torch.nn.functional.normalize(R[src] + (0.02 / d ** 0.5) * torch.randn(12, d, device=dev, generator=gq), dim=1)
    Q = torch.cat([q_pl, q_rand, q_dup]).contiguous()
    R[777_777] = R[123_456]                     # an exact duplicate pair: a genuine tie that only the id can break
    Q[0] = torch.nn.functional.normalize(R[123_456] + (1.0 / d ** 0.5) * torch.randn(d, device=dev, generator=gq), dim=0)
    return R, Q


def test_device_ids_equal_the_emulated_fp32_arithmetic_bit_for_bit():
    from revisit_anything_amd.engine import SegVLADEngine

    eng = SegVLADEngine(0)
    R, Q = _data()
    k, nq = 200, Q.shape[0]
    eng.db_add(R)
    d2_b, id_b = eng.search(Q, k)                                    # batch plan (> 128 rows would be; 64 rows take the single-image plan) ...
    d2_big, id_big = eng.search(torch.cat([Q, Q, Q]).contiguous(), k)   # ... so also as part of a 192-row batch
    eng.set_option("knn_filter", "fp32")
    d2_f, id_f = eng.search(Q, k)
    eng.set_option("knn_filter", "auto")
    assert torch.equal(id_big[:nq], id_b) and torch.equal(d2_big[:nq], d2_b) and torch.equal(id_f, id_b) and torch.equal(d2_f, d2_b)
    # contested rows (test tool: plain torch GEMM, fp32)
    approx = (Q * Q).sum(1, keepdim=True) + (R * R).sum(1)[None, :] - 2.0 * (Q @ R.T)
    kth = torch.kthvalue(approx, k, dim=1).values
    Rn, Qn = None, Q.cpu().numpy()
    q2 = E.row_sumsq(Qn)
    dd, ii = d2_b.cpu().numpy(), id_b.cpu().numpy()
    n_contested, n_tie_pairs, worst64 = 0, 0, 0.0
    for q in range(nq):
        rows = torch.nonzero(approx[q] <= kth[q] + 1e-3).flatten()
        assert rows.numel() >= k
        rr = R[rows].cpu().numpy()
        ids = rows.cpu().numpy()
        r2 = E.row_sumsq(rr)
        dist = E.d2(q2[q], r2, E.dot_chain(Qn[q], rr))
        order = np.lexsort((ids, dist))[:k]                         # (distance, id): IndexFlatL2's order, ties to the lower id
        assert np.array_equal(ids[order], ii[q]), (q, np.nonzero(ids[order] != ii[q])[0][:5])
        assert np.array_equal(dist[order].view(np.uint32), dd[q].view(np.uint32)), q
        n_contested += len(ids)
        n_tie_pairs += int((np.diff(dist[order]) == 0).sum())
        d64 = ((Qn[q].astype(np.float64)[None, :] - rr[order].astype(np.float64)) ** 2).sum(1)
        worst64 = max(worst64, float(np.abs(d64 - dist[order]).max()))
    assert worst64 < 4e-6
    assert ii[0, 0] == 123_456 and ii[0, 1] == 777_777 and dd[0, 0] == dd[0, 1]      # the planted exact tie: lower id first
    print(f"[exact ids] {nq} queries x {k}: {n_contested} contested pairs emulated, {n_tie_pairs} exact-tie neighbours among the results, "
          f"max |fp32 - fp64| {worst64:.2e}")
    eng.close()
