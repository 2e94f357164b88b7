"""Grouped exact refinement (round 5, csrc/refine_group_kernels.hip): the refine bands of 32 consecutive query rows -- the
segments of a query image share their neighbours -- are evaluated as ONE exact fp32 GEMM over the union of their rows.
The chain per (query, row) pair is the per-row kernels' (and the distance-matrix path's), so every search must return
the same BITS with the option on and off, on databases where the groups share everything, nothing, or a part, with
groups cut by the end of the batch, and next to rows that take the second tier / the redo."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine():
    from revisit_anything_amd.engine import SegVLADEngine

    return SegVLADEngine(0)


def _unit(x):
    return torch.nn.functional.normalize(x, dim=1)


def _both(R, Q, k, expect_grouped=None, hint=0, expect_policy=None):
    eng = _engine()
    eng.set_option("search_stats", 1)
    eng.set_option("query_group", hint)
    eng.db_add(R)
    eng.set_option("refine_group", 0)
    d0, i0 = eng.search(Q, k)
    st0 = eng.search_stats()
    assert st0["grp_groups"] == 0
    eng.set_option("refine_group", 2)      # every group whose union fits (the default, 1, asks a cost model first)
    d1, i1 = eng.search(Q, k)
    st1 = eng.search_stats()
    eng.set_option("refine_group", 1)
    d3, i3 = eng.search(Q, k)
    st3 = eng.search_stats()
    assert torch.equal(i0, i3) and torch.equal(d0, d3), "the default policy's result differs from the per-row refinement"
    if expect_policy is not None:
        assert expect_policy[0] <= st3["grp_groups"] <= expect_policy[1], st3
    eng.set_option("knn_filter", "fp32")
    d2, i2 = eng.search(Q, k)
    eng.close()
    assert torch.equal(i0, i1) and torch.equal(d0, d1), "grouped refinement differs from the per-row refinement"
    assert torch.equal(i0, i2) and torch.equal(d0, d2), "differs from the fp32 filter"
    if expect_grouped is not None:
        lo, hi = expect_grouped
        assert lo <= st1["grp_groups"] <= hi, st1
    return st1


def _places(g, places, per, d, noise):
    base = _unit(torch.randn(places, d, device="cuda:0", generator=g))
    R = base.repeat_interleave(per, dim=0) + noise * torch.randn(places * per, d, device="cuda:0", generator=g) / d ** 0.5
    return _unit(R)


@pytest.mark.parametrize("d", [256, 1024])
def test_groups_that_share_their_neighbours(d):
    """2000 places x 32 rows; the 32 query rows of a group are noisy copies of ONE place's rows: every band is that place."""
    g = torch.Generator(device="cuda:0")
    g.manual_seed(51)
    places, per = 2000, 32
    R = _places(g, places, per, d, 0.6)
    pick = torch.randint(0, places, (20,), device="cuda:0", generator=g)
    rows = (pick[:, None] * per + torch.arange(per, device="cuda:0")[None, :]).reshape(-1)
    Q = _unit(R[rows] + 0.3 * torch.randn(rows.numel(), d, device="cuda:0", generator=g) / d ** 0.5)
    st = _both(R, Q, 20, expect_grouped=(15, 20))
    assert st["grp_union_sum"] <= st["grp_groups"] * 40 * 4, st


def test_random_queries_keep_the_per_row_kernels():
    g = torch.Generator(device="cuda:0")
    g.manual_seed(52)
    R = _unit(torch.randn(80_000, 256, device="cuda:0", generator=g))
    Q = _unit(torch.randn(700, 256, device="cuda:0", generator=g))     # 21 full groups + one of 28 rows
    _both(R, Q, 50, expect_policy=(0, 0))     # nothing shared: the cost model keeps every group row by row


def test_mixed_batch_ragged_tail_and_k_beyond_the_union():
    """Shared groups, random groups and a cut last group in one batch; k = 200 > the 32 rows of a place, so the bands reach
    into the neighbouring random rows and the unions are a few hundred rows long."""
    g = torch.Generator(device="cuda:0")
    g.manual_seed(53)
    d, places, per = 256, 3000, 32
    R = _places(g, places, per, d, 0.5)
    parts = []
    for j in range(9):
        if j % 3 == 2:
            parts.append(_unit(torch.randn(32, d, device="cuda:0", generator=g)))
        else:
            p = int(torch.randint(0, places, (1,), device="cuda:0", generator=g))
            parts.append(_unit(R[p * per:(p + 1) * per] + 0.2 * torch.randn(per, d, device="cuda:0", generator=g) / d ** 0.5))
    Q = torch.cat(parts)[:275]      # the last group holds 19 rows
    _both(R, Q, 200)
    _both(R, Q, 1)


def test_duplicates_and_second_tier_rows_beside_grouped_rows():
    """600 exact duplicates of one row: its queries' bands outgrow the first-tier list (second tier), their group's other rows
    stay grouped; exact duplicates also pin the (distance, id) tie order."""
    g = torch.Generator(device="cuda:0")
    g.manual_seed(54)
    d, places, per = 128, 2500, 32
    R = _places(g, places, per, d, 0.5)
    R[40_000:40_600] = R[40_000]
    Q = _unit(R[torch.arange(39_990, 40_310, device="cuda:0")] + 0.05 * torch.randn(320, d, device="cuda:0", generator=g) / d ** 0.5)
    st = _both(R, Q, 30)
    assert st["n_refine2"] > 0, st


@pytest.mark.parametrize("hint,d", [(50, 1024), (50, 256), (7, 256), (64, 128), (33, 1024), (50, 160)])
def test_image_sized_groups_via_the_hint(hint, d):
    """option query_group: a query image's rows (here `hint` noisy copies of one place's rows) form one group -- two 32-row
    accumulator tiles per wave beyond 32 rows -- and the last group of the batch is cut."""
    g = torch.Generator(device="cuda:0")
    g.manual_seed(60 + hint)
    places, per = 1200, 64
    R = _places(g, places, per, d, 0.6)
    n_img = max(9, 200 // hint + 1)
    pick = torch.randint(0, places, (n_img,), device="cuda:0", generator=g)
    rows = (pick[:, None] * per + torch.arange(hint, device="cuda:0")[None, :]).reshape(-1)
    Q = _unit(R[rows] + 0.3 * torch.randn(rows.numel(), d, device="cuda:0", generator=g) / d ** 0.5)
    Q = Q[:Q.shape[0] - 3]
    st = _both(R, Q, 40, hint=hint)
    assert st["grp_groups"] >= 1, st


def test_two_chunks_of_query_rows_groups_cut_by_the_chunk_boundary():
    """More than 16 384 query rows: the search walks them in two chunks (the first chunk's exact sample level is enqueued
    before the host waits for the query scalars), the refinement groups restart at the chunk boundary -- 16 384 is no
    multiple of the 50 rows of an image, so the boundary cuts an image in two -- and every row still gets its bits."""
    g = torch.Generator(device="cuda:0")
    g.manual_seed(71)
    d, places, per, hint = 128, 1300, 50, 50
    R = _places(g, places, per, d, 0.6)
    n_img = 340                                            # 17 000 query rows
    pick = torch.randint(0, places, (n_img,), device="cuda:0", generator=g)
    rows = (pick[:, None] * per + torch.arange(hint, device="cuda:0")[None, :]).reshape(-1)
    Q = _unit(R[rows] + 0.3 * torch.randn(rows.numel(), d, device="cuda:0", generator=g) / d ** 0.5)
    st = _both(R, Q, 30, hint=hint)
    assert st["n_queries"] == 17_000 and st["grp_groups"] >= 300, st
