"""The last filter level over the COMPLEMENT of the stride-16 sample (option level_carry, round 6b): the stride-16 level has
already evaluated every 16-th database row with the last level's arithmetic, so its survivors under the next threshold stay in
the candidate lists (select mode 2) and the last level's GEMM covers the other 15/16 of the rows only (knn_f16_filter_kernel
SKIP = 16).  Results must be the ones of the plain scheme (every level from empty lists, the last one over every row) and of the
all-fp32 filter, bit for bit; the statistics say which form ran.  Run with `-m gpu` on an MI355X."""
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


def _rows(n, d, seed):
    import torch

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    # clumps of 4 near-duplicates (a reference place seen in neighbouring frames), as the bench's database has them
    base = torch.nn.functional.normalize(torch.randn((n + 3) // 4, 1, d, device=dev, generator=g), dim=2)
    R = torch.nn.functional.normalize(base + (0.05 / d ** 0.5) * torch.randn((n + 3) // 4, 4, d, device=dev, generator=g), dim=2)
    return R.reshape(-1, d)[:n].contiguous(), g


def _queries(R, g, nq, noise):
    """A third near sampled rows (multiples of 16), a third near rows of the complement, a third un-planted."""
    import torch

    n, d = R.shape
    third = nq // 3
    on = 16 * torch.randint(0, (n + 15) // 16, (third,), device=R.device, generator=g)
    off = torch.randint(0, n, (third,), device=R.device, generator=g)
    off = torch.where(off % 16 == 0, (off + 1).clamp(max=n - 1), off)
    q1 = R[on] + (noise / d ** 0.5) * torch.randn(third, d, device=R.device, generator=g)
    q2 = R[off] + (noise / d ** 0.5) * torch.randn(third, d, device=R.device, generator=g)
    q3 = torch.randn(nq - 2 * third, d, device=R.device, generator=g)
    return torch.nn.functional.normalize(torch.cat([q1, q2, q3]), dim=1).contiguous(), on, off


@pytest.mark.parametrize("n", [300000, 300001, 300015, 262144 + 16 * 7 + 2])
def test_complement_level_equals_plain_levels_and_fp32_filter(eng, n):
    import torch

    d, nq, k = 256, 600, 50
    R, g = _rows(n, d, 9100 + n % 97)
    Q, on, off = _queries(R, g, nq, 1.0)
    eng.db_reset()
    eng.db_add(R)
    try:
        eng.set_option("search_stats", 1)
        eng.set_option("level_carry", 1)
        d2c, idc = eng.search(Q, k)
        stc = eng.search_stats()
        eng.set_option("level_carry", 0)
        d2p, idp = eng.search(Q, k)
        stp = eng.search_stats()
        eng.set_option("knn_filter", "fp32")
        d2f, idf = eng.search(Q, k)
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("level_carry", 1)
        eng.set_option("search_stats", 0)
    assert stc["filter"] == "f16" and stc["levels"] >= 2 and stc["carry_rows"] == (n + 15) // 16, stc
    assert stp["carry_rows"] == 0 and stp["levels"] == stc["levels"], stp
    assert torch.equal(idc, idp) and torch.equal(d2c, d2p)
    assert torch.equal(idc, idf) and torch.equal(d2c, d2f)
    # the planted rows are found whichever side of the split they live on: the clump of 4 near-duplicates (rows 4 c .. 4 c + 3, one
    # of them a sampled row) fills the first four ranks
    third = nq // 3
    assert (idc[:third, :4] // 4 == (on // 4)[:, None]).float().mean() > 0.99
    assert (idc[third:2 * third, :4] // 4 == (off // 4)[:, None]).float().mean() > 0.99
    # the two forms hand the SAME candidate sets to the selection whenever the carried threshold is the level's own (min(A, t_in)
    # = A): the list totals agree to within the rows whose threshold was capped
    assert abs(stc["cand_sum"] - stp["cand_sum"]) <= 0.02 * stp["cand_sum"] + 64, (stc, stp)
    assert stc["n_redo"] <= 2 and stc["n_fallback"] == 0, stc


def test_complement_level_two_chunks_and_a_rigorous_redo(eng):
    """More query rows than one chunk (16 384): the carry is per chunk; and a batch whose guessed thresholds fail for some rows
    (far more near-duplicates than the rank model expects: clumps of 40) is redone with the rigorous thresholds -- without the carry,
    which guessed thresholds alone allow -- and still equals the plain scheme."""
    import torch

    d, n = 64, 280000
    R, g = _rows(n, d, 9200)
    Q, _, _ = _queries(R, g, 16384 + 300, 2.0)
    eng.db_reset()
    eng.db_add(R)
    try:
        eng.set_option("level_carry", 1)
        d2c, idc = eng.search(Q, 20)
        st = eng.search_stats()
        eng.set_option("level_carry", 0)
        d2p, idp = eng.search(Q, 20)
    finally:
        eng.set_option("level_carry", 1)
    assert st["carry_rows"] == (n + 15) // 16, st
    assert torch.equal(idc, idp) and torch.equal(d2c, d2p)
    # clumps of 40 near-duplicates: the k-th neighbour of a planted query sits INSIDE its clump, the sample sees 2-3 of its rows
    dev = R.device
    base = torch.nn.functional.normalize(torch.randn(n // 40, 1, d, device=dev, generator=g), dim=2)
    R2 = torch.nn.functional.normalize(base + (0.02 / d ** 0.5) * torch.randn(n // 40, 40, d, device=dev, generator=g), dim=2).reshape(-1, d).contiguous()
    src = torch.randint(0, R2.shape[0], (400,), device=dev, generator=g)
    Q2 = torch.nn.functional.normalize(R2[src] + (0.02 / d ** 0.5) * torch.randn(400, d, device=dev, generator=g), dim=1).contiguous()
    eng.db_reset()
    eng.db_add(R2)
    try:
        d2c, idc = eng.search(Q2, 30)
        st = eng.search_stats()
        eng.set_option("level_carry", 0)
        d2p, idp = eng.search(Q2, 30)
        eng.set_option("knn_filter", "fp32")
        d2f, idf = eng.search(Q2, 30)
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("level_carry", 1)
    print("clumps of 40:", st)
    assert torch.equal(idc, idp) and torch.equal(d2c, d2p) and torch.equal(idc, idf) and torch.equal(d2c, d2f)


def test_complement_level_with_carried_lists_beyond_the_wave_select(eng):
    """5000 sampled rows (multiples of 16) and 1000 rows of the complement are near-copies of one vector: the stride-16 level's lists of
    the queries near it hold > 4096 entries -- the workgroup form of the carrying select (select_approx_kernel, mode 2: in-place
    compaction in chunks) -- the last level appends the complement's copies behind them, the band outgrows the first refinement tier
    (second tier from the candidate list).  Same (d2, ids) as the plain levels and as the fp32 filter."""
    import torch

    d, n, k = 256, 300000, 50
    R, g = _rows(n, d, 9300)
    dev = R.device
    v0 = torch.nn.functional.normalize(torch.randn(1, d, device=dev, generator=g), dim=1)
    on = 16 * torch.randperm(n // 16, device=dev, generator=g)[:5000]
    off = 16 * torch.randperm(n // 16, device=dev, generator=g)[:1000] + 3
    R[on] = torch.nn.functional.normalize(v0 + (1e-3 / d ** 0.5) * torch.randn(5000, d, device=dev, generator=g), dim=1)
    R[off] = torch.nn.functional.normalize(v0 + (1e-3 / d ** 0.5) * torch.randn(1000, d, device=dev, generator=g), dim=1)
    Q, _, _ = _queries(R, g, 300, 1.0)
    Q[:12] = torch.nn.functional.normalize(v0 + (1e-3 / d ** 0.5) * torch.randn(12, d, device=dev, generator=g), dim=1)
    Q = Q.contiguous()
    eng.db_reset()
    eng.db_add(R)
    try:
        eng.set_option("search_stats", 1)
        d2c, idc = eng.search(Q, k)
        stc = eng.search_stats()
        eng.set_option("level_carry", 0)
        d2p, idp = eng.search(Q, k)
        eng.set_option("knn_filter", "fp32")
        d2f, idf = eng.search(Q, k)
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("level_carry", 1)
        eng.set_option("search_stats", 0)
    print("long carried lists:", stc)
    assert stc["carry_rows"] == (n + 15) // 16 and 5900 <= stc["cand_max"] <= 8192 and stc["n_fallback"] == 0 and stc["n_refine2"] >= 12, stc
    assert torch.equal(idc, idp) and torch.equal(d2c, d2p) and torch.equal(idc, idf) and torch.equal(d2c, d2f)
    near = set(on.tolist()) | set(off.tolist())
    assert all(int(i) in near for i in idc[:12].reshape(-1).tolist())
