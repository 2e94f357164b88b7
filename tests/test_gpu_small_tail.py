"""The device-driven tail of a single-image pass (round 6, csrc/small_pass_kernels.hip): segvlad_search on <= 128 query rows ends
in small_tail_kernel -- which reads the overflow counters ON THE DEVICE and finishes flagged rows there -- instead of a read-back
and a host synchronisation.  Every path of the kernel must give the bits of the read-back path (option small_tail = 0, rounds
3-5), which the other suites hold to the batch search, the fp32 filter and the fp64 oracle:
  * nothing flagged (the common case): the kernel returns at once;
  * every row forced through the exact brute force (debug_small_tail bit 0), through the second tier (bit 1), and a raised
    hand-over failure word (bit 2);
  * real overflows: a query with 600 duplicates (band > first tier) and one with 9000 (candidate list overflow);
  * twenty passes enqueued back to back without a synchronisation in between."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine():
    from revisit_anything_amd.engine import SegVLADEngine

    return SegVLADEngine(0)


def _problem(eng, n, d, nq, seed, noise=0.05):
    g = torch.Generator(device=eng.device)
    g.manual_seed(seed)
    R = torch.nn.functional.normalize(torch.randn(n, d, device=eng.device, generator=g), dim=1)
    src = torch.randint(0, n, (nq,), device=eng.device, generator=g)
    Q = torch.nn.functional.normalize(R[src] + noise * torch.randn(nq, d, device=eng.device, generator=g), dim=1)
    eng.db_reset()
    eng.db_add(R)
    return R, Q


@pytest.mark.parametrize("form", [1, 2])          # 1: flagged rows finished by the refinement kernel itself where it can; 2: always small_tail_kernel
@pytest.mark.parametrize("n,d,nq,k", [(100_000, 256, 40, 50), (70_000, 1024, 50, 200), (50_000, 64, 1, 20), (60_000, 128, 128, 300),
                                      (90_000, 1024, 128, 1000)])
def test_every_path_of_the_tail_equals_the_read_back_path(n, d, nq, k, form):
    eng = _engine()
    R, Q = _problem(eng, n, d, nq, seed=n + d)
    eng.set_option("small_tail", 0)
    ref = eng.search(Q, k)
    st0 = eng.search_stats()
    assert st0["levels"] == 1 and st0["filter"] == "f16" and st0["n_redo"] == 0 and st0["n_fallback"] == 0, st0
    eng.set_option("small_tail", form)
    got = eng.search(Q, k)
    st = eng.search_stats()
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    # (k = 1000: every band outgrows the 512-entry first tier -- all rows take the second tier on both paths)
    assert st["n_redo"] == 0 and st["n_refine2"] == st0["n_refine2"] == (nq if k > 512 else 0), (st, st0)
    # bit 3 (value 8): the fused form's own answer to a failed hand-over -- the last workgroup re-evaluates the row's band itself (query
    # row 1; never taken by the hardware so far) -- changes no bit and leaves the sticky word for the next pass's head to repair
    for bits, key in ((1, "n_redo"), (2, "n_refine2"), (4, "n_redo"), (3, "n_redo"), (8, "n_redo")):
        eng.db_reset()                 # (a fresh index: rows redone in the previous round must not switch the guessed thresholds off)
        eng.db_add(R)
        eng.set_option("debug_small_tail", bits)
        got = eng.search(Q, k)
        st = eng.search_stats()
        eng.set_option("debug_small_tail", 0)
        assert torch.equal(got[1], ref[1]), (bits, (got[1] != ref[1]).sum().item())
        assert torch.equal(got[0], ref[0]), bits
        # (bit 2 raises the word of refine_exact_small_kernel's hand-over, which only rows of d % 1024 == 0 go through)
        if k <= 512:
            assert st[key] == ((1 if d % 1024 == 0 else 0) if bits == 4 else 0 if bits == 8 else nq), (bits, st)
        again = eng.search(Q, k)       # tickets back at zero, hand-over buffers repaired: the next pass is clean
        assert torch.equal(again[0], ref[0]) and torch.equal(again[1], ref[1])
        assert eng.search_stats()["n_redo"] == 0
    eng.close()


@pytest.mark.parametrize("d", [256, 1024])     # (d % 1024 == 0: the shared-list refinement, whose fused form finishes the flagged rows itself)
def test_real_overflows_are_finished_on_the_device(d):
    """Query 7's 600 duplicates outgrow the first-tier refine list (-> second tier), query 23's 9000 overflow the candidate list
    (-> exact brute force); everything equals the read-back path and the oracle's clamp-aware top k."""
    from oracle import segvlad_oracle as O

    eng = _engine()
    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(13)
    n, nq, k = 120_000, 40, 50
    R = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
    star = torch.nn.functional.normalize(torch.randn(2, d, device=dev, generator=g), dim=1)
    dup_a = torch.arange(0, 600, device=dev) * 191 + 17
    dup_b = torch.arange(0, 9000, device=dev) * 13 + 5
    dup_b = dup_b[~torch.isin(dup_b, dup_a)]
    R[dup_a] = star[0]
    R[dup_b] = star[1]
    src = torch.randint(0, n, (nq,), device=dev, generator=g)
    Q = torch.nn.functional.normalize(R[src] + 0.05 * torch.randn(nq, d, device=dev, generator=g), dim=1)
    Q[7] = star[0]
    Q[23] = star[1]
    eng.db_add(R)
    eng.set_option("small_tail", 0)
    ref = eng.search(Q, k)
    st0 = eng.search_stats()
    assert st0["n_refine2"] >= 1 and st0["n_fallback"] + st0["n_redo"] >= 1, st0
    for form in (1, 2):
        eng.set_option("small_tail", form)
        got = eng.search(Q, k)
        st = eng.search_stats()
        assert st["n_refine2"] >= 1 and st["n_redo"] >= 1 and st["n_fallback"] == 0, (form, st)
        assert torch.equal(got[1], ref[1]) and torch.equal(got[0], ref[0]), form
    ii = got[1].cpu().numpy()
    assert np.array_equal(ii[7], np.sort(dup_a.cpu().numpy())[:k]) and np.array_equal(ii[23], np.sort(dup_b.cpu().numpy())[:k])
    rd2, ridx = O.topk_from_d2(O.l2_matrix(R.cpu().numpy(), Q.cpu().numpy()), k)
    assert np.abs(got[0].cpu().numpy() - rd2).max() < 1e-5
    eng.close()


def test_back_to_back_passes_without_a_synchronisation():
    eng = _engine()
    R, Q = _problem(eng, 90_000, 256, 50, seed=5)
    Qs = [torch.nn.functional.normalize(Q + 0.01 * j * torch.roll(Q, j, 0), dim=1) for j in range(20)]
    eng.set_option("small_tail", 0)
    refs = [eng.search(q, 100) for q in Qs]
    eng.set_option("small_tail", 1)
    torch.cuda.synchronize()
    outs = [eng.search(q, 100) for q in Qs]      # nothing in these calls waits for the device
    torch.cuda.synchronize()
    for (d2, idx), (rd2, ridx) in zip(outs, refs):
        assert torch.equal(idx, ridx) and torch.equal(d2, rd2)
    eng.close()


def test_a_database_that_keeps_failing_is_switched_to_the_rigorous_plan():
    """The host never reads a device-driven pass's counters -- but their running total reaches it through a pinned word: more than
    a quarter of >= 64 rows redone since the index changed -> no more guessing (until the index changes again)."""
    eng = _engine()
    R, Q = _problem(eng, 80_000, 128, 40, seed=9)
    ref = eng.search(Q, 30)
    eng.set_option("debug_small_tail", 1)        # every row of every pass is "redone"
    seen = []
    for _ in range(6):
        got = eng.search(Q, 30)
        seen.append(eng.search_stats()["n_redo"])          # (fetching the statistics synchronises: the pinned word has landed)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert seen[0] == 40 and seen[-1] == 0, seen           # the later passes no longer take the guessed thresholds
    eng.db_reset()
    eng.db_add(R)
    eng.set_option("debug_small_tail", 0)
    got = eng.search(Q, 30)
    assert eng.search_stats()["levels"] == 1
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    eng.close()
