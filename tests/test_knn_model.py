"""NumPy model of the mixed-precision leveled search: real fp16 operand rounding, fp32 accumulation,
the margins and threshold rules of csrc/api.hip -- and the claim that the refined result equals the exact fp32 top-k.
This checks the MATH on the CPU (the kernels themselves are checked bit for bit in tests/test_gpu_parity.py)."""
import numpy as np
import pytest


def pow2_scale(maxabs):
    if not (maxabs > 0 and np.isfinite(maxabs)):
        return 1.0
    m, e = np.frexp(np.float32(maxabs))
    return float(np.ldexp(1.0, 14 - int(e)))


def exact_d2(Q, R):
    """fp32 distances as the matrix path forms them: fma(-2, q.r, q2 + r2) with an fp32 dot product."""
    q2 = (Q * Q).sum(1, dtype=np.float32)
    r2 = (R * R).sum(1, dtype=np.float32)
    dot = (Q @ R.T).astype(np.float32)
    return (q2[:, None] + r2[None, :]) - np.float32(2) * dot, q2, r2


def approx_d2(Q, R, q2, r2):
    sq, sr = pow2_scale(np.abs(Q).max()), pow2_scale(np.abs(R).max())
    Qh = (Q * np.float32(sq)).astype(np.float16).astype(np.float32)
    Rh = (R * np.float32(sr)).astype(np.float16).astype(np.float32)
    dot = (Qh @ Rh.T).astype(np.float32) * np.float32(1.0 / (sq * sr))
    return (q2[:, None] + r2[None, :]) - np.float32(2) * dot


def leveled_search(Q, R, k, ratio=16):
    """One sampled exact level + one fp16 filter level over all rows + exact refinement (the two-level plan adds an
    approximate level in between: same rule with eps_mult = 2)."""
    n, d = R.shape
    D, q2, r2 = exact_d2(Q, R)
    Dt = approx_d2(Q, R, q2, r2)
    c_eps = 2.5 * (2.0 ** -10 + 2.0 ** -22 + 2.0 * d * 2.0 ** -24)
    eps = c_eps * np.sqrt(q2 * r2.max())
    out_d, out_i, n_ref = [], [], []
    for qi in range(Q.shape[0]):
        sample = np.arange(0, n, ratio * ratio)
        T0 = np.sort(D[qi, sample])[k - 1]                        # exact k-th of the coarsest sample
        mid = np.arange(0, n, ratio)                              # approximate level (stride 16)
        c1 = mid[Dt[qi, mid] <= T0 + eps[qi]]
        A1 = np.sort(Dt[qi, c1])[k - 1]
        c2 = np.nonzero(Dt[qi] <= A1 + 2 * eps[qi])[0]            # last level: every row
        A2 = np.sort(Dt[qi, c2])[k - 1]
        ref = c2[Dt[qi, c2] <= A2 + 2 * eps[qi]]                  # refine list
        order = np.lexsort((ref, D[qi, ref]))[:k]
        out_d.append(D[qi, ref][order])
        out_i.append(ref[order])
        n_ref.append(len(ref))
    return np.array(out_d), np.array(out_i), np.array(n_ref), D


def brute(D, k):
    idx = np.array([np.lexsort((np.arange(D.shape[1]), D[i]))[:k] for i in range(D.shape[0])])
    return np.take_along_axis(D, idx, 1), idx


@pytest.mark.parametrize("case", ["unit_random", "near_duplicates", "mixed_norms"])
def test_leveled_fp16_search_model_is_exact(case):
    rng = np.random.Generator(np.random.PCG64(77))
    n, d, nq, k = 12000, 64, 24, 20
    R = rng.standard_normal((n, d)).astype(np.float32)
    if case == "unit_random":
        R /= np.linalg.norm(R, axis=1, keepdims=True)
        Q = R[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d)).astype(np.float32) / np.sqrt(d)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    elif case == "near_duplicates":                      # many rows within the fp16 margin of the k-th distance
        R /= np.linalg.norm(R, axis=1, keepdims=True)
        base = R[:40].copy()
        R[1000:1000 + 40 * 50] = (base[:, None, :] + 2e-4 * rng.standard_normal((40, 50, d))).reshape(-1, d)
        R = R.astype(np.float32)
        Q = (base[:nq] + 1e-4 * rng.standard_normal((nq, d))).astype(np.float32)
    else:                                                # rows of different norms (the margin uses max ||r||)
        R *= rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32)
        Q = (R[rng.integers(0, n, nq)] * 0.9 + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    d_l, i_l, n_ref, D = leveled_search(Q, R, k)
    d_b, i_b = brute(D, k)
    assert np.array_equal(d_l, d_b)
    assert np.array_equal(i_l, i_b)
    assert n_ref.min() >= k                              # the refine list always holds at least the answer


def test_margin_bounds_the_observed_fp16_error():
    """|d2~ - d2| <= 2 delta ||q|| ||r|| with delta = 2^-10 + 2^-22 + 2 d 2^-24, on data that stresses the rounding."""
    rng = np.random.Generator(np.random.PCG64(78))
    for d in (64, 1024):
        R = (rng.standard_normal((2000, d)) * rng.uniform(0.5, 2.0, (2000, 1))).astype(np.float32)
        Q = (rng.standard_normal((50, d))).astype(np.float32)
        D, q2, r2 = exact_d2(Q, R)
        Dt = approx_d2(Q, R, q2, r2)
        delta = 2.0 ** -10 + 2.0 ** -22 + 2.0 * d * 2.0 ** -24
        bound = 2 * delta * np.sqrt(q2[:, None] * r2[None, :])
        assert (np.abs(Dt - D) <= bound).all()
        assert np.abs(Dt - D).max() > 0                   # the approximation is really approximate


def test_two_term_fp16_split_projection_is_fp32_class():
    """The PCA projection's arithmetic (gemm_f16x3_kernels.hip): x*s = h1 + h2 (two fp16 terms), products
    h1.g1 + h1.g2 + h2.g1 accumulated in fp32.  Its error against float64 must be of the order of a plain fp32
    GEMM's, not of fp16's (2^-11)."""
    rng = np.random.Generator(np.random.PCG64(79))
    n, kd, p = 64, 8192, 48
    X = rng.standard_normal((n, kd)).astype(np.float32)
    X[:, ::5] *= 1e-3                                            # wide dynamic range inside a row
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    mean = (0.01 * rng.standard_normal(kd)).astype(np.float32)
    W = (rng.standard_normal((p, kd)) / np.sqrt(kd)).astype(np.float32)

    def split(a, scale):
        f = a * np.float32(scale)
        h1 = f.astype(np.float16).astype(np.float32)
        h2 = (f - h1).astype(np.float16).astype(np.float32)
        return h1, h2

    sx = pow2_scale(np.abs(X).max() + np.abs(mean).max())
    sw = pow2_scale(np.abs(W).max())
    a1, a2 = split(X - mean, sx)
    b1, b2 = split(W, sw)
    y3 = ((a1 @ b1.T).astype(np.float32) + (a1 @ b2.T).astype(np.float32) + (a2 @ b1.T).astype(np.float32)) / np.float32(sx * sw)
    y32 = ((X - mean) @ W.T).astype(np.float32)
    ref = (X.astype(np.float64) - mean.astype(np.float64)) @ W.astype(np.float64).T
    scale = np.abs(ref).max()
    e3, e32 = np.abs(y3 - ref).max() / scale, np.abs(y32 - ref).max() / scale
    e16 = np.abs((a1 @ b1.T) / (sx * sw) - ref).max() / scale       # single fp16 product, for contrast
    assert e3 < 2e-6 and e3 < 8 * e32 + 1e-7, (e3, e32)
    assert e16 > 20 * e3                                             # one product alone is fp16-class


def heur_rank_small(target, ratio):
    """csrc/api.hip: heur_rank_small -- the smallest r >= 2 with P[Gamma(r, 1) < target / ratio] < 2e-5."""
    import math

    x = target / ratio
    ex, term, acc = math.exp(-x), 1.0, 0.0
    for r in range(1, target):
        acc += term
        term *= x / r
        if r >= 2 and 1.0 - ex * acc < 2e-5:
            return r
    return target


def test_single_image_plan_low_rank_threshold_rule():
    """One filter level behind a 1/stride exact sample (the single-image plan): the threshold is the r-th smallest sample
    distance; the full set then holds ~ stride * Gamma(r) rows below it.  The rule must (a) leave fewer than k rows below the
    threshold only very rarely (that query is redone, never wrong), (b) keep the expected list far below the 8192-entry
    lists.  Checked against the order-statistics model AND by simulation on uniform 'distances' (any continuous distribution
    gives the same counts: they depend on ranks only)."""
    from scipy import stats

    for k, stride in ((200, 256), (200, 64), (200, 32), (50, 256), (20, 512)):
        r = heur_rank_small(k, stride)
        assert 2 <= r <= k
        # (a) model: the fraction below the r-th of an s-row sample is Beta(r, s - r + 1) ~ Gamma(r) / s
        assert stats.gamma.cdf(k / stride, r) < 2e-5
        assert r == 2 or stats.gamma.cdf(k / stride, r - 1) >= 2e-5          # and r is the smallest such rank
        # (b) expected list length ~ stride * r rows (+ the margin's band, which the GPU tests measure)
        assert stride * r < 4096
    # simulation at the bench's shape: 1 M rows, stride 256, k = 200
    rng = np.random.Generator(np.random.PCG64(11))
    n, stride, k = 1_000_000, 256, 200
    r = heur_rank_small(k, stride)
    assert r == 7
    short, counts = 0, []
    for _ in range(300):
        x = rng.random(n, dtype=np.float32)
        t = np.partition(x[::stride], r - 1)[r - 1]
        c = int((x <= t).sum())
        counts.append(c)
        short += c < k
    assert short == 0                                     # expected 300 * 2e-5
    assert 1200 < np.mean(counts) < 2400 and max(counts) < 8192     # ~ 256 * 7 = 1792
