"""tests/fp32_emu.py (the host emulation of the device's fp32 distance arithmetic, used by tests/test_gpu_exact_ids.py) against exact
rational arithmetic: the fma must round ONCE -- including the cases where rounding the exact sum to float64 first and to float32
afterwards would give the other neighbour."""
from fractions import Fraction

import numpy as np

import fp32_emu as E


def _round32(fr: Fraction) -> np.float32:
    """nearest fp32 to an exact rational, ties to even -- by comparing the exact distances to the candidates"""
    r = np.float32(float(fr))
    cands = {float(np.nextafter(r, np.float32(-np.inf))), float(r), float(np.nextafter(r, np.float32(np.inf)))}
    best = None
    for c in sorted(cands):
        dist = abs(Fraction(c) - fr)
        even = (np.float32(c).view(np.uint32) & 1) == 0
        key = (dist, 0 if even else 1)
        if best is None or key < best[0]:
            best = (key, c)
    return np.float32(best[1])


def test_fma32_rounds_once():
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.standard_normal(4000).astype(np.float32)
    b = rng.standard_normal(4000).astype(np.float32)
    c = rng.standard_normal(4000).astype(np.float32)
    # cancellation: c ~ -a b; and sums parked next to fp32 rounding boundaries: c = a big value, a b = half an ulp of it +- a crumb
    c[:1000] = (-(a[:1000].astype(np.float64) * b[:1000].astype(np.float64))).astype(np.float32)
    big = (1.0 + rng.integers(0, 1 << 23, 1000) * 2.0 ** -23).astype(np.float32)            # in [1, 2): ulp 2^-23
    c[1000:2000] = big
    mm = rng.integers(1, 300, 1000).astype(np.float64)
    sg = rng.choice([-1.0, 1.0], 1000)
    a[1000:2000] = (2.0 ** -12 * (1 + mm * 2.0 ** -23)).astype(np.float32)          # a b = +-2^-24 (1 - m^2 2^-46): half an ulp of c, less a crumb
    b[1000:2000] = (sg * 2.0 ** -12 * (1 - mm * 2.0 ** -23)).astype(np.float32)      # far below float64's resolution of the sum
    got = E.fma32(a, b, c)
    differs = 0
    for i in range(len(a)):
        want = _round32(Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i])))
        assert got[i].view(np.uint32) == want.view(np.uint32), (i, a[i], b[i], c[i], got[i], want)
        naive = np.float32(np.float64(a[i]) * np.float64(b[i]) + np.float64(c[i]))
        differs += int(naive.view(np.uint32) != want.view(np.uint32))
    assert differs > 0          # the test does contain cases where "float64, then float32" is wrong


def test_row_sumsq_and_chain_follow_the_documented_order():
    rng = np.random.Generator(np.random.PCG64(2))
    X = rng.standard_normal((3, 1024)).astype(np.float32)
    # the same order, scalar by scalar, in exact arithmetic rounded once per operation
    for row in range(3):
        lanes = [np.float32(0)] * 64
        for j4 in range(256):
            L = j4 % 64
            for cidx in range(4):
                v = X[row, 4 * j4 + cidx]
                lanes[L] = _round32(Fraction(float(v)) * Fraction(float(v)) + Fraction(float(lanes[L])))
        for o in (32, 16, 8, 4, 2, 1):
            lanes = [np.float32(lanes[i] + lanes[i ^ o]) for i in range(64)]
        assert E.row_sumsq(X[row:row + 1])[0].view(np.uint32) == lanes[0].view(np.uint32)
    q, R = X[0], X[1:]
    for i in range(2):
        acc = np.float32(0)
        for j in range(1024):
            acc = _round32(Fraction(float(q[j])) * Fraction(float(R[i, j])) + Fraction(float(acc)))
        assert E.dot_chain(q, R)[i].view(np.uint32) == acc.view(np.uint32)
    assert E.d2(np.float32(1), np.float32(1), np.float32(1.0000001))[()] == 0.0      # the clamp
