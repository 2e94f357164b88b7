"""Host emulation of the DEVICE's documented fp32 arithmetic for one exact squared distance (test infrastructure; VERDICT r05
next #7): with it a kNN id comparison needs no "near-tie" allowance -- the device's (distance, id) lists must equal the emulated
ones bit for bit, ties included.

What the device computes for a (query q, database row r) pair on EVERY exact path (distance-matrix GEMM gemm_nt_kernel<1>,
refine_exact_*, refine_group_gemm_kernel, small_tail_kernel):

    dot   = the sequential chain  acc = fmaf(q[j], r[j], acc),  j = 0 .. d-1,  acc = 0                 (one rounding per step)
    ||x||^2 = row_sumsq (csrc/gemm_kernels.hip): lane L of a 64-lane wave sums the float4s L, L + 64, ... of the row with fmaf
              (x, y, z, w in turn), then the xor butterfly v += shfl_xor(v, o), o = 32, 16, 8, 4, 2, 1   (d % 4 == 0)
    d2    = fmaf(-2, dot, ||q||^2 + ||r||^2), negative values set to zero                                (csrc/ctx.h: sv_d2)

fp32 fma is emulated exactly in float64: the product of two fp32 numbers is exact in float64 (48 significant bits), the sum with
the fp32 addend is formed with an error-free TwoSum, and the ONE rounding to fp32 is decided from the pair (sum, error) -- the
double rounding of "round to float64, then to float32" differs from the true fma exactly when the float64 sum sits on an fp32
rounding boundary and the discarded error is not zero, which the pair resolves.  Everything is vectorised over pairs."""
import numpy as np


def fma32(a, b, c):
    """round_to_fp32(a * b + c) for fp32 arrays a, b, c (broadcasting), exactly."""
    a64, b64, c64 = np.asarray(a, np.float32).astype(np.float64), np.asarray(b, np.float32).astype(np.float64), np.asarray(c, np.float32).astype(np.float64)
    p = a64 * b64                          # exact (24 + 24 bits)
    s = p + c64                            # rounded to 53 bits ...
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)      # ... with the rounding error recovered (TwoSum: s + err == p + c exactly)
    r = s.astype(np.float32)               # float64 -> float32, ties to even
    d = s - r.astype(np.float64)           # exact
    # s exactly on the midpoint of two adjacent fp32 values: |d| is half an ulp of the fp32 grid there.  Then the TRUE value
    # s + err is off the midpoint by err, and the nearest fp32 neighbour lies on err's side.
    with np.errstate(over="ignore", invalid="ignore"):
        up = np.nextafter(r, np.float32(np.inf))
        dn = np.nextafter(r, np.float32(-np.inf))
        half_up = (up.astype(np.float64) - r.astype(np.float64)) * 0.5
        half_dn = (r.astype(np.float64) - dn.astype(np.float64)) * 0.5
    tie_hi = (d > 0) & (d == half_up)      # s is the midpoint between r and up; ties-to-even chose r
    tie_lo = (d < 0) & (-d == half_dn)     # s is the midpoint between dn and r
    r = np.where(tie_hi & (err > 0), up, r)
    r = np.where(tie_lo & (err < 0), dn, r)
    return r.astype(np.float32)


def row_sumsq(X):
    """||row||^2 of every row of X [n, d] fp32 (d % 4 == 0) as csrc/gemm_kernels.hip: row_sumsq computes it."""
    X = np.ascontiguousarray(X, np.float32)
    n, d = X.shape
    assert d % 4 == 0
    d4 = d // 4
    X4 = X.reshape(n, d4, 4)
    s = np.zeros((n, 64), np.float32)
    for j0 in range(0, d4, 64):                      # lane L takes float4 j0 + L
        w = min(64, d4 - j0)
        for c in range(4):
            v = X4[:, j0:j0 + w, c]
            s[:, :w] = fma32(v, v, s[:, :w])
    for o in (32, 16, 8, 4, 2, 1):                   # v += shfl_xor(v, o): every lane at once
        s = (s + s[:, np.arange(64) ^ o]).astype(np.float32)
    return s[:, 0].copy()


def dot_chain(q, R):
    """acc = fmaf(q[j], r[j], acc), j = 0 .. d-1, for every row r of R [n, d] against ONE query q [d] (or q [n, d] row-wise)."""
    R = np.ascontiguousarray(R, np.float32)
    q = np.asarray(q, np.float32)
    qq = np.broadcast_to(q, R.shape)
    acc = np.zeros(R.shape[0], np.float32)
    for j in range(R.shape[1]):
        acc = fma32(qq[:, j], R[:, j], acc)
    return acc


def d2(q2, r2, dot):
    """sv_d2: fmaf(-2, dot, q2 + r2) with negative results set to zero (faiss's clamp)."""
    v = fma32(np.float32(-2.0), dot, (np.asarray(q2, np.float32) + np.asarray(r2, np.float32)).astype(np.float32))
    return np.where(v < 0, np.float32(0), v).astype(np.float32)
