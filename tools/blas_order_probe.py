"""How r03 identified the operation order of the BLAS / LAPACK routines behind the reference's Kalman filters (oracle/src/lapack_order.h,
tracklab_amd/csrc/tlk_strongsort_common.hpp).  For every routine the reference calls on its small matrices, candidate evaluation orders are
written with EXACT fused multiply-adds (fractions) and compared bit-wise with the library on random inputs; the order that never mismatches is the
one restated.  Run anywhere numpy + scipy are installed:  python tools/blas_order_probe.py
On the image this repo was built in (numpy 2.2.6 / OpenBLAS 0.3.29, scipy 1.15.3 / OpenBLAS 0.3.28, SkylakeX kernels) every line prints 0 mismatches."""
from fractions import Fraction as Fr

import numpy as np
import scipy.linalg

rng = np.random.default_rng(0)


def fma(a, b, c):
    return float(Fr(a) * Fr(b) + Fr(c))


def chain(a, b):                                    # dgemm element: fma chain from 0
    acc = 0.0
    for x, y in zip(a, b):
        acc = fma(x, y, acc)
    return acc


def potf2(S):                                       # OpenBLAS potf2 (lower): fma-chain dots, column scaled by the reciprocal
    n = S.shape[0]
    a = S.copy()
    for j in range(n):
        d = np.sqrt(a[j, j] - chain(a[j, :j], a[j, :j]))
        a[j, j] = d
        r = 1.0 / d
        for i in range(j + 1, n):
            a[i, j] = (a[i, j] - chain(a[i, :j], a[j, :j])) * r
    return np.tril(a)


def trsm_fwd(L, x):                                 # dtrsm: column-oriented, reciprocal diagonal, fma updates
    n = len(x)
    x = x.copy()
    for k in range(n):
        x[k] = x[k] * (1.0 / L[k, k])
        for i in range(k + 1, n):
            x[i] = fma(-L[i, k], x[k], x[i])
    return x


def trsm_bwd(L, x):
    n = len(x)
    x = x.copy()
    for k in range(n - 1, -1, -1):
        x[k] = x[k] * (1.0 / L[k, k])
        for i in range(k):
            x[i] = fma(-L[k, i], x[k], x[i])
    return x


def trsv_fwd(L, x):                                 # dtrsv (transposed): dot form, division
    x = x.copy()
    for i in range(len(x)):
        x[i] = (x[i] - chain(L[i, :i], x[:i])) / L[i, i]
    return x


def spd(n):
    A = rng.standard_normal((n, n + 3))
    d = np.array([30., 30., 1e-2, 60.])[:n]
    return (A @ A.T + np.diag(rng.uniform(0.1, 2, n))) * d[:, None] * d[None, :]


def main():
    bad = 0
    for M, K, N in [(8, 8, 8), (8, 4, 8), (4, 4, 8), (8, 8, 4)]:
        for tA in (False, True):
            for tB in (False, True):
                for _ in range(20):
                    A = rng.standard_normal((K, M)).T if tA else rng.standard_normal((M, K))
                    B = rng.standard_normal((N, K)).T if tB else rng.standard_normal((K, N))
                    bad += not np.array_equal(np.dot(A, B), np.array([[chain(A[i], B[:, j]) for j in range(N)] for i in range(M)]))
    print("np.dot 2-D x 2-D (dgemm) = fma chain over k from 0: mismatches", bad)
    bad = 0
    for n in (4, 2):
        for _ in range(150):
            S = spd(n)
            bad += not np.array_equal(np.tril(scipy.linalg.cho_factor(S, lower=True, check_finite=False)[0]), potf2(S))
            bad += not np.array_equal(np.linalg.cholesky(S), potf2(S))
    print("scipy cho_factor(lower) and np.linalg.cholesky = potf2 order: mismatches", bad)
    bad = 0
    upd = np.eye(4, 8)
    for _ in range(150):
        S = spd(4)
        c, low = scipy.linalg.cho_factor(S, lower=True, check_finite=False)
        P = rng.standard_normal((8, 8))
        X = scipy.linalg.cho_solve((c, low), np.dot(P, upd.T).T, check_finite=False)          # as kalman_filter.py:176-178
        L = np.tril(c)
        mine = np.stack([trsm_bwd(L, trsm_fwd(L, P[r, :4].copy())) for r in range(8)], axis=1)
        bad += not np.array_equal(X, mine)
        inn = rng.standard_normal(4)
        kg = X.T
        y = np.dot(inn, kg.T)                                                                   # :181
        p = inn[None, :] * kg
        bad += not np.array_equal(y, (p[:, 0] + p[:, 2]) + (p[:, 1] + p[:, 3]))
    print("cho_solve (dpotrs, 8 right-hand sides) = column trsm order; np.dot(innovation, gain.T) = (p0+p2)+(p1+p3): mismatches", bad)
    bad = 0
    for n in (4, 2):
        for N in (1, 2, 3, 8, 110):
            for _ in range(30):
                L = np.linalg.cholesky(spd(n))
                d = rng.standard_normal((N, n))
                z = scipy.linalg.solve_triangular(L, d.T, lower=True, check_finite=False)
                f = trsv_fwd if N == 1 else trsm_fwd
                bad += not np.array_equal(z, np.stack([f(L, d[m].copy()) for m in range(N)], axis=1))
    print("solve_triangular: ONE right-hand side = trsv order, >= 2 = trsm order: mismatches", bad)
    bad = 0
    for _ in range(200):
        A = rng.standard_normal((8, 8)); x = rng.standard_normal(8)
        lanes = [[fma(A[i, j + 4], x[j + 4], A[i, j] * x[j]) for j in range(4)] for i in range(8)]
        bad += not np.array_equal(A.dot(x), np.array([(l[0] + l[2]) + (l[1] + l[3]) for l in lanes]))
        th = rng.normal(0, 0.01)
        M = [[np.cos(th), -np.sin(th), rng.normal(0, 5)], [np.sin(th), np.cos(th), rng.normal(0, 5)], [0, 0, 1]]
        v = np.array([rng.uniform(0, 1900), rng.uniform(0, 1000), 1])
        y = M @ v.T                                                                             # strong_sort/sort/track.py:232-233
        bad += not all(y[i] == fma(M[i][0], v[0], M[i][1] * v[1]) + M[i][2] * v[2] for i in range(2))
    print("8 x 8 dgemv = 4 lanes x 2 fma chunks, (l0+l2)+(l1+l3); 3 x 3 @ vec3 = fma(m0, x, m1*y) + m2*1: mismatches", bad)


if __name__ == "__main__":
    main()
