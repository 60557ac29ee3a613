/* CPU ORACLE (test infrastructure) -- see orc.h.
 * The small dense linear algebra of the Kalman filters in the operation ORDER of the libraries the reference calls, so that the oracle's
 * states are bit-identical to the reference's (crowded scenes -- a last-bit difference of a Kalman state can re-order the
 * Hungarian solver's choice among equal clamped costs).  Call sites: bpbreid_strong_sort/sort/kalman_filter.py:121-227 and
 * strong_sort/sort/kalman_filter.py (same code): scipy.linalg.cho_factor / cho_solve, np.linalg.multi_dot, np.dot, np.linalg.cholesky,
 * scipy.linalg.solve_triangular.  The libraries are third-party (numpy 2.2.6 bundles OpenBLAS 0.3.29, scipy 1.15.3 bundles OpenBLAS 0.3.28; both
 * select their SkylakeX kernels on the machine that generated the goldens); their operation order is not documented, so each routine below was
 * identified by differential testing against the library itself with exact-FMA candidates (tools/blas_order_probe.py: 0 mismatches in 200-300
 * random cases per routine) and is pinned by tests/golden/kf8_cases.npz and the golden tracker runs, which are now reproduced BIT-EXACTLY:
 *   dgemm  (np.dot 2-D x 2-D, any transposition)     C[i][j] = fma chain over k = 0..K-1 starting from 0
 *   dgemv  (np.dot of a 4-vector with a 4 x 8 view)   y[i] = (p0 + p2) + (p1 + p3), p_j = x[j] * A[j][i] rounded separately
 *   dpotrf (n <= 4: OpenBLAS potf2, lower)            d_j = sqrt(a_jj - fma-chain dot); column below: (a_ij - fma-chain dot) * (1 / d_j)
 *   dpotrs (two dtrsm, nrhs = 8) and dtrtrs, nrhs >= 2   column-oriented substitution: x_k *= 1 / l_kk, then x_i = fma(-l_ik, x_k, x_i)
 *   dtrtrs, nrhs == 1 (OpenBLAS trsv, transposed)     x_i = (b_i - fma-chain dot(l_i0.., x_0..)) / l_ii
 *   dgesv  (np.linalg.inv, 4 x 4: getf2 + getrs)      lo_inv4 below (OC-SORT / Deep-OC-SORT: filterpy-style update)
 * Matrices are row-major. */
#ifndef ORC_LAPACK_ORDER_H
#define ORC_LAPACK_ORDER_H
#include <math.h>
#include <string.h>

/* cho_factor(a, lower=True) / np.linalg.cholesky(a): reads the lower triangle of a (n x n), writes L (n x n, zero above the diagonal) */
static inline void lo_potrf_lower(const double *a, int n, double *L)
{
    memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) L[i * n + j] = a[i * n + j];
    for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int k = 0; k < j; ++k) acc = fma(L[j * n + k], L[j * n + k], acc);
        const double d = sqrt(L[j * n + j] - acc), r = 1.0 / d;
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = 0.0;
            for (int k = 0; k < j; ++k) t = fma(L[i * n + k], L[j * n + k], t);
            L[i * n + j] = (L[i * n + j] - t) * r;
        }
    }
}
/* forward substitution L y = b in place, dtrsm order (nrhs >= 2) */
static inline void lo_trsm_lower_fwd(const double *L, int n, double *x)
{
    for (int k = 0; k < n; ++k) {
        x[k] = x[k] * (1.0 / L[k * n + k]);
        for (int i = k + 1; i < n; ++i) x[i] = fma(-L[i * n + k], x[k], x[i]);
    }
}
/* back substitution L^T x = y in place, dtrsm order */
static inline void lo_trsm_lower_bwd(const double *L, int n, double *x)
{
    for (int k = n - 1; k >= 0; --k) {
        x[k] = x[k] * (1.0 / L[k * n + k]);
        for (int i = 0; i < k; ++i) x[i] = fma(-L[k * n + i], x[k], x[i]);
    }
}
/* forward substitution with ONE right-hand side: scipy.linalg.solve_triangular on a C-ordered factor solves the transposed system with dtrtrs,
 * which OpenBLAS serves with its transposed trsv (dot form, division) */
static inline void lo_trsv_lower_fwd(const double *L, int n, double *x)
{
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int k = 0; k < i; ++k) t = fma(L[i * n + k], x[k], t);
        x[i] = (x[i] - t) / L[i * n + i];
    }
}
/* fma-chain dot product of length n with strides (what every dgemm element is) */
static inline double lo_dot(const double *a, int sa, const double *b, int sb, int n)
{
    double acc = 0.0;
    for (int k = 0; k < n; ++k) acc = fma(a[k * sa], b[k * sb], acc);
    return acc;
}
/* Kalman update kalman_filter.py:154-187 given the projected mean / covariance (pm, S) of :121-152 */
static inline void lo_kf8_update(double *mean, double *cov, const double *z, const double *pm, const double *S)
{
    double L[16], K[32], B[32], inn[4];
    lo_potrf_lower(S, 4, L);                                            /* scipy.linalg.cho_factor(projected_cov, lower=True) */
    for (int c = 0; c < 8; ++c) {                                       /* cho_solve(..., (cov H^T)^T).T : row c of the gain */
        double x[4] = {cov[c * 8], cov[c * 8 + 1], cov[c * 8 + 2], cov[c * 8 + 3]};
        lo_trsm_lower_fwd(L, 4, x);
        lo_trsm_lower_bwd(L, 4, x);
        for (int j = 0; j < 4; ++j) K[c * 4 + j] = x[j];
    }
    for (int j = 0; j < 4; ++j) inn[j] = z[j] - pm[j];
    for (int i = 0; i < 8; ++i) {                                       /* mean + np.dot(innovation, kalman_gain.T) */
        const double p0 = inn[0] * K[i * 4], p1 = inn[1] * K[i * 4 + 1], p2 = inn[2] * K[i * 4 + 2], p3 = inn[3] * K[i * 4 + 3];
        mean[i] = mean[i] + ((p0 + p2) + (p1 + p3));
    }
    /* cov - multi_dot((K, S, K^T)): equal cost both ways, numpy evaluates K (S K^T) */
    for (int j = 0; j < 4; ++j) for (int c = 0; c < 8; ++c) B[j * 8 + c] = lo_dot(S + j * 4, 1, K + c * 4, 1, 4);
    for (int i = 0; i < 8; ++i) for (int c = 0; c < 8; ++c) cov[i * 8 + c] = cov[i * 8 + c] - lo_dot(K + i * 4, 1, B + c, 8, 4);
}
/* squared Mahalanobis distances kalman_filter.py:189-227 of n measurements (rows of `meas`, stride 4) in the first d dimensions */
static inline void lo_kf8_gating(const double *pm, const double *S4, int d, const double *meas, int n, double *out)
{
    double Sd[16], L[16];
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) Sd[i * d + j] = S4[i * 4 + j];
    lo_potrf_lower(Sd, d, L);                                           /* np.linalg.cholesky */
    for (int m = 0; m < n; ++m) {
        double zz[4], acc = 0;
        for (int i = 0; i < d; ++i) zz[i] = meas[m * 4 + i] - pm[i];
        if (n == 1) lo_trsv_lower_fwd(L, d, zz); else lo_trsm_lower_fwd(L, d, zz);
        for (int i = 0; i < d; ++i) acc += zz[i] * zz[i];               /* np.sum(z * z, axis=0): row by row */
        out[m] = acc;
    }
}
/* np.dot of two 2-D arrays (dgemm): C (n x m) = A (n x k) B (k x m), every element an fma chain over k ascending from 0 */
static inline void lo_gemm(const double *A, const double *B, double *C, int n, int k, int m)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) C[i * m + j] = lo_dot(A + i * k, 1, B + j, m, k);
}
/* np.dot(A (rows x 4), y (4, 1)): dgemv, the four products rounded separately, summed (p0 + p2) + (p1 + p3) */
static inline double lo_dot4_h2(const double *a, const double *y)
{
    const double p0 = a[0] * y[0], p1 = a[1] * y[1], p2 = a[2] * y[2], p3 = a[3] * y[3];
    return (p0 + p2) + (p1 + p3);
}
/* np.linalg.inv of a 4 x 4 matrix (filterpy-style Kalman filters of OC-SORT / Deep-OC-SORT: kalmanfilter.py `self.inv = np.linalg.inv`):
 * LAPACK dgesv(S, I) = OpenBLAS getf2 (left-looking LU with partial pivoting: fma-chain dots against the finished columns, the sub-diagonal
 * scaled by the RECIPROCAL of the pivot) + dgetrs on the row-permuted identity (unit-lower forward and upper backward substitution in the
 * column-oriented trsm order, diagonal by reciprocal). tools/blas_order_probe.py: 0 mismatches in 200 random matrices. */
static inline void lo_inv4(const double *S, double *SI)
{
    enum { N = 4 };
    double a[N][N], X[N][N];
    int ipiv[N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) a[i][j] = S[i * N + j];
    for (int j = 0; j < N; ++j) {
        double b[N];
        for (int i = 0; i < N; ++i) b[i] = a[i][j];
        for (int i = 0; i < j; ++i) { const int ip = ipiv[i]; if (ip != i) { const double t = b[i]; b[i] = b[ip]; b[ip] = t; } }
        for (int i = 1; i < j; ++i) { double t = 0.0; for (int q = 0; q < i; ++q) t = fma(a[i][q], b[q], t); b[i] = b[i] - t; }
        for (int i = j; i < N; ++i) { double t = 0.0; for (int q = 0; q < j; ++q) t = fma(a[i][q], b[q], t); b[i] = b[i] - t; }
        int jp = j;
        for (int i = j + 1; i < N; ++i) if (fabs(b[i]) > fabs(b[jp])) jp = i;          /* idamax: first maximum */
        ipiv[j] = jp;
        for (int i = 0; i < N; ++i) a[i][j] = b[i];
        if (b[jp] != 0.0) {
            if (jp != j) for (int q = 0; q <= j; ++q) { const double t = a[j][q]; a[j][q] = a[jp][q]; a[jp][q] = t; }
            const double r = 1.0 / a[j][j];
            for (int i = j + 1; i < N; ++i) a[i][j] = a[i][j] * r;
        }
    }
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) X[i][j] = (i == j);
    for (int i = 0; i < N; ++i) if (ipiv[i] != i) for (int j = 0; j < N; ++j) { const double t = X[i][j]; X[i][j] = X[ipiv[i]][j]; X[ipiv[i]][j] = t; }
    for (int c = 0; c < N; ++c) {
        double x[N];
        for (int i = 0; i < N; ++i) x[i] = X[i][c];
        for (int k = 0; k < N; ++k) for (int i = k + 1; i < N; ++i) x[i] = fma(-a[i][k], x[k], x[i]);
        for (int k = N - 1; k >= 0; --k) {
            x[k] = x[k] * (1.0 / a[k][k]);
            for (int i = 0; i < k; ++i) x[i] = fma(-a[i][k], x[k], x[i]);
        }
        for (int i = 0; i < N; ++i) SI[i * N + c] = x[i];
    }
}
#endif
