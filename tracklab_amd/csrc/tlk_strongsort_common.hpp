// tlk_strongsort_common.hpp -- pieces shared by the StrongSORT-family tracker kernels (tlk_bpbss.hip, tlk_ssort.hip):
// xyah Kalman update / gating in registers, the LDS work area of the association workgroup and min_cost_matching.
#pragma once
#include "tlk_common.hpp"
#include "tlk_pyset.hpp"

namespace {
using namespace tlk;

constexpr double INFTY_COST = 1e+5;                    // sort/linear_assignment.py:8
__device__ __constant__ double CHI2INV95[10] = {0, 3.8415, 5.9915, 7.8147, 9.4877, 11.070, 12.592, 14.067, 15.507, 16.919};
constexpr double W_POS = 1. / 20, W_VEL = 1. / 160;   // sort/kalman_filter.py:50-51

// The small dense linear algebra below follows the OPERATION ORDER of the libraries the reference calls (scipy.linalg.cho_factor / cho_solve,
// np.linalg.cholesky, scipy.linalg.solve_triangular, np.dot, np.linalg.multi_dot; kalman_filter.py:121-227 of both StrongSORT plugins), identified
// by differential testing and restated in oracle/src/lapack_order.h, so that Kalman states are BIT-identical to the reference's (r03; before:
// <= 1e-9, which in crowded scenes could re-order the Hungarian solver's choice among equal clamped costs):
//   dpotrf (OpenBLAS potf2, lower): d_j = sqrt(a_jj - fma-chain dot), column below = (a_ij - fma-chain dot) * (1 / d_j)
//   dtrsm  (cho_solve; solve_triangular with >= 2 right-hand sides): column-oriented, x_k *= 1 / l_kk, x_i = fma(-l_ik, x_k, x_i)
//   dtrsv  (solve_triangular with ONE right-hand side): x_i = (b_i - fma-chain dot) / l_ii
//   dgemm: fma chain over k from 0; dgemv of the 4-vector innovation: (p0 + p2) + (p1 + p3)
__device__ __forceinline__ void chol4(const double (&a)[16], int n, double (&L)[16])   // lower Cholesky of the leading n x n (stride n), lower triangle read
{
#pragma unroll
    for (int q = 0; q < 16; ++q) L[q] = 0.0;
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

__device__ __forceinline__ void chol4_full(const double (&a)[16], double (&L)[16])
{
#pragma unroll
    for (int q = 0; q < 16; ++q) L[q] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < j) acc = fma(L[j * 4 + k], L[j * 4 + k], acc);
        const double d = sqrt(a[j * 4 + j] - acc), r = 1.0 / d;
        L[j * 4 + j] = d;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i > j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < j) t = fma(L[i * 4 + k], L[j * 4 + k], t);
            L[i * 4 + j] = (a[i * 4 + j] - t) * r;
        }
    }
}

// update with measurement-noise standard deviations sd[4] (kalman_filter.py project + update)
__device__ __forceinline__ void kf8_update_sd(double (&mean)[8], double (&cov)[64], const double *z, const double (&sd)[4])
{
    double pm[4], S[16], L[16], Kg[32], B[32], inv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pm[i] = mean[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j] + (i == j ? sd[i] * sd[i] : 0.0);
    }
    chol4_full(S, L);
#pragma unroll
    for (int k = 0; k < 4; ++k) inv[k] = 1.0 / L[k * 4 + k];
#pragma unroll
    for (int c = 0; c < 8; ++c) {                      // cho_solve: row c of the gain
        double x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = cov[c * 8 + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = x[k] * inv[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i > k) x[i] = fma(-L[i * 4 + k], x[k], x[i]);
        }
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            x[k] = x[k] * inv[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i < k) x[i] = fma(-L[k * 4 + i], x[k], x[i]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Kg[c * 4 + j] = x[j];
    }
    double inn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inn[j] = z[j] - pm[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double p0 = inn[0] * Kg[i * 4], p1 = inn[1] * Kg[i * 4 + 1], p2 = inn[2] * Kg[i * 4 + 2], p3 = inn[3] * Kg[i * 4 + 3];
        mean[i] = mean[i] + ((p0 + p2) + (p1 + p3));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sacc = fma(S[j * 4 + k], Kg[c * 4 + k], sacc);
            B[j * 8 + c] = sacc;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double sacc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc = fma(Kg[i * 4 + j], B[j * 8 + c], sacc);
            cov[i * 8 + c] = cov[i * 8 + c] - sacc;
        }
}

__device__ __forceinline__ void kf8_update(double (&mean)[8], double (&cov)[64], const double *z, double conf)   // bpbreid kalman_filter.py:154-187
{
    const double sstd = (1 - conf) * (W_POS * mean[3]);
    const double sd[4] = {sstd, sstd, sstd, sstd};
    kf8_update_sd(mean, cov, z, sd);
}

// Gate row of a track: projected mean (4) + lower Cholesky factor of the projected covariance (d x d, stride d).  solve_triangular's operation
// order depends on the number of measurements of the call (above): with >= 2 the row holds the RECIPROCAL of the diagonal (`gate_row_finish`),
// with exactly one the diagonal itself.
__device__ __forceinline__ void gate_row_finish(double *g, int d, bool single)
{
    if (!single) for (int k = 0; k < d; ++k) g[4 + k * d + k] = 1.0 / g[4 + k * d + k];
}
// squared Mahalanobis distance of measurement m (xyah) to the track whose gate row was prepared (:189-227)
__device__ __forceinline__ double gating_from(const double *glrow, const double *m, int d, bool single)
{
    double zz[4], acc = 0;
    for (int i = 0; i < d; ++i) zz[i] = m[i] - glrow[i];
    if (single) {
        for (int i = 0; i < d; ++i) {
            double t = 0.0;
            for (int k = 0; k < i; ++k) t = fma(glrow[4 + i * d + k], zz[k], t);
            zz[i] = (zz[i] - t) / glrow[4 + i * d + i];
        }
    } else {
        for (int k = 0; k < d; ++k) {
            zz[k] = zz[k] * glrow[4 + k * d + k];
            for (int i = k + 1; i < d; ++i) zz[i] = fma(-glrow[4 + i * d + k], zz[k], zz[i]);
        }
    }
    for (int i = 0; i < d; ++i) acc += zz[i] * zz[i];
    return acc;
}

// the same with the track's gate row in registers (fully unrolled: no dynamic indexing, same operation order)
template <int DIM>
__device__ __forceinline__ double gating_reg(const double (&g)[20], const double *m, bool single)
{
    double zz[DIM], acc = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) zz[i] = m[i] - g[i];
    if (single) {
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < DIM; ++k) if (k < i) t = fma(g[4 + i * DIM + k], zz[k], t);
            zz[i] = (zz[i] - t) / g[4 + i * DIM + i];
        }
    } else {
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            zz[k] = zz[k] * g[4 + k * DIM + k];
#pragma unroll
            for (int i = 0; i < DIM; ++i) if (i > k) zz[i] = fma(-g[4 + i * DIM + k], zz[k], zz[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < DIM; ++i) acc += zz[i] * zz[i];
    return acc;
}

__device__ __forceinline__ double iou_ltwh(const double *b, const double *c)     // sort/iou_matching.py:7-39
{
    const double bbr0 = b[0] + b[2], bbr1 = b[1] + b[3], cbr0 = c[0] + c[2], cbr1 = c[1] + c[3];
    const double tl0 = b[0] > c[0] ? b[0] : c[0], tl1 = b[1] > c[1] ? b[1] : c[1];
    const double br0 = bbr0 < cbr0 ? bbr0 : cbr0, br1 = bbr1 < cbr1 ? bbr1 : cbr1;
    double w = br0 - tl0, h = br1 - tl1;
    w = w > 0. ? w : 0.; h = h > 0. ? h : 0.;
    const double ai = w * h;
    return ai / (b[2] * b[3] + c[2] * c[3] - ai);
}

struct BTrk {
    double *fd; int *fi; size_t stride;
    __device__ double &d(int f) const { return fd[(size_t)f * stride]; }
    __device__ int &i(int f) const { return fi[(size_t)f * stride]; }
};
// ------------------------------------------------------------------------------------------------ LDS carve
struct BLds {
    double *dxyah, *dltwh;        // MAXD*4 each (filtered detection order)
    double *d_mdist;              // MAXD
    LsaWork W;
    int *sel;                     // MAXD filtered -> input index
    int *cand, *bc;               // MAXT
    int *um_ta, *um_tb, *um_t;    // MAXT, MAXT, 2*MAXT
    int *um_da, *um_db;           // MAXD each
    int *mi_r, *mi_c;             // MAXX
    int *m_t, *m_d;               // 2*MAXX each (stage a then b)
    int *rowf, *colf, *rej;       // MAXT, MAXD, MAXX
    int *d_mname;                 // MAXD
    int *tmp;                     // MAXT
    int *scan, *sc;
    double *cost;
};

__host__ __device__ inline size_t blds_fixed(int MAXT, int MAXD)
{
    const size_t MAXX = MAXT > MAXD ? MAXT : MAXD;
    size_t b = sizeof(double) * ((size_t)MAXD * 9 + MAXX * 3);
    b += sizeof(int) * (MAXX * 4) + MAXX * 2;
    b = (b + 15) & ~(size_t)15;
    b += sizeof(int) * ((size_t)MAXD * 5 + (size_t)MAXT * 8 + MAXX * 7 + NWAVES + 32);
    return (b + 31) & ~(size_t)15;
}

__device__ inline void bcarve(unsigned char *smem, int MAXT, int MAXD, BLds &L)
{
    const int MAXX = MAXT > MAXD ? MAXT : MAXD;
    double *d = (double *)smem;
    L.dxyah = d; d += (size_t)MAXD * 4; L.dltwh = d; d += (size_t)MAXD * 4; L.d_mdist = d; d += MAXD;
    L.W.u = d; d += MAXX; L.W.v = d; d += MAXX; L.W.spc = d; d += MAXX;
    int *ip = (int *)d;
    L.W.path = ip; ip += MAXX; L.W.row4col = ip; ip += MAXX; L.W.remaining = ip; ip += MAXX; L.W.col4row = ip; ip += MAXX;
    unsigned char *bp = (unsigned char *)ip;
    L.W.SR = bp; bp += MAXX; L.W.SC = bp; bp += MAXX;
    bp = (unsigned char *)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
    ip = (int *)bp;
    L.sel = ip; ip += MAXD; L.um_da = ip; ip += MAXD; L.um_db = ip; ip += MAXD; L.colf = ip; ip += MAXD; L.d_mname = ip; ip += MAXD;
    L.cand = ip; ip += MAXT; L.bc = ip; ip += MAXT; L.um_ta = ip; ip += MAXT; L.um_tb = ip; ip += MAXT; L.um_t = ip; ip += 2 * MAXT;
    L.rowf = ip; ip += MAXT; L.tmp = ip; ip += MAXT;
    L.mi_r = ip; ip += MAXX; L.mi_c = ip; ip += MAXX; L.m_t = ip; ip += 2 * MAXX; L.m_d = ip; ip += 2 * MAXX; L.rej = ip; ip += MAXX;
    L.scan = ip; ip += NWAVES; L.sc = ip; ip += 32;
    bp = (unsigned char *)(((uintptr_t)ip + 15) & ~(uintptr_t)15);
    L.cost = (double *)bp;
}

// sort/linear_assignment.py:11-73 on a (nt x nd) cost already thresholded in `th` (row-major, ld = nd).
// trk_idx / det_idx map rows / columns to track positions / filtered detection indices.
// Appends matches at m_t/m_d[nm0..], writes unmatched lists. All 256 threads call; returns via LDS scalars.
struct McmOut { int nm, n_um_t, n_um_d, err; };      // err: the solver hit a loop bound (LSA_EINTERNAL): the caller poisons the stream
__device__ McmOut min_cost_matching(const double *th, int nt, int nd, double max_distance, const int *trk_idx, const int *det_idx,
                                    int *m_t, int *m_d, int *um_t, int *um_d, BLds &L)
{
    McmOut o{0, 0, 0, 0};
    const int tid = threadIdx.x;
    if (nd == 0 || nt == 0) {
        for (int i = tid; i < nt; i += BLOCK) um_t[i] = trk_idx[i];
        for (int j = tid; j < nd; j += BLOCK) um_d[j] = det_idx[j];
        o.n_um_t = nt; o.n_um_d = nd;
        __syncthreads();
        return o;
    }
    __syncthreads();
    if (tid < WAVE) {
        const int r = wave_lsa(th, nt, nd, (size_t)nd, (size_t)1, L.W, L.mi_r, L.mi_c);
        if (tid == 0) L.sc[0] = r == LSA_EINTERNAL ? r : (r < 0 ? 0 : r);
    }
    __syncthreads();
    int np = L.sc[0];
    if (np < 0) { o.err = 1; np = 0; }
    for (int k = tid; k < nt; k += BLOCK) L.rowf[k] = 0;
    for (int k = tid; k < nd; k += BLOCK) L.colf[k] = 0;
    __syncthreads();
    for (int k = tid; k < np; k += BLOCK) {
        L.rowf[L.mi_r[k]] = 1; L.colf[L.mi_c[k]] = 1;
        L.rej[k] = th[(size_t)L.mi_r[k] * nd + L.mi_c[k]] > max_distance ? 1 : 0;
    }
    __syncthreads();
    o.n_um_d = block_compact(nd, [&](int c) { return L.colf[c] == 0; }, [&](int c, int pos) { um_d[pos] = det_idx[c]; }, L.scan);
    o.n_um_t = block_compact(nt, [&](int r) { return L.rowf[r] == 0; }, [&](int r, int pos) { um_t[pos] = trk_idx[r]; }, L.scan);
    const int nrej = block_compact(np, [&](int k) { return L.rej[k] == 1; },
                                   [&](int k, int pos) { um_t[o.n_um_t + pos] = trk_idx[L.mi_r[k]]; um_d[o.n_um_d + pos] = det_idx[L.mi_c[k]]; }, L.scan);
    o.nm = block_compact(np, [&](int k) { return L.rej[k] == 0; },
                         [&](int k, int pos) { m_t[pos] = trk_idx[L.mi_r[k]]; m_d[pos] = det_idx[L.mi_c[k]]; }, L.scan);
    o.n_um_t += nrej; o.n_um_d += nrej;
    __syncthreads();
    return o;
}

// matching_cascade (sort/linear_assignment.py:126-128 in both StrongSORT plugins): unmatched_tracks = list(set(track_indices) -
// set(k for k, _ in matches)) in the order CPython iterates the result set (tlk_pyset.hpp). cand[0..nc) = track_indices (ascending
// track positions), L.rowf[position] = 1 for matched positions, nmatched = len(matches). Writes the list to out, returns its length.
// ws = 4 * cap ints of scratch (LDS or HBM), cap >= pyset::table_capacity(MAXT). All BLOCK threads call.
__device__ int cascade_unmatched_tracks(const int *cand, int nc, int nmatched, int *out, int *ws, unsigned cap, BLds &L)
{
    const int nu = block_compact(nc, [&](int r) { return L.rowf[cand[r]] == 0; }, [&](int r, int pos) { out[pos] = cand[r]; }, L.scan);
    __syncthreads();
    if (threadIdx.x == 0 && nc > 0 && !pyset::ascending_is_exact(nc, cand[nc - 1], nmatched, out, nu))
        pyset::difference_order_serial(cand, nc, L.rowf, nmatched, out, ws, cap, nu);
    __syncthreads();
    return nu;
}

}  // namespace
