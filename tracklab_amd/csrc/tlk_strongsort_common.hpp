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

__device__ __forceinline__ void chol4(const double (&a)[16], int n, double (&L)[16])   // lower Cholesky of the leading n x n (stride n)
{
#pragma unroll
    for (int q = 0; q < 16; ++q) L[q] = 0.0;
    for (int j = 0; j < n; ++j) {
        double sacc = a[j * n + j];
        for (int k = 0; k < j; ++k) sacc -= L[j * n + k] * L[j * n + k];
        const double d = sqrt(sacc);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = a[i * n + j];
            for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / d;
        }
    }
}

__device__ __forceinline__ void chol4_full(const double (&a)[16], double (&L)[16])
{
#pragma unroll
    for (int q = 0; q < 16; ++q) L[q] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double sacc = a[j * 4 + j];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < j) sacc -= L[j * 4 + k] * L[j * 4 + k];
        const double d = sqrt(sacc);
        L[j * 4 + j] = d;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i > j) {
            double v = a[i * 4 + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < j) v -= L[i * 4 + k] * L[j * 4 + k];
            L[i * 4 + j] = v / d;
        }
    }
}

// update with measurement-noise standard deviations sd[4] (kalman_filter.py project + update)
__device__ __forceinline__ void kf8_update_sd(double (&mean)[8], double (&cov)[64], const double *z, const double (&sd)[4])
{
    double pm[4], S[16], L[16], X[32], Kg[32], B[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pm[i] = mean[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j] + (i == j ? sd[i] * sd[i] : 0.0);
    }
    chol4_full(S, L);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = cov[c * 8 + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < i) v -= L[i * 4 + k] * y[k];
            y[i] = v / L[i * 4 + i];
        }
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            double v = y[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k > i) v -= L[k * 4 + i] * X[k * 8 + c];
            X[i * 8 + c] = v / L[i * 4 + i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Kg[i * 4 + j] = X[j * 8 + i];
    double inn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inn[j] = z[j] - pm[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double sacc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc += inn[j] * Kg[i * 4 + j];
        mean[i] = mean[i] + sacc;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sacc += S[j * 4 + k] * Kg[c * 4 + k];
            B[j * 8 + c] = sacc;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double sacc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc += Kg[i * 4 + j] * B[j * 8 + c];
            cov[i * 8 + c] = cov[i * 8 + c] - sacc;
        }
}

__device__ __forceinline__ void kf8_update(double (&mean)[8], double (&cov)[64], const double *z, double conf)   // bpbreid kalman_filter.py:154-187
{
    const double sstd = (1 - conf) * (W_POS * mean[3]);
    const double sd[4] = {sstd, sstd, sstd, sstd};
    kf8_update_sd(mean, cov, z, sd);
}

// squared Mahalanobis distance of measurement m (xyah) to the track whose (pm, L) were prepared (:189-227)
__device__ __forceinline__ double gating_from(const double *glrow, const double *m, int d)
{
    double zz[4], acc = 0;
    for (int i = 0; i < d; ++i) {
        double v = m[i] - glrow[i];
        for (int k = 0; k < i; ++k) v -= glrow[4 + i * d + k] * zz[k];
        zz[i] = v / glrow[4 + i * d + i];
    }
    for (int i = 0; i < d; ++i) acc += zz[i] * zz[i];
    return acc;
}

// the same with the track's gate row in registers (fully unrolled: no dynamic indexing, same operation order)
template <int DIM>
__device__ __forceinline__ double gating_reg(const double (&g)[20], const double *m)
{
    double zz[DIM], acc = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
        double v = m[i] - g[i];
#pragma unroll
        for (int k = 0; k < DIM; ++k) if (k < i) v -= g[4 + i * DIM + k] * zz[k];
        zz[i] = v / g[4 + i * DIM + i];
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
struct McmOut { int nm, n_um_t, n_um_d; };
__device__ McmOut min_cost_matching(const double *th, int nt, int nd, double max_distance, const int *trk_idx, const int *det_idx,
                                    int *m_t, int *m_d, int *um_t, int *um_d, BLds &L)
{
    McmOut o{0, 0, 0};
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
        if (tid == 0) L.sc[0] = r < 0 ? 0 : r;
    }
    __syncthreads();
    const int np = L.sc[0];
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
