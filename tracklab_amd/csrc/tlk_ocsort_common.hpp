// tlk_ocsort_common.hpp -- what the OC-SORT-family trackers (OC-SORT, Deep-OC-SORT: the same update() skeleton over a different
// Kalman filter and an extra embedding cost) share: the field-major slot accessor, np.linalg.inv of a 4x4, the per-stream LDS carve
// and np.setdiff1d on an LDS list.
#pragma once
#include "tlk_common.hpp"

namespace {
using namespace tlk;

struct Trk {                     // per-thread accessor of one slot
    double *fd; int *fi; size_t stride_d, stride_i;
    __device__ double &d(int f) const { return fd[(size_t)f * stride_d]; }
    __device__ int &i(int f) const { return fi[(size_t)f * stride_i]; }
};

__device__ __forceinline__ double sum5(const double *a) { return (((a[0] + a[1]) + a[2]) + a[3]) + a[4]; }


__device__ __forceinline__ void inv4(const double (&S)[16], double (&SI)[16])   // LU, partial pivoting
{
    double a[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[i][j] = S[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int p = c;
#pragma unroll
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
#pragma unroll
        for (int r = c + 1; r < 4; ++r)
            if (p == r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[r][j]; a[r][j] = t; }
            }
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            double f = a[r][c] / a[c][c];
#pragma unroll
            for (int j = c; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
#pragma unroll
    for (int c = 3; c >= 0; --c) {
#pragma unroll
        for (int j = 4; j < 8; ++j) {
            double s = a[c][j];
#pragma unroll
            for (int t = c + 1; t < 4; ++t) s -= a[c][t] * a[t][j];
            a[c][j] = s / a[c][c];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) SI[i * 4 + j] = a[i][4 + j];
}

// ------------------------------------------------------------------ LDS carve
struct Lds {
    double *trk_box;   // MAXT*4   predicted boxes by list position
    double *kobs;      // MAXT*5
    double *velp;      // MAXT*2
    double *cost;      // cost_lds_entries (or spill pointer)
    LsaWork W;         // MAXX each
    int *hi_idx, *lo_idx;             // MAXD   indices into the input detections
    int *mi_r, *mi_c;                 // MAXX   matched_indices
    int *m_d, *m_t;                   // MAXX   matches (det idx in hi list, trk position)
    int *um_d, *um_t;                 // MAXD+MAXX, MAXT+MAXX
    int *rowcnt, *colcnt, *rowhit;    // MAXD, MAXT, MAXD
    int *tmp_a, *tmp_b;               // MAXX scratch lists
    int *scan;                        // NWAVES
    double *red;                      // NWAVES
    int *sc;                          // 32 scalars
};
enum : int { SC_N = 0, SC_N2, SC_T, SC_NMI, SC_NM, SC_NUD, SC_NUT, SC_FLAG, SC_NL, SC_NREM };

__host__ __device__ inline size_t lds_fixed_bytes(int MAXT, int MAXD)
{
    const int MAXX = MAXT > MAXD ? MAXT : MAXD;
    size_t b = 0;
    b += sizeof(double) * (size_t)MAXT * 11;
    b += sizeof(double) * (size_t)MAXX * 3 + sizeof(int) * (size_t)MAXX * 4 + (size_t)MAXX * 2;   // LsaWork
    b = (b + 15) & ~(size_t)15;
    b += sizeof(int) * ((size_t)MAXD * 2 + (size_t)MAXX * 4 + (size_t)(MAXD + MAXX) + (size_t)(MAXT + MAXX)
                        + (size_t)MAXD * 2 + (size_t)MAXT + (size_t)MAXX * 2 + NWAVES + 32);
    b = (b + 15) & ~(size_t)15;
    b += sizeof(double) * NWAVES;
    return (b + 15) & ~(size_t)15;
}

__device__ inline void carve(unsigned char *smem, int MAXT, int MAXD, Lds &L)
{
    const int MAXX = MAXT > MAXD ? MAXT : MAXD;
    double *d = (double *)smem;
    L.trk_box = d; d += (size_t)MAXT * 4;
    L.kobs = d; d += (size_t)MAXT * 5;
    L.velp = d; d += (size_t)MAXT * 2;
    L.W.u = d; d += MAXX; L.W.v = d; d += MAXX; L.W.spc = d; d += MAXX;
    int *ip = (int *)d;
    L.W.path = ip; ip += MAXX; L.W.row4col = ip; ip += MAXX; L.W.remaining = ip; ip += MAXX; L.W.col4row = ip; ip += MAXX;
    unsigned char *bp = (unsigned char *)ip;
    L.W.SR = bp; bp += MAXX; L.W.SC = bp; bp += MAXX;
    bp = (unsigned char *)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
    ip = (int *)bp;
    L.hi_idx = ip; ip += MAXD; L.lo_idx = ip; ip += MAXD;
    L.mi_r = ip; ip += MAXX; L.mi_c = ip; ip += MAXX; L.m_d = ip; ip += MAXX; L.m_t = ip; ip += MAXX;
    L.um_d = ip; ip += MAXD + MAXX; L.um_t = ip; ip += MAXT + MAXX;
    L.rowcnt = ip; ip += MAXD; L.rowhit = ip; ip += MAXD; L.colcnt = ip; ip += MAXT;
    L.tmp_a = ip; ip += MAXX; L.tmp_b = ip; ip += MAXX;
    L.scan = ip; ip += NWAVES; L.sc = ip; ip += 32;
    bp = (unsigned char *)(((uintptr_t)ip + 15) & ~(uintptr_t)15);
    L.red = (double *)bp; bp += sizeof(double) * NWAVES;
    bp = (unsigned char *)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
    L.cost = (double *)bp;
}

// sorted-unique set difference on a small int list held in LDS (np.setdiff1d). All threads call.
// list[0..n) -> list[0..ret) sorted ascending without members of rem[0..nrem). Uses tmp (>= n).
__device__ int block_setdiff_sorted(int *list, int n, const int *rem, int nrem, int *tmp, int *s_scan)
{
    // rank sort (n is tiny): position = #elements smaller, duplicates dropped
    for (int k = threadIdx.x; k < n; k += BLOCK) {
        const int v = list[k];
        int rank = 0; bool dup = false, drop = false;
        for (int q = 0; q < n; ++q) { const int o = list[q]; rank += (o < v); dup |= (o == v && q < k); }
        for (int q = 0; q < nrem; ++q) drop |= (rem[q] == v);
        tmp[k] = (dup || drop) ? -1 : rank;
    }
    __syncthreads();
    // values with their ranks: emit in rank order -> compaction over rank space [0,n)
    // invert: slot[rank] = value
    int *inv = tmp + 0;   // reuse after reading: do in two steps via registers
    int myv[2], myr[2], cnt = 0;
    for (int k = threadIdx.x; k < n && cnt < 2; k += BLOCK) { myv[cnt] = list[k]; myr[cnt] = tmp[k]; ++cnt; }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += BLOCK) inv[k] = INT32_MIN;
    __syncthreads();
    for (int c = 0; c < cnt; ++c) if (myr[c] >= 0) inv[myr[c]] = myv[c];
    __syncthreads();
    const int kept = block_compact(n, [&](int r) { return inv[r] != INT32_MIN; },
                                   [&](int r, int pos) { list[pos] = inv[r]; }, s_scan);
    __syncthreads();
    return kept;
}

}  // namespace
