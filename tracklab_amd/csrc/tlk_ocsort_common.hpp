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


// np.linalg.inv of the 4 x 4 innovation covariance in LAPACK's OPERATION ORDER (r03; restated and verified in oracle/src/lapack_order.h::lo_inv4,
// 0 mismatches against numpy in 3000 matrices incl. pivoting ones): dgesv(S, I) = OpenBLAS getf2 -- left-looking LU with partial pivoting, fma-chain
// dots against the finished columns, the sub-diagonal scaled by the RECIPROCAL of the pivot -- then dgetrs on the row-permuted identity (unit-lower
// forward and upper backward substitution in column-oriented trsm order, diagonal by reciprocal). Row interchanges as predicated swaps: no
// dynamic register indexing.
__device__ __forceinline__ void inv4(const double (&S)[16], double (&SI)[16])
{
    double a[4][4], X[4][4];
    int ipiv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[i][j] = S[i * 4 + j]; X[i][j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = a[i][j];
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r > i && r == ipiv[i]) { const double t = b[i]; b[i] = b[r]; b[r] = t; }
        }
#pragma unroll
        for (int i = 1; i < 4; ++i) if (i < j) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < i) t = fma(a[i][q], b[q], t);
            b[i] = b[i] - t;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i >= j) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < j) t = fma(a[i][q], b[q], t);
            b[i] = b[i] - t;
        }
        int jp = j;
        double best = fabs(b[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i > j && fabs(b[i]) > best) { best = fabs(b[i]); jp = i; }     // idamax: the first maximum
        ipiv[j] = jp;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i][j] = b[i];
        if (best != 0.0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r > j && r == jp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q <= j) { const double t = a[j][q]; a[j][q] = a[r][q]; a[r][q] = t; }
            }
            const double rcp = 1.0 / a[j][j];
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i > j) a[i][j] = a[i][j] * rcp;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r > i && r == ipiv[i]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const double t = X[i][q]; X[i][q] = X[r][q]; X[r][q] = t; }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = X[i][c];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i > k) x[i] = fma(-a[i][k], x[k], x[i]);
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            x[k] = x[k] * (1.0 / a[k][k]);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i < k) x[i] = fma(-a[i][k], x[k], x[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) SI[i * 4 + c] = x[i];
    }
}
// np.dot(K (rows x 4), y (4, 1)): dgemv -- the four products rounded separately, summed (p0 + p2) + (p1 + p3)
__device__ __forceinline__ double dot4_h2(const double *k4, const double (&y)[4])
{
    const double p0 = k4[0] * y[0], p1 = k4[1] * y[1], p2 = k4[2] * y[2], p3 = k4[3] * y[3];
    return (p0 + p2) + (p1 + p3);
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
#ifdef TLK_LDS_CANARY
    unsigned *can[24];                // guard words between the arrays (debug build: tools/build_canary.sh, tests/test_gpu_canary.py)
    int ncan;
#endif
};
#ifdef TLK_LDS_CANARY
constexpr int CANARY_WORDS = 16;      // 64 bytes between any two arrays and after the last one
constexpr unsigned CANARY_PATTERN = 0xA5C3F00Du;
#define TLK_CANARY_BYTES_TOTAL (24 * CANARY_WORDS * 4)
#else
#define TLK_CANARY_BYTES_TOTAL 0
#endif
#ifndef TLK_LDS_PREPAD
#define TLK_LDS_PREPAD 0          // debug builds: bytes of unused LDS AHEAD of the arrays (the layout change under which deepocsort_frames_kernel faulted in r01)
#endif
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
    return ((b + 15) & ~(size_t)15) + TLK_CANARY_BYTES_TOTAL + TLK_LDS_PREPAD;
}

__device__ inline void carve(unsigned char *smem, int MAXT, int MAXD, Lds &L)
{
    const int MAXX = MAXT > MAXD ? MAXT : MAXD;
#ifdef TLK_LDS_CANARY
    L.ncan = 0;
    // a guard of CANARY_WORDS words after the array that ends at p (kept 16-byte aligned): returns the first byte after the guard
    auto guard = [&](void *p) { unsigned char *q = (unsigned char *)(((uintptr_t)p + 15) & ~(uintptr_t)15); L.can[L.ncan++] = (unsigned *)q; return q + CANARY_WORDS * 4; };
#define TLK_G(ptr, T) ptr = (T *)guard(ptr)
#else
#define TLK_G(ptr, T) (void)0
#endif
    double *d = (double *)(smem + TLK_LDS_PREPAD);
    L.trk_box = d; d += (size_t)MAXT * 4; TLK_G(d, double);
    L.kobs = d; d += (size_t)MAXT * 5; TLK_G(d, double);
    L.velp = d; d += (size_t)MAXT * 2; TLK_G(d, double);
    L.W.u = d; d += MAXX; TLK_G(d, double); L.W.v = d; d += MAXX; TLK_G(d, double); L.W.spc = d; d += MAXX; TLK_G(d, double);
    int *ip = (int *)d;
    L.W.path = ip; ip += MAXX; TLK_G(ip, int); L.W.row4col = ip; ip += MAXX; TLK_G(ip, int); L.W.remaining = ip; ip += MAXX; TLK_G(ip, int);
    L.W.col4row = ip; ip += MAXX; TLK_G(ip, int);
    unsigned char *bp = (unsigned char *)ip;
    L.W.SR = bp; bp += MAXX; L.W.SC = bp; bp += MAXX; TLK_G(bp, unsigned char);
    bp = (unsigned char *)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
    ip = (int *)bp;
    L.hi_idx = ip; ip += MAXD; TLK_G(ip, int); L.lo_idx = ip; ip += MAXD; TLK_G(ip, int);
    // (mi_r .. um_t stay contiguous: Deep-OC-SORT overlays its pair list on them, tlk_deepocsort.hip)
    L.mi_r = ip; ip += MAXX; L.mi_c = ip; ip += MAXX; L.m_d = ip; ip += MAXX; L.m_t = ip; ip += MAXX;
    L.um_d = ip; ip += MAXD + MAXX; L.um_t = ip; ip += MAXT + MAXX; TLK_G(ip, int);
    L.rowcnt = ip; ip += MAXD; TLK_G(ip, int); L.rowhit = ip; ip += MAXD; TLK_G(ip, int); L.colcnt = ip; ip += MAXT; TLK_G(ip, int);
    L.tmp_a = ip; ip += MAXX; TLK_G(ip, int); L.tmp_b = ip; ip += MAXX; TLK_G(ip, int);
    L.scan = ip; ip += NWAVES; L.sc = ip; ip += 32; TLK_G(ip, int);
    bp = (unsigned char *)(((uintptr_t)ip + 15) & ~(uintptr_t)15);
    L.red = (double *)bp; bp += sizeof(double) * NWAVES; TLK_G(bp, unsigned char);
    bp = (unsigned char *)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
    L.cost = (double *)bp;
#undef TLK_G
}

#ifdef TLK_LDS_CANARY
// Debug build only.  fill: after carve (all threads call; barrier inside).  check: returns the index + 1 of the first damaged guard, 0 if intact
// (thread 0's view after a barrier).  The last guard sits after the cost area (cost_end): an over-long matrix or list shows up there.
__device__ inline void canary_fill(Lds &L, unsigned char *cost_end)
{
    L.can[L.ncan++] = (unsigned *)(((uintptr_t)cost_end + 15) & ~(uintptr_t)15);      // (every thread holds its own, identical copy of L)
    __syncthreads();
    for (int c = threadIdx.x / CANARY_WORDS; c < L.ncan; c += BLOCK / CANARY_WORDS) L.can[c][threadIdx.x % CANARY_WORDS] = CANARY_PATTERN;
    __syncthreads();
}
__device__ inline int canary_check(const Lds &L)
{
    __syncthreads();
    int bad = 0;
    for (int c = 0; c < L.ncan && !bad; ++c)
        for (int w = 0; w < CANARY_WORDS; ++w) if (L.can[c][w] != CANARY_PATTERN) { bad = c + 1; break; }
    return bad;
}
#endif

// sorted-unique set difference on an int list (np.setdiff1d). All threads call.
// list[0..n) -> list[0..ret) sorted ascending, duplicates dropped, without members of rem[0..nrem). Uses tmp (>= n ints).  Any n (r04: the
// r01-r03 version kept each thread's elements in two registers and silently lost everything beyond 512 entries -- reachable once the
// capacity tiers let a frame hold more unmatched trackers than that); values must be in [0, 65536), n < 32768.
__device__ int block_setdiff_sorted(int *list, int n, const int *rem, int nrem, int *tmp, int *s_scan)
{
    // rank = number of smaller elements (the final order); a duplicate (not its first occurrence) or a member of rem is dropped.
    // tmp[k] = rank << 16 | value: from here on `list` itself is free to be overwritten.
    for (int k = threadIdx.x; k < n; k += BLOCK) {
        const int v = list[k];
        int rank = 0; bool dup = false, drop = false;
        for (int q = 0; q < n; ++q) { const int o = list[q]; rank += (o < v); dup |= (o == v && q < k); }
        for (int q = 0; q < nrem; ++q) drop |= (rem[q] == v);
        tmp[k] = (dup || drop) ? -1 : ((rank << 16) | v);
    }
    __threadfence_block();
    __syncthreads();
    // list[r] = the kept value of rank r, INT32_MIN where no kept value has that rank
    for (int r = threadIdx.x; r < n; r += BLOCK) {
        int val = INT32_MIN;
        for (int q = 0; q < n; ++q) { const int t = tmp[q]; if (t >= 0 && (t >> 16) == r) val = t & 0xffff; }
        list[r] = val;
    }
    __threadfence_block();
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += BLOCK) tmp[r] = list[r];
    __threadfence_block();
    __syncthreads();
    const int kept = block_compact(n, [&](int r) { return tmp[r] != INT32_MIN; }, [&](int r, int pos) { list[pos] = tmp[r]; }, s_scan);
    __threadfence_block();
    __syncthreads();
    return kept;
}

}  // namespace
