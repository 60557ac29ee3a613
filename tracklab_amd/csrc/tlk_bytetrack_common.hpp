// tlk_bytetrack_common.hpp -- what the BYTE-family trackers (ByteTrack, BoT-SORT: the same update() skeleton over different costs)
// share: the per-stream LDS carve, the float32 +1-pixel IoU of their matching.py and the lap.lapjv(extend_cost, cost_limit)
// assignment in its reduced LDS form.
#pragma once
#include "tlk_common.hpp"

namespace {
using namespace tlk;

struct ByLds {
    float *dbox, *dxyah;           // MAXD*4 each, by filtered detection index
    double *dscore;                // MAXD
    float *tbox;                   // MAXT*4 scratch (rows of the current cost matrix)
    int *sel, *hi, *lo, *rem, *udet1;                         // MAXD
    int *pool, *unconf, *rtr, *refind, *newlost, *removed, *newtrk, *pre, *alive, *ntr, *nlost, *dupa, *dupb, *ppos, *upos;   // MAXT
    int *x, *y, *m_r, *m_c, *u_r, *u_c;                       // NX
    LsaWork W;                                                // NX
    int *mi_r, *mi_c;                                         // NX
    int *scan, *sc;
    double *cost;                                             // rest of the LDS allocation: the current assignment problem
};

__host__ __device__ inline size_t bylds_bytes(int MAXT, int MAXD, int NX)
{
    size_t b = sizeof(double) * ((size_t)MAXD + 3 * (size_t)NX);
    b += sizeof(float) * ((size_t)MAXD * 8 + (size_t)MAXT * 4);
    b += sizeof(int) * ((size_t)MAXD * 5 + (size_t)MAXT * 15 + (size_t)NX * (6 + 4 + 2) + NWAVES + 32);
    b += (size_t)NX * 2 + 64;
    return (b + 15) & ~(size_t)15;
}
__device__ inline void bycarve(unsigned char *smem, int MAXT, int MAXD, int NX, ByLds &L)
{
    double *d = (double *)smem;
    L.dscore = d; d += MAXD; L.W.u = d; d += NX; L.W.v = d; d += NX; L.W.spc = d; d += NX;
    float *f = (float *)d;
    L.dbox = f; f += (size_t)MAXD * 4; L.dxyah = f; f += (size_t)MAXD * 4; L.tbox = f; f += (size_t)MAXT * 4;
    int *ip = (int *)f;
    L.sel = ip; ip += MAXD; L.hi = ip; ip += MAXD; L.lo = ip; ip += MAXD; L.rem = ip; ip += MAXD; L.udet1 = ip; ip += MAXD;
    L.pool = ip; ip += MAXT; L.unconf = ip; ip += MAXT; L.rtr = ip; ip += MAXT; L.refind = ip; ip += MAXT; L.newlost = ip; ip += MAXT;
    L.removed = ip; ip += MAXT; L.newtrk = ip; ip += MAXT; L.pre = ip; ip += MAXT; L.alive = ip; ip += MAXT; L.ntr = ip; ip += MAXT;
    L.nlost = ip; ip += MAXT; L.dupa = ip; ip += MAXT; L.dupb = ip; ip += MAXT; L.ppos = ip; ip += MAXT; L.upos = ip; ip += MAXT;
    L.x = ip; ip += NX; L.y = ip; ip += NX; L.m_r = ip; ip += NX; L.m_c = ip; ip += NX; L.u_r = ip; ip += NX; L.u_c = ip; ip += NX;
    L.W.path = ip; ip += NX; L.W.row4col = ip; ip += NX; L.W.remaining = ip; ip += NX; L.W.col4row = ip; ip += NX;
    L.mi_r = ip; ip += NX; L.mi_c = ip; ip += NX;
    L.scan = ip; ip += NWAVES; L.sc = ip; ip += 32;
    unsigned char *bp = (unsigned char *)ip;
    L.W.SR = bp; bp += NX; L.W.SC = bp; bp += NX;
    L.cost = (double *)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
}

// List / solver work area of ONE frame (r04: capacity is an allocation size, the reference's lists simply grow -- byte_tracker.py:167-320):
// the SMALLEST tier that holds (tracked + lost + detections, detections) -- 256 x 128 or 384 x 128 carved out of LDS, the rest of the LDS being
// the cost matrix -- or, for a scene beyond that, the bank's full capacity carved out of HBM (`big_ws`; matrices then go to the HBM spill
// area).  Same code either way, generic pointers.  Returns the number of cost entries that fit the LDS (0 in the HBM tier).  `alive` is
// indexed by SLOT, not by list position, so it always lives in HBM at capacity.
__device__ inline int by_carve_frame(unsigned char *smem, int lds_bytes, unsigned char *big_ws, int MAXT, int MAXD, int need_t, int n_in,
                                     int *alive_g, ByLds &L)
{
    const int tiers[2][2] = {{256, 128}, {384, 128}};
    int entries = -1;
    for (int k = 0; k < 2 && entries < 0; ++k) {
        const int tt = MAXT < tiers[k][0] ? MAXT : tiers[k][0], td = MAXD < tiers[k][1] ? MAXD : tiers[k][1];
        const size_t fixed = bylds_bytes(tt, td, tt + td) + 16;
        if (need_t <= tt && n_in <= td && fixed + 4096 <= (size_t)lds_bytes) {
            bycarve(smem, tt, td, tt + td, L);
            entries = (int)(((size_t)lds_bytes - fixed) / sizeof(double));
        }
    }
    if (entries < 0) { bycarve(big_ws, MAXT, MAXD, MAXT + MAXD, L); entries = 0; }
    L.alive = alive_g;
    return entries;
}

__device__ __forceinline__ float bbox_iou32(const float *b, const float *q)       // matching.py:181-217
{
    const float box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    const float iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
    if (iw > 0) {
        const float ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
        if (ih > 0) {
            const float uaf = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
            return (float)((double)(iw * ih) / (double)uaf);
        }
    }
    return 0.f;
}

struct AsgOut { int nm, n_ur, n_uc, err; };      // err: the solver hit a loop bound (LSA_EINTERNAL)
// linear_assignment (matching.py:37-48): lap.lapjv(extend_cost=True, cost_limit=thresh) on cost(i, j), i < nr, j < nc.
// lap embeds the problem in an (nr+nc)^2 one with thresh/2 padding; its objective is
//   sum over matched pairs of c_ij  +  (unmatched rows + unmatched columns) * thresh / 2  =  const + sum over matched (c_ij - thresh),
// i.e. a rectangular assignment on min(c_ij - thresh, 0) where a row sitting on a 0 entry is "unmatched". That problem is nr x nc
// (4x fewer entries, and it fits the LDS cost area) and has the same optimal pair set whenever no two real costs tie
// (tlk_lsa_lapjv_limit_f64 keeps the literal embedding; tests compare the two). Fills L.m_r/m_c (matches, ascending rows),
// L.u_r, L.u_c (ascending). All 256 threads call.
template <class CostFn>
__device__ AsgOut lapjv_assign(int nr, int nc, double thresh, CostFn cost, double *ebuf, double *lds_cost, int lds_entries, ByLds &L)
{
    AsgOut o{0, 0, 0, 0};
    const int tid = threadIdx.x;
    if (nr == 0 || nc == 0) {
        for (int i = tid; i < nr; i += BLOCK) L.u_r[i] = i;
        for (int j = tid; j < nc; j += BLOCK) L.u_c[j] = j;
        o.n_ur = nr; o.n_uc = nc;
        __syncthreads();
        return o;
    }
    double *cm = (nr * nc <= lds_entries) ? lds_cost : ebuf;
    for (int e = tid; e < nr * nc; e += BLOCK) {
        const int r = e / nc, c = e - r * nc;
        const double v = cost(r, c) - thresh;
        cm[e] = v < 0.0 ? v : 0.0;
    }
    for (int i = tid; i < nr; i += BLOCK) L.x[i] = -1;
    for (int j = tid; j < nc; j += BLOCK) L.y[j] = -1;
    __threadfence_block();
    __syncthreads();
    if (tid < WAVE) {
        const int r = wave_lsa(cm, nr, nc, (size_t)nc, (size_t)1, L.W, L.mi_r, L.mi_c);
        if (tid == 0) L.sc[0] = r == LSA_EINTERNAL ? r : (r < 0 ? 0 : r);
    }
    __syncthreads();
    int np = L.sc[0];
    if (np < 0) { o.err = 1; np = 0; }
    for (int k = tid; k < np; k += BLOCK) {
        const int r = L.mi_r[k], c = L.mi_c[k];
        if (cm[(size_t)r * nc + c] < 0.0) { L.x[r] = c; L.y[c] = r; }
    }
    __syncthreads();
    o.nm = block_compact(nr, [&](int i) { return L.x[i] >= 0; }, [&](int i, int pos) { L.m_r[pos] = i; L.m_c[pos] = L.x[i]; }, L.scan);
    o.n_ur = block_compact(nr, [&](int i) { return L.x[i] < 0; }, [&](int i, int pos) { L.u_r[pos] = i; }, L.scan);
    o.n_uc = block_compact(nc, [&](int j) { return L.y[j] < 0; }, [&](int j, int pos) { L.u_c[pos] = j; }, L.scan);
    __syncthreads();
    return o;
}

}  // namespace
