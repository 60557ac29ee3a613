// tlk_common.hpp -- shared host/device helpers of libtlk.so (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "tlk.h"

namespace tlk {

// ------------------------------------------------------------------------------------ host errors
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
// negative per-stream frame count written by a tracker kernel -> status of the update call (`fn`: name of the entry point)
inline int fail_stream(int code, const char *fn, const char *capacity_detail = "max_tracks/max_dets")
{
    if (code == TLK_EINTERNAL)
        return fail(TLK_EINTERNAL, std::string(fn) + ": a device loop hit its iteration bound (assignment solver state no consistent run can reach); "
                                   "the stream is poisoned until it is reset");
    if (code <= -100) return fail(code, std::string(fn) + ": guard word damaged (debug build), array " + std::to_string(-100 - code));
    return fail(code, std::string(fn) + ": tracker capacity exceeded (" + capacity_detail + ")");
}
#define TLK_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::tlk::fail(TLK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

int conv_stem3_f32(const float *x, const float *w, const float *bias, float *y, int n, int h, int wd, int cout, int kh, int kw, int stride, int pad, int act,
                   int x_pix, int y_pix, hipStream_t st);      // tlk_conv_stem.hip: 1 = not a stem it has a kernel for
const int *conv_dynamic_batch();      // tlk_conv_set_dynamic_batch (tlk_conv.hip): device pointer to the live image count, or NULL

constexpr int WAVE = 64;
constexpr int BLOCK = 256;          // 4 wavefronts, one per SIMD of a CU
constexpr int NWAVES = BLOCK / WAVE;

// ------------------------------------------------------------------------------------ device helpers
#if defined(__HIPCC__)

__device__ __forceinline__ double dmax_(double a, double b) { return a > b ? a : b; }   // np.maximum (non-NaN)
__device__ __forceinline__ double dmin_(double a, double b) { return a < b ? a : b; }

// Similarity of two xyxy boxes; same operation order as oc_sort/association.py:5-147 so that the
// fp64 result is bit-identical to numpy's (compiled with -ffp-contract=off). TLK_CT returns the raw
// centre distance; the matrix-wide rescale of association.py:169-171 is applied by the caller.
__device__ __forceinline__ double box_similarity(int variant, const double *a, const double *b)
{
    if (variant == TLK_CT) {
        double cx1 = (a[0] + a[2]) / 2.0, cy1 = (a[1] + a[3]) / 2.0;
        double cx2 = (b[0] + b[2]) / 2.0, cy2 = (b[1] + b[3]) / 2.0;
        double dx = cx1 - cx2, dy = cy1 - cy2;
        return sqrt(dx * dx + dy * dy);
    }
    double xx1 = dmax_(a[0], b[0]), yy1 = dmax_(a[1], b[1]);
    double xx2 = dmin_(a[2], b[2]), yy2 = dmin_(a[3], b[3]);
    double w = dmax_(0., xx2 - xx1), h = dmax_(0., yy2 - yy1);
    double wh = w * h;
    double iou = wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
    if (variant == TLK_IOU) return iou;
    double xxc1 = dmin_(a[0], b[0]), yyc1 = dmin_(a[1], b[1]);
    double xxc2 = dmax_(a[2], b[2]), yyc2 = dmax_(a[3], b[3]);
    if (variant == TLK_GIOU) {
        double wc = xxc2 - xxc1, hc = yyc2 - yyc1;
        double area_enclose = wc * hc;
        double giou = iou - (area_enclose - wh) / area_enclose;
        return (giou + 1.) / 2.0;
    }
    double cx1 = (a[0] + a[2]) / 2.0, cy1 = (a[1] + a[3]) / 2.0;
    double cx2 = (b[0] + b[2]) / 2.0, cy2 = (b[1] + b[3]) / 2.0;
    double inner = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
    double outer = (xxc2 - xxc1) * (xxc2 - xxc1) + (yyc2 - yyc1) * (yyc2 - yyc1);
    if (variant == TLK_DIOU) return ((iou - inner / outer) + 1) / 2.0;
    double w1 = a[2] - a[0], h1 = a[3] - a[1], w2 = b[2] - b[0], h2 = b[3] - b[1];
    h2 = h2 + 1.; h1 = h1 + 1.;
    double arct = atan(w2 / h2) - atan(w1 / h1);
    const double PI = 3.141592653589793;
    double v = (4 / (PI * PI)) * (arct * arct);
    double S = 1 - iou;
    double alpha = v / (S + v);
    return ((iou - inner / outer - alpha * v) + 1) / 2.0;
}

// Stable stream compaction over i in [0,n): every thread of the 256-thread block must call.
// emit(i, position) runs for each i with pred(i) true; returns the number of kept items.
// s_scan: >= NWAVES ints of LDS scratch.
template <class Pred, class Emit>
__device__ __forceinline__ int block_compact(int n, Pred pred, Emit emit, int *s_scan)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int base = 0;
    for (int start = 0; start < n; start += BLOCK) {
        const int i = start + (int)threadIdx.x;
        const bool f = (i < n) && pred(i);
        const unsigned long long m = __ballot(f);
        const int prefix = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[w] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < NWAVES; ++k) { const int c = s_scan[k]; if (k < w) woff += c; tot += c; }
        if (f) emit(i, base + woff + prefix);
        base += tot;
        __syncthreads();
    }
    return base;
}

// Block-wide max with numpy semantics (NaN propagates). s_red: >= NWAVES doubles of LDS.
__device__ __forceinline__ double block_max_nan(double v, bool valid, double *s_red)
{
    // identity: -inf for invalid lanes
    double x = valid ? v : -INFINITY;
    bool isn = valid && (v != v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(x, off);
        x = (o > x) ? o : x;
    }
    const unsigned long long nanm = __ballot(isn);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_red[w] = nanm ? NAN : x;
    __syncthreads();
    double r = s_red[0];
#pragma unroll
    for (int k = 1; k < NWAVES; ++k) { double o = s_red[k]; r = (o != o || r != r) ? NAN : (o > r ? o : r); }
    __syncthreads();
    return r;
}

// Return codes of the wave solvers besides a pair count: -1 = infeasible (scipy raises ValueError there), LSA_EINTERNAL = a loop bound
// that no consistent solver state can reach was hit (corrupted work area); the banks turn it into TLK_EINTERNAL for the stream.
constexpr int LSA_EINTERNAL = -3;

// LDS work arrays of one LSA problem (sized for max(nr,nc) entries each).
struct LsaWork {
    double *u, *v, *spc;
    int *path, *row4col, *remaining, *col4row;
    unsigned char *SR, *SC;
    int hop_limit = 0;      // > 0: cap on the augmenting walk below its natural bound (tlk_debug_lsa_hop_limit: lets a test trip LSA_EINTERNAL)
};

// scipy-identical rectangular LSAP solved by ONE wavefront (all 64 lanes must call, converged).
// cost(i, j) element = cost[i*rs + j*cs] for the ORIGINAL (nr0 x nc0) orientation; tall inputs are
// solved transposed, exactly like scipy's rectangular_lsap.cpp. Writes pairs sorted by original
// row to rows_out/cols_out and returns their number (min(nr0,nc0)), or -1 if infeasible.
// Scan semantics restated: remaining[] is filled in reverse; among minimum shortest-path costs an
// unassigned column wins (the one scanned last), otherwise the first scanned column.
__device__ __noinline__ int wave_lsa_lds(const double *cost, int nr0, int nc0, size_t rs0, size_t cs0,
                                        const LsaWork &W, int *rows_out, int *cols_out)
{
    const int lane = threadIdx.x & 63;
    if (nr0 == 0 || nc0 == 0) return 0;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    const size_t rs = transpose ? cs0 : rs0, cs = transpose ? rs0 : cs0;
    for (int k = lane; k < nr; k += WAVE) { W.u[k] = 0.0; W.col4row[k] = -1; }
    for (int k = lane; k < nc; k += WAVE) { W.v[k] = 0.0; W.row4col[k] = -1; W.path[k] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int ret = nr;
    for (int cur = 0; cur < nr; ++cur) {
        for (int k = lane; k < nc; k += WAVE) { W.remaining[k] = nc - k - 1; W.SC[k] = 0; W.spc[k] = INFINITY; }
        for (int k = lane; k < nr; k += WAVE) W.SR[k] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        double minval = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            // (bounded by construction: every pass takes one column out of `remaining`; with none left the scan finds no finite candidate and
            //  the loop leaves through the infeasible exit below -- at most nc + 1 passes whatever the work area holds.  A guard of its own here
            //  was measured at +10..15 us per 100-object frame and removed.)
            if (lane == 0) W.SR[i] = 1;
            const double ui = W.u[i];
            double best = INFINITY;
            int best_s = -1, best_it = -1;
            for (int it = lane; it < num_remaining; it += WAVE) {
                const int j = W.remaining[it];
                const double r = minval + cost[(size_t)i * rs + (size_t)j * cs] - ui - W.v[j];
                double sp = W.spc[j];
                if (r < sp) { W.path[j] = i; W.spc[j] = r; sp = r; }
                const int s = (W.row4col[j] == -1) ? (nc + it) : (nc - 1 - it);
                if (sp < best || (sp == best && s > best_s)) { best = sp; best_s = s; best_it = it; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int os = __shfl_xor(best_s, off);
                const int oi = __shfl_xor(best_it, off);
                if (ob < best || (ob == best && os > best_s)) { best = ob; best_s = os; best_it = oi; }
            }
            minval = best;
            if (!(minval < INFINITY)) { ret = -1; break; }     // infeasible (or NaN cost)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int j = W.remaining[best_it];
            const int r4c = W.row4col[j];
            if (r4c == -1) sink = j; else i = r4c;
            --num_remaining;
            const int last = W.remaining[num_remaining];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { W.SC[j] = 1; W.remaining[best_it] = last; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        if (ret < 0) break;
        // dual update (rectangular_lsap.cpp: "update dual variables")
        for (int k = lane; k < nr; k += WAVE)
            if (W.SR[k] && k != cur) W.u[k] += minval - W.spc[W.col4row[k]];
        if (lane == 0) W.u[cur] += minval;
        for (int k = lane; k < nc; k += WAVE)
            if (W.SC[k]) W.v[k] -= minval - W.spc[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // augment (uniform across lanes; lane 0 writes).  An alternating path visits every row at most once: a walk longer than nr steps
        // (or one that leaves the index range) means the work area was corrupted under the solver -- give up with LSA_EINTERNAL instead
        // of walking garbage for ever (r05: every data-dependent device loop of the trackers is bounded)
        int j = sink;
        for (int hops = 0;; ++hops) {
            if (hops > (W.hop_limit > 0 ? W.hop_limit : nr) || (unsigned)j >= (unsigned)nc) return LSA_EINTERNAL;
            const int pi = W.path[j];
            if ((unsigned)pi >= (unsigned)nr) return LSA_EINTERNAL;
            const int old = W.col4row[pi];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { W.row4col[j] = pi; W.col4row[pi] = j; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            j = old;
            if (pi == cur) break;
        }
    }
    if (ret < 0) return ret;
    if (!transpose) {
        for (int k = lane; k < nr; k += WAVE) { rows_out[k] = k; cols_out[k] = W.col4row[k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        return nr;
    }
    // transposed: original row r == transposed column r; emit (r, row4col[r]) for assigned r, ascending
    int base = 0;
    for (int start = 0; start < nc; start += WAVE) {
        const int r = start + lane;
        const bool f = (r < nc) && (W.row4col[r] != -1);
        const unsigned long long m = __ballot(f);
        if (f) { const int p = base + __popcll(m & ((1ull << lane) - 1ull)); rows_out[p] = r; cols_out[p] = W.row4col[r]; }
        base += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return base;
}


// ---------------------------------------------------------------------------------------------
// Register-resident variant (max(nr,nc) <= 512): lane l owns columns j = c*64 + l and keeps their dual v,
// shortest-path cost, path, row4col and position-in-`remaining` in VGPRs, so one scan step is ONE coalesced
// LDS read of a cost row segment + a DPP min-reduction + a few ballots/readlanes -- no per-step LDS
// bookkeeping. `remaining[]` of scipy is represented by its inverse permutation pos[j] (scan order only
// matters for tie-breaks): initially pos[j] = nc-1-j; removing column j* at position p moves the column at
// the last position to p. Row duals u[] and col4row[] live in LDS (W.u, W.col4row).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_ror_i32(int v, int n)
{
    switch (n) {
        case 1: return __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false);
        case 2: return __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false);
        case 4: return __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);
        default: return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);
    }
}
__device__ __forceinline__ double dpp_ror_f64(double v, int n)
{
    return __hiloint2double(dpp_ror_i32(__double2hiint(v), n), dpp_ror_i32(__double2loint(v), n));
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_min_f64(double v)      // all lanes get the minimum (no NaNs expected)
{
#pragma unroll
    for (int n = 1; n <= 8; n <<= 1) { const double o = dpp_ror_f64(v, n); v = o < v ? o : v; }
    const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    const double ab = b < a ? b : a, cd = d < c ? d : c;
    return cd < ab ? cd : ab;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int n = 1; n <= 8; n <<= 1) { const int o = dpp_ror_i32(v, n); v = o > v ? o : v; }
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// (mov_dpp, i.e. an UNDEFINED old operand, is what lets the compiler fold the lane rotation into the min / max itself:
//  v_min_u32_dpp, one instruction per level instead of copy + rotate + min)
// r03: the cross-row step stays in the VALU too -- row_bcast15 hands every row the minimum of the row before it, row_bcast31 hands rows 2-3 the minimum
// of rows 0-1, so lane 63 holds the wave minimum after six DPP-folded v_min / v_max and ONE v_readlane (was: four v_readlane + s_min + two
// v_mov + v_min3, a VALU -> SALU -> VALU round trip with its wait states inside the Hungarian solver's dependent chain)
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v)      // wave-uniform result
{
    v = min(v, (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xf, 0xf, true));
    v = min(v, (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x122, 0xf, 0xf, true));
    v = min(v, (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xf, 0xf, true));
    v = min(v, (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true));
    v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));      // row_bcast15 -> rows 1, 3
    v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));      // row_bcast31 -> rows 2, 3
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_max_i32_fast(int v)
{
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x122, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
// order-preserving map double -> (hi, lo) unsigned pair: numeric < on non-NaN doubles == lexicographic unsigned < on (hi, lo)
__device__ __forceinline__ void f64_key(double x, unsigned int &hi, unsigned int &lo)
{
    const int h = __double2hiint(x);
    const unsigned int m = (unsigned int)(h >> 31);
    hi = (unsigned int)h ^ (m | 0x80000000u);
    lo = (unsigned int)__double2loint(x) ^ m;
}
__device__ __forceinline__ double f64_unkey(unsigned int hi, unsigned int lo)
{
    const unsigned int m = (hi & 0x80000000u) ? 0u : 0xffffffffu;
    return __hiloint2double((int)(hi ^ (m | 0x80000000u)), (int)(lo ^ m));
}

// Address spaces matter here: through generic pointers every access of the solver became a flat_load / flat_store (and the LsaWork
// members were re-read from the caller's stack inside the loops); with LDS-qualified pointers they are ds_read / ds_write with
// immediate offsets and the work-area pointers stay in registers.
#define TLK_LDS __attribute__((address_space(3)))
#define TLK_GLOBAL __attribute__((address_space(1)))

template <int CPL, typename CostPtr>
__device__ __noinline__ int wave_lsa_reg(CostPtr cost, int nr0, int nc0, unsigned rs0, unsigned cs0,
                                         TLK_LDS double *u_lds, TLK_LDS int *col4row, int *rows_out, int *cols_out, int hop_limit)
{
    const int lane = threadIdx.x & 63;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    const unsigned rs = transpose ? cs0 : rs0, cs = transpose ? rs0 : cs0;
    const int hop_cap = hop_limit > 0 ? hop_limit : nr;
    double v_[CPL], spc_[CPL];
    int r4c_[CPL], path_[CPL], pos_[CPL];
    unsigned coff_[CPL];                                   // element offset of the lane's columns inside a cost row
#pragma unroll
    for (int c = 0; c < CPL; ++c) { v_[c] = 0.0; r4c_[c] = -1; path_[c] = -1; const int j = c * WAVE + lane; coff_[c] = (unsigned)(j < nc ? j : 0) * cs; }
    for (int k = lane; k < nr; k += WAVE) { u_lds[k] = 0.0; col4row[k] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int cur = 0; cur < nr; ++cur) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = c * WAVE + lane; pos_[c] = j < nc ? nc - 1 - j : -2; spc_[c] = INFINITY; }
        double minval = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            // (bounded by construction, as in wave_lsa_lds: after nc passes every column is out of the scan, `best` stays INFINITY and the
            //  loop leaves through the infeasible exit)
            const double ui = u_lds[i];
            const unsigned rbase = (unsigned)i * rs;
            double cval[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) cval[c] = cost[rbase + coff_[c]];      // unconditional (offsets of absent columns point at column 0): all loads in flight together
            // branch-free scan of the lane's columns (select instead of exec-mask regions), then the wave minimum through 32-bit
            // DPP reductions of an order-preserving integer key (the fp64 compare-and-select ladder was ~60 dependent instructions)
            double best = INFINITY;
            int best_s = -1, best_c = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int pos = pos_[c];
                const int act = -(int)(pos >= 0);                   // all-ones mask for a column still in `remaining`
                const double r = minval + cval[c] - ui - v_[c];
                const bool upd = (act != 0) & (r < spc_[c]);
                path_[c] = upd ? i : path_[c];
                spc_[c] = upd ? r : spc_[c];
                const double sp = act ? spc_[c] : INFINITY;
                // tie-break rank: an unassigned column ranks by nc + pos (the one scanned last wins), an assigned one by nc - 1 - pos
                // (first scanned wins); -1 for a column that is out of the scan. Arithmetic only: no divergent regions.
                const int un = -(int)(r4c_[c] == -1);
                const int sr = ((nc + pos) & un) | ((nc - 1 - pos) & ~un);
                const int s = (sr & act) | ~act;
                const bool better = (sp < best) | ((sp == best) & (s > best_s));
                best = better ? sp : best; best_s = better ? s : best_s; best_c = better ? c : best_c;
            }
            unsigned int khi, klo;
            f64_key(best + 0.0, khi, klo);                          // + 0.0: -0.0 and +0.0 compare equal, so must their keys
            const unsigned int mhi = wave_min_u32(khi);
            // r03: distinct shortest-path costs almost always differ in their upper 32 bits already -- then ONE lane carries the minimum and its
            // lower word is read with one v_readlane; the second reduction runs only for equal upper words (exact ties: clamped / gated costs)
            const unsigned long long m_hi = __ballot(khi == mhi);
            const unsigned int mlo = __popcll(m_hi) == 1 ? (unsigned int)__builtin_amdgcn_readlane((int)klo, __ffsll((long long)m_hi) - 1)
                                                        : wave_min_u32(khi == mhi ? klo : 0xffffffffu);
            minval = f64_unkey(mhi, mlo);
            if (!(minval < INFINITY)) return -1;                    // infeasible
            const bool is_min = khi == mhi && klo == mlo;
            unsigned long long m_eq = __ballot(is_min && best_s >= 0);
            if (__popcll(m_eq) != 1) {                              // ties: unassigned column scanned last wins, else first scanned
                const int ms = wave_max_i32_fast(is_min ? best_s : -1);
                m_eq = __ballot(is_min && best_s == ms);
            }
            const int wl = __ffsll((long long)m_eq) - 1;
            const int wc = __builtin_amdgcn_readlane(best_c, wl);
            int pos_sel = 0, r4c_sel = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) if (c == wc) { pos_sel = pos_[c]; r4c_sel = r4c_[c]; }
            const int wpos = __builtin_amdgcn_readlane(pos_sel, wl);
            const int wr4c = __builtin_amdgcn_readlane(r4c_sel, wl);
            const int jstar = wc * WAVE + wl;
            const int last = num_remaining - 1;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const bool is_star = (lane == wl) && (c == wc);
                if (is_star) pos_[c] = -1;                          // SC[j*] = true
                else if (pos_[c] == last) pos_[c] = wpos;           // remaining[index] = remaining[--num_remaining]
            }
            num_remaining = last;
            if (wr4c == -1) sink = jstar; else i = wr4c;
        }
        // dual update: SR \ {cur} == { row4col[j] : j in SC, assigned }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (pos_[c] == -1) {
                const double d = minval - spc_[c];
                if (r4c_[c] != -1) u_lds[r4c_[c]] += d;
                v_[c] -= d;
            }
        }
        if (lane == 0) u_lds[cur] += minval;
        // augment along path[] (uniform walk; owner lanes update their registers)
        int j = sink;
        for (int hops = 0;; ++hops) {
#ifndef TLK_LSA_UNGUARDED
            if (hops > hop_cap) return LSA_EINTERNAL;      // (see wave_lsa_lds: bounded walk)
#endif
            const int c = j >> 6, l = j & 63;
            int path_sel = 0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) if (q == c) path_sel = path_[q];
            const int pi = __builtin_amdgcn_readlane(path_sel, l);
#pragma unroll
            for (int q = 0; q < CPL; ++q) if (q == c && lane == l) r4c_[q] = pi;
            const int old = col4row[pi];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) col4row[pi] = j;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            j = old;
            if (pi == cur) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (!transpose) {
        for (int k = lane; k < nr; k += WAVE) { rows_out[k] = k; cols_out[k] = col4row[k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        return nr;
    }
    int base = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {            // original row r == transposed column r, ascending
        const int r = c * WAVE + lane;
        const bool f = (r < nc) && (r4c_[c] != -1);
        const unsigned long long m = __ballot(f);
        if (f) { const int p = base + __popcll(m & ((1ull << lane) - 1ull)); rows_out[p] = r; cols_out[p] = r4c_[c]; }
        base += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return base;
}

template <typename CostPtr>
__device__ __forceinline__ int wave_lsa_reg_cpl(CostPtr cost, int nr0, int nc0, size_t rs0, size_t cs0, const LsaWork &W, int *rows_out, int *cols_out)
{
    // the work area of every caller is carved out of its workgroup's LDS
    TLK_LDS double *u = (TLK_LDS double *)W.u;
    TLK_LDS int *c4r = (TLK_LDS int *)W.col4row;
    const int mx = nr0 > nc0 ? nr0 : nc0;
    if (mx <= 64) return wave_lsa_reg<1>(cost, nr0, nc0, (unsigned)rs0, (unsigned)cs0, u, c4r, rows_out, cols_out, W.hop_limit);
    if (mx <= 128) return wave_lsa_reg<2>(cost, nr0, nc0, (unsigned)rs0, (unsigned)cs0, u, c4r, rows_out, cols_out, W.hop_limit);
    if (mx <= 256) return wave_lsa_reg<4>(cost, nr0, nc0, (unsigned)rs0, (unsigned)cs0, u, c4r, rows_out, cols_out, W.hop_limit);
    return wave_lsa_reg<8>(cost, nr0, nc0, (unsigned)rs0, (unsigned)cs0, u, c4r, rows_out, cols_out, W.hop_limit);
}

__device__ __forceinline__ int wave_lsa(const double *cost, int nr0, int nc0, size_t rs0, size_t cs0,
                                        const LsaWork &W, int *rows_out, int *cols_out)
{
    if (nr0 == 0 || nc0 == 0) return 0;
    const int mx = nr0 > nc0 ? nr0 : nc0;
    // (the register-resident solver keeps u / col4row behind LDS-qualified pointers: a work area carved out of HBM -- the big-scene tier of
    //  the StrongSORT banks -- takes the generic-pointer solver whatever the size)
    if (mx > 512 || !__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)W.u))
        return wave_lsa_lds(cost, nr0, nc0, rs0, cs0, W, rows_out, cols_out);
    if (__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)cost))       // wave-uniform: the matrix sits in LDS ...
        return wave_lsa_reg_cpl((const TLK_LDS double *)cost, nr0, nc0, rs0, cs0, W, rows_out, cols_out);
    return wave_lsa_reg_cpl((const TLK_GLOBAL double *)cost, nr0, nc0, rs0, cs0, W, rows_out, cols_out);       // ... or spilled to HBM
}

#endif  // __HIPCC__
}  // namespace tlk
