// tlk_ssort.hip -- plain StrongSORT (plugins/track/strong_sort) on gfx950: three launches per frame for ALL streams of a bank.
//
//   1. ssort_detnorm_kernel  one wavefront per detection feature: float32 L2 norm
//   2. ssort_cosine_kernel   f32 MFMA cosine gallery minimum (tlk_cosine.hpp): workgroup = (confirmed track, 16 detections),
//                            gallery = the track's ring of <= nn_budget EMA features kept in HBM with their norms
//   3. ssort_assoc_kernel    one 256-thread workgroup per stream: KF predict, gate + appearance stage, IoU stage (both with the
//                            scipy-identical wavefront LSA), KF update, feature EMA + renormalisation, predictions without
//                            assignment, track lifecycle, gallery append, output rows.
// Reference: strong_sort.py:41-84 (update), sort/tracker.py:53-193, sort/track.py:69-301, sort/linear_assignment.py:11-174,
// sort/iou_matching.py:7-82, sort/kalman_filter.py:50-214, sort/nn_matching.py:94-161.
// The arithmetic follows the reference's dtype trail: Detection boxes/features are float32 (sort/detection.py:33-36), a new
// track's KF state is float32 until its first predict (initiate() on a float32 measurement), everything else is float64.
#include "tlk_common.hpp"
#include "tlk_cosine.hpp"
#include "tlk_strongsort_common.hpp"

using namespace tlk;

namespace {

constexpr double CHI2_4 = 9.4877;                       // kalman_filter.py:16 chi2inv95[4]
enum : int { SD_MEAN = 0, SD_COV = 8, SD_CONF = 72, SD_TLID = 73, SD_COUNT = 74 };
enum : int { SI_TID = 0, SI_HITS, SI_AGE, SI_TSU, SI_STATE, SI_UWA, SI_F32, SI_CLS, SI_GLEN, SI_GPOS, SI_COUNT };
enum : int { H_NTRK = 0, H_NEXTID, H_NFREE, H_ERR, H_COUNT = 8 };
enum : int { ST_TENTATIVE = 1, ST_CONFIRMED = 2, ST_DELETED = 3 };     // track.py:17-19
constexpr int SGL = 20;                                  // per-track gating scratch: projected mean (4) + Cholesky factor (16)

struct SsDev {
    double *fd;              // SD_COUNT x S x MAXT
    int *fi;                 // SI_COUNT x S x MAXT
    int *hdr, *order, *freestk;
    float *feat;             // S x MAXT x D        features[-1] by slot
    float *gal;              // S x MAXT x B x D    metric.samples[track] as a ring by slot
    float *gnorm;            // S x MAXT x B        norm of every gallery row
    float *dnorm;            // S x MAXD            norm of every detection feature (input order)
    double *reid;            // S x MAXT x MAXD     cosine gallery minimum, row = list position, col = input detection index
    double *gl;              // S x MAXT x SGL
    double *cost_g;          // S x MAXT x MAXD     cost-matrix spill
    int *ps_ws;              // S x 4 x ps_cap       hash tables of the set-order emulation when they do not fit the LDS cost area
    unsigned char *big_ws;   // S x big_stride: list / solver work area of the big-scene tier (see ssort_assoc_kernel)
    size_t big_stride;
    int S, MAXT, MAXD, D, B, lds_bytes, ps_cap, unbounded;      // unbounded: budget=None -- B rows of room, overflow is an error
};

struct SsP {
    double max_dist, max_iou_dist, mc_lambda, ema_alpha, min_conf;
    int max_age, max_unmatched_preds, n_init, wrapper_mode, img_w, img_h;
};

struct SsIn {                // dets index = s*stream_stride_dets + i
    const double *dets;      // (.., 7) [x1,y1,x2,y2,conf,cls,tracklab_id]
    const float *feat;       // (.., D)
    const int *counts; size_t stream_stride_dets, count_stride;
};

// ------------------------------------------------------------------------------------------------ norms + cosine stage
__global__ void __launch_bounds__(BLOCK) ssort_detnorm_kernel(SsDev Dv, SsIn in)
{
    const int s = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int N = in.counts[(size_t)s * in.count_stride];
    const int n = blockIdx.x * NWAVES + w;
    if (N <= 0 || N > Dv.MAXD || n >= N) return;
    const float *x = in.feat + ((size_t)s * in.stream_stride_dets + n) * Dv.D;
    float ss = 0.f;
    for (int d = lane; d < Dv.D; d += WAVE) { const float v = x[d]; ss += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) Dv.dnorm[(size_t)s * Dv.MAXD + n] = sqrtf(ss);
}

template <int DS>
__global__ void __launch_bounds__(BLOCK) ssort_cosine_kernel(SsDev Dv, SsIn in)
{
    __shared__ float s_min[NWAVES][16];
    const int s = blockIdx.z, t = blockIdx.y, n0 = blockIdx.x * 16;
    const int T = Dv.hdr[(size_t)s * H_COUNT + H_NTRK];
    const int N = in.counts[(size_t)s * in.count_stride];
    if (t >= T || N <= 0 || N > Dv.MAXD || n0 >= N) return;
    const int slot = Dv.order[(size_t)s * Dv.MAXT + t];
    const size_t stride = (size_t)Dv.S * Dv.MAXT, at = (size_t)s * Dv.MAXT + slot;
    if (Dv.fi[(size_t)SI_STATE * stride + at] != ST_CONFIRMED) return;          // only confirmed tracks enter the appearance stage
    const int glen = Dv.fi[(size_t)SI_GLEN * stride + at];
    if (glen <= 0) return;
    const int g_lo = (int)(at * Dv.B);
    cosine_gallery_tile<DS>(Dv.gal, g_lo, g_lo + glen, Dv.gnorm, in.feat + (size_t)s * in.stream_stride_dets * Dv.D, N, n0,
                            Dv.dnorm + (size_t)s * Dv.MAXD, Dv.reid + ((size_t)s * Dv.MAXT + t) * Dv.MAXD, s_min);
}

// r03: workgroup = one track, wavefronts own detection tiles (cosine_gallery_track): the gallery crosses HBM once per frame instead of once per detection tile
template <int DS>
__global__ void __launch_bounds__(BLOCK) ssort_cosine_track_kernel(SsDev Dv, SsIn in)
{
    const int s = blockIdx.y, t = blockIdx.x;
    const int T = Dv.hdr[(size_t)s * H_COUNT + H_NTRK];
    const int N = in.counts[(size_t)s * in.count_stride];
    if (t >= T || N <= 0 || N > Dv.MAXD) return;
    const int slot = Dv.order[(size_t)s * Dv.MAXT + t];
    const size_t stride = (size_t)Dv.S * Dv.MAXT, at = (size_t)s * Dv.MAXT + slot;
    if (Dv.fi[(size_t)SI_STATE * stride + at] != ST_CONFIRMED) return;          // only confirmed tracks enter the appearance stage
    const int glen = Dv.fi[(size_t)SI_GLEN * stride + at];
    if (glen <= 0) return;
    const int g_lo = (int)(at * Dv.B);
    cosine_gallery_track<DS>(Dv.gal, g_lo, g_lo + glen, Dv.gnorm, in.feat + (size_t)s * in.stream_stride_dets * Dv.D, N,
                             Dv.dnorm + (size_t)s * Dv.MAXD, Dv.reid + ((size_t)s * Dv.MAXT + t) * Dv.MAXD);
}

// ------------------------------------------------------------------------------------------------ KF pieces that differ from bpbreid's
// predict (kalman_filter.py:85-119): process noise relative to x, y, a, h; float32 arithmetic while the state is still the
// float32 one initiate() made
__device__ __forceinline__ void kf8p_predict(double (&mean)[8], double (&cov)[64], bool f32)
{
    double q[8];
    if (f32) {
        const float f0 = (float)mean[0], f1 = (float)mean[1], f2 = (float)mean[2], f3 = (float)mean[3];
        const float sd[8] = {(float)W_POS * f0, (float)W_POS * f1, f2, (float)W_POS * f3, (float)W_VEL * f0, (float)W_VEL * f1, (float)0.1 * f2, (float)W_VEL * f3};
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float v = sd[i] * sd[i]; q[i] = (double)v; }
    } else {
        const double sd[8] = {W_POS * mean[0], W_POS * mean[1], 1 * mean[2], W_POS * mean[3], W_VEL * mean[0], W_VEL * mean[1], 0.1 * mean[2], W_VEL * mean[3]};
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = sd[i] * sd[i];
    }
    // F (cov F^T): numpy's multi_dot picks A(BC) on the equal-cost tie
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cov[i * 8 + j] = cov[i * 8 + j] + cov[i * 8 + j + 4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) cov[i * 8 + j] = cov[i * 8 + j] + cov[(i + 4) * 8 + j];
#pragma unroll
    for (int i = 0; i < 8; ++i) cov[i * 9] += q[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = mean[i] + mean[i + 4];
}

// project()'s measurement noise (kalman_filter.py:121-152): (1 - confidence) * [h/20, h/20, 1e-1, h/20]
__device__ __forceinline__ void nsa_sd(double h, double conf, double (&sd)[4])
{
    const double base[4] = {W_POS * h, W_POS * h, 1e-1, W_POS * h};
#pragma unroll
    for (int i = 0; i < 4; ++i) sd[i] = (1 - conf) * base[i];
}

__device__ __forceinline__ void strk_tlwh(const BTrk &T, double *o)     // track.py:99-111
{
    const double w = T.d(SD_MEAN + 2) * T.d(SD_MEAN + 3), h = T.d(SD_MEAN + 3);
    o[2] = w; o[3] = h; o[0] = T.d(SD_MEAN) - w / 2; o[1] = T.d(SD_MEAN + 1) - h / 2;
}

// iou_matching.py:7-39 with a float64 track box and a float32 candidate (its bottom-right corner and area are float32)
__device__ __forceinline__ double iou_f32cand(const double *b, const double *cd)
{
    const float c0 = (float)cd[0], c1 = (float)cd[1], c2 = (float)cd[2], c3 = (float)cd[3];
    const double bbr0 = b[0] + b[2], bbr1 = b[1] + b[3];
    const float cbr0 = c0 + c2, cbr1 = c1 + c3;
    const double tl0 = b[0] > (double)c0 ? b[0] : (double)c0, tl1 = b[1] > (double)c1 ? b[1] : (double)c1;
    const double br0 = bbr0 < (double)cbr0 ? bbr0 : (double)cbr0, br1 = bbr1 < (double)cbr1 ? bbr1 : (double)cbr1;
    double w = br0 - tl0, h = br1 - tl1;
    w = w > 0. ? w : 0.; h = h > 0. ? h : 0.;
    const double ai = w * h;
    const float ac = c2 * c3;
    return ai / (b[2] * b[3] + (double)ac - ai);
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------ association kernel
__global__ void __launch_bounds__(BLOCK, 1)
ssort_assoc_kernel(SsDev Dv, SsP P, SsIn in, tlk_ssort_row *__restrict__ rows_all, size_t rows_stream_stride, int out_cap,
                   int *__restrict__ out_counts, size_t oc_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int MAXT = Dv.MAXT, MAXD = Dv.MAXD, D = Dv.D, B = Dv.B;
    int *hdr = Dv.hdr + (size_t)s * H_COUNT;
    // List / solver work area of this frame: the SMALLEST tier that holds (tracks before the frame + detections, detections) -- 256 x 128 or
    // 1024 x 256 carved out of LDS (the rest of the LDS is the cost matrix), or, for a scene beyond that, the bank's full capacity carved out
    // of HBM (r04: the reference's track list simply grows, strong_sort/sort/tracker.py:130-141; the arrays are allocated at capacity in
    // HBM and only this frame's lists move out of LDS).  Same code, generic pointers.
    BLds L;
    int cost_lds_entries = 0;
    {
        const int t_now = hdr[H_NTRK], n_now = in.counts[(size_t)s * in.count_stride];
        const int need_t = t_now + (n_now > 0 ? n_now : 0);
        int ct = 0, cd = 0;
        const int tiers[2][2] = {{256, 128}, {1024, 256}};
        for (int k = 0; k < 2 && ct == 0; ++k) {
            const int tt = MAXT < tiers[k][0] ? MAXT : tiers[k][0], td = MAXD < tiers[k][1] ? MAXD : tiers[k][1];
            if (need_t <= tt && n_now <= td && blds_fixed(tt, td) + 4096 <= (size_t)Dv.lds_bytes) { ct = tt; cd = td; }
        }
        if (ct) { bcarve(smem, ct, cd, L); cost_lds_entries = (int)(((size_t)Dv.lds_bytes - blds_fixed(ct, cd)) / sizeof(double)); }
        else bcarve(Dv.big_ws + (size_t)s * Dv.big_stride, MAXT, MAXD, L);      // (cost_lds_entries = 0: matrices go to cost_g, the set table to ps_ws)
    }
    int *order = Dv.order + (size_t)s * MAXT;
    int *freestk = Dv.freestk + (size_t)s * MAXT;
    const size_t stride = (size_t)Dv.S * MAXT;
    auto trk_at = [&](int slot) { BTrk T; T.fd = Dv.fd + (size_t)s * MAXT + slot; T.fi = Dv.fi + (size_t)s * MAXT + slot; T.stride = stride; return T; };
    float *featS = Dv.feat + (size_t)s * MAXT * D;
    float *galS = Dv.gal + (size_t)s * MAXT * B * D;
    float *gnormS = Dv.gnorm + (size_t)s * MAXT * B;
    const float *dnormS = Dv.dnorm + (size_t)s * MAXD;
    const double *reid = Dv.reid + (size_t)s * MAXT * MAXD;
    double *gl = Dv.gl + (size_t)s * MAXT * SGL;
    tlk_ssort_row *rows = rows_all + (size_t)s * rows_stream_stride;
    int *out_count = out_counts + (size_t)s * oc_stride;
    const size_t dbase = (size_t)s * in.stream_stride_dets;
    const int n_in = in.counts[(size_t)s * in.count_stride];
    const int w = tid >> 6, lane = tid & 63;

    if (hdr[H_ERR] != 0) { if (tid == 0) *out_count = hdr[H_ERR]; return; }
    if (n_in > MAXD || n_in < 0) { if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    if (P.wrapper_mode && n_in == 0) { if (tid == 0) *out_count = 0; return; }      // strong_sort_api.py:68-69

    // inputs[inputs[:, 4] > min_confidence] (strong_sort_api.py:71)
    const int N = block_compact(n_in, [&](int i) { return in.dets[(dbase + i) * 7 + 4] > P.min_conf; }, [&](int i, int pos) { L.sel[pos] = i; }, L.scan);
    int T = hdr[H_NTRK];
    __syncthreads();
    // Tracker.predict (tracker.py:53-58, track.py:243-256)
    for (int p = tid; p < T; p += BLOCK) {
        const BTrk Kt = trk_at(order[p]);
        double mean[8], cov[64];
#pragma unroll
        for (int k = 0; k < 8; ++k) mean[k] = Kt.d(SD_MEAN + k);
#pragma unroll
        for (int k = 0; k < 64; ++k) cov[k] = Kt.d(SD_COV + k);
        kf8p_predict(mean, cov, Kt.i(SI_F32) != 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) Kt.d(SD_MEAN + k) = mean[k];
#pragma unroll
        for (int k = 0; k < 64; ++k) Kt.d(SD_COV + k) = cov[k];
        Kt.i(SI_F32) = 0;
        Kt.i(SI_AGE) = Kt.i(SI_AGE) + 1;
        Kt.i(SI_TSU) = Kt.i(SI_TSU) + 1;
    }
    // Detections (strong_sort.py:43-60): xyxy -> xywh -> tlwh in float64, stored float32; to_xyah in float32 (detection.py:46-53)
    for (int j = tid; j < N; j += BLOCK) {
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        const double cx = (d[0] + d[2]) / 2, cy = (d[1] + d[3]) / 2, bw = d[2] - d[0], bh = d[3] - d[1];
        const float b0 = (float)(cx - bw / 2.), b1 = (float)(cy - bh / 2.), b2 = (float)bw, b3 = (float)bh;
        L.dltwh[j * 4] = b0; L.dltwh[j * 4 + 1] = b1; L.dltwh[j * 4 + 2] = b2; L.dltwh[j * 4 + 3] = b3;
        const float z0 = b0 + b2 / 2, z1 = b1 + b3 / 2, z2 = b2 / b3;
        L.dxyah[j * 4] = z0; L.dxyah[j * 4 + 1] = z1; L.dxyah[j * 4 + 2] = z2; L.dxyah[j * 4 + 3] = b3;
    }
    __syncthreads();
    // per-track gating factors: projected mean + Cholesky of the projected covariance (confidence 0), kalman_filter.py:189-214; gate_cost_matrix
    // hands all N detections to one solve_triangular call per track, whose operation order depends on N == 1 (tlk_strongsort_common.hpp)
    const bool gate_single = N == 1;
    for (int p = tid; p < T; p += BLOCK) {
        const BTrk Kt = trk_at(order[p]);
        double sd[4], Sd[16], Lc[16];
        nsa_sd(Kt.d(SD_MEAN + 3), 0.0, sd);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Sd[i * 4 + j] = Kt.d(SD_COV + i * 8 + j) + (i == j ? sd[i] * sd[i] : 0.0);
        chol4(Sd, 4, Lc);
        double *g = gl + (size_t)p * SGL;
        for (int i = 0; i < 4; ++i) g[i] = Kt.d(SD_MEAN + i);
        for (int q = 0; q < 16; ++q) g[4 + q] = Lc[q];
        gate_row_finish(g, 4, gate_single);
    }
    __syncthreads();
    // ---------------- Tracker._match (tracker.py:152-188) ----------------
    const int nc = block_compact(T, [&](int p) { return trk_at(order[p]).i(SI_STATE) == ST_CONFIRMED; }, [&](int p, int pos) { L.cand[pos] = p; }, L.scan);
    const int nu = block_compact(T, [&](int p) { return trk_at(order[p]).i(SI_STATE) != ST_CONFIRMED; }, [&](int p, int pos) { L.bc[pos] = p; }, L.scan);
    double *cm = ((size_t)nc * N <= (size_t)cost_lds_entries) ? L.cost : (Dv.cost_g + (size_t)s * MAXT * MAXD);
    // gated_metric: cosine gallery minimum + gate_cost_matrix (linear_assignment.py:131-174) + thresholding (:54)
    // one wavefront per track row, lanes over the detections; the row's gate factors are prefetched into registers one row ahead
    // (the flat (row, det) sweep re-read 20 doubles of global memory per entry: as in tlk_bpbss.hip, 70 -> 31 us for 110 x 98)
    {
        const int wv = tid >> 6, lane = tid & 63;
        double gA[20];
        int r = wv;
        if (r < nc) {
            const double *g = gl + (size_t)L.cand[r] * SGL;
#pragma unroll
            for (int q = 0; q < 20; ++q) gA[q] = g[q];
        }
        for (; r < nc; r += NWAVES) {
            const int p = L.cand[r];
            double gB[20];
            const int rn = r + NWAVES;
            if (rn < nc) {
                const double *g = gl + (size_t)L.cand[rn] * SGL;
#pragma unroll
                for (int q = 0; q < 20; ++q) gB[q] = g[q];
            }
            for (int j = lane; j < N; j += WAVE) {
                double c = reid[(size_t)p * MAXD + L.sel[j]];
                const double gd = gating_reg<4>(gA, L.dxyah + j * 4, gate_single);
                if (gd > CHI2_4) c = INFTY_COST;
                c = P.mc_lambda * c + (1 - P.mc_lambda) * gd;
                cm[(size_t)r * N + j] = c > P.max_dist ? P.max_dist + 1e-5 : c;
            }
#pragma unroll
            for (int q = 0; q < 20; ++q) gA[q] = gB[q];
        }
    }
    for (int j = tid; j < N; j += BLOCK) L.um_db[j] = j;
    __syncthreads();
    const McmOut A = min_cost_matching(cm, nc, N, P.max_dist, L.cand, L.um_db, L.m_t, L.m_d, L.um_ta, L.um_da, L);
    for (int p = tid; p < T; p += BLOCK) L.rowf[p] = 0;
    __syncthreads();
    for (int k = tid; k < A.nm; k += BLOCK) L.rowf[L.m_t[k]] = 1;     // by track position
    __syncthreads();
    // unmatched confirmed tracks = list(set(track_indices) - matched) in CPython's set-iteration order (linear_assignment.py:126-127;
    // cm has been consumed: the LDS cost area doubles as the hash-table scratch), split by time_since_update == 1 (tracker.py:174-179)
    int *psw = ((size_t)cost_lds_entries * sizeof(double) >= (size_t)16 * Dv.ps_cap) ? (int *)L.cost : Dv.ps_ws + (size_t)s * 4 * Dv.ps_cap;
    const int n_unm = cascade_unmatched_tracks(L.cand, nc, A.nm, L.tmp, psw, (unsigned)Dv.ps_cap, L);
    const int nb_extra = block_compact(n_unm, [&](int r) { return trk_at(order[L.tmp[r]]).i(SI_TSU) == 1; },
                                       [&](int r, int pos) { L.bc[nu + pos] = L.tmp[r]; }, L.scan);
    const int n_uta = block_compact(n_unm, [&](int r) { return trk_at(order[L.tmp[r]]).i(SI_TSU) != 1; },
                                    [&](int r, int pos) { L.um_t[pos] = L.tmp[r]; }, L.scan);
    const int nb = nu + nb_extra, n_uda = A.n_um_d;
    __syncthreads();
    double *cb = ((size_t)nb * n_uda <= (size_t)cost_lds_entries) ? L.cost : (Dv.cost_g + (size_t)s * MAXT * MAXD);
    for (int e = tid; e < nb * n_uda; e += BLOCK) {          // iou_cost (iou_matching.py:42-82) + thresholding
        const int r = e / n_uda, c = e - r * n_uda;
        const BTrk Kt = trk_at(order[L.bc[r]]);
        double tl[4];
        strk_tlwh(Kt, tl);
        const double v = Kt.i(SI_TSU) > 1 ? INFTY_COST : 1. - iou_f32cand(tl, L.dltwh + L.um_da[c] * 4);
        cb[e] = v > P.max_iou_dist ? P.max_iou_dist + 1e-5 : v;
    }
    __syncthreads();
    const McmOut Bm = min_cost_matching(cb, nb, n_uda, P.max_iou_dist, L.bc, L.um_da, L.m_t + A.nm, L.m_d + A.nm, L.um_tb, L.um_db, L);
    const int nm = A.nm + Bm.nm;
    if (A.err | Bm.err) { if (tid == 0) { hdr[H_ERR] = TLK_EINTERNAL; *out_count = TLK_EINTERNAL; } return; }      // uniform: an assignment solver hit its loop bound
    for (int k = tid; k < Bm.n_um_t; k += BLOCK) L.um_t[n_uta + k] = L.um_tb[k];
    const int n_umt = n_uta + Bm.n_um_t, n_umd = Bm.n_um_d;
    const int *um_d_final = L.um_db;
    __syncthreads();

    // ---------------- Tracker.update (tracker.py:90-104) ----------------
    for (int k = tid; k < nm; k += BLOCK) {                   // Track.update: KF part (track.py:267-296)
        const int j = L.m_d[k];
        const BTrk Kt = trk_at(order[L.m_t[k]]);
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        double mean[8], cov[64], sd[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) mean[q] = Kt.d(SD_MEAN + q);
#pragma unroll
        for (int q = 0; q < 64; ++q) cov[q] = Kt.d(SD_COV + q);
        nsa_sd(mean[3], d[4], sd);
        kf8_update_sd(mean, cov, L.dxyah + j * 4, sd);
#pragma unroll
        for (int q = 0; q < 8; ++q) Kt.d(SD_MEAN + q) = mean[q];
#pragma unroll
        for (int q = 0; q < 64; ++q) Kt.d(SD_COV + q) = cov[q];
        Kt.d(SD_CONF) = d[4]; Kt.i(SI_CLS) = (int)d[5]; Kt.d(SD_TLID) = d[6];
        const int hits = Kt.i(SI_HITS) + 1;
        Kt.i(SI_HITS) = hits; Kt.i(SI_TSU) = 0;
        if (Kt.i(SI_STATE) == ST_TENTATIVE && hits >= P.n_init) Kt.i(SI_STATE) = ST_CONFIRMED;
    }
    // feature EMA + renormalisation in float32 (track.py:282-286): one wavefront per match
    {
        const float a_t = (float)P.ema_alpha, a_d = (float)(1 - P.ema_alpha);
        for (int k = w; k < nm; k += NWAVES) {
            const int di = L.sel[L.m_d[k]];
            float *f = featS + (size_t)order[L.m_t[k]] * D;
            const float *df = in.feat + (dbase + di) * D;
            const float nf = dnormS[di];
            float ss = 0.f;
            for (int e = lane; e < D; e += WAVE) {
                const float x = a_t * f[e], y = a_d * (df[e] / nf);
                const float sm = x + y;
                f[e] = sm; ss += sm * sm;
            }
            const float ns = sqrtf(wave_sum_f32(ss));
            for (int e = lane; e < D; e += WAVE) f[e] = f[e] / ns;
        }
    }
    for (int k = tid; k < n_umt; k += BLOCK) {                // mark_missed (track.py:298-303) + update_kf on the own box (tracker.py:98-101)
        const BTrk Kt = trk_at(order[L.um_t[k]]);
        if (Kt.i(SI_STATE) == ST_TENTATIVE) Kt.i(SI_STATE) = ST_DELETED;
        else if (Kt.i(SI_TSU) > P.max_age) Kt.i(SI_STATE) = ST_DELETED;
        if (P.max_unmatched_preds != 0 && Kt.i(SI_UWA) < 7) {             // track.py:258-265; max_num_updates_wo_assignment = 7 (:76)
            double b[4], z[4], mean[8], cov[64], sd[4];
            strk_tlwh(Kt, b);
            z[0] = b[0] + b[2] / 2; z[1] = b[1] + b[3] / 2; z[2] = b[2] / b[3]; z[3] = b[3];       // detection.py:55-62
            Kt.i(SI_UWA) = Kt.i(SI_UWA) + 1;
#pragma unroll
            for (int q = 0; q < 8; ++q) mean[q] = Kt.d(SD_MEAN + q);
#pragma unroll
            for (int q = 0; q < 64; ++q) cov[q] = Kt.d(SD_COV + q);
            nsa_sd(mean[3], 0.5, sd);
            kf8_update_sd(mean, cov, z, sd);
#pragma unroll
            for (int q = 0; q < 8; ++q) Kt.d(SD_MEAN + q) = mean[q];
#pragma unroll
            for (int q = 0; q < 64; ++q) Kt.d(SD_COV + q) = cov[q];
        }
    }
    __syncthreads();
    // _initiate_track (tracker.py:190-193, Track.__init__ track.py:69-98) in the order of unmatched_detections
    int nfree = hdr[H_NFREE], nextid = hdr[H_NEXTID];
    if (T + n_umd > MAXT) { if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    for (int k = tid; k < n_umd; k += BLOCK) {
        const int j = um_d_final[k];
        const int slot = freestk[nfree - 1 - k];
        order[T + k] = slot;
        const BTrk Kt = trk_at(slot);
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        // kalman_filter.py:55-83 on a float32 measurement: float32 std, squared in float32
        const float m0 = (float)L.dxyah[j * 4], m1 = (float)L.dxyah[j * 4 + 1], m2 = (float)L.dxyah[j * 4 + 2], m3 = (float)L.dxyah[j * 4 + 3];
        const float sd[8] = {(float)(2 * W_POS) * m0, (float)(2 * W_POS) * m1, m2, (float)(2 * W_POS) * m3,
                             (float)(10 * W_VEL) * m0, (float)(10 * W_VEL) * m1, (float)0.1 * m2, (float)(10 * W_VEL) * m3};
#pragma unroll
        for (int q = 0; q < 64; ++q) Kt.d(SD_COV + q) = 0.0;
        Kt.d(SD_MEAN) = m0; Kt.d(SD_MEAN + 1) = m1; Kt.d(SD_MEAN + 2) = m2; Kt.d(SD_MEAN + 3) = m3;
#pragma unroll
        for (int q = 0; q < 4; ++q) Kt.d(SD_MEAN + 4 + q) = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float v = sd[q] * sd[q]; Kt.d(SD_COV + q * 9) = (double)v; }
        Kt.i(SI_F32) = 1;
        Kt.i(SI_TID) = nextid + k; Kt.i(SI_HITS) = 1; Kt.i(SI_AGE) = 1; Kt.i(SI_TSU) = 0; Kt.i(SI_UWA) = 0;
        Kt.i(SI_STATE) = ST_TENTATIVE; Kt.i(SI_GLEN) = 0; Kt.i(SI_GPOS) = 0;
        Kt.i(SI_CLS) = (int)d[5]; Kt.d(SD_CONF) = d[4]; Kt.d(SD_TLID) = d[6];
    }
    for (int k = w; k < n_umd; k += NWAVES) {                 // feature /= np.linalg.norm(feature) (track.py:82-84)
        const int di = L.sel[um_d_final[k]];
        float *f = featS + (size_t)freestk[nfree - 1 - k] * D;
        const float *df = in.feat + (dbase + di) * D;
        const float nf = dnormS[di];
        for (int e = lane; e < D; e += WAVE) f[e] = df[e] / nf;
    }
    __syncthreads();
    nfree -= n_umd; nextid += n_umd; T += n_umd;
    // drop deleted tracks (stable), tracker.py:104
    for (int p = tid; p < T; p += BLOCK) { L.tmp[p] = order[p]; L.rowf[p] = trk_at(order[p]).i(SI_STATE) == ST_DELETED ? 1 : 0; }
    __syncthreads();
    const int kept = block_compact(T, [&](int p) { return L.rowf[p] == 0; }, [&](int p, int pos) { order[pos] = L.tmp[p]; }, L.scan);
    if (kept != T) {
        block_compact(T, [&](int p) { return L.rowf[p] != 0; }, [&](int p, int pos) { freestk[nfree + pos] = L.tmp[p]; }, L.scan);
        nfree += T - kept;
    }
    T = kept;
    __syncthreads();
    if (tid == 0) { hdr[H_NTRK] = T; hdr[H_NFREE] = nfree; hdr[H_NEXTID] = nextid; }
    // metric.partial_fit (tracker.py:106-114, nn_matching.py:124-142): every confirmed track appends its current feature; ring of B rows
    for (int p = w; p < T; p += NWAVES) {
        const int slot = order[p];
        const BTrk Kt = trk_at(slot);
        if (Kt.i(SI_STATE) != ST_CONFIRMED) continue;
        const int pos = Kt.i(SI_GPOS);
        const float *f = featS + (size_t)slot * D;
        float *g = galS + ((size_t)slot * B + pos) * D;
        float ss = 0.f;
        for (int e = lane; e < D; e += WAVE) { const float v = f[e]; g[e] = v; ss += v * v; }
        ss = wave_sum_f32(ss);
        if (lane == 0) {
            gnormS[(size_t)slot * B + pos] = sqrtf(ss);
            Kt.i(SI_GPOS) = pos + 1 == B ? 0 : pos + 1;
            const int gl_ = Kt.i(SI_GLEN);
            if (Dv.unbounded && gl_ == B) hdr[H_ERR] = TLK_ECAPACITY;       // budget=None: the oldest sample may not be overwritten
            Kt.i(SI_GLEN) = gl_ < B ? gl_ + 1 : B;
        }
    }
    __syncthreads();
    // outputs (strong_sort.py:62-79): confirmed tracks with time_since_update <= 1, list order; _tlwh_to_xyxy (:111-122)
    const int nrows = block_compact(T, [&](int p) { const BTrk Kt = trk_at(order[p]); return Kt.i(SI_STATE) == ST_CONFIRMED && Kt.i(SI_TSU) <= 1; },
                                    [&](int p, int pos) {
                                        if (pos >= out_cap) return;
                                        const BTrk Kt = trk_at(order[p]);
                                        double b[4];
                                        strk_tlwh(Kt, b);
                                        int x1 = (int)b[0], x2 = (int)(b[0] + b[2]), y1 = (int)b[1], y2 = (int)(b[1] + b[3]);
                                        x1 = x1 < 0 ? 0 : x1; y1 = y1 < 0 ? 0 : y1;
                                        x2 = x2 > P.img_w - 1 ? P.img_w - 1 : x2; y2 = y2 > P.img_h - 1 ? P.img_h - 1 : y2;
                                        tlk_ssort_row r;
                                        r.det_id = (long long)Kt.d(SD_TLID); r.track_id = Kt.i(SI_TID);
                                        r.ltrb[0] = x1; r.ltrb[1] = y1; r.ltrb[2] = x2; r.ltrb[3] = y2;
                                        r.conf = Kt.d(SD_CONF); r.class_id = Kt.i(SI_CLS); r.time_since_update = Kt.i(SI_TSU);
                                        rows[pos] = r;
                                    }, L.scan);
    if (tid == 0) *out_count = hdr[H_ERR] != 0 ? hdr[H_ERR] : (nrows > out_cap ? TLK_ECAPACITY : nrows);
}

__global__ void ssort_reset_kernel(SsDev D, int stream)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        int *hdr = D.hdr + (size_t)s * H_COUNT;
        for (int k = threadIdx.x; k < D.MAXT; k += blockDim.x) D.freestk[(size_t)s * D.MAXT + k] = D.MAXT - 1 - k;
        if (threadIdx.x == 0) { hdr[H_NTRK] = 0; hdr[H_NEXTID] = 1; hdr[H_NFREE] = D.MAXT; hdr[H_ERR] = 0; }   // _next_id = 1 (tracker.py:51)
    }
}

// Tracker.camera_update -> Track.camera_update (sort/tracker.py:66-68, sort/track.py:221-239) with the ECC estimate passed in:
// one thread per live track; [0, 0, 1] appended, get_matrix's identity fallback (||I - M||_F >= 100), corners through the matrix,
// mean[:4] = [cx, cy, w / h, h] in the mean's own dtype. A launch of its own: the frame kernels are untouched.
struct SsWarp { double m[6]; };
__global__ void ssort_camera_kernel(SsDev D, int stream, SsWarp W)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    double M[6];
    double d2 = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double v = i < 6 ? W.m[i] : (i == 8 ? 1.0 : 0.0); const double e = (i % 4 == 0 ? 1.0 : 0.0) - v; d2 += e * e; }
    const bool ok = sqrt(d2) < 100;
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = ok ? W.m[i] : ((i == 0 || i == 4) ? 1.0 : 0.0);
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        const int T = D.hdr[(size_t)s * H_COUNT + H_NTRK];
        for (int p = threadIdx.x; p < T; p += blockDim.x) {
            const int slot = D.order[(size_t)s * D.MAXT + p];
            double *fd = D.fd + (size_t)s * D.MAXT + slot;
            const bool f32 = D.fi[(size_t)SI_F32 * stride + (size_t)s * D.MAXT + slot] != 0;
            double x1, y1, x2, y2;
            if (f32) {
                float r0 = (float)fd[(size_t)(SD_MEAN + 0) * stride], r1 = (float)fd[(size_t)(SD_MEAN + 1) * stride], r2 = (float)fd[(size_t)(SD_MEAN + 2) * stride];
                const float r3 = (float)fd[(size_t)(SD_MEAN + 3) * stride];
                r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
                x1 = r0; y1 = r1; x2 = (float)(r0 + r2); y2 = (float)(r1 + r3);
            } else {
                double r0 = fd[(size_t)(SD_MEAN + 0) * stride], r1 = fd[(size_t)(SD_MEAN + 1) * stride], r2 = fd[(size_t)(SD_MEAN + 2) * stride];
                const double r3 = fd[(size_t)(SD_MEAN + 3) * stride];
                r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
                x1 = r0; y1 = r1; x2 = r0 + r2; y2 = r1 + r3;
            }
            // numpy's 3 x 3 matrix-vector product: fma(m0, x, m1 * y) + m2 * 1 (oracle/src/ssort.c, identified against numpy itself)
            const double x1_ = fma(M[0], x1, M[1] * y1) + M[2] * 1.0, y1_ = fma(M[3], x1, M[4] * y1) + M[5] * 1.0;
            const double x2_ = fma(M[0], x2, M[1] * y2) + M[2] * 1.0, y2_ = fma(M[3], x2, M[4] * y2) + M[5] * 1.0;
            const double w = x2_ - x1_, h = y2_ - y1_, cx = x1_ + w / 2, cy = y1_ + h / 2;
            const double nm[4] = {cx, cy, w / h, h};
#pragma unroll
            for (int q = 0; q < 4; ++q) fd[(size_t)(SD_MEAN + q) * stride] = f32 ? (double)(float)nm[q] : nm[q];
        }
    }
}

__global__ void ssort_gather_kernel(SsDev D, int stream, long long *ids, double *mean, double *cov, float *feat, long long *state5,
                                    long long *glen, int cap, int *n_out)
{
    const int T = D.hdr[(size_t)stream * H_COUNT + H_NTRK];
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = T;
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int p = blockIdx.x; p < T && p < cap; p += gridDim.x) {
        const int slot = D.order[(size_t)stream * D.MAXT + p];
        const double *fd = D.fd + (size_t)stream * D.MAXT + slot;
        const int *fi = D.fi + (size_t)stream * D.MAXT + slot;
        if (ids && threadIdx.x == 0) ids[p] = fi[(size_t)SI_TID * stride];
        if (mean) for (int k = threadIdx.x; k < 8; k += blockDim.x) mean[(size_t)p * 8 + k] = fd[(size_t)(SD_MEAN + k) * stride];
        if (cov) for (int k = threadIdx.x; k < 64; k += blockDim.x) cov[(size_t)p * 64 + k] = fd[(size_t)(SD_COV + k) * stride];
        if (feat) for (int k = threadIdx.x; k < D.D; k += blockDim.x) feat[(size_t)p * D.D + k] = D.feat[((size_t)stream * D.MAXT + slot) * D.D + k];
        if (state5 && threadIdx.x == 0) {
            state5[(size_t)p * 5] = fi[(size_t)SI_HITS * stride]; state5[(size_t)p * 5 + 1] = fi[(size_t)SI_AGE * stride];
            state5[(size_t)p * 5 + 2] = fi[(size_t)SI_TSU * stride]; state5[(size_t)p * 5 + 3] = fi[(size_t)SI_STATE * stride];
            state5[(size_t)p * 5 + 4] = fi[(size_t)SI_UWA * stride];
        }
        if (glen && threadIdx.x == 0) glen[p] = fi[(size_t)SI_GLEN * stride];
    }
}

}  // namespace

struct tlk_ssort {
    SsDev D; SsP P; int device; size_t smem;
    double *d_dets; float *d_feat; int *d_cnt, *d_ocnt; tlk_ssort_row *d_rows;     // staging of the host-buffer entry point
    int out_cap;
};

static void ss_free(tlk_ssort *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    SsDev &D = h->D;
    void *ptrs[] = {D.fd, D.fi, D.hdr, D.order, D.freestk, D.feat, D.gal, D.gnorm, D.dnorm, D.reid, D.gl, D.cost_g, D.ps_ws, D.big_ws,
                    h->d_dets, h->d_feat, h->d_cnt, h->d_ocnt, h->d_rows};
    for (void *p : ptrs) if (p) hipFree(p);
    delete h;
}

static int ss_launch_frame(tlk_ssort *h, const SsDev &Dv, int n_streams, const SsIn &in, tlk_ssort_row *rows, size_t rows_stream_stride,
                           int out_cap, int *out_counts, size_t oc_stride, hipStream_t st)
{
    hipLaunchKernelGGL(ssort_detnorm_kernel, dim3((Dv.MAXD + NWAVES - 1) / NWAVES, n_streams), dim3(BLOCK), 0, st, Dv, in);
    static const int by_track = [] { const char *e = getenv("TLK_SSORT_COSINE"); return e ? atoi(e) : 1; }();     // 0: the (track, 16 detections) workgroups of r01 / r02
    if (by_track) {
        const dim3 gt(Dv.MAXT, n_streams);
        switch (Dv.D) {
        case 512: hipLaunchKernelGGL((ssort_cosine_track_kernel<32>), gt, dim3(BLOCK), 0, st, Dv, in); break;
        case 256: hipLaunchKernelGGL((ssort_cosine_track_kernel<16>), gt, dim3(BLOCK), 0, st, Dv, in); break;
        case 128: hipLaunchKernelGGL((ssort_cosine_track_kernel<8>), gt, dim3(BLOCK), 0, st, Dv, in); break;
        case 64: hipLaunchKernelGGL((ssort_cosine_track_kernel<4>), gt, dim3(BLOCK), 0, st, Dv, in); break;
        default: hipLaunchKernelGGL((ssort_cosine_track_kernel<2>), gt, dim3(BLOCK), 0, st, Dv, in); break;      // D == 32
        }
    } else {
        const dim3 grid((Dv.MAXD + 15) / 16, Dv.MAXT, n_streams);
        switch (Dv.D) {
        case 512: hipLaunchKernelGGL((ssort_cosine_kernel<32>), grid, dim3(BLOCK), 0, st, Dv, in); break;
        case 256: hipLaunchKernelGGL((ssort_cosine_kernel<16>), grid, dim3(BLOCK), 0, st, Dv, in); break;
        case 128: hipLaunchKernelGGL((ssort_cosine_kernel<8>), grid, dim3(BLOCK), 0, st, Dv, in); break;
        case 64: hipLaunchKernelGGL((ssort_cosine_kernel<4>), grid, dim3(BLOCK), 0, st, Dv, in); break;
        default: hipLaunchKernelGGL((ssort_cosine_kernel<2>), grid, dim3(BLOCK), 0, st, Dv, in); break;      // D == 32
        }
    }
    hipLaunchKernelGGL(ssort_assoc_kernel, dim3(n_streams), dim3(BLOCK), h->smem, st, Dv, h->P, in, rows, rows_stream_stride, out_cap,
                       out_counts, oc_stride);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_ssort_create(const tlk_ssort_params *p, int n_streams, int device, tlk_ssort **out)
{
    if (!p || !out) return fail(TLK_EINVAL, "tlk_ssort_create: null pointer");
    if (n_streams < 1) return fail(TLK_EINVAL, "tlk_ssort_create: n_streams must be >= 1");
    if (p->dim != 32 && p->dim != 64 && p->dim != 128 && p->dim != 256 && p->dim != 512)
        return fail(TLK_EINVAL, "tlk_ssort_create: dim must be one of 32, 64, 128, 256, 512");
    // nn_budget > 0: the reference's budget (a ring of that many rows per track). nn_budget < 0: the reference's budget=None -- every feature is
    // kept -- with room for -nn_budget rows per track in HBM; a track that outgrows it is a loud TLK_ECAPACITY, never a dropped sample.
    if (p->nn_budget == 0 || p->nn_budget > 1024 || p->nn_budget < -65536)
        return fail(TLK_EINVAL, "tlk_ssort_create: nn_budget must be in [1, 1024], or -rows (rows <= 65536) for the reference's unbounded budget=None");
    if (p->img_w < 1 || p->img_h < 1) return fail(TLK_EINVAL, "tlk_ssort_create: image size must be positive");
    const int MAXT = p->max_tracks > 0 ? p->max_tracks : 256, MAXD = p->max_dets > 0 ? p->max_dets : 128;
    // capacity = allocation size (r04): LDS tiers while the scene fits, HBM lists beyond (ssort_assoc_kernel); the gallery is
    // max_tracks x |nn_budget| x dim floats per stream, the caller's choice of HBM
    if (MAXT > 16384 || MAXD > 1024) return fail(TLK_ECAPACITY, "tlk_ssort_create: max_tracks <= 16384 and max_dets <= 1024");
    if ((unsigned long long)n_streams * MAXT * (unsigned long long)(p->nn_budget > 0 ? p->nn_budget : -p->nn_budget) > 0x7fffffffULL)
        return fail(TLK_ECAPACITY, "tlk_ssort_create: n_streams x max_tracks x gallery rows must stay below 2^31");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_ssort_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_ssort_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_ssort *h = new tlk_ssort();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->P = SsP{p->max_dist, p->max_iou_dist, p->mc_lambda, p->ema_alpha, p->min_confidence, p->max_age, p->max_unmatched_preds, p->n_init,
               p->wrapper_mode, p->img_w, p->img_h};
    SsDev &D = h->D;
    D.S = n_streams; D.MAXT = MAXT; D.MAXD = MAXD; D.D = p->dim; D.B = p->nn_budget > 0 ? p->nn_budget : -p->nn_budget; D.unbounded = p->nn_budget < 0;
    const size_t budget = 160 * 1024 - 256;
    D.lds_bytes = (int)(budget & ~(size_t)15);
    h->smem = (size_t)D.lds_bytes;
    D.big_stride = (blds_fixed(MAXT, MAXD) + 255) & ~(size_t)255;
    const size_t slots = (size_t)n_streams * MAXT;
    h->out_cap = MAXT;
#define SS_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { ss_free(h); return fail(TLK_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    SS_ALLOC(D.fd, sizeof(double) * SD_COUNT * slots);
    SS_ALLOC(D.fi, sizeof(int) * SI_COUNT * slots);
    SS_ALLOC(D.hdr, sizeof(int) * H_COUNT * n_streams);
    SS_ALLOC(D.order, sizeof(int) * slots);
    SS_ALLOC(D.freestk, sizeof(int) * slots);
    SS_ALLOC(D.feat, sizeof(float) * D.D * slots);
    SS_ALLOC(D.gal, sizeof(float) * D.D * D.B * slots);
    SS_ALLOC(D.gnorm, sizeof(float) * D.B * slots);
    SS_ALLOC(D.dnorm, sizeof(float) * (size_t)n_streams * MAXD);
    SS_ALLOC(D.reid, sizeof(double) * slots * MAXD);
    SS_ALLOC(D.gl, sizeof(double) * SGL * slots);
    SS_ALLOC(D.cost_g, sizeof(double) * slots * MAXD);
    SS_ALLOC(D.big_ws, D.big_stride * (size_t)n_streams);
    D.ps_cap = (int)pyset::table_capacity((unsigned)MAXT);
    SS_ALLOC(D.ps_ws, sizeof(int) * 4 * (size_t)D.ps_cap * n_streams);
    SS_ALLOC(h->d_dets, sizeof(double) * 7 * MAXD);
    SS_ALLOC(h->d_feat, sizeof(float) * D.D * MAXD);
    SS_ALLOC(h->d_cnt, sizeof(int));
    SS_ALLOC(h->d_ocnt, sizeof(int));
    SS_ALLOC(h->d_rows, sizeof(tlk_ssort_row) * h->out_cap);
#undef SS_ALLOC
    hipError_t e = hipMemset(D.fd, 0, sizeof(double) * SD_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.fi, 0, sizeof(int) * SI_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.feat, 0, sizeof(float) * D.D * slots);
    if (e == hipSuccess) e = hipMemset(D.reid, 0, sizeof(double) * slots * MAXD);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)ssort_assoc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) { ss_free(h); return fail(TLK_EHIP, std::string("tlk_ssort_create: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(ssort_reset_kernel, dim3(n_streams < 256 ? n_streams : 256), dim3(BLOCK), 0, 0, D, -1);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { ss_free(h); return fail(TLK_EHIP, std::string("tlk_ssort_create: ") + hipGetErrorString(e)); }
    *out = h;
    return TLK_OK;
}

extern "C" int tlk_ssort_destroy(tlk_ssort *h) { ss_free(h); return TLK_OK; }

extern "C" int tlk_ssort_reset(tlk_ssort *h, int stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_ssort_reset: null handle");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ssort_reset: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(ssort_reset_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, 0, h->D, stream);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipStreamSynchronize(0));
    return TLK_OK;
}

extern "C" int tlk_ssort_camera_update(tlk_ssort *h, int stream, const double *warp6, void *hip_stream)
{
    if (!h || !warp6) return fail(TLK_EINVAL, "tlk_ssort_camera_update: null pointer");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ssort_camera_update: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    SsWarp W;
    for (int i = 0; i < 6; ++i) W.m[i] = warp6[i];
    hipLaunchKernelGGL(ssort_camera_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, (hipStream_t)hip_stream, h->D, stream, W);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_ssort_update_dev(tlk_ssort *h, const double *dets_dev, const float *feat_dev, const int32_t *counts_dev, int n_frames,
                                    tlk_ssort_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_ssort_update_dev: null handle");
    if (n_frames < 0 || out_cap < 0) return fail(TLK_EINVAL, "tlk_ssort_update_dev: negative size");
    if (n_frames == 0) return TLK_OK;
    if (!dets_dev || !feat_dev || !counts_dev || !rows_dev || !out_counts_dev) return fail(TLK_EINVAL, "tlk_ssort_update_dev: null pointer");
    TLK_HIP(hipSetDevice(h->device));
    const SsDev &D = h->D;
    for (int f = 0; f < n_frames; ++f) {
        const size_t off = (size_t)f * D.MAXD;          // dets index within a stream block of n_frames*MAXD
        SsIn in;
        in.dets = dets_dev + off * 7; in.feat = feat_dev + off * D.D; in.counts = (const int *)counts_dev + f;
        in.stream_stride_dets = (size_t)n_frames * D.MAXD; in.count_stride = (size_t)n_frames;
        const int rc = ss_launch_frame(h, D, D.S, in, rows_dev + (size_t)f * out_cap, (size_t)n_frames * out_cap, out_cap,
                                       (int *)out_counts_dev + f, (size_t)n_frames, (hipStream_t)hip_stream);
        if (rc != TLK_OK) return rc;
    }
    return TLK_OK;
}

extern "C" int tlk_ssort_update(tlk_ssort *h, int stream, const double *dets, const float *feat, int n, tlk_ssort_row *rows, int cap, int *n_out)
{
    if (!h || !n_out) return fail(TLK_EINVAL, "tlk_ssort_update: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ssort_update: stream out of range");
    if (n < 0 || (n > 0 && (!dets || !feat))) return fail(TLK_EINVAL, "tlk_ssort_update: bad detections");
    if (n > h->D.MAXD) return fail(TLK_ECAPACITY, "tlk_ssort_update: more detections than max_dets");
    TLK_HIP(hipSetDevice(h->device));
    hipStream_t st = 0;
    if (n) {
        TLK_HIP(hipMemcpyAsync(h->d_dets, dets, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_feat, feat, sizeof(float) * (size_t)h->D.D * n, hipMemcpyHostToDevice, st));
    }
    TLK_HIP(hipMemcpyAsync(h->d_cnt, &n, sizeof(int), hipMemcpyHostToDevice, st));
    SsDev V = h->D;        // single-stream view: shift per-stream bases, keep strides
    const size_t sl = (size_t)stream * V.MAXT;
    V.fd += sl; V.fi += sl; V.hdr += (size_t)stream * H_COUNT; V.order += sl; V.freestk += sl;
    V.feat += sl * V.D; V.gal += sl * V.B * V.D; V.gnorm += sl * V.B; V.dnorm += (size_t)stream * V.MAXD;
    V.reid += sl * V.MAXD; V.gl += sl * SGL; V.cost_g += sl * V.MAXD; V.ps_ws += (size_t)stream * 4 * V.ps_cap; V.big_ws += (size_t)stream * V.big_stride;
    SsIn in;
    in.dets = h->d_dets; in.feat = h->d_feat; in.counts = h->d_cnt; in.stream_stride_dets = 0; in.count_stride = 0;
    const int rc = ss_launch_frame(h, V, 1, in, h->d_rows, 0, h->out_cap, h->d_ocnt, 0, st);
    if (rc != TLK_OK) return rc;
    int rows_n = 0;
    TLK_HIP(hipMemcpyAsync(&rows_n, h->d_ocnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TLK_HIP(hipStreamSynchronize(st));
    if (rows_n < 0) return fail_stream(rows_n, "tlk_ssort_update");
    if (rows_n > cap) return fail(TLK_ECAPACITY, "tlk_ssort_update: output buffer too small");
    if (rows_n) TLK_HIP(hipMemcpy(rows, h->d_rows, sizeof(tlk_ssort_row) * rows_n, hipMemcpyDeviceToHost));
    *n_out = rows_n;
    return TLK_OK;
}

extern "C" int tlk_ssort_get_tracks(tlk_ssort *h, int stream, int64_t *ids, double *mean, double *cov, float *feat, int64_t *state5,
                                    int64_t *gallery_rows, int cap, int *n_tracks)
{
    if (!h || !n_tracks) return fail(TLK_EINVAL, "tlk_ssort_get_tracks: null pointer");
    if (stream < 0 || stream >= h->D.S || cap < 0) return fail(TLK_EINVAL, "tlk_ssort_get_tracks: bad argument");
    TLK_HIP(hipSetDevice(h->device));
    const size_t c = cap > 0 ? cap : 1;
    long long *d_ids = nullptr, *d_st = nullptr, *d_gl = nullptr; double *d_mean = nullptr, *d_cov = nullptr; float *d_feat = nullptr; int *d_n = nullptr;
    TLK_HIP(hipMalloc((void **)&d_ids, sizeof(long long) * c)); TLK_HIP(hipMalloc((void **)&d_st, sizeof(long long) * 5 * c));
    TLK_HIP(hipMalloc((void **)&d_gl, sizeof(long long) * c)); TLK_HIP(hipMalloc((void **)&d_mean, sizeof(double) * 8 * c));
    TLK_HIP(hipMalloc((void **)&d_cov, sizeof(double) * 64 * c)); TLK_HIP(hipMalloc((void **)&d_feat, sizeof(float) * h->D.D * c));
    TLK_HIP(hipMalloc((void **)&d_n, sizeof(int)));
    hipLaunchKernelGGL(ssort_gather_kernel, dim3(64), dim3(64), 0, 0, h->D, stream, d_ids, d_mean, d_cov, d_feat, d_st, d_gl, cap, d_n);
    int n = 0;
    hipError_t e = hipMemcpy(&n, d_n, sizeof(int), hipMemcpyDeviceToHost);
    const int m = n < cap ? n : cap;
    if (e == hipSuccess && m > 0) {
        if (ids) e = hipMemcpy(ids, d_ids, sizeof(long long) * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && mean) e = hipMemcpy(mean, d_mean, sizeof(double) * 8 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && cov) e = hipMemcpy(cov, d_cov, sizeof(double) * 64 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && feat) e = hipMemcpy(feat, d_feat, sizeof(float) * h->D.D * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && state5) e = hipMemcpy(state5, d_st, sizeof(long long) * 5 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && gallery_rows) e = hipMemcpy(gallery_rows, d_gl, sizeof(long long) * m, hipMemcpyDeviceToHost);
    }
    hipFree(d_ids); hipFree(d_st); hipFree(d_gl); hipFree(d_mean); hipFree(d_cov); hipFree(d_feat); hipFree(d_n);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_ssort_get_tracks: ") + hipGetErrorString(e));
    *n_tracks = n;
    return TLK_OK;
}
