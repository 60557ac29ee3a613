// tlk_bpbss.hip -- BPBReID-StrongSORT on gfx950: three launches per frame for ALL streams of a bank.
//
//   1. partnorm_kernel   one wavefront per (row, part) embedding: L2 norm + squared norm of the normalised vector
//   2. partdist_kernel   f32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains): one workgroup per 16x16 tile of
//                        the track x detection matrix, one wavefront per body part; masked mean over common parts
//   3. bpbss_assoc_kernel one 256-thread workgroup per stream: KF predict, KF gating, two-stage (or BoT-SORT style)
//                        assignment with the scipy-identical wavefront LSA, KF update, visibility-aware EMA of the
//                        part embeddings, track lifecycle, output rows.
// fp64 association math in the reference's operation order (-ffp-contract=off); fp32 embeddings like torch.
// State: field-major SoA per slot; list order kept as an indirection (order[]), so deaths compact 4-byte indices.
#include "tlk_common.hpp"
#include "tlk_strongsort_common.hpp"

using namespace tlk;

namespace {


enum : int { BD_MEAN = 0, BD_COV = 8, BD_PRED = 72, BD_MDIST = 76, BD_KP = 77, BD_COUNT = 77 + 51 };   // KP: last_detection.keypoints (17,3)
constexpr int GLN = 24;   // per-track scratch: projected mean (4), Cholesky factor (16), OKS scale, OKS visible count
enum : int { BI_TID = 0, BI_HITS, BI_AGE, BI_TSU, BI_STATE, BI_MNAME, BI_PVALID, BI_COUNT };
enum : int { H_NTRK = 0, H_NEXTID, H_NFREE, H_ERR, H_COUNT = 8 };
enum : int { ST_TENTATIVE = 0, ST_CONFIRMED = 1, ST_DELETED = 2 };

struct BpbDev {
    double *fd;              // BD_COUNT x S x MAXT
    int *fi;                 // BI_COUNT x S x MAXT
    long long *detid;        // S x MAXT   last_detection.id per slot
    int *hdr, *order, *freestk;
    float *feat;             // S x MAXT x K x D   by slot
    unsigned char *fvis;     // S x MAXT x K       by slot
    float *tnorm;            // S x MAXT x K x 2   by list position: (norm, sum of squares of the normalised vector)
    float *dnorm;            // S x MAXD x K x 2   by input detection index
    double *reid;            // S x MAXT x MAXD    part-based distance, row = list position, col = input detection index
    double *gl;              // S x MAXT x GLN     per track: projected mean (4) + Cholesky factor (16) for gating, OKS scale + count
    double *cost_g;          // S x MAXT x MAXD    cost-matrix spill
    int *ema_list, *ema_n;   // S x MAXD x 2 (slot, input detection index) / S: matches of the frame, consumed by bpbss_ema_kernel
    long long *prof;         // optional S x 16 phase accumulators in 100 MHz ticks (diagnostics: TLK_BPBSS_PROF)
    int *ps_ws;              // S x 4 x ps_cap      hash tables of the set-order emulation when they do not fit the LDS cost area
    unsigned char *big_ws;   // S x big_stride      list / solver work area of the big-scene tier (scenes beyond what the LDS lists hold)
    size_t big_stride;
    int S, MAXT, MAXD, K, D, lds_bytes, ps_cap;
};

struct BpbP {
    double ema_alpha, mc_lambda, max_dist, max_iou_distance, min_conf, gating_thres_factor, w_kfgd, w_reid, w_st;
    int max_age, n_init, only_position, max_pred, strategy, wrapper_mode, motion;
    double max_oks_distance;
};

struct FrameIn {            // per-(stream, frame) strides in elements
    const long long *ids; const double *ltwh; const float *emb; const unsigned char *vis; const double *conf;
    const double *kps;       // (.., 17, 3) COCO keypoints per detection or nullptr
    const int *counts; size_t stream_stride_dets; size_t count_stride;   // dets index = s*stream_stride_dets + i
};

// ------------------------------------------------------------------------------------------------ norms
__global__ void __launch_bounds__(BLOCK) partnorm_kernel(BpbDev Dv, FrameIn in, int wrapper_mode)
{
    const int s = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int K = Dv.K, D = Dv.D;
    const int T = Dv.hdr[(size_t)s * H_COUNT + H_NTRK];
    const int N = in.counts[(size_t)s * in.count_stride];
    if (N <= 0 || N > Dv.MAXD) return;
    for (int vec = blockIdx.x * NWAVES + w; vec < (T + N) * K; vec += gridDim.x * NWAVES) {        // [0, (T+N)*K): the grid is bounded, the capacity is not
    const bool is_trk = vec < T * K;
    const int row = is_trk ? vec / K : (vec - T * K) / K, k = is_trk ? vec % K : (vec - T * K) % K;
    const float *x = is_trk ? Dv.feat + (((size_t)s * Dv.MAXT + Dv.order[(size_t)s * Dv.MAXT + row]) * K + k) * D
                            : in.emb + (((size_t)s * in.stream_stride_dets + row) * K + k) * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += WAVE) { const float v = x[d]; ss += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    float nrm = sqrtf(ss);
    if (nrm < 1e-12f) nrm = 1e-12f;                       // F.normalize(eps=1e-12)
    float s2 = 0.f;
    for (int d = lane; d < D; d += WAVE) { const float v = x[d] / nrm; s2 += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
    if (lane == 0) {
        float *o = is_trk ? Dv.tnorm + (((size_t)s * Dv.MAXT + row) * K + k) * 2 : Dv.dnorm + (((size_t)s * Dv.MAXD + row) * K + k) * 2;
        o[0] = nrm; o[1] = s2;
    }
    }
}

// ------------------------------------------------------------------------------------------------ MFMA distance
typedef float f32x4 __attribute__((ext_vector_type(4)));

// blockDim = 64*K. Tile (blockIdx.y = track tile, blockIdx.x = det tile) of stream blockIdx.z.
__global__ void __launch_bounds__(512) partdist_kernel(BpbDev Dv, FrameIn in)
{
    __shared__ float s_dist[8][256];
    __shared__ unsigned char s_valid[8][256];
    const int s = blockIdx.z, k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int K = Dv.K, D = Dv.D;
    const int T = Dv.hdr[(size_t)s * H_COUNT + H_NTRK];
    const int N = in.counts[(size_t)s * in.count_stride];
    if (N <= 0 || N > Dv.MAXD) return;
    const int tiles_n = (N + 15) >> 4, tiles = ((T + 15) >> 4) * tiles_n;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {          // bounded grid over the (track tile, detection tile) pairs of the frame
    const int t0 = (tile / tiles_n) * 16, n0 = (tile % tiles_n) * 16;
    __syncthreads();                                                        // s_dist / s_valid of the previous tile have been read
    const int i = lane & 15, g = lane >> 4;
    const int tp = min(t0 + i, T - 1), dn = min(n0 + i, N - 1);
    const int slot = Dv.order[(size_t)s * Dv.MAXT + tp];
    const float *qrow = Dv.feat + (((size_t)s * Dv.MAXT + slot) * K + k) * D;
    const float *grow = in.emb + (((size_t)s * in.stream_stride_dets + dn) * K + k) * D;
    const float nq = Dv.tnorm[(((size_t)s * Dv.MAXT + tp) * K + k) * 2];
    const float ng = Dv.dnorm[(((size_t)s * Dv.MAXD + dn) * K + k) * 2];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float rq = 1.f / nq, rg = 1.f / ng;      // reciprocal multiply: the loop was VALU-bound on fp32 divisions
#pragma unroll 4
    for (int c = 0; c < D; c += 16) {
        const float4 a = *reinterpret_cast<const float4 *>(qrow + c + 4 * g);
        const float4 b = *reinterpret_cast<const float4 *>(grow + c + 4 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x * rq, b.x * rg, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y * rq, b.y * rg, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z * rq, b.z * rg, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w * rq, b.w * rg, acc, 0, 0, 0);
    }
    // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int cn = min(n0 + i, N - 1);
    const float gs = Dv.dnorm[(((size_t)s * Dv.MAXD + cn) * K + k) * 2 + 1];
    const bool gv = in.vis[((size_t)s * in.stream_stride_dets + cn) * K + k] != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = g * 4 + r;
        const int rt = min(t0 + row, T - 1);
        const float qs = Dv.tnorm[(((size_t)s * Dv.MAXT + rt) * K + k) * 2 + 1];
        const bool qv = Dv.fvis[((size_t)s * Dv.MAXT + Dv.order[(size_t)s * Dv.MAXT + rt]) * K + k] != 0;
        float d2 = qs - 2 * acc[r] + gs;
        d2 = d2 < 0.f ? 0.f : d2;
        const bool valid = qv && gv;
        s_dist[k][row * 16 + i] = valid ? sqrtf(d2) : 0.f;
        s_valid[k][row * 16 + i] = valid ? 1 : 0;
    }
    __syncthreads();
    if (k == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = e * 64 + lane;
            const int row = idx >> 4, col = idx & 15;
            float sum = 0.f; int cnt = 0;
            for (int p = 0; p < K; ++p) { sum += s_dist[p][idx]; cnt += s_valid[p][idx]; }
            const float pair = cnt ? sum / (float)cnt : -1.0f;
            if (t0 + row < T && n0 + col < N)
                Dv.reid[((size_t)s * Dv.MAXT + t0 + row) * Dv.MAXD + n0 + col] = (double)(pair / 2);
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------ KF8 in registers
__device__ __forceinline__ void kf8_predict(double (&mean)[8], double (&cov)[64])     // kalman_filter.py:85-119
{
    const double sp = W_POS * mean[3], sv = W_VEL * mean[3];
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
    for (int i = 0; i < 4; ++i) { cov[i * 9] += sp * sp; cov[(4 + i) * 9] += sv * sv; mean[i] = mean[i] + mean[i + 4]; }
}

// sort/oks_matching.py:7-92. Scale of the track's keypoints (visible / all, axis-aligned and 45-degree rotated extents).
__device__ __constant__ double KAPPA[17] = {0.026, 0.025, 0.025, 0.035, 0.035, 0.079, 0.079, 0.072, 0.072, 0.062, 0.062, 0.107, 0.107, 0.087, 0.087, 0.089, 0.089};
template <class KP>
__device__ double oks_scale(KP kp, int *nvis_out)
{
    const double c45 = 0.7071067811865476, s45 = 0.7071067811865475;       // np.cos / np.sin(np.deg2rad(45))
    double tl[2] = {INFINITY, INFINITY}, br[2] = {-INFINITY, -INFINITY}, ttl[2] = {INFINITY, INFINITY}, tbr[2] = {-INFINITY, -INFINITY};
    double tl4[2] = {INFINITY, INFINITY}, br4[2] = {-INFINITY, -INFINITY}, ttl4[2] = {INFINITY, INFINITY}, tbr4[2] = {-INFINITY, -INFINITY};
    int nvis = 0;
    for (int k = 0; k < 17; ++k) {
        const double x = kp(3 * k), y = kp(3 * k + 1);
        const double r[2] = {c45 * x + (-s45) * y, s45 * x + c45 * y};
        const double p[2] = {x, y};
        const bool v = kp(3 * k + 2) > 0.0;
        nvis += v ? 1 : 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (p[a] < ttl[a]) ttl[a] = p[a];
            if (p[a] > tbr[a]) tbr[a] = p[a];
            if (r[a] < ttl4[a]) ttl4[a] = r[a];
            if (r[a] > tbr4[a]) tbr4[a] = r[a];
            if (v) {
                if (p[a] < tl[a]) tl[a] = p[a];
                if (p[a] > br[a]) br[a] = p[a];
                if (r[a] < tl4[a]) tl4[a] = r[a];
                if (r[a] > br4[a]) br4[a] = r[a];
            }
        }
    }
    *nvis_out = nvis;
    const double area = (br[0] - tl[0]) * (br[1] - tl[1]), total_area = (tbr[0] - ttl[0]) * (tbr[1] - ttl[1]);
    const double area45 = (br4[0] - tl4[0]) * (br4[1] - tl4[1]), total45 = (tbr4[0] - ttl4[0]) * (tbr4[1] - ttl4[1]);
    const double f1 = area > 0.1 ? total_area / area : INFINITY, f2 = area45 > 0.1 ? total45 / area45 : INFINITY;
    const double factor = sqrt(f1 < f2 ? f1 : f2);
    const double fc = factor < 5.0 ? factor : 5.0;
    double scale = sqrt(area) * fc;
    if (scale < 0.1) scale = NAN;
    return scale;
}
template <class KP>
__device__ double oks_one(KP kp, double scale, int nvis, const double *c)
{
    double sum = 0;
    for (int k = 0; k < 17; ++k) {
        const double dx = kp(3 * k) - c[3 * k], dy = kp(3 * k + 1) - c[3 * k + 1];
        const double d = sqrt(dx * dx + dy * dy);
        const double e = exp(-(d * d) / (2 * (scale * scale) * (KAPPA[k] * KAPPA[k])));
        sum += e * (kp(3 * k + 2) > 0.0 ? 1.0 : 0.0);
    }
    return sum / (double)nvis;
}

__device__ __forceinline__ void trk_ltwh(const BTrk &T, double *o)     // track.py:97-100
{
    const double x = T.d(BD_MEAN), y = T.d(BD_MEAN + 1), a = T.d(BD_MEAN + 2), h = T.d(BD_MEAN + 3);
    const double w = a * h;
    o[0] = x - w / 2; o[1] = y - h / 2; o[2] = w; o[3] = h;
}

// ------------------------------------------------------------------------------------------------ association kernel
__global__ void __launch_bounds__(BLOCK, 1)
bpbss_assoc_kernel(BpbDev Dv, BpbP P, FrameIn in, tlk_bpbss_row *__restrict__ rows_all, size_t rows_stream_stride, int out_cap,
                   int *__restrict__ out_counts, size_t oc_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int MAXT = Dv.MAXT, MAXD = Dv.MAXD, K = Dv.K, D = Dv.D;
    const size_t FD = (size_t)K * D;
    int *hdr = Dv.hdr + (size_t)s * H_COUNT;
    // List / solver work area of this frame.  The lists are sized by (tracks before the frame + detections, detections): the SMALLEST tier that
    // holds them is used -- 256 x 128 or 1024 x 256 carved out of LDS (what r01-r03 ran on; the rest of the LDS is the cost matrix), or, for a
    // scene beyond that, the bank's full capacity carved out of HBM (r04: the reference's lists simply grow, tracker.py:427-441; here the
    // arrays are allocated at capacity in the 288 GB of HBM and only this frame's lists move out of LDS).  Same code, generic pointers.
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
        if (ct) {
            bcarve(smem, ct, cd, L);
            cost_lds_entries = (int)(((size_t)Dv.lds_bytes - blds_fixed(ct, cd)) / sizeof(double));
        } else {
            bcarve(Dv.big_ws + (size_t)s * Dv.big_stride, MAXT, MAXD, L);           // (L.cost is never used with cost_lds_entries = 0: matrices go to cost_g)
        }
    }
    int *order = Dv.order + (size_t)s * MAXT;
    int *freestk = Dv.freestk + (size_t)s * MAXT;
    const size_t stride = (size_t)Dv.S * MAXT;
    auto trk_at = [&](int slot) { BTrk T; T.fd = Dv.fd + (size_t)s * MAXT + slot; T.fi = Dv.fi + (size_t)s * MAXT + slot; T.stride = stride; return T; };
    float *featS = Dv.feat + (size_t)s * MAXT * FD;
    unsigned char *fvisS = Dv.fvis + (size_t)s * MAXT * K;
    long long *detidS = Dv.detid + (size_t)s * MAXT;
    const double *reid = Dv.reid + (size_t)s * MAXT * MAXD;
    double *gl = Dv.gl + (size_t)s * MAXT * GLN;
    tlk_bpbss_row *rows = rows_all + (size_t)s * rows_stream_stride;
    int *out_count = out_counts + (size_t)s * oc_stride;
    const size_t dbase = (size_t)s * in.stream_stride_dets;
    const int n_in = in.counts[(size_t)s * in.count_stride];

    if (tid == 0) Dv.ema_n[s] = 0;                       // nothing for bpbss_ema_kernel unless matches are published below
    if (hdr[H_ERR] != 0) { if (tid == 0) *out_count = hdr[H_ERR]; return; }
    if (n_in > MAXD || n_in < 0) { if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    if (P.wrapper_mode && n_in == 0) { if (tid == 0) *out_count = 0; return; }      // bpbreid_strong_sort_api.py:103-104
    long long t_prev = 0;
#define BPB_PROF(i) do { if (Dv.prof) { __syncthreads(); if (tid == 0) { const long long t_ = wall_clock64(); Dv.prof[(size_t)s * 16 + (i)] += t_ - t_prev; t_prev = t_; } } } while (0)
    long long c_start = 0, w_start = 0;
    if (Dv.prof && tid == 0) { t_prev = wall_clock64(); w_start = t_prev; c_start = clock64(); }

    // filter_detections (strong_sort.py:143-147)
    const int N = block_compact(n_in, [&](int i) { return in.conf[dbase + i] > P.min_conf; }, [&](int i, int pos) { L.sel[pos] = i; }, L.scan);
    int T = hdr[H_NTRK];
    __syncthreads();
    // Tracker.predict (tracker.py:92-99, track.py:128-135)
    for (int p = tid; p < T; p += BLOCK) {
        const BTrk Kt = trk_at(order[p]);
        const int tsu = Kt.i(BI_TSU);
        if (tsu < P.max_pred) {
            double mean[8], cov[64];
#pragma unroll
            for (int k = 0; k < 8; ++k) mean[k] = Kt.d(BD_MEAN + k);
#pragma unroll
            for (int k = 0; k < 64; ++k) cov[k] = Kt.d(BD_COV + k);
            kf8_predict(mean, cov);
#pragma unroll
            for (int k = 0; k < 8; ++k) Kt.d(BD_MEAN + k) = mean[k];
#pragma unroll
            for (int k = 0; k < 64; ++k) Kt.d(BD_COV + k) = cov[k];
        }
        Kt.i(BI_AGE) = Kt.i(BI_AGE) + 1;
        Kt.i(BI_TSU) = tsu + 1;
    }
    __syncthreads();
    BPB_PROF(0);                                          // filter + predict
    if (N > 0) {
        const int gdim = P.only_position ? 2 : 4;
        const bool use_oks = P.motion == 1 && in.kps != nullptr;
        const double motion_max = P.motion == 1 ? P.max_oks_distance : P.max_iou_distance;
        // spatio-temporal cost of track position p vs filtered detection j: 1 - IoU (iou_matching.py:42-78) or 1 - OKS (oks_matching.py:95-128)
        auto motion_cost = [&](int p, int j) {
            const BTrk Kt = trk_at(order[p]);
            if (use_oks) {
                const double *g = gl + (size_t)p * GLN;
                return 1.0 - oks_one([&](int q) { return Kt.d(BD_KP + q); }, g[20], (int)g[21], in.kps + (dbase + L.sel[j]) * 51);
            }
            double tl[4];
            trk_ltwh(Kt, tl);
            return 1. - iou_ltwh(tl, L.dltwh + j * 4);
        };
        for (int j = tid; j < N; j += BLOCK) {              // detection.py:31-58
            const double *b = in.ltwh + (dbase + L.sel[j]) * 4;
            L.dltwh[j * 4] = b[0]; L.dltwh[j * 4 + 1] = b[1]; L.dltwh[j * 4 + 2] = b[2]; L.dltwh[j * 4 + 3] = b[3];
            L.dxyah[j * 4] = b[0] + b[2] / 2; L.dxyah[j * 4 + 1] = b[1] + b[3] / 2; L.dxyah[j * 4 + 2] = b[2] / b[3]; L.dxyah[j * 4 + 3] = b[3];
            L.d_mname[j] = 0; L.d_mdist[j] = 0.0;
        }
        // per-track gating factors: projected mean + Cholesky of the projected covariance (conf = 0). Every gating_distance call of the frame
        // (gate_cost_matrix / _full_cost_metric through the single-level matching_cascade) sees all N filtered detections: one
        // solve_triangular call per track whose operation order depends on N == 1 (tlk_strongsort_common.hpp)
        const bool gate_single = N == 1;
        for (int p = tid; p < T; p += BLOCK) {
            const BTrk Kt = trk_at(order[p]);
            const double h = Kt.d(BD_MEAN + 3);
            const double sstd = (1 - 0.0) * (W_POS * h);
            double Sd[16], Lc[16];
            for (int i = 0; i < gdim; ++i)
                for (int j = 0; j < gdim; ++j) Sd[i * gdim + j] = Kt.d(BD_COV + i * 8 + j) + (i == j ? sstd * sstd : 0.0);
            chol4(Sd, gdim, Lc);
            double *g = gl + (size_t)p * GLN;
            for (int i = 0; i < 4; ++i) g[i] = Kt.d(BD_MEAN + i);
            for (int q = 0; q < 16; ++q) g[4 + q] = Lc[q];
            gate_row_finish(g, gdim, gate_single);
            if (use_oks) { int nv; g[20] = oks_scale([&](int q) { return Kt.d(BD_KP + q); }, &nv); g[21] = (double)nv; }
        }
        __syncthreads();
        BPB_PROF(1);                                      // detection + gate preparation
        int nm = 0, n_umt = 0, n_umd = 0, lsa_err = 0;
        int *um_d_final = L.um_db;
        if (P.strategy == 0) {
            // ---------------- strong_sort_matching (tracker.py:242-333) ----------------
            const int nc = block_compact(T, [&](int p) { return trk_at(order[p]).i(BI_STATE) == ST_CONFIRMED; },
                                         [&](int p, int pos) { L.cand[pos] = p; }, L.scan);
            const int nu = block_compact(T, [&](int p) { return trk_at(order[p]).i(BI_STATE) != ST_CONFIRMED; },
                                         [&](int p, int pos) { L.bc[pos] = p; }, L.scan);
            double *cm = ((size_t)nc * N <= (size_t)cost_lds_entries) ? L.cost : (Dv.cost_g + (size_t)s * MAXT * MAXD);
            // gate_cost_matrix (linear_assignment.py:132-175) + thresholding (:54-55). One wavefront per track row: the row's gate
            // (projected mean + Cholesky factor, 20 doubles in HBM) sits in registers and the NEXT row's is already in flight while
            // this one is computed -- the entry-per-thread loop re-read it for every entry and ran at the latency of ~25 dependent
            // global loads per entry (70 us for 110 x 98)
            {
                const int wv = tid >> 6, lane = tid & 63;
                const double chi = CHI2INV95[gdim];
                double gA[20];
                int r = wv;
                if (r < nc) {
                    const double *g = gl + (size_t)L.cand[r] * GLN;
#pragma unroll
                    for (int q = 0; q < 20; ++q) gA[q] = g[q];
                }
                for (; r < nc; r += NWAVES) {
                    const int p = L.cand[r];
                    double gB[20];
                    const int rn = r + NWAVES;
                    if (rn < nc) {
                        const double *g = gl + (size_t)L.cand[rn] * GLN;
#pragma unroll
                        for (int q = 0; q < 20; ++q) gB[q] = g[q];
                    }
                    for (int j = lane; j < N; j += WAVE) {
                        double c = reid[(size_t)p * MAXD + L.sel[j]];
                        const double gd = gdim == 4 ? gating_reg<4>(gA, L.dxyah + j * 4, gate_single) : gating_reg<2>(gA, L.dxyah + j * 4, gate_single);
                        if (gd > chi) c = INFTY_COST;
                        c = P.mc_lambda * c + (1 - P.mc_lambda) * gd;
                        cm[(size_t)r * N + j] = c > P.max_dist ? P.max_dist + 1e-5 : c;
                    }
#pragma unroll
                    for (int q = 0; q < 20; ++q) gA[q] = gB[q];
                }
            }
            // all detections as columns: det_idx = identity -> reuse um_db as identity scratch
            for (int j = tid; j < N; j += BLOCK) L.um_db[j] = j;
            __syncthreads();
            BPB_PROF(2);                                  // appearance cost fill
            const McmOut A = min_cost_matching(cm, nc, N, P.max_dist, L.cand, L.um_db, L.m_t, L.m_d, L.um_ta, L.um_da, L);
            lsa_err |= A.err;
            BPB_PROF(3);                                  // LSA + match lists, stage A
            if (nc > 0)
                for (int k = tid; k < A.nm; k += BLOCK) {       // add_matching_information "R": un-thresholded gated cost (tracker.py:409-425)
                    const int p = L.m_t[k], j = L.m_d[k];
                    double c = reid[(size_t)p * MAXD + L.sel[j]];
                    const double gd = gating_from(gl + (size_t)p * GLN, L.dxyah + j * 4, gdim, gate_single);
                    if (gd > CHI2INV95[gdim]) c = INFTY_COST;
                    L.d_mname[j] = 1; L.d_mdist[j] = P.mc_lambda * c + (1 - P.mc_lambda) * gd;
                }
            __syncthreads();
            // matching_cascade: unmatched_tracks = list(set(track_indices) - matched), linear_assignment.py:128 ...
            for (int p = tid; p < T; p += BLOCK) L.rowf[p] = 0;
            __syncthreads();
            for (int k = tid; k < A.nm; k += BLOCK) L.rowf[L.m_t[k]] = 1;     // by track position
            __syncthreads();
            // ... in CPython's set-iteration order (cm has been consumed: the LDS cost area doubles as the hash-table scratch)
            int *psw = ((size_t)cost_lds_entries * sizeof(double) >= (size_t)16 * Dv.ps_cap) ? (int *)L.cost : Dv.ps_ws + (size_t)s * 4 * Dv.ps_cap;
            const int n_unm = cascade_unmatched_tracks(L.cand, nc, A.nm, L.tmp, psw, (unsigned)Dv.ps_cap, L);
            // split by time_since_update == 1 (tracker.py:308-313), order kept
            const int nb_extra = block_compact(n_unm, [&](int r) { return trk_at(order[L.tmp[r]]).i(BI_TSU) == 1; },
                                               [&](int r, int pos) { L.bc[nu + pos] = L.tmp[r]; }, L.scan);
            const int n_uta = block_compact(n_unm, [&](int r) { return trk_at(order[L.tmp[r]]).i(BI_TSU) != 1; },
                                            [&](int r, int pos) { L.um_t[pos] = L.tmp[r]; }, L.scan);
            const int nb = nu + nb_extra, n_uda = A.n_um_d;
            __syncthreads();
            double *cb = ((size_t)nb * n_uda <= (size_t)cost_lds_entries) ? L.cost : (Dv.cost_g + (size_t)s * MAXT * MAXD);
            for (int e = tid; e < nb * n_uda; e += BLOCK) {          // iou_cost (iou_matching.py:42-78) + thresholding
                const int r = e / n_uda, c = e - r * n_uda;
                const double v = motion_cost(L.bc[r], L.um_da[c]);
                cb[e] = v > motion_max ? motion_max + 1e-5 : v;
            }
            __syncthreads();
            BPB_PROF(4);                                  // matched info, set order, motion cost fill
            const McmOut Bm = min_cost_matching(cb, nb, n_uda, motion_max, L.bc, L.um_da, L.m_t + A.nm, L.m_d + A.nm,
                                                L.um_tb, L.um_db, L);
            lsa_err |= Bm.err;
            if (nb > 0 && n_uda > 0)
                for (int k = tid; k < Bm.nm; k += BLOCK) {        // "S"
                    const int p = L.m_t[A.nm + k], j = L.m_d[A.nm + k];
                    L.d_mname[j] = 2; L.d_mdist[j] = motion_cost(p, j);
                }
            nm = A.nm + Bm.nm;
            for (int k = tid; k < Bm.n_um_t; k += BLOCK) L.um_t[n_uta + k] = L.um_tb[k];
            n_umt = n_uta + Bm.n_um_t; n_umd = Bm.n_um_d;
            um_d_final = L.um_db;
            __syncthreads();
            BPB_PROF(5);                                  // LSA + match lists, stage B
        } else {
            // ---------------- bot_sort_matching (tracker.py:335-363, _full_cost_metric :169-240) ----------------
            for (int p = tid; p < T; p += BLOCK) L.cand[p] = p;
            for (int j = tid; j < N; j += BLOCK) L.um_db[j] = j;
            double *cm = ((size_t)T * N <= (size_t)cost_lds_entries) ? L.cost : (Dv.cost_g + (size_t)s * MAXT * MAXD);
            const double GT = sqrt(CHI2INV95[gdim]);
            const double wsum = P.w_kfgd + P.w_reid + P.w_st;
            auto full_cost = [&](int p, int j) {
                const double gd = gating_from(gl + (size_t)p * GLN, L.dxyah + j * 4, gdim, gate_single);
                const double pos = sqrt(gd) / (GT * P.gating_thres_factor);
                const double app = reid[(size_t)p * MAXD + L.sel[j]];
                const double st = motion_cost(p, j);
                const bool pos_gate = P.w_kfgd > 0 ? pos > 1.0 : false;
                const bool app_gate = P.w_reid > 0 ? app > P.max_dist : false;
                const bool st_gate = P.w_st > 0 ? st > motion_max : false;
                const double c = (P.w_kfgd * pos + P.w_reid * app + st * P.w_st) / wsum;
                // np.logical_or(pos_gate, app_gate, st_gate): the third argument is out= -> st_gate is not part of the mask
                const bool m = P.w_kfgd > 0 ? (pos_gate || app_gate) : (P.w_st > 0 ? (app_gate || st_gate) : app_gate);
                return m ? INFTY_COST : c;
            };
            __syncthreads();
            for (int e = tid; e < T * N; e += BLOCK) {
                const int p = e / N, j = e - p * N;
                const double c = full_cost(p, j);
                cm[e] = c > P.max_dist ? P.max_dist + 1e-5 : c;
            }
            __syncthreads();
            const McmOut A = min_cost_matching(cm, T, N, P.max_dist, L.cand, L.um_db, L.m_t, L.m_d, L.um_ta, L.um_da, L);
            lsa_err |= A.err;
            if (T > 0)
                for (int k = tid; k < A.nm; k += BLOCK) { const int p = L.m_t[k], j = L.m_d[k]; L.d_mname[j] = 1; L.d_mdist[j] = full_cost(p, j); }
            for (int r = tid; r < T; r += BLOCK) L.rowf[r] = 0;
            __syncthreads();
            for (int k = tid; k < A.nm; k += BLOCK) L.rowf[L.m_t[k]] = 1;
            __syncthreads();
            n_umt = block_compact(T, [&](int r) { return L.rowf[r] == 0; }, [&](int r, int pos) { L.um_t[pos] = r; }, L.scan);
            nm = A.nm; n_umd = A.n_um_d; um_d_final = L.um_da;
            __syncthreads();
        }

        // ---------------- Tracker.update (tracker.py:134-167) ----------------
        for (int k = tid; k < nm; k += BLOCK) {               // Track.update: KF part (track.py:137-148)
            const int j = L.m_d[k];
            const BTrk Kt = trk_at(order[L.m_t[k]]);
            double mean[8], cov[64], tl[4];
            trk_ltwh(Kt, tl);
#pragma unroll
            for (int q = 0; q < 4; ++q) Kt.d(BD_PRED + q) = tl[q];
            Kt.i(BI_PVALID) = 1;
#pragma unroll
            for (int q = 0; q < 8; ++q) mean[q] = Kt.d(BD_MEAN + q);
#pragma unroll
            for (int q = 0; q < 64; ++q) cov[q] = Kt.d(BD_COV + q);
            kf8_update(mean, cov, L.dxyah + j * 4, in.conf[dbase + L.sel[j]]);
#pragma unroll
            for (int q = 0; q < 8; ++q) Kt.d(BD_MEAN + q) = mean[q];
#pragma unroll
            for (int q = 0; q < 64; ++q) Kt.d(BD_COV + q) = cov[q];
            detidS[order[L.m_t[k]]] = in.ids[dbase + L.sel[j]];
            if (in.kps) for (int q = 0; q < 51; ++q) Kt.d(BD_KP + q) = in.kps[(dbase + L.sel[j]) * 51 + q];
            Kt.i(BI_MNAME) = L.d_mname[j];
            Kt.d(BD_MDIST) = L.d_mdist[j];
            const int hits = Kt.i(BI_HITS) + 1;
            Kt.i(BI_HITS) = hits; Kt.i(BI_TSU) = 0;
            if (Kt.i(BI_STATE) == ST_TENTATIVE && hits >= P.n_init) Kt.i(BI_STATE) = ST_CONFIRMED;
        }
        BPB_PROF(6);                                      // Kalman update of the matched tracks
        // visibility-aware EMA of the part embeddings (track.py:150-170): 0.6 MB read + written per side per frame -- far too much for
        // ONE workgroup (it ran at 18 GB/s: 102 us of a 440 us frame). The association kernel only publishes the match list
        // (slot, input detection index); bpbss_ema_kernel, launched right behind it on the same stream, sweeps it with one
        // wavefront per (match, part) across the whole chip.
        for (int k = tid; k < nm; k += BLOCK) { Dv.ema_list[((size_t)s * MAXD + k) * 2] = order[L.m_t[k]]; Dv.ema_list[((size_t)s * MAXD + k) * 2 + 1] = L.sel[L.m_d[k]]; }
        if (tid == 0) Dv.ema_n[s] = nm;
        BPB_PROF(7);                                      // embedding EMA
        for (int k = tid; k < n_umt; k += BLOCK) {            // mark_missed (track.py:181-187)
            const BTrk Kt = trk_at(order[L.um_t[k]]);
            if (Kt.i(BI_STATE) == ST_TENTATIVE) Kt.i(BI_STATE) = ST_DELETED;
            else if (Kt.i(BI_TSU) > P.max_age) Kt.i(BI_STATE) = ST_DELETED;
        }
        __syncthreads();
        // _initiate_track (tracker.py:427-441) in the order of unmatched_detections
        int nfree = hdr[H_NFREE], nextid = hdr[H_NEXTID];
        if (lsa_err) { if (tid == 0) { hdr[H_ERR] = TLK_EINTERNAL; *out_count = TLK_EINTERNAL; Dv.ema_n[s] = 0; } return; }      // uniform: an assignment solver hit its loop bound
        if (T + n_umd > MAXT) { if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; Dv.ema_n[s] = 0; } return; }
        for (int k = tid; k < n_umd; k += BLOCK) {
            const int j = um_d_final[k];
            const int slot = freestk[nfree - 1 - k];
            order[T + k] = slot;
            const BTrk Kt = trk_at(slot);
            const double *m = L.dxyah + j * 4;                // kalman_filter.py:53-83
            const double sp = 2 * W_POS * m[3], sv = 10 * W_VEL * m[3];
#pragma unroll
            for (int q = 0; q < 64; ++q) Kt.d(BD_COV + q) = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { Kt.d(BD_MEAN + q) = m[q]; Kt.d(BD_MEAN + 4 + q) = 0.0; Kt.d(BD_COV + q * 9) = sp * sp; Kt.d(BD_COV + (4 + q) * 9) = sv * sv; }
            Kt.i(BI_TID) = nextid + k; Kt.i(BI_HITS) = 1; Kt.i(BI_AGE) = 1; Kt.i(BI_TSU) = 0;
            Kt.i(BI_STATE) = (1 >= P.n_init) ? ST_CONFIRMED : ST_TENTATIVE;
            Kt.i(BI_MNAME) = L.d_mname[j]; Kt.d(BD_MDIST) = L.d_mdist[j]; Kt.i(BI_PVALID) = 0;
            detidS[slot] = in.ids[dbase + L.sel[j]];
            if (in.kps) for (int q = 0; q < 51; ++q) Kt.d(BD_KP + q) = in.kps[(dbase + L.sel[j]) * 51 + q];
        }
        {   // copy the new tracks' part embeddings + visibility
            const int w = tid >> 6, lane = tid & 63;
            for (int job = w; job < n_umd * K; job += NWAVES) {
                const int k = job / K, p = job - k * K;
                const int slot = freestk[nfree - 1 - k], di = L.sel[um_d_final[k]];
                float *f = featS + (size_t)slot * FD + (size_t)p * D;
                const float *df = in.emb + (dbase + di) * FD + (size_t)p * D;
                for (int d = lane; d < D; d += WAVE) f[d] = df[d];
                if (lane == 0) fvisS[(size_t)slot * K + p] = in.vis[(dbase + di) * K + p] != 0 ? 1 : 0;
            }
        }
        __syncthreads();
        nfree -= n_umd; nextid += n_umd; T += n_umd;
        BPB_PROF(8);                                      // mark_missed + births
        // drop deleted tracks (stable), tracker.py:154
        for (int p = tid; p < T; p += BLOCK) { L.tmp[p] = order[p]; L.rowf[p] = trk_at(order[p]).i(BI_STATE) == ST_DELETED ? 1 : 0; }
        __syncthreads();
        const int kept = block_compact(T, [&](int p) { return L.rowf[p] == 0; }, [&](int p, int pos) { order[pos] = L.tmp[p]; }, L.scan);
        if (kept != T) {
            block_compact(T, [&](int p) { return L.rowf[p] != 0; }, [&](int p, int pos) { freestk[nfree + pos] = L.tmp[p]; }, L.scan);
            nfree += T - kept;
        }
        T = kept;
        __syncthreads();
        if (tid == 0) { hdr[H_NTRK] = T; hdr[H_NFREE] = nfree; hdr[H_NEXTID] = nextid; }
    }
    __syncthreads();
    // outputs (strong_sort.py:93-141): confirmed tracks updated in this frame, list order
    const int nrows = block_compact(T, [&](int p) { const BTrk Kt = trk_at(order[p]); return Kt.i(BI_STATE) == ST_CONFIRMED && Kt.i(BI_TSU) == 0; },
                                    [&](int p, int pos) {
                                        if (pos >= out_cap) return;
                                        const int slot = order[p];
                                        const BTrk Kt = trk_at(slot);
                                        tlk_bpbss_row r;
                                        r.det_id = detidS[slot]; r.track_id = Kt.i(BI_TID);
                                        trk_ltwh(Kt, r.kf_ltwh);
                                        r.pred_valid = Kt.i(BI_PVALID);
                                        for (int q = 0; q < 4; ++q) r.pred_ltwh[q] = r.pred_valid ? Kt.d(BD_PRED + q) : NAN;
                                        r.matched_name = Kt.i(BI_MNAME);
                                        r.matched_dist = r.matched_name ? Kt.d(BD_MDIST) : NAN;
                                        r.hits = Kt.i(BI_HITS); r.age = Kt.i(BI_AGE); r.time_since_update = Kt.i(BI_TSU); r.state = Kt.i(BI_STATE);
                                        rows[pos] = r;
                                    }, L.scan);
    if (tid == 0) *out_count = nrows > out_cap ? TLK_ECAPACITY : nrows;
    BPB_PROF(9);                                          // deaths + output rows
    if (Dv.prof && tid == 0) { Dv.prof[(size_t)s * 16 + 14] += clock64() - c_start; Dv.prof[(size_t)s * 16 + 15] += wall_clock64() - w_start; }   // shader cycles vs 100 MHz ticks: the clock the kernel ran at
#undef BPB_PROF
}

// visibility-aware EMA of the part embeddings of the matched tracks (track.py:150-170): one wavefront per (match, part);
// rows with both weights zero are set to 1 (:166-169); the track's visibility becomes max(track, detection) (:170)
__global__ void __launch_bounds__(BLOCK) bpbss_ema_kernel(BpbDev Dv, BpbP P, FrameIn in)
{
    const int s = blockIdx.y, K = Dv.K, D = Dv.D;
    const int nm = Dv.ema_n[s];
    const int job = blockIdx.x * NWAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (job >= nm * K) return;
    const int k = job / K, p = job - k * K;
    const int slot = Dv.ema_list[((size_t)s * Dv.MAXD + k) * 2], di = Dv.ema_list[((size_t)s * Dv.MAXD + k) * 2 + 1];
    const size_t FD = (size_t)K * D, dbase = (size_t)s * in.stream_stride_dets;
    unsigned char *tvp = Dv.fvis + ((size_t)s * Dv.MAXT + slot) * K + p;
    const bool tv = *tvp != 0, dv = in.vis[(dbase + di) * K + p] != 0;
    const bool both = tv && dv, x = tv != dv;
    const float a_t = (float)P.ema_alpha, a_d = (float)(1 - P.ema_alpha);
    const float et = (float)both * a_t + (float)(x && tv);
    const float ed = (float)both * a_d + (float)(x && dv);
    float4 *f = reinterpret_cast<float4 *>(Dv.feat + ((size_t)s * Dv.MAXT + slot) * FD + (size_t)p * D);
    const float4 *df = reinterpret_cast<const float4 *>(in.emb + (dbase + di) * FD + (size_t)p * D);
    for (int d4 = lane; d4 < (D >> 2); d4 += WAVE) {
        float4 o;
        if (et == 0.f && ed == 0.f) { o.x = 1.f; o.y = 1.f; o.z = 1.f; o.w = 1.f; }
        else {
            const float4 tf = f[d4], dd = df[d4];
            float a, b;
            a = et * tf.x; b = ed * dd.x; o.x = a + b;
            a = et * tf.y; b = ed * dd.y; o.y = a + b;
            a = et * tf.z; b = ed * dd.z; o.z = a + b;
            a = et * tf.w; b = ed * dd.w; o.w = a + b;
        }
        f[d4] = o;
    }
    if (lane == 0 && dv) *tvp = 1;
}

__global__ void bpbss_reset_kernel(BpbDev D, int stream)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        int *hdr = D.hdr + (size_t)s * H_COUNT;
        for (int k = threadIdx.x; k < D.MAXT; k += blockDim.x) D.freestk[(size_t)s * D.MAXT + k] = D.MAXT - 1 - k;
        if (threadIdx.x == 0) { hdr[H_NTRK] = 0; hdr[H_NEXTID] = 1; hdr[H_NFREE] = D.MAXT; hdr[H_ERR] = 0; }   // _next_id = 1 (tracker.py:87)
    }
}

__global__ void bpbss_gather_kernel(BpbDev D, int stream, long long *ids, double *mean, double *cov, float *feat, unsigned char *fvis,
                                    int cap, int *n_out)
{
    const int T = D.hdr[(size_t)stream * H_COUNT + H_NTRK];
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = T;
    const size_t stride = (size_t)D.S * D.MAXT, FD = (size_t)D.K * D.D;
    for (int p = blockIdx.x; p < T && p < cap; p += gridDim.x) {
        const int slot = D.order[(size_t)stream * D.MAXT + p];
        const double *fd = D.fd + (size_t)stream * D.MAXT + slot;
        if (ids && threadIdx.x == 0) ids[p] = D.fi[(size_t)BI_TID * stride + (size_t)stream * D.MAXT + slot];
        if (mean) for (int k = threadIdx.x; k < 8; k += blockDim.x) mean[(size_t)p * 8 + k] = fd[(size_t)(BD_MEAN + k) * stride];
        if (cov) for (int k = threadIdx.x; k < 64; k += blockDim.x) cov[(size_t)p * 64 + k] = fd[(size_t)(BD_COV + k) * stride];
        if (feat) for (size_t k = threadIdx.x; k < FD; k += blockDim.x) feat[(size_t)p * FD + k] = D.feat[((size_t)stream * D.MAXT + slot) * FD + k];
        if (fvis) for (int k = threadIdx.x; k < D.K; k += blockDim.x) fvis[(size_t)p * D.K + k] = D.fvis[((size_t)stream * D.MAXT + slot) * D.K + k];
    }
}

// ------------------------------------------------------------------ stateless KF8 / motion-cost entry points
__device__ __forceinline__ void ld8(const double *m, const double *c, size_t i, double (&mean)[8], double (&cov)[64])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) mean[k] = m[i * 8 + k];
#pragma unroll
    for (int k = 0; k < 64; ++k) cov[k] = c[i * 64 + k];
}
__device__ __forceinline__ void st8(double *m, double *c, size_t i, const double (&mean)[8], const double (&cov)[64])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) m[i * 8 + k] = mean[k];
#pragma unroll
    for (int k = 0; k < 64; ++k) c[i * 64 + k] = cov[k];
}
__global__ void __launch_bounds__(BLOCK) kf8_initiate_kernel(const double *__restrict__ meas, double *__restrict__ means, double *__restrict__ covs, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;      // kalman_filter.py:53-83
    if (i >= n) return;
    double mean[8], cov[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) cov[k] = 0.0;
    const double h = meas[(size_t)i * 4 + 3];
    const double sp = 2 * W_POS * h, sv = 10 * W_VEL * h;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = meas[(size_t)i * 4 + k]; mean[4 + k] = 0.0; cov[k * 9] = sp * sp; cov[(4 + k) * 9] = sv * sv; }
    st8(means, covs, (size_t)i, mean, cov);
}
__global__ void __launch_bounds__(BLOCK) kf8_predict_kernel(double *__restrict__ means, double *__restrict__ covs, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double mean[8], cov[64];
    ld8(means, covs, (size_t)i, mean, cov);
    kf8_predict(mean, cov);
    st8(means, covs, (size_t)i, mean, cov);
}
__global__ void __launch_bounds__(BLOCK) kf8_project_kernel(const double *__restrict__ means, const double *__restrict__ covs, const double *__restrict__ conf,
                                                            double *__restrict__ pm, double *__restrict__ pc, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;      // kalman_filter.py:121-152
    if (i >= n) return;
    const double *m = means + (size_t)i * 8, *c = covs + (size_t)i * 64;
    const double sstd = (1 - (conf ? conf[i] : 0.0)) * (W_POS * m[3]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        pm[(size_t)i * 4 + a] = m[a];
#pragma unroll
        for (int b = 0; b < 4; ++b) pc[(size_t)i * 16 + a * 4 + b] = c[a * 8 + b] + (a == b ? sstd * sstd : 0.0);
    }
}
__global__ void __launch_bounds__(BLOCK) kf8_update_kernel(double *__restrict__ means, double *__restrict__ covs, const double *__restrict__ z,
                                                           const double *__restrict__ conf, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double mean[8], cov[64];
    ld8(means, covs, (size_t)i, mean, cov);
    kf8_update(mean, cov, z + (size_t)i * 4, conf ? conf[i] : 0.0);
    st8(means, covs, (size_t)i, mean, cov);
}
// grid.x = track, threads stride over the measurements; every thread factors the track's projected covariance itself (4x4)
__global__ void __launch_bounds__(BLOCK) kf8_gate_kernel(const double *__restrict__ means, const double *__restrict__ covs, int T,
                                                         const double *__restrict__ meas, int N, int d, double *__restrict__ out)
{
    const int t = blockIdx.x;                            // kalman_filter.py:189-227
    const double *m = means + (size_t)t * 8, *c = covs + (size_t)t * 64;
    double gl[GLN], Sd[16], Lc[16];
    const double sstd = W_POS * m[3];
#pragma unroll
    for (int q = 0; q < 16; ++q) Sd[q] = 0.0;
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) Sd[a * d + b] = c[a * 8 + b] + (a == b ? sstd * sstd : 0.0);
    chol4(Sd, d, Lc);
#pragma unroll
    for (int a = 0; a < 4; ++a) gl[a] = m[a];
#pragma unroll
    for (int q = 0; q < 16; ++q) gl[4 + q] = Lc[q];
    gate_row_finish(gl, d, N == 1);
    for (int j = threadIdx.x; j < N; j += BLOCK) out[(size_t)t * N + j] = gating_from(gl, meas + (size_t)j * 4, d, N == 1);
}
__global__ void __launch_bounds__(BLOCK) iou_ltwh_cost_kernel(const double *__restrict__ trk, int T, const double *__restrict__ det, int N,
                                                              double *__restrict__ out)
{
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;     // sort/iou_matching.py:42-78 without the age gate
    if (gid >= (long long)T * N) return;
    const int t = (int)(gid / N), j = (int)(gid - (long long)t * N);
    out[gid] = 1.0 - iou_ltwh(trk + (size_t)t * 4, det + (size_t)j * 4);
}
struct KpPtr { const double *p; __device__ double operator()(int i) const { return p[i]; } };
__global__ void __launch_bounds__(BLOCK) oks_cost_kernel(const double *__restrict__ trk_kp, int T, const double *__restrict__ det_kp, int N,
                                                         double *__restrict__ out)
{
    const int t = blockIdx.x;                            // sort/oks_matching.py:95-128
    const KpPtr kp{trk_kp + (size_t)t * 51};
    int nvis = 0;
    const double sc = oks_scale(kp, &nvis);
    for (int j = threadIdx.x; j < N; j += BLOCK) out[(size_t)t * N + j] = 1.0 - oks_one(kp, sc, nvis, det_kp + (size_t)j * 51);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
struct tlk_bpbss {
    BpbDev D; BpbP P; int device; size_t smem;
    // staging for the host-buffer entry point
    long long *d_ids; double *d_ltwh; float *d_emb; unsigned char *d_vis; double *d_conf, *d_kps; int *d_cnt, *d_ocnt; tlk_bpbss_row *d_rows;
    int out_cap;
};

static void bpb_free(tlk_bpbss *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    BpbDev &D = h->D;
    void *ptrs[] = {D.fd, D.fi, D.detid, D.hdr, D.order, D.freestk, D.feat, D.fvis, D.tnorm, D.dnorm, D.reid, D.gl, D.cost_g, D.big_ws, D.ps_ws, D.prof, D.ema_list, D.ema_n,
                    h->d_ids, h->d_ltwh, h->d_emb, h->d_vis, h->d_conf, h->d_kps, h->d_cnt, h->d_ocnt, h->d_rows};
    for (void *p : ptrs) if (p) hipFree(p);
    delete h;
}

static int launch_frame(tlk_bpbss *h, const BpbDev &Dv, int n_streams, const FrameIn &in, tlk_bpbss_row *rows, size_t rows_stream_stride,
                        int out_cap, int *out_counts, size_t oc_stride, hipStream_t st)
{
    const int K = Dv.K;
    const int nvec = (Dv.MAXT + Dv.MAXD) * K;
    // bounded grids (the kernels loop): a bank created with room for thousands of tracks launches no more idle workgroups than a small one
    const int nb_norm = (nvec + NWAVES - 1) / NWAVES, nb_dist = ((Dv.MAXD + 15) / 16) * ((Dv.MAXT + 15) / 16);
    hipLaunchKernelGGL(partnorm_kernel, dim3(nb_norm < 512 ? nb_norm : 512, n_streams), dim3(BLOCK), 0, st, Dv, in, h->P.wrapper_mode);
    hipLaunchKernelGGL(partdist_kernel, dim3(nb_dist < 256 ? nb_dist : 256, 1, n_streams), dim3(64 * K), 0, st, Dv, in);
    hipLaunchKernelGGL(bpbss_assoc_kernel, dim3(n_streams), dim3(BLOCK), h->smem, st, Dv, h->P, in, rows, rows_stream_stride, out_cap,
                       out_counts, oc_stride);
    hipLaunchKernelGGL(bpbss_ema_kernel, dim3((Dv.MAXD * K + NWAVES - 1) / NWAVES, n_streams), dim3(BLOCK), 0, st, Dv, h->P, in);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_bpbss_create(const tlk_bpbss_params *p, int n_streams, int device, tlk_bpbss **out)
{
    if (!p || !out) return fail(TLK_EINVAL, "tlk_bpbss_create: null pointer");
    if (n_streams < 1) return fail(TLK_EINVAL, "tlk_bpbss_create: n_streams must be >= 1");
    if (p->parts < 1 || p->parts > 8) return fail(TLK_EINVAL, "tlk_bpbss_create: parts must be in [1, 8]");
    if (p->dim < 16 || p->dim % 16 != 0) return fail(TLK_EINVAL, "tlk_bpbss_create: dim must be a positive multiple of 16");
    if (p->matching_strategy < 0 || p->matching_strategy > 1) return fail(TLK_EINVAL, "tlk_bpbss_create: unknown matching_strategy");
    const int MAXT = p->max_tracks > 0 ? p->max_tracks : 256, MAXD = p->max_dets > 0 ? p->max_dets : 128;
    // Capacity is an ALLOCATION size now (r04): per-track state lives in HBM at capacity; the per-frame lists use LDS while the scene fits a
    // 256 x 128 or 1024 x 256 tier and HBM beyond (bpbss_assoc_kernel).  The bounds below only keep one stream's arrays below a few GB.
    if (MAXT > 16384 || MAXD > 1024) return fail(TLK_ECAPACITY, "tlk_bpbss_create: max_tracks <= 16384 and max_dets <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_bpbss_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_bpbss_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_bpbss *h = new tlk_bpbss();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->P = BpbP{p->ema_alpha, p->mc_lambda, p->max_dist, p->max_iou_distance, p->min_bbox_confidence, p->gating_thres_factor,
                p->w_kfgd, p->w_reid, p->w_st, p->max_age, p->n_init, p->only_position_for_kf_gating,
                p->max_kalman_prediction_without_update, p->matching_strategy, p->wrapper_mode, p->motion_criterium, p->max_oks_distance};
    if (p->motion_criterium < 0 || p->motion_criterium > 1) { delete h; return fail(TLK_EINVAL, "tlk_bpbss_create: motion_criterium must be 0 (iou) or 1 (oks)"); }
    BpbDev &D = h->D;
    D.S = n_streams; D.MAXT = MAXT; D.MAXD = MAXD; D.K = p->parts; D.D = p->dim;
    const size_t budget = 160 * 1024 - 256;
    D.lds_bytes = (int)(budget & ~(size_t)15);
    h->smem = (size_t)D.lds_bytes;
    D.big_stride = (blds_fixed(MAXT, MAXD) + 255) & ~(size_t)255;
    const size_t slots = (size_t)n_streams * MAXT, FD = (size_t)D.K * D.D;
    h->out_cap = MAXD;
#define BPB_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { bpb_free(h); return fail(TLK_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    BPB_ALLOC(D.fd, sizeof(double) * BD_COUNT * slots);
    BPB_ALLOC(D.fi, sizeof(int) * BI_COUNT * slots);
    BPB_ALLOC(D.detid, sizeof(long long) * slots);
    BPB_ALLOC(D.hdr, sizeof(int) * H_COUNT * n_streams);
    BPB_ALLOC(D.order, sizeof(int) * slots);
    BPB_ALLOC(D.freestk, sizeof(int) * slots);
    BPB_ALLOC(D.feat, sizeof(float) * FD * slots);
    BPB_ALLOC(D.fvis, (size_t)D.K * slots);
    BPB_ALLOC(D.tnorm, sizeof(float) * 2 * D.K * slots);
    BPB_ALLOC(D.dnorm, sizeof(float) * 2 * D.K * (size_t)n_streams * MAXD);
    BPB_ALLOC(D.reid, sizeof(double) * slots * MAXD);
    BPB_ALLOC(D.gl, sizeof(double) * GLN * slots);
    BPB_ALLOC(D.cost_g, sizeof(double) * slots * MAXD);
    BPB_ALLOC(D.big_ws, D.big_stride * (size_t)n_streams);
    D.ps_cap = (int)pyset::table_capacity((unsigned)MAXT);
    BPB_ALLOC(D.ps_ws, sizeof(int) * 4 * (size_t)D.ps_cap * n_streams);
    BPB_ALLOC(D.ema_list, sizeof(int) * 2 * (size_t)MAXD * n_streams);
    BPB_ALLOC(D.ema_n, sizeof(int) * n_streams);
    if (getenv("TLK_BPBSS_PROF")) { BPB_ALLOC(D.prof, sizeof(long long) * 16 * n_streams); hipMemset(D.prof, 0, sizeof(long long) * 16 * n_streams); }
    BPB_ALLOC(h->d_ids, sizeof(long long) * MAXD);
    BPB_ALLOC(h->d_ltwh, sizeof(double) * 4 * MAXD);
    BPB_ALLOC(h->d_emb, sizeof(float) * FD * MAXD);
    BPB_ALLOC(h->d_vis, (size_t)D.K * MAXD);
    BPB_ALLOC(h->d_conf, sizeof(double) * MAXD);
    BPB_ALLOC(h->d_kps, sizeof(double) * 51 * MAXD);
    BPB_ALLOC(h->d_cnt, sizeof(int));
    BPB_ALLOC(h->d_ocnt, sizeof(int));
    BPB_ALLOC(h->d_rows, sizeof(tlk_bpbss_row) * h->out_cap);
#undef BPB_ALLOC
    hipError_t e = hipMemset(D.fd, 0, sizeof(double) * BD_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.fi, 0, sizeof(int) * BI_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.feat, 0, sizeof(float) * FD * slots);
    if (e == hipSuccess) e = hipMemset(D.fvis, 0, (size_t)D.K * slots);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)bpbss_assoc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) { bpb_free(h); return fail(TLK_EHIP, std::string("tlk_bpbss_create: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(bpbss_reset_kernel, dim3(n_streams < 256 ? n_streams : 256), dim3(BLOCK), 0, 0, D, -1);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { bpb_free(h); return fail(TLK_EHIP, std::string("tlk_bpbss_create: ") + hipGetErrorString(e)); }
    *out = h;
    return TLK_OK;
}

extern "C" int tlk_bpbss_destroy(tlk_bpbss *h) { bpb_free(h); return TLK_OK; }

extern "C" int tlk_bpbss_get_profile(tlk_bpbss *h, int stream, long long *ticks16)
{
    if (!h || !ticks16) return fail(TLK_EINVAL, "tlk_bpbss_get_profile: null pointer");
    if (!h->D.prof) return fail(TLK_EINVAL, "tlk_bpbss_get_profile: create the bank with TLK_BPBSS_PROF=1 in the environment");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_bpbss_get_profile: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    TLK_HIP(hipDeviceSynchronize());
    TLK_HIP(hipMemcpy(ticks16, h->D.prof + (size_t)stream * 16, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    return TLK_OK;
}

extern "C" int tlk_bpbss_reset(tlk_bpbss *h, int stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_bpbss_reset: null handle");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_bpbss_reset: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(bpbss_reset_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, 0, h->D, stream);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipStreamSynchronize(0));
    return TLK_OK;
}

extern "C" int tlk_bpbss_update_dev(tlk_bpbss *h, const int64_t *ids_dev, const double *ltwh_dev, const float *emb_dev,
                                    const uint8_t *vis_dev, const double *conf_dev, const double *kps_dev, const int32_t *counts_dev,
                                    int n_frames, tlk_bpbss_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_bpbss_update_dev: null handle");
    if (n_frames < 0 || out_cap < 0) return fail(TLK_EINVAL, "tlk_bpbss_update_dev: negative size");
    if (n_frames == 0) return TLK_OK;
    if (!ids_dev || !ltwh_dev || !emb_dev || !vis_dev || !conf_dev || !counts_dev || !rows_dev || !out_counts_dev)
        return fail(TLK_EINVAL, "tlk_bpbss_update_dev: null pointer");
    TLK_HIP(hipSetDevice(h->device));
    const BpbDev &D = h->D;
    const size_t FD = (size_t)D.K * D.D;
    for (int f = 0; f < n_frames; ++f) {
        const size_t off = (size_t)f * D.MAXD;          // dets index within a stream block of n_frames*MAXD
        FrameIn in;
        in.ids = (const long long *)ids_dev + off; in.ltwh = ltwh_dev + off * 4; in.emb = emb_dev + off * FD;
        in.vis = vis_dev + off * D.K; in.conf = conf_dev + off; in.counts = (const int *)counts_dev + f;
        in.kps = kps_dev ? kps_dev + off * 51 : nullptr;
        in.stream_stride_dets = (size_t)n_frames * D.MAXD; in.count_stride = (size_t)n_frames;
        const int rc = launch_frame(h, D, D.S, in, rows_dev + (size_t)f * out_cap, (size_t)n_frames * out_cap, out_cap,
                                    (int *)out_counts_dev + f, (size_t)n_frames, (hipStream_t)hip_stream);
        if (rc != TLK_OK) return rc;
    }
    return TLK_OK;
}

extern "C" int tlk_bpbss_update(tlk_bpbss *h, int stream, const int64_t *ids, const double *ltwh, const float *emb,
                                const uint8_t *vis, const double *conf, const double *kps, int n, tlk_bpbss_row *rows, int cap,
                                int *n_out)
{
    if (!h || !n_out) return fail(TLK_EINVAL, "tlk_bpbss_update: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_bpbss_update: stream out of range");
    if (n < 0 || (n > 0 && (!ids || !ltwh || !emb || !vis || !conf))) return fail(TLK_EINVAL, "tlk_bpbss_update: bad detections");
    if (n > h->D.MAXD) return fail(TLK_ECAPACITY, "tlk_bpbss_update: more detections than max_dets");
    TLK_HIP(hipSetDevice(h->device));
    hipStream_t st = 0;
    const size_t FD = (size_t)h->D.K * h->D.D;
    if (n) {
        TLK_HIP(hipMemcpyAsync(h->d_ids, ids, sizeof(long long) * n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_ltwh, ltwh, sizeof(double) * 4 * n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_emb, emb, sizeof(float) * FD * n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_vis, vis, (size_t)h->D.K * n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_conf, conf, sizeof(double) * n, hipMemcpyHostToDevice, st));
        if (kps) TLK_HIP(hipMemcpyAsync(h->d_kps, kps, sizeof(double) * 51 * n, hipMemcpyHostToDevice, st));
    }
    TLK_HIP(hipMemcpyAsync(h->d_cnt, &n, sizeof(int), hipMemcpyHostToDevice, st));
    BpbDev V = h->D;       // single-stream view: shift per-stream bases, keep strides
    const size_t sl = (size_t)stream * V.MAXT;
    V.fd += sl; V.fi += sl; V.detid += sl; V.hdr += (size_t)stream * H_COUNT; V.order += sl; V.freestk += sl;
    V.feat += sl * FD; V.fvis += sl * V.K; V.tnorm += sl * V.K * 2; V.dnorm += (size_t)stream * V.MAXD * V.K * 2;
    V.reid += sl * V.MAXD; V.gl += sl * GLN; V.cost_g += sl * V.MAXD; V.big_ws += (size_t)stream * V.big_stride; V.ps_ws += (size_t)stream * 4 * V.ps_cap; if (V.prof) V.prof += (size_t)stream * 16;
    V.ema_list += (size_t)stream * V.MAXD * 2; V.ema_n += stream;
    FrameIn in;
    in.ids = h->d_ids; in.ltwh = h->d_ltwh; in.emb = h->d_emb; in.vis = h->d_vis; in.conf = h->d_conf; in.counts = h->d_cnt;
    in.kps = kps ? h->d_kps : nullptr;
    if (h->P.motion == 1 && n > 0 && !kps) return fail(TLK_EINVAL, "tlk_bpbss_update: motion_criterium 'oks' needs keypoints");
    in.stream_stride_dets = 0; in.count_stride = 0;
    const int rc = launch_frame(h, V, 1, in, h->d_rows, 0, h->out_cap, h->d_ocnt, 0, st);
    if (rc != TLK_OK) return rc;
    int rows_n = 0;
    TLK_HIP(hipMemcpyAsync(&rows_n, h->d_ocnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TLK_HIP(hipStreamSynchronize(st));
    if (rows_n < 0) return fail_stream(rows_n, "tlk_bpbss_update");
    if (rows_n > cap) return fail(TLK_ECAPACITY, "tlk_bpbss_update: output buffer too small");
    if (rows_n) TLK_HIP(hipMemcpy(rows, h->d_rows, sizeof(tlk_bpbss_row) * rows_n, hipMemcpyDeviceToHost));
    *n_out = rows_n;
    return TLK_OK;
}

extern "C" int tlk_bpbss_get_tracks(tlk_bpbss *h, int stream, int64_t *ids, double *mean, double *cov, float *feat, uint8_t *fvis,
                                    int cap, int *n_tracks)
{
    if (!h || !n_tracks) return fail(TLK_EINVAL, "tlk_bpbss_get_tracks: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_bpbss_get_tracks: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    const size_t c = cap > 0 ? cap : 1, FD = (size_t)h->D.K * h->D.D;
    long long *di = nullptr; double *dm = nullptr, *dc = nullptr; float *df = nullptr; unsigned char *dv = nullptr; int *dn = nullptr;
    TLK_HIP(hipMalloc((void **)&di, sizeof(long long) * c));
    TLK_HIP(hipMalloc((void **)&dm, sizeof(double) * 8 * c));
    TLK_HIP(hipMalloc((void **)&dc, sizeof(double) * 64 * c));
    TLK_HIP(hipMalloc((void **)&df, sizeof(float) * FD * c));
    TLK_HIP(hipMalloc((void **)&dv, (size_t)h->D.K * c));
    TLK_HIP(hipMalloc((void **)&dn, sizeof(int)));
    hipLaunchKernelGGL(bpbss_gather_kernel, dim3(64), dim3(BLOCK), 0, 0, h->D, stream, di, dm, dc, df, dv, cap, dn);
    int n = 0;
    hipError_t e = hipMemcpy(&n, dn, sizeof(int), hipMemcpyDeviceToHost);
    const int k = n < cap ? n : cap;
    if (e == hipSuccess && k > 0) {
        if (ids) e = hipMemcpy(ids, di, sizeof(long long) * k, hipMemcpyDeviceToHost);
        if (e == hipSuccess && mean) e = hipMemcpy(mean, dm, sizeof(double) * 8 * k, hipMemcpyDeviceToHost);
        if (e == hipSuccess && cov) e = hipMemcpy(cov, dc, sizeof(double) * 64 * k, hipMemcpyDeviceToHost);
        if (e == hipSuccess && feat) e = hipMemcpy(feat, df, sizeof(float) * FD * k, hipMemcpyDeviceToHost);
        if (e == hipSuccess && fvis) e = hipMemcpy(fvis, dv, (size_t)h->D.K * k, hipMemcpyDeviceToHost);
    }
    hipFree(di); hipFree(dm); hipFree(dc); hipFree(df); hipFree(dv); hipFree(dn);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_bpbss_get_tracks: ") + hipGetErrorString(e));
    *n_tracks = n;
    return TLK_OK;
}

// stateless part distance: builds a throw-away identity-ordered view
extern "C" int tlk_partdist_f32(const float *q_dev, const uint8_t *qvis_dev, int T, const float *g_dev, const uint8_t *gvis_dev,
                                int N, int K, int D, double *out_dev, void *hip_stream)
{
    if (T < 0 || N < 0 || K < 1 || K > 8 || D < 16 || D % 16 != 0) return fail(TLK_EINVAL, "tlk_partdist_f32: bad shape (K in [1,8], D % 16 == 0)");
    if (T == 0 || N == 0) return TLK_OK;
    if (!q_dev || !qvis_dev || !g_dev || !gvis_dev || !out_dev) return fail(TLK_EINVAL, "tlk_partdist_f32: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    BpbDev V;
    memset(&V, 0, sizeof(V));
    V.S = 1; V.MAXT = T; V.MAXD = N; V.K = K; V.D = D;
    int *hdr = nullptr, *order = nullptr, *cnt = nullptr; float *tn = nullptr, *dn = nullptr;
    TLK_HIP(hipMallocAsync((void **)&hdr, sizeof(int) * H_COUNT, st));
    TLK_HIP(hipMallocAsync((void **)&order, sizeof(int) * T, st));
    TLK_HIP(hipMallocAsync((void **)&cnt, sizeof(int), st));
    TLK_HIP(hipMallocAsync((void **)&tn, sizeof(float) * 2 * K * T, st));
    TLK_HIP(hipMallocAsync((void **)&dn, sizeof(float) * 2 * K * N, st));
    std::string tmp((size_t)sizeof(int) * (H_COUNT + T + 1), '\0');
    int *hp = (int *)tmp.data();
    hp[H_NTRK] = T;
    for (int i = 0; i < T; ++i) hp[H_COUNT + i] = i;
    hp[H_COUNT + T] = N;
    TLK_HIP(hipMemcpyAsync(hdr, hp, sizeof(int) * H_COUNT, hipMemcpyHostToDevice, st));
    TLK_HIP(hipMemcpyAsync(order, hp + H_COUNT, sizeof(int) * T, hipMemcpyHostToDevice, st));
    TLK_HIP(hipMemcpyAsync(cnt, hp + H_COUNT + T, sizeof(int), hipMemcpyHostToDevice, st));
    TLK_HIP(hipStreamSynchronize(st));           // staging buffer `tmp` goes out of scope below
    V.hdr = hdr; V.order = order; V.feat = const_cast<float *>(q_dev); V.fvis = const_cast<unsigned char *>(qvis_dev);
    V.tnorm = tn; V.dnorm = dn; V.reid = out_dev;
    FrameIn in;
    memset(&in, 0, sizeof(in));
    in.emb = g_dev; in.vis = gvis_dev; in.counts = cnt;
    const int nvec = (T + N) * K;
    hipLaunchKernelGGL(partnorm_kernel, dim3((nvec + NWAVES - 1) / NWAVES, 1), dim3(BLOCK), 0, st, V, in, 0);
    hipLaunchKernelGGL(partdist_kernel, dim3((N + 15) / 16, (T + 15) / 16, 1), dim3(64 * K), 0, st, V, in);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipFreeAsync(hdr, st)); TLK_HIP(hipFreeAsync(order, st)); TLK_HIP(hipFreeAsync(cnt, st));
    TLK_HIP(hipFreeAsync(tn, st)); TLK_HIP(hipFreeAsync(dn, st));
    return TLK_OK;
}

// ------------------------------------------------------------------------------------------------ stateless KF8 / motion costs
#define KF8_LAUNCH(name, kern, n, ...)                                                                                   \
    do {                                                                                                                 \
        if ((n) < 0) return fail(TLK_EINVAL, name ": n < 0");                                                            \
        if ((n) == 0) return TLK_OK;                                                                                     \
        hipLaunchKernelGGL(kern, dim3(((n) + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)hip_stream, __VA_ARGS__); \
        TLK_HIP(hipGetLastError());                                                                                      \
        return TLK_OK;                                                                                                   \
    } while (0)

extern "C" int tlk_kf8_initiate_f64(const double *meas_xyah_dev, double *mean_dev, double *cov_dev, int n, void *hip_stream)
{
    if (n > 0 && (!meas_xyah_dev || !mean_dev || !cov_dev)) return fail(TLK_EINVAL, "tlk_kf8_initiate_f64: null pointer");
    KF8_LAUNCH("tlk_kf8_initiate_f64", kf8_initiate_kernel, n, meas_xyah_dev, mean_dev, cov_dev, n);
}
extern "C" int tlk_kf8_predict_f64(double *mean_dev, double *cov_dev, int n, void *hip_stream)
{
    if (n > 0 && (!mean_dev || !cov_dev)) return fail(TLK_EINVAL, "tlk_kf8_predict_f64: null pointer");
    KF8_LAUNCH("tlk_kf8_predict_f64", kf8_predict_kernel, n, mean_dev, cov_dev, n);
}
extern "C" int tlk_kf8_project_f64(const double *mean_dev, const double *cov_dev, const double *conf_dev, double *pmean_dev, double *pcov_dev,
                                   int n, void *hip_stream)
{
    if (n > 0 && (!mean_dev || !cov_dev || !pmean_dev || !pcov_dev)) return fail(TLK_EINVAL, "tlk_kf8_project_f64: null pointer");
    KF8_LAUNCH("tlk_kf8_project_f64", kf8_project_kernel, n, mean_dev, cov_dev, conf_dev, pmean_dev, pcov_dev, n);
}
extern "C" int tlk_kf8_update_f64(double *mean_dev, double *cov_dev, const double *meas_xyah_dev, const double *conf_dev, int n, void *hip_stream)
{
    if (n > 0 && (!mean_dev || !cov_dev || !meas_xyah_dev)) return fail(TLK_EINVAL, "tlk_kf8_update_f64: null pointer");
    KF8_LAUNCH("tlk_kf8_update_f64", kf8_update_kernel, n, mean_dev, cov_dev, meas_xyah_dev, conf_dev, n);
}
extern "C" int tlk_kf8_gate_f64(const double *mean_dev, const double *cov_dev, int n_tracks, const double *meas_xyah_dev, int n_meas,
                                int only_position, double *out_dev, void *hip_stream)
{
    if (n_tracks < 0 || n_meas < 0) return fail(TLK_EINVAL, "tlk_kf8_gate_f64: negative size");
    if (n_tracks == 0 || n_meas == 0) return TLK_OK;
    if (!mean_dev || !cov_dev || !meas_xyah_dev || !out_dev) return fail(TLK_EINVAL, "tlk_kf8_gate_f64: null pointer");
    hipLaunchKernelGGL(kf8_gate_kernel, dim3(n_tracks), dim3(BLOCK), 0, (hipStream_t)hip_stream, mean_dev, cov_dev, n_tracks, meas_xyah_dev, n_meas,
                       only_position ? 2 : 4, out_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
extern "C" int tlk_iou_ltwh_cost_f64(const double *tracks_ltwh_dev, int n_tracks, const double *dets_ltwh_dev, int n_dets, double *out_dev,
                                     void *hip_stream)
{
    if (n_tracks < 0 || n_dets < 0) return fail(TLK_EINVAL, "tlk_iou_ltwh_cost_f64: negative size");
    if (n_tracks == 0 || n_dets == 0) return TLK_OK;
    if (!tracks_ltwh_dev || !dets_ltwh_dev || !out_dev) return fail(TLK_EINVAL, "tlk_iou_ltwh_cost_f64: null pointer");
    const long long tot = (long long)n_tracks * n_dets;
    hipLaunchKernelGGL(iou_ltwh_cost_kernel, dim3((unsigned)((tot + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)hip_stream, tracks_ltwh_dev,
                       n_tracks, dets_ltwh_dev, n_dets, out_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
extern "C" int tlk_oks_cost_f64(const double *track_kps_dev, int n_tracks, const double *det_kps_dev, int n_dets, double *out_dev, void *hip_stream)
{
    if (n_tracks < 0 || n_dets < 0) return fail(TLK_EINVAL, "tlk_oks_cost_f64: negative size");
    if (n_tracks == 0 || n_dets == 0) return TLK_OK;
    if (!track_kps_dev || !det_kps_dev || !out_dev) return fail(TLK_EINVAL, "tlk_oks_cost_f64: null pointer");
    hipLaunchKernelGGL(oks_cost_kernel, dim3(n_tracks), dim3(BLOCK), 0, (hipStream_t)hip_stream, track_kps_dev, n_tracks, det_kps_dev, n_dets, out_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
