// tlk_botsort.hip -- BoT-SORT (plugins/track/bot_sort, cmc_method "none") on gfx950: three launches per frame for ALL streams of a bank.
//
//   1. botsort_prep_kernel     one wavefront per detection: the float32 feature normalised twice (STrack.update_features on a
//                              fresh STrack, where curr_feat and smooth_feat are one array: bot_sort.py:42-50)
//   2. botsort_cosine_kernel   float64 cosine distance of every live track's smooth_feat to every high-score detection
//                              (matching.embedding_distance = max(0, cdist(.., "cosine")), matching.py:127-142), 16 x 16 tiles
//                              staged through LDS, accumulated in feature order like scipy's C loop
//   3. botsort_kernel          one 256-thread workgroup per stream walks BoTSORT.update (bot_sort.py:275-485): score split,
//                              multi_predict, first association on lambda * embedding + (1 - lambda) * Mahalanobis with the
//                              chi2 gate (fuse_motion, matching.py:159-171), second association on IoU, unconfirmed tracks on
//                              min(IoU x score, embedding / 2 under both thresholds), KF updates, feature EMA, class vote,
//                              new tracks, time-outs, tracked / lost bookkeeping, duplicate removal, output rows.
// The three assignment problems are lap.lapjv(extend_cost=True, cost_limit) in the reduced LDS form of tlk_bytetrack_common.hpp.
// dtype trail as in the reference: STrack._tlwh / features float32, a new track's mean and covariance float32 until
// multi_predict / multi_gmc (identity warp: values unchanged, arrays float64) / update, the rest float64 (-ffp-contract=off).
// The camera-motion estimators (gmc.py: orb / sift / ecc / sparseOptFlow) are cv2 and out of scope.
#include "tlk_common.hpp"
#include "tlk_strongsort_common.hpp"
#include "tlk_bytetrack_common.hpp"

using namespace tlk;

namespace {

constexpr int MAXCLS = 16;                               // distinct classes one track may collect in cls_hist
constexpr double CHI2_4 = 9.4877;                        // kalman_filter.py chi2inv95[4]
constexpr int GLD = 20;                                  // per-track gate: projected mean (4) + Cholesky factor (16)
enum : int { OD_MEAN = 0, OD_COV = 8, OD_SCORE = 72, OD_CLS = 73, OD_TLID = 74, OD_HCLS = 75, OD_HFREQ = OD_HCLS + MAXCLS, OD_COUNT = OD_HFREQ + MAXCLS };
enum : int { OI_TID = 0, OI_STATE, OI_ACT, OI_TLEN, OI_FID, OI_SFID, OI_F32, OI_INREM, OI_NHIST, OI_COUNT };
enum : int { OH_NTRK = 0, OH_NLOST, OH_NFREE, OH_COUNT_ID, OH_FRAME, OH_ERR, OH_COUNT = 8 };
enum : int { OT_NEW = 0, OT_TRACKED = 1, OT_LOST = 2, OT_LONGLOST = 3, OT_REMOVED = 4 };      // basetrack.py:5-10

struct BoDev {
    double *fd;              // OD_COUNT x S x MAXT
    int *fi;                 // OI_COUNT x S x MAXT
    int *hdr, *tracked, *lost, *freestk;
    float *feat;             // S x MAXT x D     smooth_feat by slot
    float *dfeat;            // S x MAXD x D     this frame's detection features after the two normalisations (input order)
    double *dist;            // S x MAXT x MAXD  embedding distance, row = position in [tracked list, lost list], col = input index
    double *gl;              // S x MAXT x GLD   gate of pool track p
    double *ebuf;            // S x MAXT x MAXD  assignment problem spill
    int *alive;              // S x MAXT   slot-indexed marks of the end-of-frame free-slot sweep
    unsigned char *big_ws;   // S x big_stride: list / solver work area of the big-scene tier (by_carve_frame)
    size_t big_stride;
    int S, MAXT, MAXD, D, lds_bytes;
};
struct BoP { double track_high, new_track, match_thresh, proximity, appearance, lambda_, min_conf; int max_time_lost, wrapper_mode; };
struct BoIn { const double *dets; const float *feats; const int *counts; size_t stream_stride_dets, count_stride;
              const double *warps; size_t warp_stride; };      // warps: per stream a (2,3) float64 camera-motion warp of this frame, nullptr = identity

// STrack.multi_gmc with a (2,3) warp H (bot_sort.py:93-109): mean = kron(I4, R) mean, mean[:2] += t, cov = kron(I4, R) cov kron(I4, R)^T
__device__ __forceinline__ void bo_gmc_apply(double (&mean)[8], double (&cov)[64], const double *H)
{
    const double R[4] = {H[0], H[1], H[3], H[4]};
    double m[8], t1[64];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        m[2 * b] = R[0] * mean[2 * b] + R[1] * mean[2 * b + 1];
        m[2 * b + 1] = R[2] * mean[2 * b] + R[3] * mean[2 * b + 1];
    }
    m[0] += H[2]; m[1] += H[5];
#pragma unroll
    for (int i = 0; i < 8; ++i)                 // t1 = R8 cov: row i mixes rows 2*(i/2), 2*(i/2)+1
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int r0 = i & ~1; t1[i * 8 + j] = fma(R[(i & 1) * 2 + 1], cov[(r0 + 1) * 8 + j], R[(i & 1) * 2] * cov[r0 * 8 + j]); }   // dgemm: fma chain over k
#pragma unroll
    for (int i = 0; i < 8; ++i)                 // cov = t1 R8^T: column j mixes columns 2*(j/2), 2*(j/2)+1
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c0 = j & ~1; cov[i * 8 + j] = fma(t1[i * 8 + c0 + 1], R[(j & 1) * 2 + 1], t1[i * 8 + c0] * R[(j & 1) * 2]); }
#pragma unroll
    for (int k = 0; k < 8; ++k) mean[k] = m[k];
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------ 1. detection features
__global__ void __launch_bounds__(BLOCK) botsort_prep_kernel(BoDev Dv, BoP P, BoIn in)
{
    const int s = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int N = in.counts[(size_t)s * in.count_stride];
    const int n = blockIdx.x * NWAVES + w;
    if (N <= 0 || N > Dv.MAXD || n >= N) return;
    const double conf = in.dets[((size_t)s * in.stream_stride_dets + n) * 7 + 4];
    if (!(conf > P.min_conf && conf > P.track_high)) return;                        // only these become STracks with a feature
    const float *x = in.feats + ((size_t)s * in.stream_stride_dets + n) * Dv.D;
    float *o = Dv.dfeat + ((size_t)s * Dv.MAXD + n) * Dv.D;
    float ss = 0.f;
    for (int d = lane; d < Dv.D; d += WAVE) { const float v = x[d]; ss += v * v; }
    const float n1 = sqrtf(wave_sum(ss));
    ss = 0.f;
    for (int d = lane; d < Dv.D; d += WAVE) { const float v = x[d] / n1; o[d] = v; ss += v * v; }
    const float n2 = sqrtf(wave_sum(ss));
    for (int d = lane; d < Dv.D; d += WAVE) o[d] = o[d] / n2;
}

// ------------------------------------------------------------------------------------------------ 2. embedding distance
constexpr int CT = 16, CK = 64;
__global__ void __launch_bounds__(BLOCK) botsort_cosine_kernel(BoDev Dv, BoP P, BoIn in)
{
    __shared__ float s_t[CT][CK + 1], s_d[CT][CK + 1];
    const int s = blockIdx.z, p0 = blockIdx.y * CT, n0 = blockIdx.x * CT, tid = threadIdx.x;
    const int *hdr = Dv.hdr + (size_t)s * OH_COUNT;
    const int n_trk = hdr[OH_NTRK], T = n_trk + hdr[OH_NLOST];
    const int N = in.counts[(size_t)s * in.count_stride];
    if (hdr[OH_ERR] != 0 || p0 >= T || N <= 0 || N > Dv.MAXD || n0 >= N) return;
    const int ti = tid >> 4, nj = tid & 15;
    const int D = Dv.D;
    // rows staged by this thread: row (tid >> 4) of each tile, 4 consecutive floats at column (tid & 15) * 4
    const int lp = p0 + ti, ln = n0 + ti;
    const float *trow = nullptr, *drow = nullptr;
    if (lp < T) {
        const int slot = lp < n_trk ? Dv.tracked[(size_t)s * Dv.MAXT + lp] : Dv.lost[(size_t)s * Dv.MAXT + lp - n_trk];
        trow = Dv.feat + ((size_t)s * Dv.MAXT + slot) * D;
    }
    if (ln < N) drow = Dv.dfeat + ((size_t)s * Dv.MAXD + ln) * D;
    double uv = 0, uu = 0, vv = 0;
    for (int k0 = 0; k0 < D; k0 += CK) {
        const int kc = (tid & 15) * 4;
        float4 a = make_float4(0, 0, 0, 0), b = a;
        if (trow && k0 + kc < D) a = *(const float4 *)(trow + k0 + kc);
        if (drow && k0 + kc < D) b = *(const float4 *)(drow + k0 + kc);
        s_t[ti][kc] = a.x; s_t[ti][kc + 1] = a.y; s_t[ti][kc + 2] = a.z; s_t[ti][kc + 3] = a.w;
        s_d[ti][kc] = b.x; s_d[ti][kc + 1] = b.y; s_d[ti][kc + 2] = b.z; s_d[ti][kc + 3] = b.w;
        __syncthreads();
        const int kn = D - k0 < CK ? D - k0 : CK;
        for (int k = 0; k < kn; ++k) {
            const double u = (double)s_t[ti][k], v = (double)s_d[nj][k];
            uv += u * v; uu += u * u; vv += v * v;
        }
        __syncthreads();
    }
    const int p = p0 + ti, n = n0 + nj;
    if (p >= T || n >= N) return;
    const double conf = in.dets[((size_t)s * in.stream_stride_dets + n) * 7 + 4];
    if (!(conf > P.min_conf && conf > P.track_high)) return;
    double c = uv / (sqrt(uu) * sqrt(vv));
    if (fabs(c) > 1.) c = c < 0 ? -1. : 1.;             // scipy clips the cosine
    const double r = 1. - c;
    Dv.dist[((size_t)s * Dv.MAXT + p) * Dv.MAXD + n] = r > 0.0 ? r : 0.0;
}

// ------------------------------------------------------------------------------------------------ 3. association
// multi_predict (kalman_filter.py:154-191): noise relative to w and h; float32 while every mean in the batch still is
__device__ __forceinline__ void kfo_predict(double (&mean)[8], double (&cov)[64], bool all_f32)
{
    double q[8];
    if (all_f32) {
        const float w = (float)mean[2], h = (float)mean[3];
        const float sd[8] = {(float)W_POS * w, (float)W_POS * h, (float)W_POS * w, (float)W_POS * h, (float)W_VEL * w, (float)W_VEL * h, (float)W_VEL * w, (float)W_VEL * h};
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float v = sd[i] * sd[i]; q[i] = (double)v; }
    } else {
        const double w = mean[2], h = mean[3];
        const double sd[8] = {W_POS * w, W_POS * h, W_POS * w, W_POS * h, W_VEL * w, W_VEL * h, W_VEL * w, W_VEL * h};
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = sd[i] * sd[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) cov[i * 8 + j] = cov[i * 8 + j] + cov[(i + 4) * 8 + j];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cov[i * 8 + j] = cov[i * 8 + j] + cov[i * 8 + j + 4];
#pragma unroll
    for (int i = 0; i < 8; ++i) cov[i * 9] += q[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = mean[i] + mean[i + 4];
}

// STrack.tlbr (bot_sort.py:168-186) in the mean's dtype, cast to float32 (matching.py:63-66)
__device__ __forceinline__ void bo_tlbr32(const BTrk &T, float *o)
{
    if (T.i(OI_F32)) {
        float r0 = (float)T.d(OD_MEAN), r1 = (float)T.d(OD_MEAN + 1);
        const float r2 = (float)T.d(OD_MEAN + 2), r3 = (float)T.d(OD_MEAN + 3);
        r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = r0; o[1] = r1; o[2] = r2 + r0; o[3] = r3 + r1;
    } else {
        double r0 = T.d(OD_MEAN), r1 = T.d(OD_MEAN + 1);
        const double r2 = T.d(OD_MEAN + 2), r3 = T.d(OD_MEAN + 3);
        r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = (float)r0; o[1] = (float)r1; o[2] = (float)(r2 + r0); o[3] = (float)(r3 + r1);
    }
}

// update_cls (bot_sort.py:52-69); returns false when the history is full
__device__ __forceinline__ bool bo_update_cls(const BTrk &K, double cls, double score)
{
    const int nh = K.i(OI_NHIST);
    if (nh > 0) {
        double max_freq = 0; bool found = false;
        for (int i = 0; i < nh; ++i) {
            if (cls == K.d(OD_HCLS + i)) { K.d(OD_HFREQ + i) = K.d(OD_HFREQ + i) + score; found = true; }
            if (K.d(OD_HFREQ + i) > max_freq) { max_freq = K.d(OD_HFREQ + i); K.d(OD_CLS) = K.d(OD_HCLS + i); }
        }
        if (!found) {
            if (nh >= MAXCLS) return false;
            K.d(OD_HCLS + nh) = cls; K.d(OD_HFREQ + nh) = score; K.i(OI_NHIST) = nh + 1; K.d(OD_CLS) = cls;
        }
    } else { K.d(OD_HCLS) = cls; K.d(OD_HFREQ) = score; K.i(OI_NHIST) = 1; K.d(OD_CLS) = cls; }
    return true;
}

__global__ void __launch_bounds__(BLOCK, 1)
botsort_kernel(BoDev Dv, BoP P, BoIn in, tlk_botsort_row *__restrict__ rows_all, size_t rows_stream_stride, int out_cap,
               int *__restrict__ out_counts, size_t oc_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int MAXT = Dv.MAXT, MAXD = Dv.MAXD, D = Dv.D;
    ByLds L;
    int *hdr = Dv.hdr + (size_t)s * OH_COUNT;
    int *tracked = Dv.tracked + (size_t)s * MAXT, *lost = Dv.lost + (size_t)s * MAXT, *freestk = Dv.freestk + (size_t)s * MAXT;
    double *ebuf = Dv.ebuf + (size_t)s * MAXT * MAXD;
    const double *dist = Dv.dist + (size_t)s * MAXT * MAXD;
    double *gl = Dv.gl + (size_t)s * MAXT * GLD;
    float *feat = Dv.feat + (size_t)s * MAXT * D, *dfeat = Dv.dfeat + (size_t)s * MAXD * D;
    const size_t stride = (size_t)Dv.S * MAXT;
    auto trk_at = [&](int slot) { BTrk T; T.fd = Dv.fd + (size_t)s * MAXT + slot; T.fi = Dv.fi + (size_t)s * MAXT + slot; T.stride = stride; return T; };
    tlk_botsort_row *rows = rows_all + (size_t)s * rows_stream_stride;
    int *out_count = out_counts + (size_t)s * oc_stride;
    const size_t dbase = (size_t)s * in.stream_stride_dets;
    const int n_in = in.counts[(size_t)s * in.count_stride];

    if (hdr[OH_ERR] != 0) { if (tid == 0) *out_count = hdr[OH_ERR]; return; }
    if (n_in > MAXD || n_in < 0) { if (tid == 0) { hdr[OH_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    if (P.wrapper_mode && n_in == 0) { if (tid == 0) *out_count = 0; return; }      // bot_sort_api.py:59-60
    const int fid = hdr[OH_FRAME] + 1;
    int n_trk = hdr[OH_NTRK], n_lost = hdr[OH_NLOST], nfree = hdr[OH_NFREE], next_id = hdr[OH_COUNT_ID];
    const int cost_lds_entries = by_carve_frame(smem, Dv.lds_bytes, Dv.big_ws + (size_t)s * Dv.big_stride, MAXT, MAXD, n_trk + n_lost + n_in, n_in,
                                                Dv.alive + (size_t)s * MAXT, L);

    // wrapper filter inputs[:, 4] > min_confidence (bot_sort_api.py:62), then the score split (:293-309)
    const int N = block_compact(n_in, [&](int i) { return in.dets[(dbase + i) * 7 + 4] > P.min_conf; }, [&](int i, int pos) { L.sel[pos] = i; }, L.scan);
    __syncthreads();
    for (int j = tid; j < N; j += BLOCK) {
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        const double cx = (d[0] + d[2]) / 2, cy = (d[1] + d[3]) / 2, w = d[2] - d[0], h = d[3] - d[1];      // xyxy2xywh in float64
        float t0 = (float)cx, t1 = (float)cy, t2, t3;
        if (d[4] > P.track_high) { t2 = (float)w; t3 = (float)h; }             // STrack(xywh row, ...): the row is kept as "_tlwh" (:320)
        else { t2 = (float)(w - cx); t3 = (float)(h - cy); }                   // STrack(tlbr_to_tlwh(xywh row), ...) (:397)
        L.dbox[j * 4] = t0; L.dbox[j * 4 + 1] = t1; L.dbox[j * 4 + 2] = t2 + t0; L.dbox[j * 4 + 3] = t3 + t1;                 // tlbr
        L.dxyah[j * 4] = t0 + t2 / 2; L.dxyah[j * 4 + 1] = t1 + t3 / 2; L.dxyah[j * 4 + 2] = t2; L.dxyah[j * 4 + 3] = t3;     // tlwh_to_xywh
        L.dscore[j] = d[4];
    }
    __syncthreads();
    const int nhi = block_compact(N, [&](int j) { return L.dscore[j] > P.track_high; }, [&](int j, int pos) { L.hi[pos] = j; }, L.scan);
    const int nlo = block_compact(N, [&](int j) { const double c = L.dscore[j]; return c > 0.1 && c < P.track_high; },
                                  [&](int j, int pos) { L.lo[pos] = j; }, L.scan);
    // unconfirmed / pool = joint(activated tracked, lost) (:326-335); ppos / upos = row of the precomputed distance matrix
    const int n_unconf = block_compact(n_trk, [&](int p) { return trk_at(tracked[p]).i(OI_ACT) == 0; }, [&](int p, int pos) { L.unconf[pos] = tracked[p]; L.upos[pos] = p; }, L.scan);
    const int n_act0 = block_compact(n_trk, [&](int p) { return trk_at(tracked[p]).i(OI_ACT) != 0; }, [&](int p, int pos) { L.pool[pos] = tracked[p]; L.ppos[pos] = p; }, L.scan);
    for (int q = tid; q < n_lost; q += BLOCK) { L.pool[n_act0 + q] = lost[q]; L.ppos[n_act0 + q] = n_trk + q; }
    const int n_pool = n_act0 + n_lost;
    for (int p = tid; p < MAXT; p += BLOCK) L.alive[p] = 0;
    __syncthreads();
    // multi_predict (:79-91), then multi_gmc with this frame's camera-motion warp (:93-109; bot_sort.py:341-343): pool and
    // unconfirmed become float64, and each pool track's gate (project(): mean, Cholesky of the projected covariance) is prepared
    // for fuse_motion from the WARPED state
    const double *warp = in.warps ? in.warps + (size_t)s * in.warp_stride : nullptr;
    {
        int f32all = 1;
        for (int p = tid; p < n_pool; p += BLOCK) f32all &= trk_at(L.pool[p]).i(OI_F32) != 0;
        const int all_f32 = __syncthreads_and(f32all);
        for (int p = tid; p < n_pool; p += BLOCK) {
            const BTrk Kt = trk_at(L.pool[p]);
            double mean[8], cov[64];
#pragma unroll
            for (int k = 0; k < 8; ++k) mean[k] = Kt.d(OD_MEAN + k);
#pragma unroll
            for (int k = 0; k < 64; ++k) cov[k] = Kt.d(OD_COV + k);
            if (Kt.i(OI_STATE) != OT_TRACKED) { mean[6] = 0; mean[7] = 0; }
            kfo_predict(mean, cov, all_f32 != 0);
            if (warp) bo_gmc_apply(mean, cov, warp);
#pragma unroll
            for (int k = 0; k < 8; ++k) Kt.d(OD_MEAN + k) = mean[k];
#pragma unroll
            for (int k = 0; k < 64; ++k) Kt.d(OD_COV + k) = cov[k];
            Kt.i(OI_F32) = 0;
            double S[16], Lc[16];
            const double sw = W_POS * mean[2], sh = W_POS * mean[3];
            const double sd2[4] = {sw * sw, sh * sh, sw * sw, sh * sh};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j] + (i == j ? sd2[i] : 0.0);
            chol4_full(S, Lc);
            double *g = gl + (size_t)p * GLD;
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = mean[i];
#pragma unroll
            for (int i = 0; i < 16; ++i) g[4 + i] = Lc[i];
        }
        for (int p = tid; p < n_unconf; p += BLOCK) {
            const BTrk Kt = trk_at(L.unconf[p]);
            Kt.i(OI_F32) = 0;
            if (warp) {
                double mean[8], cov[64];
#pragma unroll
                for (int k = 0; k < 8; ++k) mean[k] = Kt.d(OD_MEAN + k);
#pragma unroll
                for (int k = 0; k < 64; ++k) cov[k] = Kt.d(OD_COV + k);
                bo_gmc_apply(mean, cov, warp);
#pragma unroll
                for (int k = 0; k < 8; ++k) Kt.d(OD_MEAN + k) = mean[k];
#pragma unroll
                for (int k = 0; k < 64; ++k) Kt.d(OD_COV + k) = cov[k];
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // KF update of a matched track (STrack.update :142-166 / re_activate :125-140); one thread per match
    auto apply = [&](int slot, int j, bool reactivate) {
        const BTrk Kt = trk_at(slot);
        double mean[8], cov[64], z[4], sd[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) mean[k] = Kt.d(OD_MEAN + k);
#pragma unroll
        for (int k = 0; k < 64; ++k) cov[k] = Kt.d(OD_COV + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (double)L.dxyah[j * 4 + k];
        sd[0] = W_POS * mean[2]; sd[1] = W_POS * mean[3]; sd[2] = sd[0]; sd[3] = sd[1];          // project() (kalman_filter.py:142-147)
        kf8_update_sd(mean, cov, z, sd);
#pragma unroll
        for (int k = 0; k < 8; ++k) Kt.d(OD_MEAN + k) = mean[k];
#pragma unroll
        for (int k = 0; k < 64; ++k) Kt.d(OD_COV + k) = cov[k];
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        Kt.i(OI_F32) = 0;
        if (reactivate) Kt.i(OI_TLEN) = 0; else Kt.i(OI_TLEN) = Kt.i(OI_TLEN) + 1;
        Kt.i(OI_FID) = fid; Kt.i(OI_STATE) = OT_TRACKED; Kt.i(OI_ACT) = 1;
        Kt.d(OD_SCORE) = d[4]; Kt.d(OD_TLID) = d[6];
        if (!bo_update_cls(Kt, d[5], d[4])) hdr[OH_ERR] = TLK_ECAPACITY;
    };
    // update_features(new_track.curr_feat) (:42-50) for nm matches of slots(k) with filtered detections dets(k): one wavefront
    // per match; the detection's array is normalised in place a third time, the EMA and its renormalisation are float32
    auto smooth = [&](int nm, auto slot_of, auto det_of) {
        const float a = (float)0.9, b1 = (float)(1 - 0.9);
        for (int k = wv; k < nm; k += NWAVES) {
            float *sf = feat + (size_t)slot_of(k) * D, *df = dfeat + (size_t)L.sel[det_of(k)] * D;
            float ss = 0.f;
            for (int d = lane; d < D; d += WAVE) { const float v = df[d]; ss += v * v; }
            const float n1 = sqrtf(wave_sum(ss));
            ss = 0.f;
            for (int d = lane; d < D; d += WAVE) { const float v = a * sf[d] + b1 * (df[d] / n1); sf[d] = v; ss += v * v; }
            const float n2 = sqrtf(wave_sum(ss));
            for (int d = lane; d < D; d += WAVE) sf[d] = sf[d] / n2;
        }
    };

    // ---- first association: embedding distance fused with the Mahalanobis gate (:346-379) ----
    // fuse_motion (matching.py:188-233) solves its triangular systems for all nhi measurements in one scipy.linalg.solve_triangular call, whose
    // operation order depends on nhi == 1 (tlk_strongsort_common.hpp): finish the gate rows for this frame's count
    const bool gate_single = nhi == 1;
    for (int p = tid; p < n_pool; p += BLOCK) { L.pre[p] = trk_at(L.pool[p]).i(OI_STATE); gate_row_finish(gl + (size_t)p * GLD, 4, gate_single); }
    __syncthreads();
    const AsgOut A1 = lapjv_assign(n_pool, nhi, P.match_thresh, [&](int r, int c) {
        const int j = L.hi[c];
        double m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = (double)L.dxyah[j * 4 + k];
        const double gd = gating_from(gl + (size_t)r * GLD, m, 4, gate_single);
        double e = dist[(size_t)L.ppos[r] * MAXD + L.sel[j]];
        if (gd > CHI2_4) e = INFINITY;
        return P.lambda_ * e + (1 - P.lambda_) * gd;
    }, ebuf, L.cost, cost_lds_entries, L);
    for (int k = tid; k < A1.nm; k += BLOCK) apply(L.pool[L.m_r[k]], L.hi[L.m_c[k]], L.pre[L.m_r[k]] != OT_TRACKED);
    smooth(A1.nm, [&](int k) { return L.pool[L.m_r[k]]; }, [&](int k) { return L.hi[L.m_c[k]]; });
    const int n_ref = block_compact(A1.nm, [&](int k) { return L.pre[L.m_r[k]] != OT_TRACKED; }, [&](int k, int pos) { L.refind[pos] = L.pool[L.m_r[k]]; }, L.scan);
    for (int k = tid; k < A1.n_uc; k += BLOCK) L.udet1[k] = L.hi[L.u_c[k]];
    const int n_udet1 = A1.n_uc;
    // ---- second association: low-score detections against the still-Tracked rest, IoU only (:394-419) ----
    const int n_rtr = block_compact(A1.n_ur, [&](int q) { return L.pre[L.u_r[q]] == OT_TRACKED; }, [&](int q, int pos) { L.rtr[pos] = L.pool[L.u_r[q]]; }, L.scan);
    __syncthreads();
    for (int p = tid; p < n_rtr; p += BLOCK) bo_tlbr32(trk_at(L.rtr[p]), L.tbox + p * 4);
    __syncthreads();
    const AsgOut A2 = lapjv_assign(n_rtr, nlo, 0.5, [&](int r, int c) {
        return (double)(float)(1 - bbox_iou32(L.tbox + r * 4, L.dbox + L.lo[c] * 4));
    }, ebuf, L.cost, cost_lds_entries, L);
    for (int k = tid; k < A2.nm; k += BLOCK) apply(L.rtr[L.m_r[k]], L.lo[L.m_c[k]], false);
    for (int k = tid; k < A2.n_ur; k += BLOCK) { const int slot = L.rtr[L.u_r[k]]; trk_at(slot).i(OI_STATE) = OT_LOST; L.newlost[k] = slot; }   // mark_lost
    const int n_newlost = A2.n_ur;
    __syncthreads();
    // ---- unconfirmed tracks against the remaining high-score detections (:421-441) ----
    for (int p = tid; p < n_unconf; p += BLOCK) bo_tlbr32(trk_at(L.unconf[p]), L.tbox + p * 4);
    __syncthreads();
    const AsgOut A3 = lapjv_assign(n_unconf, n_udet1, 0.7, [&](int r, int c) {
        const int j = L.udet1[c];
        const float c32 = 1 - bbox_iou32(L.tbox + r * 4, L.dbox + j * 4);        // iou_distance (float32)
        const float sim = 1 - c32;
        const double iou_d = 1 - (double)sim * L.dscore[j];                      // fuse_score in float64
        double e = dist[(size_t)L.upos[r] * MAXD + L.sel[j]] / 2.0;
        if (e > P.appearance) e = 1.0;
        if (c32 > (float)P.proximity) e = 1.0;                                   // ious_dists_mask on the float32 distances
        return iou_d < e ? iou_d : e;
    }, ebuf, L.cost, cost_lds_entries, L);
    for (int k = tid; k < A3.nm; k += BLOCK) apply(L.unconf[L.m_r[k]], L.udet1[L.m_c[k]], false);
    smooth(A3.nm, [&](int k) { return L.unconf[L.m_r[k]]; }, [&](int k) { return L.udet1[L.m_c[k]]; });
    for (int k = tid; k < A3.n_ur; k += BLOCK) { const int slot = L.unconf[L.u_r[k]]; trk_at(slot).i(OI_STATE) = OT_REMOVED; L.removed[k] = slot; }
    int n_removed = A3.n_ur;
    if (A1.err | A2.err | A3.err) { if (tid == 0) { hdr[OH_ERR] = TLK_EINTERNAL; *out_count = TLK_EINTERNAL; } return; }      // uniform: an assignment solver hit its loop bound
    __syncthreads();
    // ---- new tracks from the still unmatched detections with score >= new_track_thresh (:443-450) ----
    const int n_new = block_compact(A3.n_uc, [&](int q) { return !(L.dscore[L.udet1[L.u_c[q]]] < P.new_track); }, [&](int q, int pos) { L.rem[pos] = L.udet1[L.u_c[q]]; }, L.scan);
    if (n_new > nfree) { if (tid == 0) { hdr[OH_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    __syncthreads();
    for (int k = tid; k < n_new; k += BLOCK) {
        const int j = L.rem[k], slot = freestk[nfree - 1 - k];
        L.newtrk[k] = slot;
        const BTrk Kt = trk_at(slot);
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        // kalman_filter.py:55-86 on a float32 measurement: a list of float32 products, np.square keeps float32
        const float w = L.dxyah[j * 4 + 2], h = L.dxyah[j * 4 + 3];
        const float pw = (float)(2 * W_POS) * w, ph = (float)(2 * W_POS) * h, vw = (float)(10 * W_VEL) * w, vh = (float)(10 * W_VEL) * h;
        const float sd[8] = {pw, ph, pw, ph, vw, vh, vw, vh};
#pragma unroll
        for (int q = 0; q < 64; ++q) Kt.d(OD_COV + q) = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { Kt.d(OD_MEAN + q) = (double)L.dxyah[j * 4 + q]; Kt.d(OD_MEAN + 4 + q) = 0.0; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float v = sd[q] * sd[q]; Kt.d(OD_COV + q * 9) = (double)v; }
        Kt.i(OI_F32) = 1; Kt.i(OI_TID) = next_id + 1 + k;
        Kt.i(OI_TLEN) = 0; Kt.i(OI_STATE) = OT_TRACKED; Kt.i(OI_ACT) = fid == 1 ? 1 : 0; Kt.i(OI_FID) = fid; Kt.i(OI_SFID) = fid; Kt.i(OI_INREM) = 0;
        Kt.d(OD_SCORE) = d[4]; Kt.d(OD_TLID) = d[6];
        Kt.i(OI_NHIST) = 0;
        bo_update_cls(Kt, d[5], d[4]);                                           // the detection STrack's own one-entry history
    }
    __syncthreads();
    for (int k = wv; k < n_new; k += NWAVES) {                                   // the detection's (twice normalised) array IS the track's smooth_feat
        float *sf = feat + (size_t)L.newtrk[k] * D;
        const float *df = dfeat + (size_t)L.sel[L.rem[k]] * D;
        for (int d = lane; d < D; d += WAVE) sf[d] = df[d];
    }
    nfree -= n_new; next_id += n_new;
    __syncthreads();
    // ---- lost tracks that timed out (:452-456) ----
    {
        const int nto = block_compact(n_lost, [&](int q) { const BTrk Kt = trk_at(lost[q]); return fid - Kt.i(OI_FID) > P.max_time_lost; },
                                      [&](int q, int pos) { L.removed[n_removed + pos] = lost[q]; }, L.scan);
        __syncthreads();
        for (int k = tid; k < nto; k += BLOCK) trk_at(L.removed[n_removed + k]).i(OI_STATE) = OT_REMOVED;
        n_removed += nto;
    }
    __syncthreads();
    // ---- list bookkeeping (:458-466), identical to ByteTrack's ----
    for (int p = tid; p < n_trk; p += BLOCK) L.alive[tracked[p]] = 1;
    for (int q = tid; q < n_lost; q += BLOCK) L.alive[lost[q]] = 1;
    for (int k = tid; k < n_new; k += BLOCK) L.alive[L.newtrk[k]] = 1;
    __syncthreads();
    int nn = block_compact(n_trk, [&](int p) { return trk_at(tracked[p]).i(OI_STATE) == OT_TRACKED; }, [&](int p, int pos) { L.ntr[pos] = tracked[p]; }, L.scan);
    // activated_starcks order: first-stage, second-stage and unconfirmed matches are already in tracked; only the new tracks append
    for (int k = tid; k < n_new; k += BLOCK) L.ntr[nn + k] = L.newtrk[k];
    for (int k = tid; k < n_ref; k += BLOCK) L.ntr[nn + n_new + k] = L.refind[k];
    nn += n_new + n_ref;
    int nl = block_compact(n_lost, [&](int q) { const BTrk Kt = trk_at(lost[q]); return Kt.i(OI_STATE) != OT_TRACKED && Kt.i(OI_INREM) == 0; },
                           [&](int q, int pos) { L.nlost[pos] = lost[q]; }, L.scan);
    nl += block_compact(n_newlost, [&](int k) { return trk_at(L.newlost[k]).i(OI_INREM) == 0; }, [&](int k, int pos) { L.nlost[nl + pos] = L.newlost[k]; }, L.scan);
    __syncthreads();
    for (int k = tid; k < n_removed; k += BLOCK) trk_at(L.removed[k]).i(OI_INREM) = 1;
    if (nn > MAXT || nl > MAXT) { if (tid == 0) { hdr[OH_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    // remove_duplicate_stracks (:532-545)
    for (int p = tid; p < nn; p += BLOCK) { L.dupa[p] = 0; bo_tlbr32(trk_at(L.ntr[p]), L.tbox + p * 4); }
    for (int q = tid; q < nl; q += BLOCK) L.dupb[q] = 0;
    __syncthreads();
    for (int e = tid; e < nn * nl; e += BLOCK) {
        const int p = e / nl, q = e - p * nl;
        float lb[4];
        bo_tlbr32(trk_at(L.nlost[q]), lb);
        const float pd = 1 - bbox_iou32(L.tbox + p * 4, lb);
        if (pd < (float)0.15) {
            const BTrk Ka = trk_at(L.ntr[p]), Kb = trk_at(L.nlost[q]);
            const int timep = Ka.i(OI_FID) - Ka.i(OI_SFID), timeq = Kb.i(OI_FID) - Kb.i(OI_SFID);
            if (timep > timeq) L.dupb[q] = 1; else L.dupa[p] = 1;
        }
    }
    __syncthreads();
    const int nn2 = block_compact(nn, [&](int p) { return L.dupa[p] == 0; }, [&](int p, int pos) { tracked[pos] = L.ntr[p]; }, L.scan);
    const int nl2 = block_compact(nl, [&](int q) { return L.dupb[q] == 0; }, [&](int q, int pos) { lost[pos] = L.nlost[q]; }, L.scan);
    __syncthreads();
    for (int p = tid; p < nn2; p += BLOCK) L.alive[tracked[p]] = 0;
    for (int q = tid; q < nl2; q += BLOCK) L.alive[lost[q]] = 0;
    __syncthreads();
    const int ndead = block_compact(MAXT, [&](int slot) { return L.alive[slot] != 0; }, [&](int slot, int pos) { freestk[nfree + pos] = slot; }, L.scan);
    nfree += ndead;
    __syncthreads();
    if (tid == 0) { hdr[OH_NTRK] = nn2; hdr[OH_NLOST] = nl2; hdr[OH_NFREE] = nfree; hdr[OH_COUNT_ID] = next_id; hdr[OH_FRAME] = fid; }
    // ---- outputs (:468-485): activated tracks of the tracked list, xywh2xyxy of tlwh in the mean's dtype ----
    const int nrows = block_compact(nn2, [&](int p) { return trk_at(tracked[p]).i(OI_ACT) != 0; },
                                    [&](int p, int pos) {
                                        if (pos >= out_cap) return;
                                        const BTrk Kt = trk_at(tracked[p]);
                                        tlk_botsort_row r;
                                        if (Kt.i(OI_F32)) {
                                            float r0 = (float)Kt.d(OD_MEAN), r1 = (float)Kt.d(OD_MEAN + 1);
                                            const float r2 = (float)Kt.d(OD_MEAN + 2), r3 = (float)Kt.d(OD_MEAN + 3);
                                            r0 -= r2 / 2; r1 -= r3 / 2;
                                            const float hw = r2 / 2, hh = r3 / 2;
                                            r.ltrb[0] = r0 - hw; r.ltrb[1] = r1 - hh; r.ltrb[2] = r0 + hw; r.ltrb[3] = r1 + hh;
                                        } else {
                                            double r0 = Kt.d(OD_MEAN), r1 = Kt.d(OD_MEAN + 1);
                                            const double r2 = Kt.d(OD_MEAN + 2), r3 = Kt.d(OD_MEAN + 3);
                                            r0 -= r2 / 2; r1 -= r3 / 2;
                                            const double hw = r2 / 2, hh = r3 / 2;
                                            r.ltrb[0] = r0 - hw; r.ltrb[1] = r1 - hh; r.ltrb[2] = r0 + hw; r.ltrb[3] = r1 + hh;
                                        }
                                        r.det_id = (long long)Kt.d(OD_TLID); r.track_id = Kt.i(OI_TID);
                                        r.score = Kt.d(OD_SCORE); r.cls = Kt.d(OD_CLS);
                                        rows[pos] = r;
                                    }, L.scan);
    __syncthreads();
    if (tid == 0) { const int err = hdr[OH_ERR]; *out_count = err != 0 ? err : (nrows > out_cap ? TLK_ECAPACITY : nrows); }
}

__global__ void botsort_reset_kernel(BoDev D, int stream, int keep_ids)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        int *hdr = D.hdr + (size_t)s * OH_COUNT;
        for (int k = threadIdx.x; k < D.MAXT; k += blockDim.x) D.freestk[(size_t)s * D.MAXT + k] = D.MAXT - 1 - k;
        if (threadIdx.x == 0) { hdr[OH_NTRK] = 0; hdr[OH_NLOST] = 0; hdr[OH_NFREE] = D.MAXT; if (!keep_ids) hdr[OH_COUNT_ID] = 0; hdr[OH_FRAME] = 0; hdr[OH_ERR] = 0; }
    }
}

__global__ void botsort_gather_kernel(BoDev D, int stream, int which, long long *ids, double *mean, double *cov, long long *state5, float *feat,
                                      int cap, int *n_out)
{
    const int n = D.hdr[(size_t)stream * OH_COUNT + (which ? OH_NLOST : OH_NTRK)];
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = n;
    const int *list = (which ? D.lost : D.tracked) + (size_t)stream * D.MAXT;
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int p = blockIdx.x; p < n && p < cap; p += gridDim.x) {
        const int slot = list[p];
        const double *fd = D.fd + (size_t)stream * D.MAXT + slot;
        const int *fi = D.fi + (size_t)stream * D.MAXT + slot;
        if (ids && threadIdx.x == 0) ids[p] = fi[(size_t)OI_TID * stride];
        if (mean) for (int k = threadIdx.x; k < 8; k += blockDim.x) mean[(size_t)p * 8 + k] = fd[(size_t)(OD_MEAN + k) * stride];
        if (cov) for (int k = threadIdx.x; k < 64; k += blockDim.x) cov[(size_t)p * 64 + k] = fd[(size_t)(OD_COV + k) * stride];
        if (feat) for (int k = threadIdx.x; k < D.D; k += blockDim.x) feat[(size_t)p * D.D + k] = D.feat[((size_t)stream * D.MAXT + slot) * D.D + k];
        if (state5 && threadIdx.x == 0) {
            state5[(size_t)p * 5] = fi[(size_t)OI_STATE * stride]; state5[(size_t)p * 5 + 1] = fi[(size_t)OI_ACT * stride];
            state5[(size_t)p * 5 + 2] = fi[(size_t)OI_FID * stride]; state5[(size_t)p * 5 + 3] = fi[(size_t)OI_SFID * stride];
            state5[(size_t)p * 5 + 4] = fi[(size_t)OI_TLEN * stride];
        }
    }
}

}  // namespace

struct tlk_botsort {
    BoDev D; BoP P; int device; size_t smem;
    double *d_dets; float *d_feats; int *d_cnt, *d_ocnt; tlk_botsort_row *d_rows; double *d_warp;
    int out_cap, cmc_method;
};

static void bo_free(tlk_botsort *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    void *ptrs[] = {h->D.fd, h->D.fi, h->D.hdr, h->D.tracked, h->D.lost, h->D.freestk, h->D.feat, h->D.dfeat, h->D.dist, h->D.gl, h->D.ebuf, h->D.alive, h->D.big_ws,
                    h->d_dets, h->d_feats, h->d_cnt, h->d_ocnt, h->d_rows, h->d_warp};
    for (void *p : ptrs) if (p) hipFree(p);
    delete h;
}

static int bo_launch_frame(tlk_botsort *h, const BoDev &Dv, int n_streams, const BoIn &in, tlk_botsort_row *rows, size_t rows_stream_stride,
                           int out_cap, int *out_counts, size_t oc_stride, hipStream_t st)
{
    hipLaunchKernelGGL(botsort_prep_kernel, dim3((Dv.MAXD + NWAVES - 1) / NWAVES, n_streams), dim3(BLOCK), 0, st, Dv, h->P, in);
    hipLaunchKernelGGL(botsort_cosine_kernel, dim3((Dv.MAXD + CT - 1) / CT, (Dv.MAXT + CT - 1) / CT, n_streams), dim3(BLOCK), 0, st, Dv, h->P, in);
    hipLaunchKernelGGL(botsort_kernel, dim3(n_streams), dim3(BLOCK), h->smem, st, Dv, h->P, in, rows, rows_stream_stride, out_cap, out_counts, oc_stride);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_botsort_create(const tlk_botsort_params *p, int n_streams, int device, tlk_botsort **out)
{
    if (!p || !out) return fail(TLK_EINVAL, "tlk_botsort_create: null pointer");
    if (n_streams < 1) return fail(TLK_EINVAL, "tlk_botsort_create: n_streams must be >= 1");
    if (p->dim < 4 || p->dim > 4096 || p->dim % 4) return fail(TLK_EINVAL, "tlk_botsort_create: dim must be a multiple of 4 in [4, 4096]");
    if (p->cmc_method < 0 || p->cmc_method > 5) return fail(TLK_EINVAL, "tlk_botsort_create: cmc_method out of range (gmc.py:18-78: 0 none, 1 orb, 2 sift, 3 ecc, 4 sparseOptFlow, 5 file)");
    const int MAXT = p->max_tracks > 0 ? p->max_tracks : 256, MAXD = p->max_dets > 0 ? p->max_dets : 128;
    // capacity = allocation size (r04): LDS tiers while the scene fits, HBM lists beyond (by_carve_frame)
    if (MAXT > 16384 || MAXD > 1024) return fail(TLK_ECAPACITY, "tlk_botsort_create: max_tracks <= 16384 and max_dets <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_botsort_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_botsort_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_botsort *h = new tlk_botsort();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->cmc_method = p->cmc_method;
    h->P = BoP{p->track_high_thresh, p->new_track_thresh, p->match_thresh, p->proximity_thresh, p->appearance_thresh, p->lambda_, p->min_confidence,
               (int)(p->frame_rate / 30.0 * p->track_buffer), p->wrapper_mode};
    BoDev &D = h->D;
    D.S = n_streams; D.MAXT = MAXT; D.MAXD = MAXD; D.D = p->dim;
    const size_t budget = 160 * 1024 - 256;
    D.lds_bytes = (int)(budget & ~(size_t)15);
    D.big_stride = (bylds_bytes(MAXT, MAXD, MAXT + MAXD) + 16 + 255) & ~(size_t)255;
    h->smem = (size_t)D.lds_bytes;
    const size_t slots = (size_t)n_streams * MAXT;
    h->out_cap = MAXT;
#define BO_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { bo_free(h); return fail(TLK_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    BO_ALLOC(D.fd, sizeof(double) * OD_COUNT * slots);
    BO_ALLOC(D.fi, sizeof(int) * OI_COUNT * slots);
    BO_ALLOC(D.hdr, sizeof(int) * OH_COUNT * n_streams);
    BO_ALLOC(D.tracked, sizeof(int) * slots);
    BO_ALLOC(D.lost, sizeof(int) * slots);
    BO_ALLOC(D.freestk, sizeof(int) * slots);
    BO_ALLOC(D.feat, sizeof(float) * slots * D.D);
    BO_ALLOC(D.dfeat, sizeof(float) * (size_t)n_streams * MAXD * D.D);
    BO_ALLOC(D.dist, sizeof(double) * slots * MAXD);
    BO_ALLOC(D.gl, sizeof(double) * slots * GLD);
    BO_ALLOC(D.ebuf, sizeof(double) * slots * MAXD);
    BO_ALLOC(D.alive, sizeof(int) * slots);
    BO_ALLOC(D.big_ws, D.big_stride * (size_t)n_streams);
    BO_ALLOC(h->d_dets, sizeof(double) * 7 * MAXD);
    BO_ALLOC(h->d_feats, sizeof(float) * (size_t)MAXD * D.D);
    BO_ALLOC(h->d_cnt, sizeof(int));
    BO_ALLOC(h->d_ocnt, sizeof(int));
    BO_ALLOC(h->d_warp, sizeof(double) * 6);
    BO_ALLOC(h->d_rows, sizeof(tlk_botsort_row) * h->out_cap);
#undef BO_ALLOC
    hipError_t e = hipMemset(D.fd, 0, sizeof(double) * OD_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.fi, 0, sizeof(int) * OI_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.feat, 0, sizeof(float) * slots * D.D);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)botsort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) { bo_free(h); return fail(TLK_EHIP, std::string("tlk_botsort_create: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(botsort_reset_kernel, dim3(n_streams < 256 ? n_streams : 256), dim3(BLOCK), 0, 0, D, -1, 0);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { bo_free(h); return fail(TLK_EHIP, std::string("tlk_botsort_create: ") + hipGetErrorString(e)); }
    *out = h;
    return TLK_OK;
}

extern "C" int tlk_botsort_destroy(tlk_botsort *h) { bo_free(h); return TLK_OK; }

static int botsort_reset_impl(tlk_botsort *h, int stream, int keep_ids)
{
    if (!h) return fail(TLK_EINVAL, "tlk_botsort_reset: null handle");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_botsort_reset: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(botsort_reset_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, 0, h->D, stream, keep_ids);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipStreamSynchronize(0));
    return TLK_OK;
}

extern "C" int tlk_botsort_reset(tlk_botsort *h, int stream) { return botsort_reset_impl(h, stream, 0); }
extern "C" int tlk_botsort_reset_keep_ids(tlk_botsort *h, int stream) { return botsort_reset_impl(h, stream, 1); }

extern "C" int tlk_botsort_update_dev_gmc(tlk_botsort *h, const double *dets_dev, const float *feats_dev, const int32_t *counts_dev, const double *warps_dev,
                                          int n_frames, tlk_botsort_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_botsort_update_dev: null handle");
    if (n_frames < 0 || out_cap < 0) return fail(TLK_EINVAL, "tlk_botsort_update_dev: negative size");
    if (n_frames == 0) return TLK_OK;
    if (!dets_dev || !feats_dev || !counts_dev || !rows_dev || !out_counts_dev) return fail(TLK_EINVAL, "tlk_botsort_update_dev: null pointer");
    if (h->cmc_method != 0 && !warps_dev)
        return fail(TLK_EINVAL, "tlk_botsort_update_dev: this tracker was created with a camera-motion method: pass the frames' warps (tlk_botsort_update_dev_gmc)");
    TLK_HIP(hipSetDevice(h->device));
    const BoDev &D = h->D;
    for (int f = 0; f < n_frames; ++f) {
        BoIn in;
        in.dets = dets_dev + (size_t)f * D.MAXD * 7; in.feats = feats_dev + (size_t)f * D.MAXD * D.D; in.counts = (const int *)counts_dev + f;
        in.stream_stride_dets = (size_t)n_frames * D.MAXD; in.count_stride = (size_t)n_frames;
        in.warps = warps_dev ? warps_dev + (size_t)f * 6 : nullptr; in.warp_stride = (size_t)n_frames * 6;
        const int rc = bo_launch_frame(h, D, D.S, in, rows_dev + (size_t)f * out_cap, (size_t)n_frames * out_cap, out_cap, (int *)out_counts_dev + f,
                                       (size_t)n_frames, (hipStream_t)hip_stream);
        if (rc != TLK_OK) return rc;
    }
    return TLK_OK;
}

extern "C" int tlk_botsort_update_dev(tlk_botsort *h, const double *dets_dev, const float *feats_dev, const int32_t *counts_dev, int n_frames,
                                      tlk_botsort_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    return tlk_botsort_update_dev_gmc(h, dets_dev, feats_dev, counts_dev, nullptr, n_frames, rows_dev, out_cap, out_counts_dev, hip_stream);
}

extern "C" int tlk_botsort_update_gmc(tlk_botsort *h, int stream, const double *dets, const float *feats, int n, const double *warp6,
                                      tlk_botsort_row *rows, int cap, int *n_out)
{
    if (!h || !n_out) return fail(TLK_EINVAL, "tlk_botsort_update: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_botsort_update: stream out of range");
    if (n < 0 || (n > 0 && (!dets || !feats))) return fail(TLK_EINVAL, "tlk_botsort_update: bad detections");
    if (n > h->D.MAXD) return fail(TLK_ECAPACITY, "tlk_botsort_update: more detections than max_dets");
    if (h->cmc_method != 0 && !warp6)
        return fail(TLK_EINVAL, "tlk_botsort_update: this tracker was created with a camera-motion method: pass the frame's warp (tlk_botsort_update_gmc)");
    TLK_HIP(hipSetDevice(h->device));
    hipStream_t st = 0;
    if (n) {
        TLK_HIP(hipMemcpyAsync(h->d_dets, dets, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_feats, feats, sizeof(float) * (size_t)n * h->D.D, hipMemcpyHostToDevice, st));
    }
    TLK_HIP(hipMemcpyAsync(h->d_cnt, &n, sizeof(int), hipMemcpyHostToDevice, st));
    if (warp6) TLK_HIP(hipMemcpyAsync(h->d_warp, warp6, sizeof(double) * 6, hipMemcpyHostToDevice, st));
    BoDev V = h->D;
    const size_t sl = (size_t)stream * V.MAXT;
    V.fd += sl; V.fi += sl; V.hdr += (size_t)stream * OH_COUNT; V.tracked += sl; V.lost += sl; V.freestk += sl;
    V.feat += sl * V.D; V.dfeat += (size_t)stream * V.MAXD * V.D; V.dist += sl * V.MAXD; V.gl += sl * GLD; V.ebuf += sl * V.MAXD; V.alive += sl; V.big_ws += (size_t)stream * V.big_stride;
    BoIn in;
    in.dets = h->d_dets; in.feats = h->d_feats; in.counts = h->d_cnt; in.stream_stride_dets = 0; in.count_stride = 0;
    in.warps = warp6 ? h->d_warp : nullptr; in.warp_stride = 0;
    const int rc = bo_launch_frame(h, V, 1, in, h->d_rows, (size_t)0, h->out_cap, h->d_ocnt, (size_t)0, st);
    if (rc != TLK_OK) return rc;
    int rows_n = 0;
    TLK_HIP(hipMemcpyAsync(&rows_n, h->d_ocnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TLK_HIP(hipStreamSynchronize(st));
    if (rows_n < 0) return fail_stream(rows_n, "tlk_botsort_update", "max_tracks / max_dets / 16 classes per track");
    if (rows_n > cap) return fail(TLK_ECAPACITY, "tlk_botsort_update: output buffer too small");
    if (rows_n) TLK_HIP(hipMemcpy(rows, h->d_rows, sizeof(tlk_botsort_row) * rows_n, hipMemcpyDeviceToHost));
    *n_out = rows_n;
    return TLK_OK;
}

extern "C" int tlk_botsort_update(tlk_botsort *h, int stream, const double *dets, const float *feats, int n, tlk_botsort_row *rows, int cap, int *n_out)
{
    return tlk_botsort_update_gmc(h, stream, dets, feats, n, nullptr, rows, cap, n_out);
}

extern "C" int tlk_botsort_get_tracks(tlk_botsort *h, int stream, int which, int64_t *ids, double *mean, double *cov, int64_t *state5,
                                      float *smooth_feat, int cap, int *n_tracks)
{
    if (!h || !n_tracks) return fail(TLK_EINVAL, "tlk_botsort_get_tracks: null pointer");
    if (stream < 0 || stream >= h->D.S || cap < 0 || which < 0 || which > 1) return fail(TLK_EINVAL, "tlk_botsort_get_tracks: bad argument");
    TLK_HIP(hipSetDevice(h->device));
    const size_t c = cap > 0 ? cap : 1;
    long long *d_ids = nullptr, *d_st = nullptr; double *d_mean = nullptr, *d_cov = nullptr; float *d_feat = nullptr; int *d_n = nullptr;
    TLK_HIP(hipMalloc((void **)&d_ids, sizeof(long long) * c)); TLK_HIP(hipMalloc((void **)&d_st, sizeof(long long) * 5 * c));
    TLK_HIP(hipMalloc((void **)&d_mean, sizeof(double) * 8 * c)); TLK_HIP(hipMalloc((void **)&d_cov, sizeof(double) * 64 * c));
    TLK_HIP(hipMalloc((void **)&d_feat, sizeof(float) * c * h->D.D));
    TLK_HIP(hipMalloc((void **)&d_n, sizeof(int)));
    hipLaunchKernelGGL(botsort_gather_kernel, dim3(64), dim3(64), 0, 0, h->D, stream, which, d_ids, d_mean, d_cov, d_st, d_feat, cap, d_n);
    int n = 0;
    hipError_t e = hipMemcpy(&n, d_n, sizeof(int), hipMemcpyDeviceToHost);
    const int m = n < cap ? n : cap;
    if (e == hipSuccess && m > 0) {
        if (ids) e = hipMemcpy(ids, d_ids, sizeof(long long) * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && mean) e = hipMemcpy(mean, d_mean, sizeof(double) * 8 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && cov) e = hipMemcpy(cov, d_cov, sizeof(double) * 64 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && state5) e = hipMemcpy(state5, d_st, sizeof(long long) * 5 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && smooth_feat) e = hipMemcpy(smooth_feat, d_feat, sizeof(float) * (size_t)m * h->D.D, hipMemcpyDeviceToHost);
    }
    hipFree(d_ids); hipFree(d_st); hipFree(d_mean); hipFree(d_cov); hipFree(d_feat); hipFree(d_n);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_botsort_get_tracks: ") + hipGetErrorString(e));
    *n_tracks = n;
    return TLK_OK;
}
