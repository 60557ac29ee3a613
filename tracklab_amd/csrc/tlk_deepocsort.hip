// tlk_deepocsort.hip -- Deep-OC-SORT (plugins/track/deep_oc_sort, cmc_off, the (x, y, w, h) Kalman filter) as ONE fused HIP kernel
// per (stream, frame batch), the same shape as tlk_ocsort.hip: one 256-thread workgroup owns a stream and walks its frames.
//
// On top of OC-SORT's skeleton (ocsort.py:392-534):
//   * KalmanFilterNew with dim_x = 8 (ocsort.py:112-146): process / measurement noise relative to the state's w, h per call
//     (new_kf_process_noise / new_kf_measurement_noise, :77-89), R taken from the state BEFORE the ORU replay, the replay itself
//     (kalmanfilter.py:433-478) run with the filter's default R = I, Q = I on boxes that read (x, y, w, h) as (x, y, s, r);
//     `frozen` tracks predict with zero w / h velocity (:293-297)
//   * the embedding cost dets_embs @ trk_embs.T in float32, needed only where IoU > 0 (association.py:343): pairs with IoU > 0 are
//     found by a wavefront ballot over 64 entries and reduced one dot product per wavefront at a time (D <= 4096)
//   * compute_aw_max_metric (association.py:263-288): float32 row / column weights from the two largest entries
//   * update_emb (ocsort.py:254-256): float32 EMA with the detection's confidence-dependent alpha, float32 renormalisation
// Reference quirks kept: OCSort.update never increments frame_count (min_hits is inert), a track's conf is the one of its first
// detection, "scores" in the angle cost is the CLASS column. embedding_off crashes in the reference from the second frame
// (get_emb on a numpy array) and is rejected here; cmc needs cv2 and must be off; new_kf_off is not built.
#include "tlk_common.hpp"
#include "tlk_ocsort_common.hpp"

using namespace tlk;

namespace {

constexpr int RINGN = 8;       // observation ring slots; requires delta_t < RINGN

enum : int {
    GD_X = 0,               // 8   state (x, y, w, h, and their velocities)
    GD_P = GD_X + 8,        // 64
    GD_SX = GD_P + 64,      // 8   frozen state
    GD_SP = GD_SX + 8,      // 64
    GD_LZ = GD_SP + 64,     // 4   last non-None history_obs entry
    GD_CONF = GD_LZ + 4, GD_CLS, GD_TID,
    GD_LO,                  // 5   last_observation
    GD_VEL = GD_LO + 5,     // 2
    GD_OB = GD_VEL + 2,     // RINGN*5
    GD_COUNT = GD_OB + RINGN * 5
};
enum : int {
    GI_ID = 0, GI_TSU, GI_HITS, GI_STREAK, GI_AGE, GI_OBSERVED, GI_HAS_SAVED, GI_GAP, GI_HAS_VEL, GI_NOBS, GI_FROZEN,
    GI_OBAGE,               // RINGN
    GI_COUNT = GI_OBAGE + RINGN
};
enum : int { H_NTRK = 0, H_NEXTID, H_NFREE, H_ERR, H_COUNT = 8 };

struct DocDev {
    double *fd;      // GD_COUNT x S x MAXT
    int *fi;         // GI_COUNT x S x MAXT
    int *hdr, *order, *freestk;
    float *emb;      // S x MAXT x D      track embedding by slot
    double *lastb;   // S x MAXT x 5
    double *cost_g;  // S x MAXD x MAXT   cost-matrix spill
    long long *prof; // optional S x 16 cycle accumulators (diagnostics)
    int *pairs;      // S x MAXD x MAXT   spill of the (detection, track) pairs with IoU > 0 that do not fit the LDS list
    unsigned char *big_ws;   // S x big_stride: list / solver work area of the big-scene tier (see deepocsort_frames_kernel)
    size_t big_stride;
    int S, MAXT, MAXD, D, cost_lds_entries, lds_bytes;      // cost_lds_entries: the guard-word debug build only (one layout at capacity)
};
struct DocP {
    double det_thresh, iou_threshold, inertia, w_emb, alpha_fixed, aw_param, min_confidence;
    int max_age, min_hits, delta_t, asso_func, aw_off, wrapper_mode;
};

__device__ __forceinline__ void bbox_to_z(const double *b, double *z)          // convert_bbox_to_z_new (ocsort.py:49-54)
{
    const double w = b[2] - b[0], h = b[3] - b[1];
    z[0] = b[0] + w / 2.0; z[1] = b[1] + h / 2.0; z[2] = w; z[3] = h;
}
__device__ __forceinline__ void x_to_bbox(const double *x, double *b)          // convert_x_to_bbox_new (:57-59)
{
    b[0] = x[0] - x[2] / 2; b[1] = x[1] - x[3] / 2; b[2] = x[0] + x[2] / 2; b[3] = x[1] + x[3] / 2;
}
__device__ __forceinline__ void noise8(double w, double h, double (&q)[8])     // new_kf_process_noise (:77-82)
{
    const double p = 1. / 20, v = 1. / 160;
    q[0] = (p * w) * (p * w); q[1] = (p * h) * (p * h); q[2] = q[0]; q[3] = q[1];
    q[4] = (v * w) * (v * w); q[5] = (v * h) * (v * h); q[6] = q[4]; q[7] = q[5];
}

__device__ __forceinline__ void kf8n_predict(double (&x)[8], double (&P)[64], const double (&q)[8])     // kalmanfilter.py:340-379
{
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = x[i] + x[i + 4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) P[i * 8 + j] = P[i * 8 + j] + P[(i + 4) * 8 + j];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i * 8 + j] = P[i * 8 + j] + P[i * 8 + j + 4];
#pragma unroll
    for (int i = 0; i < 64; ++i) P[i] = 1.0 * P[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) P[i * 9] += q[i];
}

__device__ __forceinline__ void kf8n_update_core(double (&x)[8], double (&P)[64], const double *z, const double (&R)[4])   // kalmanfilter.py:522-564
{
    double y[4], S[16], SI[16], K[32], IKH[32], t1[64];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = z[i] - x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = P[i * 8 + j] + (i == j ? R[i] : 0.0);
    inv4(S, SI);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fma(P[i * 8 + t], SI[t * 4 + j], s);      // every np.dot of two matrices: fma chain over k (dgemm order, r03)
            K[i * 4 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = x[i] + dot4_h2(K + i * 4, y);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) IKH[i * 4 + j] = (i == j ? 1.0 : 0.0) - K[i * 4 + j];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fma(IKH[i * 4 + t], P[t * 8 + j], s);
            if (i >= 4) s += P[i * 8 + j];
            t1[i * 8 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fma(t1[i * 8 + t], IKH[j * 4 + t], s);
            if (j >= 4) s += t1[i * 8 + j];
            double s3 = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s3 = fma(K[i * 4 + t] * R[t], K[j * 4 + t], s3);
            P[i * 8 + j] = s + s3;
        }
}

__device__ __forceinline__ void load_xp(const Trk &T, double (&x)[8], double (&P)[64])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = T.d(GD_X + k);
#pragma unroll
    for (int k = 0; k < 64; ++k) P[k] = T.d(GD_P + k);
}
__device__ __forceinline__ void store_xp(const Trk &T, const double (&x)[8], const double (&P)[64])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) T.d(GD_X + k) = x[k];
#pragma unroll
    for (int k = 0; k < 64; ++k) T.d(GD_P + k) = P[k];
}

// KalmanFilterNew.update(z, R=R) incl. the unfreeze replay (kalmanfilter.py:433-478, :480-569)
__device__ __noinline__ void kf_update_obs(const Trk &T, const double *z, const double (&R)[4])
{
    double x[8], P[64];
    const bool observed = T.i(GI_OBSERVED) != 0, has_saved = T.i(GI_HAS_SAVED) != 0;
    if (!observed && has_saved) {
        const double ONE4[4] = {1, 1, 1, 1}, ONE8[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        const double x1 = T.d(GD_LZ + 0), y1 = T.d(GD_LZ + 1), s1 = T.d(GD_LZ + 2), r1 = T.d(GD_LZ + 3);
        const double w1 = sqrt(s1 * r1), h1 = sqrt(s1 / r1);
        const double x2 = z[0], y2 = z[1], s2 = z[2], r2 = z[3];
        const double w2 = sqrt(s2 * r2), h2 = sqrt(s2 / r2);
        const int time_gap = T.i(GI_GAP) + 1;
        const double dx = (x2 - x1) / time_gap, dy = (y2 - y1) / time_gap, dw = (w2 - w1) / time_gap, dh = (h2 - h1) / time_gap;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = T.d(GD_SX + k);
#pragma unroll
        for (int k = 0; k < 64; ++k) P[k] = T.d(GD_SP + k);
        T.i(GI_HAS_SAVED) = 0;
        double nb[4] = {0, 0, 0, 0};
        for (int i = 0; i < time_gap; ++i) {
            const double xx = x1 + (i + 1) * dx, yy = y1 + (i + 1) * dy, w = w1 + (i + 1) * dw, h = h1 + (i + 1) * dh;
            nb[0] = xx; nb[1] = yy; nb[2] = w * h; nb[3] = w / h;
            kf8n_update_core(x, P, nb, ONE4);
            if (i != time_gap - 1) kf8n_predict(x, P, ONE8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) T.d(GD_LZ + k) = nb[k];
    } else {
        load_xp(T, x, P);
#pragma unroll
        for (int k = 0; k < 4; ++k) T.d(GD_LZ + k) = z[k];
    }
    T.i(GI_GAP) = 0;
    T.i(GI_OBSERVED) = 1;
    kf8n_update_core(x, P, z, R);
    store_xp(T, x, P);
}

__device__ __forceinline__ void kf_update_none(const Trk &T)   // KalmanBoxTracker.update(None) (ocsort.py:247-249), kalmanfilter.py:507-520
{
    if (T.i(GI_OBSERVED)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) T.d(GD_SX + k) = T.d(GD_X + k);
#pragma unroll
        for (int k = 0; k < 64; ++k) T.d(GD_SP + k) = T.d(GD_P + k);
        T.i(GI_HAS_SAVED) = 1;
    }
    T.i(GI_OBSERVED) = 0;
    T.i(GI_GAP) = T.i(GI_GAP) + 1;
    T.i(GI_FROZEN) = 1;
}

__device__ __forceinline__ bool obs_lookup(const Trk &T, int age, double *box)
{
    if (age < 0) return false;
    const int s = age % RINGN;
    if (T.i(GI_OBAGE + s) != age) return false;
#pragma unroll
    for (int k = 0; k < 5; ++k) box[k] = T.d(GD_OB + s * 5 + k);
    return true;
}

// KalmanBoxTracker.update(bbox, cls, tracklab_id) (ocsort.py:208-252). det = 7-vector
__device__ __noinline__ void kbt_update(const Trk &T, const double *det, int delta_t)
{
    double lo[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) lo[k] = T.d(GD_LO + k);
    const int age = T.i(GI_AGE);
    T.i(GI_FROZEN) = 0;
    T.d(GD_CLS) = det[5];                                   // conf stays the first detection's
    if (sum5(lo) >= 0) {
        double prev[5];
        bool found = false;
        for (int i = 0; i < delta_t && !found; ++i) found = obs_lookup(T, age - (delta_t - i), prev);
        if (!found) {
#pragma unroll
            for (int k = 0; k < 5; ++k) prev[k] = lo[k];
        }
        const double cx1 = (prev[0] + prev[2]) / 2.0, cy1 = (prev[1] + prev[3]) / 2.0;
        const double cx2 = (det[0] + det[2]) / 2.0, cy2 = (det[1] + det[3]) / 2.0;
        const double norm = sqrt((cy2 - cy1) * (cy2 - cy1) + (cx2 - cx1) * (cx2 - cx1)) + 1e-6;
        T.d(GD_VEL + 0) = (cy2 - cy1) / norm;
        T.d(GD_VEL + 1) = (cx2 - cx1) / norm;
        T.i(GI_HAS_VEL) = 1;
    }
    const int s = age % RINGN;
    T.i(GI_OBAGE + s) = age;
#pragma unroll
    for (int k = 0; k < 5; ++k) { T.d(GD_LO + k) = det[k]; T.d(GD_OB + s * 5 + k) = det[k]; }
    T.i(GI_NOBS) = T.i(GI_NOBS) + 1;
    T.i(GI_TSU) = 0;
    T.i(GI_HITS) = T.i(GI_HITS) + 1;
    T.i(GI_STREAK) = T.i(GI_STREAK) + 1;
    const double m = 1. / 20, w = T.d(GD_X + 2), h = T.d(GD_X + 3);     // new_kf_measurement_noise of the state before kf.update
    const double R[4] = {(m * w) * (m * w), (m * h) * (m * h), (m * w) * (m * w), (m * h) * (m * h)};
    double z[4];
    bbox_to_z(det, z);
    kf_update_obs(T, z, R);
    T.d(GD_TID) = det[6];
}

__device__ __noinline__ void kbt_init(const Trk &T, const double *det, int id)   // ocsort.py:100-206 (new_kf)
{
    double z[4], q[8];
    bbox_to_z(det, z);
    noise8(z[2], z[3], q);
#pragma unroll
    for (int k = 0; k < 64; ++k) T.d(GD_P + k) = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { T.d(GD_P + k * 9) = k < 4 ? q[k] * 4 : q[k] * 100; T.d(GD_X + k) = k < 4 ? z[k] : 0.0; }
    T.i(GI_ID) = id; T.i(GI_TSU) = 0; T.i(GI_HITS) = 0; T.i(GI_STREAK) = 0; T.i(GI_AGE) = 0;
    T.i(GI_OBSERVED) = 0; T.i(GI_HAS_SAVED) = 0; T.i(GI_GAP) = 0; T.i(GI_HAS_VEL) = 0; T.i(GI_NOBS) = 0; T.i(GI_FROZEN) = 0;
#pragma unroll
    for (int k = 0; k < RINGN; ++k) T.i(GI_OBAGE + k) = -1;
    T.d(GD_CONF) = det[4]; T.d(GD_CLS) = det[5]; T.d(GD_TID) = det[6];
#pragma unroll
    for (int k = 0; k < 5; ++k) T.d(GD_LO + k) = -1.0;
    T.d(GD_VEL) = 0.0; T.d(GD_VEL + 1) = 0.0;
}

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// row / column weight of compute_aw_max_metric from the two largest entries (float32 arithmetic, association.py:266-286)
__device__ __forceinline__ float aw_weight(float top1, float top2, double bottom)
{
    if (top1 == 0.f) return 0.f;
    const float r = top2 / top1;
    const float m = r - (float)bottom;
    if (!(m > 0.f)) return 1.f;
    return 1.f - m / (float)(1 - bottom);
}

// The two float32 embedding phases live in their own (non-inlined) functions: inside the 256-VGPR tracker kernel the register
// allocator serialised every load behind a scratch spill; on their own they keep sixteen float4 loads in flight per lane.
//
// dets_embs @ trk_embs.T for the listed (detection, track) pairs (association.py:343: only where IoU > 0): eight dot products per
// wavefront at a time, 8 lanes each. All threads of the block call; the result does not depend on the list order.
__device__ __noinline__ void emb_pair_dots(const float *__restrict__ demb, const float *__restrict__ temb, const int *hi_idx, const int *order,
                                           const int *plist, int plist_cap, const int *pairs_g, int npairs, int T, int DIM, double *cost)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 3, j = lane & 7;
    const int nblk = (DIM & 3) == 0 ? DIM / 256 : 0;           // full 8 x float4 x 8-lane blocks
    for (int k0 = wv * 8; k0 < npairs; k0 += NWAVES * 8) {
        const int k = k0 + g;
        const bool act = k < npairs;
        int e = 0;
        float sacc = 0.f;
        if (act) {
            e = k < plist_cap ? plist[k] : pairs_g[k - plist_cap];
            const int d = e / T, t = e - d * T;
            const float *a = demb + (size_t)hi_idx[d] * DIM, *b = temb + (size_t)order[t] * DIM;
            for (int blk = 0; blk < nblk; ++blk) {
                const float4 *a4 = (const float4 *)(a + blk * 256) + j, *b4 = (const float4 *)(b + blk * 256) + j;
                float4 x[8], y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { x[i] = a4[8 * i]; y[i] = b4[8 * i]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) { sacc += x[i].x * y[i].x; sacc += x[i].y * y[i].y; sacc += x[i].z * y[i].z; sacc += x[i].w * y[i].w; }
            }
            for (int q = nblk * 256 + j; q < DIM; q += 8) sacc += a[q] * b[q];
        }
        sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); sacc += __shfl_xor(sacc, 4);
        if (act && j == 0) cost[e] = (double)sacc;
    }
}

// update_emb (ocsort.py:254-256) for `cnt` (track position, input detection index) pairs: float32 EMA with the detection's
// confidence-dependent alpha (:433-436), float32 renormalisation. D <= 512 and D % 4 == 0: four pairs per wavefront, the new
// embedding stays in registers between the EMA and the division; otherwise one wavefront per pair through memory.
__device__ __noinline__ void emb_update_pairs(const double *__restrict__ dets, const float *__restrict__ demb, float *temb, const int *order,
                                              const int *trk_pos, const int *det_in, int cnt, int DIM, double det_thresh, double alpha_fixed)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto alpha_of = [&](int di) {
        const double conf = dets[(size_t)di * 7 + 4];
        const double trust = (conf - det_thresh) / (1 - det_thresh);
        return alpha_fixed + (1 - alpha_fixed) * (1 - trust);
    };
    if (DIM <= 512 && (DIM & 3) == 0) {
        const int g = lane >> 4, j = lane & 15;
        const int nq = DIM >> 2;                               // float4 per row
        for (int k0 = wv * 4; k0 < cnt; k0 += NWAVES * 4) {
            const int k = k0 + g;
            const bool act = k < cnt;
            float4 r[8];
            float ss = 0.f;
            float4 *te4 = (float4 *)temb;
            if (act) {
                const int di = det_in[k];
                const double alpha = alpha_of(di);
                const float a = (float)alpha, b = (float)(1 - alpha);
                te4 = (float4 *)(temb + (size_t)order[trk_pos[k]] * DIM);
                const float4 *de4 = (const float4 *)(demb + (size_t)di * DIM);
                float4 x[8], y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const int q = j + 16 * i; const int qq = q < nq ? q : 0; x[i] = te4[qq]; y[i] = de4[qq]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float4 o;
                    { const float u = a * x[i].x, v = b * y[i].x; o.x = u + v; }
                    { const float u = a * x[i].y, v = b * y[i].y; o.y = u + v; }
                    { const float u = a * x[i].z, v = b * y[i].z; o.z = u + v; }
                    { const float u = a * x[i].w, v = b * y[i].w; o.w = u + v; }
                    r[i] = o;
                    if (j + 16 * i < nq) { ss += o.x * o.x; ss += o.y * o.y; ss += o.z * o.z; ss += o.w * o.w; }
                }
            }
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
            const float n = sqrtf(ss);
            if (act) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (j + 16 * i < nq) te4[j + 16 * i] = make_float4(r[i].x / n, r[i].y / n, r[i].z / n, r[i].w / n);
            }
        }
        return;
    }
    for (int k = wv; k < cnt; k += NWAVES) {
        const int di = det_in[k];
        const double alpha = alpha_of(di);
        const float a = (float)alpha, b = (float)(1 - alpha);
        float *te = temb + (size_t)order[trk_pos[k]] * DIM;
        const float *de = demb + (size_t)di * DIM;
        float ss = 0.f;
        for (int q = lane; q < DIM; q += WAVE) { const float u = a * te[q], v = b * de[q]; const float r = u + v; te[q] = r; ss += r * r; }
        const float n = sqrtf(wave_sum_f(ss));
        for (int q = lane; q < DIM; q += WAVE) te[q] = te[q] / n;
    }
}

__global__ void __launch_bounds__(BLOCK, 1)
deepocsort_frames_kernel(DocDev D, DocP P, const double *__restrict__ dets_all, const float *__restrict__ embs_all, const int *__restrict__ counts,
                         int n_frames, size_t det_stream_stride, size_t det_frame_stride, double *__restrict__ out_all, int out_cap,
                         int *__restrict__ out_counts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int MAXT = D.MAXT, MAXD = D.MAXD, S = D.S, DIM = D.D;
    Lds L;
    int lds_t = MAXT, lds_d = MAXD;          // capacity of the list / solver work area in use (a tier, or the bank's capacity)
    int cost_lds_entries = 0;
#ifdef TLK_LDS_CANARY
    // guard-word debug build: ONE layout at the bank's capacity (created small), guards written once and checked every frame
    carve(smem, MAXT, MAXD, L);
    cost_lds_entries = D.cost_lds_entries - TLK_CANARY_BYTES_TOTAL / 24 / 8 * 2;
    canary_fill(L, (unsigned char *)(L.cost + (D.cost_lds_entries > 16 ? D.cost_lds_entries - 16 : 0)));
#endif
    int *hdr = D.hdr + (size_t)s * H_COUNT;
    int *order = D.order + (size_t)s * MAXT;
    int *freestk = D.freestk + (size_t)s * MAXT;
    double *lastb = D.lastb + (size_t)s * MAXT * 5;
    int *pairs_g = D.pairs + (size_t)s * MAXD * MAXT;
    float *temb = D.emb + (size_t)s * MAXT * DIM;
    const size_t stride_d = (size_t)S * MAXT, stride_i = (size_t)S * MAXT;
    auto trk_at = [&](int slot) {
        Trk T; T.fd = D.fd + (size_t)s * MAXT + slot; T.fi = D.fi + (size_t)s * MAXT + slot;
        T.stride_d = stride_d; T.stride_i = stride_i; return T;
    };

    long long t_prev = 0;
#define PROF(i) do { if (D.prof && tid == 0) { const long long t_ = wall_clock64(); D.prof[(size_t)s * 16 + (i)] += t_ - t_prev; t_prev = t_; } } while (0)
    for (int f = 0; f < n_frames; ++f) {
        if (D.prof && tid == 0) t_prev = wall_clock64();
        const double *dets = dets_all + (size_t)s * det_stream_stride * 7 + (size_t)f * det_frame_stride * 7;
        const float *demb = embs_all + ((size_t)s * det_stream_stride + (size_t)f * det_frame_stride) * DIM;
        double *out = out_all + ((size_t)s * n_frames + f) * (size_t)out_cap * 8;
        int *out_count = out_counts + (size_t)s * n_frames + f;
        const int n_in = counts[(size_t)s * n_frames + f];
        __syncthreads();
#ifdef TLK_LDS_CANARY
        { const int bad = canary_check(L); if (bad && tid == 0) hdr[H_ERR] = -100 - bad; __syncthreads(); }
#endif
        if (hdr[H_ERR] != 0) { if (tid == 0) *out_count = hdr[H_ERR]; continue; }
        if (n_in > MAXD || n_in < 0) { if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } continue; }
        if (P.wrapper_mode && n_in == 0) { if (tid == 0) *out_count = 0; continue; }   // deep_oc_sort_api.py:59-60
#ifndef TLK_LDS_CANARY
        // List / solver work area of THIS frame: the smallest tier that holds (tracks + detections, detections) -- 256 x 128 or 512 x 256 carved
        // out of LDS (the rest of the LDS is the cost matrix), or the bank's full capacity carved out of HBM for a scene beyond that (r04: the
        // reference's list of trackers just grows, deep_oc_sort/ocsort.py:563-574; track state sits in HBM at capacity either way).
        {
            const int need_t = hdr[H_NTRK] + n_in;
            int ct = 0, cd = 0;
            const int tiers[2][2] = {{256, 128}, {512, 256}};
            for (int k = 0; k < 2 && ct == 0; ++k) {
                const int tt = MAXT < tiers[k][0] ? MAXT : tiers[k][0], td = MAXD < tiers[k][1] ? MAXD : tiers[k][1];
                if (need_t <= tt && n_in <= td && lds_fixed_bytes(tt, td) + 4096 <= (size_t)D.lds_bytes) { ct = tt; cd = td; }
            }
            if (ct) { carve(smem, ct, cd, L); cost_lds_entries = (int)(((size_t)D.lds_bytes - lds_fixed_bytes(ct, cd)) / sizeof(double)); lds_t = ct; lds_d = cd; }
            else { carve(D.big_ws + (size_t)s * D.big_stride, MAXT, MAXD, L); cost_lds_entries = 0; lds_t = MAXT; lds_d = MAXD; }
        }
#endif

        // wrapper filter (deep_oc_sort_api.py:62), then scores > det_thresh (ocsort.py:407-408)
        const int N = block_compact(n_in, [&](int i) { const double c = dets[(size_t)i * 7 + 4]; return (!P.wrapper_mode || c > P.min_confidence) && c > P.det_thresh; },
                                    [&](int i, int pos) { L.hi_idx[pos] = i; }, L.scan);
        int T = hdr[H_NTRK];
        __syncthreads();

        PROF(0);
        // ---- predict (ocsort.py:439-456 -> :283-309)
        for (int p = tid; p < T; p += BLOCK) {
            const Trk K = trk_at(order[p]);
            double x[8], Pm[64], q[8];
            load_xp(K, x, Pm);
            if (x[2] + x[6] <= 0) x[6] = 0;
            if (x[3] + x[7] <= 0) x[7] = 0;
            if (K.i(GI_FROZEN)) { x[6] = 0; x[7] = 0; }
            noise8(x[2], x[3], q);
            kf8n_predict(x, Pm, q);
            store_xp(K, x, Pm);
            K.i(GI_AGE) = K.i(GI_AGE) + 1;
            const int tsu = K.i(GI_TSU);
            if (tsu > 0) K.i(GI_STREAK) = 0;
            K.i(GI_TSU) = tsu + 1;
            double b[4];
            x_to_bbox(x, b);
            const bool bad = (b[0] != b[0]) || (b[1] != b[1]) || (b[2] != b[2]) || (b[3] != b[3]);
            L.tmp_a[p] = bad ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) L.kobs[(size_t)p * 5 + k] = b[k];
        }
        __syncthreads();
        PROF(1);
        {   // drop NaN trackers (stable), free their slots
            for (int p = tid; p < T; p += BLOCK) L.tmp_b[p] = order[p];
            __syncthreads();
            const int kept = block_compact(T, [&](int p) { return L.tmp_a[p] == 0; },
                                           [&](int p, int pos) {
                                               order[pos] = L.tmp_b[p];
#pragma unroll
                                               for (int k = 0; k < 4; ++k) L.trk_box[(size_t)pos * 4 + k] = L.kobs[(size_t)p * 5 + k];
                                           }, L.scan);
            if (kept != T) {
                const int nfree = hdr[H_NFREE];
                block_compact(T, [&](int p) { return L.tmp_a[p] != 0; }, [&](int p, int pos) { freestk[nfree + pos] = L.tmp_b[p]; }, L.scan);
                __syncthreads();
                if (tid == 0) { hdr[H_NFREE] = nfree + (T - kept); hdr[H_NTRK] = kept; }
                T = kept;
            }
        }
        __syncthreads();

        PROF(2);
        // ---- velocities / last_boxes / k_observations (ocsort.py:458-460)
        for (int p = tid; p < T; p += BLOCK) {
            const Trk K = trk_at(order[p]);
            const bool hv = K.i(GI_HAS_VEL) != 0;
            L.velp[p * 2] = hv ? K.d(GD_VEL) : 0.0;
            L.velp[p * 2 + 1] = hv ? K.d(GD_VEL + 1) : 0.0;
            double lo[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) { lo[k] = K.d(GD_LO + k); lastb[(size_t)p * 5 + k] = lo[k]; }
            double ko[5];
            if (K.i(GI_NOBS) == 0) {
#pragma unroll
                for (int k = 0; k < 5; ++k) ko[k] = -1.0;
            } else {
                const int age = K.i(GI_AGE);
                bool found = false;
                for (int i = 0; i < P.delta_t && !found; ++i) found = obs_lookup(K, age - (P.delta_t - i), ko);
                if (!found) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) ko[k] = lo[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) L.kobs[(size_t)p * 5 + k] = ko[k];
        }
        for (int k = tid; k < N; k += BLOCK) { L.rowcnt[k] = 0; L.rowhit[k] = -1; }
        for (int k = tid; k < T; k += BLOCK) L.colcnt[k] = 0;
        if (tid == 0) L.sc[SC_NREM] = 0;
        // list of the pairs with IoU > 0: the assignment scratch lists (mi_r .. um_t, contiguous, unused until the LSA) first, HBM beyond
        int *plist = L.mi_r;
        const int plist_cap = 6 * (lds_t > lds_d ? lds_t : lds_d) + lds_d + lds_t;
        __syncthreads();

        PROF(3);
        // ---- first association (association.py:291-364)
        double *cost = ((size_t)N * T <= (size_t)cost_lds_entries) ? L.cost : (D.cost_g + (size_t)s * MAXD * MAXT);
        // until the cost fill, cost[e] holds the float32 embedding cost of the pair (exact in a double): no second N x T array
        int n_mi = 0;
        if (T > 0 && N > 0) {
            for (int e = tid; e < N * T; e += BLOCK) {
                const int d = e / T, t = e - d * T;
                const double iou = box_similarity(TLK_IOU, dets + (size_t)L.hi_idx[d] * 7, L.trk_box + (size_t)t * 4);
                cost[e] = 0.0;                              // emb_cost[iou_matrix <= 0] = 0 (association.py:343)
                if (iou > 0) { const int pos = atomicAdd(&L.sc[SC_NREM], 1); if (pos < plist_cap) plist[pos] = e; else pairs_g[pos - plist_cap] = e; }
                if (iou > P.iou_threshold) { atomicAdd(&L.rowcnt[d], 1); atomicAdd(&L.colcnt[t], 1); L.rowhit[d] = t; }
            }
            __threadfence_block();
            __syncthreads();
            int mx = 0;
            for (int k = tid; k < N; k += BLOCK) mx = max(mx, L.rowcnt[k]);
            int mxc = 0;
            for (int k = tid; k < T; k += BLOCK) mxc = max(mxc, L.colcnt[k]);
            if (tid == 0) { L.sc[SC_FLAG] = 0; L.sc[SC_NL] = 0; }
            __syncthreads();
            atomicMax(&L.sc[SC_FLAG], mx);
            atomicMax(&L.sc[SC_NL], mxc);
            __syncthreads();
            const bool one2one = (L.sc[SC_FLAG] == 1) && (L.sc[SC_NL] == 1);
            __syncthreads();
            if (one2one) {
                n_mi = block_compact(N, [&](int d) { return L.rowcnt[d] == 1; },
                                     [&](int d, int pos) { L.mi_r[pos] = d; L.mi_c[pos] = L.rowhit[d]; }, L.scan);
            } else {
                PROF(4);
                emb_pair_dots(demb, temb, L.hi_idx, order, plist, plist_cap, pairs_g, L.sc[SC_NREM], T, DIM, cost);
                __threadfence_block();
                __syncthreads();
                PROF(5);
                // compute_aw_max_metric: row weights (per detection) and column weights (per track) from the top two entries
                float *roww = (float *)L.tmp_a, *colw = (float *)L.tmp_b;
                if (!P.aw_off) {
                    for (int d = tid; d < N; d += BLOCK) {
                        float w = 1.f;
                        if (T >= 2) {
                            float a = -INFINITY, b = -INFINITY;
                            for (int t = 0; t < T; ++t) { const float v = (float)cost[(size_t)d * T + t]; if (v > a) { b = a; a = v; } else if (v > b) b = v; }
                            w = aw_weight(a, b, P.aw_param);
                        }
                        roww[d] = w;
                    }
                    for (int t = tid; t < T; t += BLOCK) {
                        float w = 1.f;
                        if (N >= 2) {
                            float a = -INFINITY, b = -INFINITY;
                            for (int d = 0; d < N; ++d) { const float v = (float)cost[(size_t)d * T + t]; if (v > a) { b = a; a = v; } else if (v > b) b = v; }
                            w = aw_weight(a, b, P.aw_param);
                        }
                        colw[t] = w;
                    }
                }
                __syncthreads();
                const double PI = 3.141592653589793;
                for (int e = tid; e < N * T; e += BLOCK) {
                    const int d = e / T, t = e - d * T;
                    const double *de = dets + (size_t)L.hi_idx[d] * 7;
                    const double *ko = L.kobs + (size_t)t * 5;
                    const double iou = box_similarity(TLK_IOU, de, L.trk_box + (size_t)t * 4);
                    const double valid = ko[4] < 0 ? 0.0 : 1.0;
                    double adc = 0.0;
                    if (valid != 0.0 && P.inertia != 0.0 && de[5] != 0.0) {
                        const double cx1 = (de[0] + de[2]) / 2.0, cy1 = (de[1] + de[3]) / 2.0;
                        const double cx2 = (ko[0] + ko[2]) / 2.0, cy2 = (ko[1] + ko[3]) / 2.0;
                        double dx = cx1 - cx2, dy = cy1 - cy2;
                        const double norm = sqrt(dx * dx + dy * dy) + 1e-6;
                        dx = dx / norm; dy = dy / norm;
                        double c = L.velp[t * 2 + 1] * dx + L.velp[t * 2] * dy;
                        c = c < -1 ? -1 : (c > 1 ? 1 : c);
                        double ang = acos(c);
                        ang = (PI / 2.0 - fabs(ang)) / PI;
                        adc = ((valid * ang) * P.inertia) * de[5];
                    }
                    float w;
                    if (P.aw_off) w = (float)P.w_emb;
                    else { w = (float)P.w_emb; if (T >= 2) w *= roww[d]; if (N >= 2) w *= colw[t]; }
                    const float ecv = (float)cost[e];
                    const float ecw = P.aw_off ? ecv * w : w * ecv;
                    cost[e] = -((iou + adc) + (double)ecw);
                }
                __threadfence_block();
                __syncthreads();
                PROF(6);
                if (tid < WAVE) {
                    const int r = wave_lsa(cost, N, T, (size_t)T, (size_t)1, L.W, L.mi_r, L.mi_c);
                    if (tid == 0) { L.sc[SC_NMI] = r < 0 ? 0 : r; if (r == LSA_EINTERNAL) hdr[H_ERR] = TLK_EINTERNAL; }
                }
                __syncthreads();
                n_mi = L.sc[SC_NMI];
            }
        }
        __syncthreads();
        PROF(7);
        // unmatched lists + low-IoU rejection (association.py:344-362)
        int nud = 0, nut = 0, nm = 0;
        if (T == 0) {
            for (int k = tid; k < N; k += BLOCK) L.um_d[k] = k;
            nud = N;
        } else {
            for (int k = tid; k < N; k += BLOCK) L.rowcnt[k] = 0;
            for (int k = tid; k < T; k += BLOCK) L.colcnt[k] = 0;
            __syncthreads();
            for (int k = tid; k < n_mi; k += BLOCK) {
                L.rowcnt[L.mi_r[k]] = 1; L.colcnt[L.mi_c[k]] = 1;
                const double iou = box_similarity(TLK_IOU, dets + (size_t)L.hi_idx[L.mi_r[k]] * 7, L.trk_box + (size_t)L.mi_c[k] * 4);
                L.tmp_a[k] = (iou < P.iou_threshold) ? 1 : 0;
            }
            __syncthreads();
            nud = block_compact(N, [&](int d) { return L.rowcnt[d] == 0; }, [&](int d, int pos) { L.um_d[pos] = d; }, L.scan);
            nut = block_compact(T, [&](int t) { return L.colcnt[t] == 0; }, [&](int t, int pos) { L.um_t[pos] = t; }, L.scan);
            const int nrej = block_compact(n_mi, [&](int k) { return L.tmp_a[k] == 1; },
                                           [&](int k, int pos) { L.um_d[nud + pos] = L.mi_r[k]; L.um_t[nut + pos] = L.mi_c[k]; }, L.scan);
            nm = block_compact(n_mi, [&](int k) { return L.tmp_a[k] == 0; }, [&](int k, int pos) { L.m_d[pos] = L.mi_r[k]; L.m_t[pos] = L.mi_c[k]; }, L.scan);
            nud += nrej; nut += nrej;
        }
        __syncthreads();
        PROF(8);
        // update_emb (ocsort.py:254-256) for `cnt` (track position, input detection index) pairs: one wavefront per pair
        auto update_embs = [&](int cnt, const int *trk_pos, const int *det_in) {
            emb_update_pairs(dets, demb, temb, order, trk_pos, det_in, cnt, DIM, P.det_thresh, P.alpha_fixed);
        };
        for (int k = tid; k < nm; k += BLOCK) {                                 // ocsort.py:475-477
            kbt_update(trk_at(order[L.m_t[k]]), dets + (size_t)L.hi_idx[L.m_d[k]] * 7, P.delta_t);
            L.tmp_a[k] = L.hi_idx[L.m_d[k]];
        }
        __syncthreads();
        update_embs(nm, L.m_t, L.tmp_a);
        __syncthreads();

        PROF(9);
        // ---- second round by OCR on the last observations (ocsort.py:480-513)
        if (nud > 0 && nut > 0) {
            const int nrow = nud, ncol = nut;
            double *mat = ((size_t)nrow * ncol <= (size_t)cost_lds_entries) ? L.cost : (D.cost_g + (size_t)s * MAXD * MAXT);
            double lmax = -INFINITY; bool lnan = false;
            for (int e = tid; e < nrow * ncol; e += BLOCK) {
                const int r = e / ncol, c = e - r * ncol;
                const double v = box_similarity(P.asso_func, dets + (size_t)L.hi_idx[L.um_d[r]] * 7, lastb + (size_t)L.um_t[c] * 5);
                mat[e] = v;
                lnan |= (v != v); lmax = v > lmax ? v : lmax;
            }
            double mx = block_max_nan(lnan ? NAN : lmax, true, L.red);
            if (P.asso_func == TLK_CT) {
                lmax = -INFINITY; lnan = false;
                for (int e = tid; e < nrow * ncol; e += BLOCK) { double v = mat[e] / mx; mat[e] = v; lnan |= (v != v); lmax = v > lmax ? v : lmax; }
                const double m2 = block_max_nan(lnan ? NAN : lmax, true, L.red);
                lmax = -INFINITY; lnan = false;
                for (int e = tid; e < nrow * ncol; e += BLOCK) { double v = m2 - mat[e]; mat[e] = v; lnan |= (v != v); lmax = v > lmax ? v : lmax; }
                mx = block_max_nan(lnan ? NAN : lmax, true, L.red);
            }
            if (mx > P.iou_threshold) {                                            // uniform across the block
                for (int e = tid; e < nrow * ncol; e += BLOCK) mat[e] = -mat[e];
                __threadfence_block();
                __syncthreads();
                if (tid < WAVE) {
                    const int r = wave_lsa(mat, nrow, ncol, (size_t)ncol, (size_t)1, L.W, L.mi_r, L.mi_c);
                    if (tid == 0) { L.sc[SC_NL] = r < 0 ? 0 : r; if (r == LSA_EINTERNAL) hdr[H_ERR] = TLK_EINTERNAL; }
                }
                __syncthreads();
                const int nl = L.sc[SC_NL];
                const int nacc = block_compact(nl, [&](int k) { return !((-mat[(size_t)L.mi_r[k] * ncol + L.mi_c[k]]) < P.iou_threshold); },
                                               [&](int k, int pos) {
                                                   L.tmp_a[pos] = L.hi_idx[L.um_d[L.mi_r[k]]];      // input det index
                                                   L.tmp_b[pos] = L.um_t[L.mi_c[k]];                // track position
                                                   L.m_d[pos] = L.um_d[L.mi_r[k]];                  // det index in the filtered list
                                               }, L.scan);
                __syncthreads();
                for (int k = tid; k < nacc; k += BLOCK) kbt_update(trk_at(order[L.tmp_b[k]]), dets + (size_t)L.tmp_a[k] * 7, P.delta_t);
                update_embs(nacc, L.tmp_b, L.tmp_a);
                __syncthreads();
                nut = block_setdiff_sorted(L.um_t, nut, L.tmp_b, nacc, L.mi_c, L.scan);
                nud = block_setdiff_sorted(L.um_d, nud, L.m_d, nacc, L.mi_c, L.scan);
                __syncthreads();
            }
        }
        __syncthreads();

        PROF(10);
        for (int k = tid; k < nut; k += BLOCK) kf_update_none(trk_at(order[L.um_t[k]]));   // ocsort.py:515-516
        // ---- births (ocsort.py:519-524)
        int nfree = hdr[H_NFREE], nextid = hdr[H_NEXTID];
        __syncthreads();
        if (T + nud > MAXT) {
            if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; }
            continue;
        }
        for (int k = tid; k < nud; k += BLOCK) {
            const int slot = freestk[nfree - 1 - k];
            order[T + k] = slot;
            kbt_init(trk_at(slot), dets + (size_t)L.hi_idx[L.um_d[k]] * 7, nextid + k);
        }
        __threadfence_block();
        __syncthreads();
        for (int k = wv; k < nud; k += NWAVES) {                                   // emb = dets_embs[i], as delivered (not normalised)
            float *te = temb + (size_t)order[T + k] * DIM;
            const float *de = demb + (size_t)L.hi_idx[L.um_d[k]] * DIM;
            for (int q = lane; q < DIM; q += WAVE) te[q] = de[q];
        }
        __syncthreads();
        nfree -= nud; nextid += nud; T += nud;
        PROF(11);
        // ---- emit rows in reversed list order + drop dead tracklets (ocsort.py:525-531; frame_count stays 0 in the reference)
        for (int q = tid; q < T; q += BLOCK) {
            const Trk K = trk_at(order[T - 1 - q]);
            const int tsu = K.i(GI_TSU);
            L.tmp_a[q] = (tsu < 1 && (K.i(GI_STREAK) >= P.min_hits || 0 <= P.min_hits)) ? 1 : 0;
            L.tmp_b[T - 1 - q] = (tsu > P.max_age) ? 1 : 0;
            L.mi_r[T - 1 - q] = order[T - 1 - q];
        }
        __syncthreads();
        const int rows = block_compact(T, [&](int q) { return L.tmp_a[q] == 1; },
                                       [&](int q, int pos) {
                                           if (pos >= out_cap) return;
                                           const Trk K = trk_at(L.mi_r[T - 1 - q]);
                                           double lo[5], d4[4];
#pragma unroll
                                           for (int k = 0; k < 5; ++k) lo[k] = K.d(GD_LO + k);
                                           if (sum5(lo) < 0) {
                                               double x[4] = {K.d(GD_X), K.d(GD_X + 1), K.d(GD_X + 2), K.d(GD_X + 3)};
                                               x_to_bbox(x, d4);
                                           } else { d4[0] = lo[0]; d4[1] = lo[1]; d4[2] = lo[2]; d4[3] = lo[3]; }
                                           double *r = out + (size_t)pos * 8;
                                           r[0] = d4[0]; r[1] = d4[1]; r[2] = d4[2]; r[3] = d4[3];
                                           r[4] = (double)(K.i(GI_ID) + 1); r[5] = K.d(GD_CLS); r[6] = K.d(GD_CONF); r[7] = K.d(GD_TID);
                                       }, L.scan);
        const int kept = block_compact(T, [&](int p) { return L.tmp_b[p] == 0; }, [&](int p, int pos) { order[pos] = L.mi_r[p]; }, L.scan);
        if (kept != T) {
            block_compact(T, [&](int p) { return L.tmp_b[p] != 0; }, [&](int p, int pos) { freestk[nfree + pos] = L.mi_r[p]; }, L.scan);
            nfree += T - kept;
        }
        __syncthreads();
        PROF(12);
        if (tid == 0) {
            hdr[H_NTRK] = kept; hdr[H_NFREE] = nfree; hdr[H_NEXTID] = nextid;
            *out_count = hdr[H_ERR] != 0 ? hdr[H_ERR] : (rows > out_cap ? TLK_ECAPACITY : rows);      // (H_ERR: TLK_EINTERNAL from a solver's loop bound)
        }
        __threadfence_block();
        __syncthreads();
    }
#ifdef TLK_LDS_CANARY
    { const int bad = canary_check(L); if (bad && tid == 0) { hdr[H_ERR] = -100 - bad; out_counts[(size_t)s * n_frames + n_frames - 1] = -100 - bad; } }
#endif
}

// KalmanBoxTracker.apply_affine_correction + KalmanFilterNew.apply_affine_correction (ocsort.py:261-281, kalmanfilter.py:387-405, new_kf
// branch) with the camera-motion estimate passed in: one thread per live track, a launch of its own ahead of the frame kernel (the
// reference applies it before predict, ocsort.py:425-428). last_observation and observations[age of the last update] are one numpy
// array in the reference (update() stores the same view in both): the ring slot with the newest age mirrors last_observation, and a
// box still inside the delta_t window is warped twice.
struct DocWarp { double m[6]; };
__device__ __forceinline__ void doc_affine_pts(const double (&A)[6], double (&b)[4])
{
    const double x1 = A[0] * b[0] + A[1] * b[1], y1 = A[3] * b[0] + A[4] * b[1];
    const double x2 = A[0] * b[2] + A[1] * b[3], y2 = A[3] * b[2] + A[4] * b[3];
    b[0] = x1 + A[2]; b[1] = y1 + A[5]; b[2] = x2 + A[2]; b[3] = y2 + A[5];
}
// x = kron(I4, R) x, x[:2] += t, P = kron(I4, R) P kron(I4, R)^T on the slot fields starting at fx / fP
__device__ void doc_affine_state(const Trk &T, int fx, int fP, const double (&A)[6])
{
    const double R[4] = {A[0], A[1], A[3], A[4]};
    double m[8];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const double u = T.d(fx + 2 * b), v = T.d(fx + 2 * b + 1);
        m[2 * b] = R[0] * u + R[1] * v; m[2 * b + 1] = R[2] * u + R[3] * v;
    }
    m[0] += A[2]; m[1] += A[5];
#pragma unroll
    for (int k = 0; k < 8; ++k) T.d(fx + k) = m[k];
    for (int bi = 0; bi < 4; ++bi)             // 2x2 blocks: P'[bi][bj] = R P[bi][bj] R^T, rows first, then columns
        for (int bj = 0; bj < 4; ++bj) {
            const int r0 = 2 * bi, c0 = 2 * bj;
            const double p00 = T.d(fP + r0 * 8 + c0), p01 = T.d(fP + r0 * 8 + c0 + 1), p10 = T.d(fP + (r0 + 1) * 8 + c0), p11 = T.d(fP + (r0 + 1) * 8 + c0 + 1);
            const double t00 = R[0] * p00 + R[1] * p10, t01 = R[0] * p01 + R[1] * p11;
            const double t10 = R[2] * p00 + R[3] * p10, t11 = R[2] * p01 + R[3] * p11;
            T.d(fP + r0 * 8 + c0) = t00 * R[0] + t01 * R[1]; T.d(fP + r0 * 8 + c0 + 1) = t00 * R[2] + t01 * R[3];
            T.d(fP + (r0 + 1) * 8 + c0) = t10 * R[0] + t11 * R[1]; T.d(fP + (r0 + 1) * 8 + c0 + 1) = t10 * R[2] + t11 * R[3];
        }
}
__global__ void deepocsort_affine_kernel(DocDev D, int stream, int delta_t, DocWarp W)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    double A[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[i] = W.m[i];
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        const int T = D.hdr[(size_t)s * H_COUNT + H_NTRK];
        for (int p = threadIdx.x; p < T; p += blockDim.x) {
            const int slot = D.order[(size_t)s * D.MAXT + p];
            Trk K; K.fd = D.fd + (size_t)s * D.MAXT + slot; K.fi = D.fi + (size_t)s * D.MAXT + slot; K.stride_d = stride; K.stride_i = stride;
            const int age = K.i(GI_AGE);
            int alias = -1, alias_age = -1;                                     // ring slot of the last update == last_observation's array
            if (K.i(GI_NOBS) > 0)
                for (int q = 0; q < RINGN; ++q) { const int a = K.i(GI_OBAGE + q); if (a > alias_age) { alias_age = a; alias = q; } }
            double lo[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) lo[k] = K.d(GD_LO + k);
            if (sum5(lo) > 0) {
                double b[4] = {lo[0], lo[1], lo[2], lo[3]};
                doc_affine_pts(A, b);
#pragma unroll
                for (int k = 0; k < 4; ++k) { K.d(GD_LO + k) = b[k]; if (alias >= 0) K.d(GD_OB + alias * 5 + k) = b[k]; }
            }
            for (int dt = delta_t; dt >= 0; --dt) {
                const int a = age - dt;
                if (a < 0) continue;
                const int q = a % RINGN;
                if (K.i(GI_OBAGE + q) != a) continue;
                double b[4] = {K.d(GD_OB + q * 5), K.d(GD_OB + q * 5 + 1), K.d(GD_OB + q * 5 + 2), K.d(GD_OB + q * 5 + 3)};
                doc_affine_pts(A, b);
#pragma unroll
                for (int k = 0; k < 4; ++k) { K.d(GD_OB + q * 5 + k) = b[k]; if (q == alias) K.d(GD_LO + k) = b[k]; }
            }
            doc_affine_state(K, GD_X, GD_P, A);
            if (K.i(GI_OBSERVED) == 0 && K.i(GI_HAS_SAVED) != 0) {
                doc_affine_state(K, GD_SX, GD_SP, A);
                const double l0 = K.d(GD_LZ), l1 = K.d(GD_LZ + 1), l2 = K.d(GD_LZ + 2), l3 = K.d(GD_LZ + 3);
                K.d(GD_LZ) = A[0] * l0 + A[1] * l1 + A[2]; K.d(GD_LZ + 1) = A[3] * l0 + A[4] * l1 + A[5];
                K.d(GD_LZ + 2) = A[0] * l2 + A[1] * l3; K.d(GD_LZ + 3) = A[3] * l2 + A[4] * l3;
            }
        }
    }
}

__global__ void deepocsort_reset_kernel(DocDev D, int stream)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        int *hdr = D.hdr + (size_t)s * H_COUNT;
        for (int k = threadIdx.x; k < D.MAXT; k += blockDim.x) D.freestk[(size_t)s * D.MAXT + k] = D.MAXT - 1 - k;
        if (threadIdx.x == 0) { hdr[H_NTRK] = 0; hdr[H_NEXTID] = 0; hdr[H_NFREE] = D.MAXT; hdr[H_ERR] = 0; }
    }
}

__global__ void deepocsort_gather_kernel(DocDev D, int stream, long long *ids, double *x, double *Pm, float *emb, long long *state6, double *vel,
                                         double *last, int cap, int *n_out)
{
    const int T = D.hdr[(size_t)stream * H_COUNT + H_NTRK];
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = T;
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int p = blockIdx.x; p < T && p < cap; p += gridDim.x) {
        const int slot = D.order[(size_t)stream * D.MAXT + p];
        const double *fd = D.fd + (size_t)stream * D.MAXT + slot;
        const int *fi = D.fi + (size_t)stream * D.MAXT + slot;
        for (int k = threadIdx.x; k < 8; k += blockDim.x) x[(size_t)p * 8 + k] = fd[(size_t)(GD_X + k) * stride];
        for (int k = threadIdx.x; k < 64; k += blockDim.x) Pm[(size_t)p * 64 + k] = fd[(size_t)(GD_P + k) * stride];
        for (int k = threadIdx.x; k < D.D; k += blockDim.x) emb[(size_t)p * D.D + k] = D.emb[((size_t)stream * D.MAXT + slot) * D.D + k];
        for (int k = threadIdx.x; k < 5; k += blockDim.x) last[(size_t)p * 5 + k] = fd[(size_t)(GD_LO + k) * stride];
        if (threadIdx.x == 0) {
            ids[p] = fi[(size_t)GI_ID * stride];
            const bool hv = fi[(size_t)GI_HAS_VEL * stride] != 0;
            vel[(size_t)p * 2] = hv ? fd[(size_t)GD_VEL * stride] : 0.0; vel[(size_t)p * 2 + 1] = hv ? fd[(size_t)(GD_VEL + 1) * stride] : 0.0;
            long long *st = state6 + (size_t)p * 6;
            st[0] = fi[(size_t)GI_TSU * stride]; st[1] = fi[(size_t)GI_HITS * stride]; st[2] = fi[(size_t)GI_STREAK * stride];
            st[3] = fi[(size_t)GI_AGE * stride]; st[4] = fi[(size_t)GI_FROZEN * stride]; st[5] = fi[(size_t)GI_OBSERVED * stride];
        }
    }
}

}  // namespace

struct tlk_deepocsort {
    DocDev D; DocP P; int device; size_t smem;
    double *d_dets, *d_out; float *d_embs; int *d_cnt, *d_ocnt;
    int out_cap;
};

static int doc_free(tlk_deepocsort *h)
{
    if (!h) return TLK_OK;
    hipSetDevice(h->device);
    void *ptrs[] = {h->D.fd, h->D.fi, h->D.hdr, h->D.order, h->D.freestk, h->D.emb, h->D.lastb, h->D.cost_g, h->D.big_ws, h->D.prof, h->D.pairs,
                    h->d_dets, h->d_out, h->d_embs, h->d_cnt, h->d_ocnt};
    for (void *p : ptrs) if (p) hipFree(p);
    delete h;
    return TLK_OK;
}

extern "C" int tlk_deepocsort_create(const tlk_deepocsort_params *p, int n_streams, int device, tlk_deepocsort **out)
{
    if (!p || !out) return fail(TLK_EINVAL, "tlk_deepocsort_create: null pointer");
    if (n_streams < 1) return fail(TLK_EINVAL, "tlk_deepocsort_create: n_streams must be >= 1");
    if (p->asso_func < TLK_IOU || p->asso_func > TLK_CT) return fail(TLK_EINVAL, "tlk_deepocsort_create: unknown asso_func");
    if (p->delta_t < 0 || p->delta_t >= RINGN) return fail(TLK_EINVAL, "tlk_deepocsort_create: delta_t must be in [0, 8)");
    if (p->dim < 1 || p->dim > 4096) return fail(TLK_EINVAL, "tlk_deepocsort_create: dim must be in [1, 4096]");
    if (!p->cmc_off) return fail(TLK_EUNSUPPORTED, "tlk_deepocsort_create: camera-motion compensation (cmc.py, cv2 optical flow) is not implemented; set cmc_off");
    if (p->embedding_off) return fail(TLK_EUNSUPPORTED, "tlk_deepocsort_create: embedding_off is not supported (the reference itself fails on it: ocsort.py:429 get_emb)");
    if (p->new_kf_off) return fail(TLK_EUNSUPPORTED, "tlk_deepocsort_create: new_kf_off (the 7-state filter) is not built; use tlk_ocsort for it");
    const int MAXT = p->max_tracks > 0 ? p->max_tracks : 256, MAXD = p->max_dets > 0 ? p->max_dets : 128;
    // capacity = allocation size (r04): LDS tiers while the scene fits, HBM lists beyond (deepocsort_frames_kernel)
    if (MAXT > 16384 || MAXD > 1024) return fail(TLK_ECAPACITY, "tlk_deepocsort_create: max_tracks <= 16384 and max_dets <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_deepocsort_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_deepocsort_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_deepocsort *h = new tlk_deepocsort();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->P = DocP{p->det_thresh, p->iou_threshold, p->inertia, p->w_association_emb, p->alpha_fixed_emb, p->aw_param, p->min_confidence,
                p->max_age, p->min_hits, p->delta_t, p->asso_func, p->aw_off, p->wrapper_mode};
    DocDev &D = h->D;
    D.S = n_streams; D.MAXT = MAXT; D.MAXD = MAXD; D.D = p->dim;
    const size_t budget = 160 * 1024 - 256;
#ifdef TLK_LDS_CANARY
    const size_t fixed = lds_fixed_bytes(MAXT, MAXD);
    if (fixed + 4096 > budget) { delete h; return fail(TLK_ECAPACITY, "tlk_deepocsort_create: LDS budget exceeded (guard-word build: one layout at capacity)"); }
    D.cost_lds_entries = (int)((budget - fixed) / sizeof(double));
    h->smem = fixed + (size_t)D.cost_lds_entries * sizeof(double);
#else
    D.cost_lds_entries = 0;
    h->smem = budget & ~(size_t)15;
#endif
    D.lds_bytes = (int)h->smem;
    D.big_stride = (lds_fixed_bytes(MAXT, MAXD) + 255) & ~(size_t)255;
    const size_t slots = (size_t)n_streams * MAXT;
    h->out_cap = MAXT + MAXD;
#define DOC_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { doc_free(h); return fail(TLK_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    DOC_ALLOC(D.fd, sizeof(double) * GD_COUNT * slots);
    DOC_ALLOC(D.fi, sizeof(int) * GI_COUNT * slots);
    DOC_ALLOC(D.hdr, sizeof(int) * H_COUNT * n_streams);
    DOC_ALLOC(D.order, sizeof(int) * slots);
    DOC_ALLOC(D.freestk, sizeof(int) * slots);
    DOC_ALLOC(D.emb, sizeof(float) * slots * D.D);
    DOC_ALLOC(D.lastb, sizeof(double) * 5 * slots);
    DOC_ALLOC(D.cost_g, sizeof(double) * (size_t)n_streams * MAXD * MAXT);
    DOC_ALLOC(D.big_ws, D.big_stride * (size_t)n_streams);
    DOC_ALLOC(D.pairs, sizeof(int) * (size_t)n_streams * MAXD * MAXT);
    if (getenv("TLK_DEEPOCSORT_PROF")) { DOC_ALLOC(D.prof, sizeof(long long) * 16 * n_streams); hipMemset(D.prof, 0, sizeof(long long) * 16 * n_streams); }
    DOC_ALLOC(h->d_dets, sizeof(double) * 7 * MAXD);
    DOC_ALLOC(h->d_embs, sizeof(float) * (size_t)MAXD * D.D);
    DOC_ALLOC(h->d_out, sizeof(double) * 8 * h->out_cap);
    DOC_ALLOC(h->d_cnt, sizeof(int));
    DOC_ALLOC(h->d_ocnt, sizeof(int));
#undef DOC_ALLOC
    hipError_t e = hipMemset(D.fd, 0, sizeof(double) * GD_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.fi, 0, sizeof(int) * GI_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.emb, 0, sizeof(float) * slots * D.D);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)deepocsort_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) { doc_free(h); return fail(TLK_EHIP, std::string("tlk_deepocsort_create: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(deepocsort_reset_kernel, dim3(n_streams < 256 ? n_streams : 256), dim3(BLOCK), 0, 0, D, -1);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { doc_free(h); return fail(TLK_EHIP, std::string("tlk_deepocsort_create: ") + hipGetErrorString(e)); }
    *out = h;
    return TLK_OK;
}

extern "C" int tlk_deepocsort_destroy(tlk_deepocsort *h) { return doc_free(h); }

extern "C" int tlk_deepocsort_reset(tlk_deepocsort *h, int stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_deepocsort_reset: null handle");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_deepocsort_reset: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(deepocsort_reset_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, 0, h->D, stream);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipStreamSynchronize(0));
    return TLK_OK;
}

extern "C" int tlk_deepocsort_affine_correction(tlk_deepocsort *h, int stream, const double *warp6, void *hip_stream)
{
    if (!h || !warp6) return fail(TLK_EINVAL, "tlk_deepocsort_affine_correction: null pointer");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_deepocsort_affine_correction: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    DocWarp W;
    for (int i = 0; i < 6; ++i) W.m[i] = warp6[i];
    hipLaunchKernelGGL(deepocsort_affine_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, (hipStream_t)hip_stream, h->D, stream,
                       h->P.delta_t, W);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_deepocsort_update_dev(tlk_deepocsort *h, const double *dets_dev, const float *embs_dev, const int32_t *counts_dev, int n_frames,
                                         double *out_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_deepocsort_update_dev: null handle");
    if (n_frames < 0 || out_cap < 0) return fail(TLK_EINVAL, "tlk_deepocsort_update_dev: negative size");
    if (n_frames == 0) return TLK_OK;
    if (!dets_dev || !embs_dev || !counts_dev || !out_dev || !out_counts_dev) return fail(TLK_EINVAL, "tlk_deepocsort_update_dev: null pointer");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(deepocsort_frames_kernel, dim3(h->D.S), dim3(BLOCK), h->smem, (hipStream_t)hip_stream, h->D, h->P, dets_dev, embs_dev,
                       (const int *)counts_dev, n_frames, (size_t)h->D.MAXD * n_frames, (size_t)h->D.MAXD, out_dev, out_cap, (int *)out_counts_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_deepocsort_update(tlk_deepocsort *h, int stream, const double *dets, const float *embs, int n, double *out, int out_cap, int *n_out)
{
    if (!h || !n_out) return fail(TLK_EINVAL, "tlk_deepocsort_update: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_deepocsort_update: stream out of range");
    if (n < 0 || (n > 0 && (!dets || !embs))) return fail(TLK_EINVAL, "tlk_deepocsort_update: bad detections");
    if (n > h->D.MAXD) return fail(TLK_ECAPACITY, "tlk_deepocsort_update: more detections than max_dets");
    TLK_HIP(hipSetDevice(h->device));
    hipStream_t st = 0;
    if (n) {
        TLK_HIP(hipMemcpyAsync(h->d_dets, dets, sizeof(double) * 7 * (size_t)n, hipMemcpyHostToDevice, st));
        TLK_HIP(hipMemcpyAsync(h->d_embs, embs, sizeof(float) * (size_t)n * h->D.D, hipMemcpyHostToDevice, st));
    }
    TLK_HIP(hipMemcpyAsync(h->d_cnt, &n, sizeof(int), hipMemcpyHostToDevice, st));
    DocDev V = h->D;
    const size_t sl = (size_t)stream * V.MAXT;
    V.fd += sl; V.fi += sl; V.hdr += (size_t)stream * H_COUNT; V.order += sl; V.freestk += sl; V.emb += sl * V.D; V.lastb += sl * 5;
    V.cost_g += (size_t)stream * V.MAXD * V.MAXT; V.big_ws += (size_t)stream * V.big_stride; V.pairs += (size_t)stream * V.MAXD * V.MAXT; if (V.prof) V.prof += (size_t)stream * 16;
    hipLaunchKernelGGL(deepocsort_frames_kernel, dim3(1), dim3(BLOCK), h->smem, st, V, h->P, (const double *)h->d_dets, (const float *)h->d_embs,
                       (const int *)h->d_cnt, 1, (size_t)0, (size_t)0, h->d_out, h->out_cap, h->d_ocnt);
    TLK_HIP(hipGetLastError());
    int rows = 0;
    TLK_HIP(hipMemcpyAsync(&rows, h->d_ocnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TLK_HIP(hipStreamSynchronize(st));
    if (rows < 0) return fail_stream(rows, "tlk_deepocsort_update");
    if (rows > out_cap) return fail(TLK_ECAPACITY, "tlk_deepocsort_update: output buffer too small");
    if (rows) TLK_HIP(hipMemcpy(out, h->d_out, sizeof(double) * 8 * (size_t)rows, hipMemcpyDeviceToHost));
    *n_out = rows;
    return TLK_OK;
}

extern "C" int tlk_deepocsort_get_tracks(tlk_deepocsort *h, int stream, int64_t *ids, double *x, double *P, float *emb, int64_t *state6, double *vel,
                                         double *last, int cap, int *n_tracks)
{
    if (!h || !n_tracks || !ids || !x || !P || !emb || !state6 || !vel || !last) return fail(TLK_EINVAL, "tlk_deepocsort_get_tracks: null pointer");
    if (stream < 0 || stream >= h->D.S || cap < 0) return fail(TLK_EINVAL, "tlk_deepocsort_get_tracks: bad argument");
    TLK_HIP(hipSetDevice(h->device));
    const size_t c = cap > 0 ? cap : 1;
    long long *d_ids = nullptr, *d_st = nullptr; double *d_x = nullptr, *d_P = nullptr, *d_vel = nullptr, *d_last = nullptr; float *d_emb = nullptr; int *d_n = nullptr;
    TLK_HIP(hipMalloc((void **)&d_ids, sizeof(long long) * c)); TLK_HIP(hipMalloc((void **)&d_st, sizeof(long long) * 6 * c));
    TLK_HIP(hipMalloc((void **)&d_x, sizeof(double) * 8 * c)); TLK_HIP(hipMalloc((void **)&d_P, sizeof(double) * 64 * c));
    TLK_HIP(hipMalloc((void **)&d_vel, sizeof(double) * 2 * c)); TLK_HIP(hipMalloc((void **)&d_last, sizeof(double) * 5 * c));
    TLK_HIP(hipMalloc((void **)&d_emb, sizeof(float) * c * h->D.D)); TLK_HIP(hipMalloc((void **)&d_n, sizeof(int)));
    hipLaunchKernelGGL(deepocsort_gather_kernel, dim3(64), dim3(64), 0, 0, h->D, stream, d_ids, d_x, d_P, d_emb, d_st, d_vel, d_last, cap, d_n);
    int n = 0;
    hipError_t e = hipMemcpy(&n, d_n, sizeof(int), hipMemcpyDeviceToHost);
    const int m = n < cap ? n : cap;
    if (e == hipSuccess && m > 0) {
        e = hipMemcpy(ids, d_ids, sizeof(long long) * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(x, d_x, sizeof(double) * 8 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(P, d_P, sizeof(double) * 64 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(emb, d_emb, sizeof(float) * (size_t)m * h->D.D, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(state6, d_st, sizeof(long long) * 6 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(vel, d_vel, sizeof(double) * 2 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(last, d_last, sizeof(double) * 5 * m, hipMemcpyDeviceToHost);
    }
    hipFree(d_ids); hipFree(d_st); hipFree(d_x); hipFree(d_P); hipFree(d_vel); hipFree(d_last); hipFree(d_emb); hipFree(d_n);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_deepocsort_get_tracks: ") + hipGetErrorString(e));
    *n_tracks = n;
    return TLK_OK;
}

extern "C" int tlk_deepocsort_get_profile(tlk_deepocsort *h, int stream, long long *cycles16)
{
    if (!h || !cycles16) return fail(TLK_EINVAL, "tlk_deepocsort_get_profile: null pointer");
    if (!h->D.prof) return fail(TLK_EINVAL, "tlk_deepocsort_get_profile: create the bank with TLK_DEEPOCSORT_PROF=1 in the environment");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_deepocsort_get_profile: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    TLK_HIP(hipMemcpy(cycles16, h->D.prof + (size_t)stream * 16, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    return TLK_OK;
}
