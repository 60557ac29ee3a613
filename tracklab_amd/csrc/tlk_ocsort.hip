// tlk_ocsort.hip -- OC-SORT as ONE fused HIP kernel per (stream, frame batch).
//
// One 256-thread workgroup (4 wavefronts = one per SIMD of a CU) owns one video stream and walks
// its frames in order; S streams run as S workgroups of the same launch. All tracker state
// (Kalman x/P, frozen copies, observation ring, counters) lives in HBM as field-major SoA so that
// thread-per-track phases read/write coalesced; the per-frame working set (predicted boxes, velocity
// /k-obs, cost matrix, LSA duals) lives in LDS. fp64 throughout, compiled with -ffp-contract=off and
// with the reference's operation order so that track ids / assignments are bit-identical to
// plugins/track/oc_sort (ocsort.py:203-334, association.py:242-298, kalmanfilter.py:339-526).
#include "tlk_common.hpp"
#include "tlk_ocsort_common.hpp"

using namespace tlk;

namespace {

constexpr int RINGN = 8;       // observation ring slots; requires delta_t < RINGN

// ---- per-slot double fields (field-major: fd[(field*S + s)*MAXT + slot]) ----
enum : int {
    FD_X = 0,               // 7   state
    FD_P = FD_X + 7,        // 49  covariance
    FD_SX = FD_P + 49,      // 7   frozen state      (KalmanFilterNew.freeze, kalmanfilter.py:383-387)
    FD_SP = FD_SX + 7,      // 49  frozen covariance
    FD_LZ = FD_SP + 49,     // 4   last non-None history_obs entry
    FD_CONF = FD_LZ + 4,    // 1
    FD_CLS = FD_CONF + 1,   // 1
    FD_TID = FD_CLS + 1,    // 1   tracklab detection id
    FD_LO = FD_TID + 1,     // 5   last_observation
    FD_VEL = FD_LO + 5,     // 2
    FD_OB = FD_VEL + 2,     // RINGN*5 observation ring (observations[age])
    FD_COUNT = FD_OB + RINGN * 5
};
enum : int {
    FI_ID = 0, FI_TSU, FI_HITS, FI_STREAK, FI_AGE, FI_OBSERVED, FI_HAS_SAVED, FI_GAP, FI_HAS_VEL, FI_NOBS,
    FI_OBAGE,               // RINGN
    FI_COUNT = FI_OBAGE + RINGN
};
enum : int { H_NTRK = 0, H_FRAME, H_NEXTID, H_NFREE, H_ERR, H_COUNT = 8 };

struct OcsDev {
    double *fd;      // FD_COUNT x S x MAXT
    int *fi;         // FI_COUNT x S x MAXT
    int *hdr;        // S x H_COUNT
    int *order;      // S x MAXT   list position -> slot
    int *freestk;    // S x MAXT   free slots (stack)
    double *lastb;   // S x MAXT x 5   last_boxes snapshot by position (ocsort.py:248)
    double *cost_g;  // S x MAXD x MAXT   cost-matrix spill when it does not fit LDS
    long long *prof; // optional S x 16 cycle accumulators (diagnostics)
    unsigned char *big_ws;   // S x big_stride: list / solver work area of the big-scene tier (see ocsort_frames_kernel)
    size_t big_stride;
    int S, MAXT, MAXD, lds_bytes;
};

struct OcsP {
    double det_thresh, iou_threshold, inertia, min_confidence;
    int max_age, min_hits, delta_t, asso_func, use_byte, wrapper_mode;
};

// ------------------------------------------------------------------ small math (ocsort.py:21-54)
__device__ __forceinline__ void bbox_to_z(const double *b, double *z)
{
    double w = b[2] - b[0], h = b[3] - b[1];
    z[0] = b[0] + w / 2.; z[1] = b[1] + h / 2.; z[2] = w * h; z[3] = w / (h + 1e-6);
}
__device__ __forceinline__ void x_to_bbox(const double *x, double *b)
{
    double w = sqrt(x[2] * x[3]), h = x[2] / w;
    b[0] = x[0] - w / 2.; b[1] = x[1] - h / 2.; b[2] = x[0] + w / 2.; b[3] = x[1] + h / 2.;
}
// ------------------------------------------------------------------ KalmanFilterNew in registers
__device__ __forceinline__ void kf7_predict(double (&x)[7], double (&P)[49])   // kalmanfilter.py:368-379
{
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = x[i] + x[i + 4];
    // t1 = F P : rows 0..2 add rows 4..6
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) P[i * 7 + j] = P[i * 7 + j] + P[(i + 4) * 7 + j];
    // t2 = t1 F^T : cols 0..2 add cols 4..6
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) P[i * 7 + j] = P[i * 7 + j] + P[i * 7 + j + 4];
    // + Q  (ocsort.py:83-84: diag(1,1,1,1,.01,.01,.01*.01))
    double q4 = 1.0; q4 *= 0.01;
    double q6 = 1.0; q6 *= 0.01; q6 *= 0.01;
#pragma unroll
    for (int i = 0; i < 49; ++i) P[i] = 1.0 * P[i];
    P[0] += 1.0; P[8] += 1.0; P[16] += 1.0; P[24] += 1.0; P[32] += q4; P[40] += q4; P[48] += q6;
}

__device__ __forceinline__ void kf7_update_core(double (&x)[7], double (&P)[49], const double *z)  // kalmanfilter.py:480-526
{
    const double R[4] = {1., 1., 10., 10.};
    double y[4], S[16], SI[16], K[28], IKH[28], t1[49];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = z[i] - x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = P[i * 7 + j] + (i == j ? R[i] : 0.0);
    inv4(S, SI);
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fma(P[i * 7 + t], SI[t * 4 + j], s);      // every np.dot of two matrices: fma chain over k (dgemm order, r03)
            K[i * 4 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        x[i] = x[i] + dot4_h2(K + i * 4, y);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) IKH[i * 4 + j] = (i == j ? 1.0 : 0.0) - K[i * 4 + j];
    // t1 = (I-KH) P   (columns >= 4 of I-KH are identity columns)
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fma(IKH[i * 4 + t], P[t * 7 + j], s);
            if (i >= 4) s += P[i * 7 + j];
            t1[i * 7 + j] = s;
        }
    // P = t1 (I-KH)^T + (K R) K^T
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fma(t1[i * 7 + t], IKH[j * 4 + t], s);
            if (j >= 4) s += t1[i * 7 + j];
            double s3 = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s3 = fma(K[i * 4 + t] * R[t], K[j * 4 + t], s3);
            P[i * 7 + j] = s + s3;
        }
}

__device__ __forceinline__ void load_xp(const Trk &T, double (&x)[7], double (&P)[49])
{
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = T.d(FD_X + k);
#pragma unroll
    for (int k = 0; k < 49; ++k) P[k] = T.d(FD_P + k);
}
__device__ __forceinline__ void store_xp(const Trk &T, const double (&x)[7], const double (&P)[49])
{
#pragma unroll
    for (int k = 0; k < 7; ++k) T.d(FD_X + k) = x[k];
#pragma unroll
    for (int k = 0; k < 49; ++k) T.d(FD_P + k) = P[k];
}

// KalmanFilterNew.update(z) incl. unfreeze replay (kalmanfilter.py:390-434,437-526)
__device__ __noinline__ void kf_update_obs(const Trk &T, const double *z)
{
    double x[7], P[49];
    const bool observed = T.i(FI_OBSERVED) != 0, has_saved = T.i(FI_HAS_SAVED) != 0;
    if (!observed && has_saved) {
        double x1 = T.d(FD_LZ + 0), y1 = T.d(FD_LZ + 1), s1 = T.d(FD_LZ + 2), r1 = T.d(FD_LZ + 3);
        double w1 = sqrt(s1 * r1), h1 = sqrt(s1 / r1);
        double x2 = z[0], y2 = z[1], s2 = z[2], r2 = z[3];
        double w2 = sqrt(s2 * r2), h2 = sqrt(s2 / r2);
        const int time_gap = T.i(FI_GAP) + 1;
        double dx = (x2 - x1) / time_gap, dy = (y2 - y1) / time_gap;
        double dw = (w2 - w1) / time_gap, dh = (h2 - h1) / time_gap;
#pragma unroll
        for (int k = 0; k < 7; ++k) x[k] = T.d(FD_SX + k);
#pragma unroll
        for (int k = 0; k < 49; ++k) P[k] = T.d(FD_SP + k);
        T.i(FI_HAS_SAVED) = 0;
        double nb[4] = {0, 0, 0, 0};
        for (int i = 0; i < time_gap; ++i) {
            double xx = x1 + (i + 1) * dx, yy = y1 + (i + 1) * dy;
            double w = w1 + (i + 1) * dw, h = h1 + (i + 1) * dh;
            nb[0] = xx; nb[1] = yy; nb[2] = w * h; nb[3] = w / h;
            kf7_update_core(x, P, nb);
            if (i != time_gap - 1) kf7_predict(x, P);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) T.d(FD_LZ + k) = nb[k];
    } else {
        load_xp(T, x, P);
#pragma unroll
        for (int k = 0; k < 4; ++k) T.d(FD_LZ + k) = z[k];
    }
    T.i(FI_GAP) = 0;
    T.i(FI_OBSERVED) = 1;
    kf7_update_core(x, P, z);
    store_xp(T, x, P);
}

__device__ __forceinline__ void kf_update_none(const Trk &T)   // kalmanfilter.py:465-477
{
    if (T.i(FI_OBSERVED)) {
#pragma unroll
        for (int k = 0; k < 7; ++k) T.d(FD_SX + k) = T.d(FD_X + k);
#pragma unroll
        for (int k = 0; k < 49; ++k) T.d(FD_SP + k) = T.d(FD_P + k);
        T.i(FI_HAS_SAVED) = 1;
    }
    T.i(FI_OBSERVED) = 0;
    T.i(FI_GAP) = T.i(FI_GAP) + 1;
}

__device__ __forceinline__ bool obs_lookup(const Trk &T, int age, double *box)
{
    if (age < 0) return false;
    const int s = age % RINGN;
    if (T.i(FI_OBAGE + s) != age) return false;
#pragma unroll
    for (int k = 0; k < 5; ++k) box[k] = T.d(FD_OB + s * 5 + k);
    return true;
}

// KalmanBoxTracker.update(bbox, cls, tracklab_id), ocsort.py:109-148. det = 7-vector
__device__ __noinline__ void kbt_update(const Trk &T, const double *det, int delta_t)
{
    double lo[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) lo[k] = T.d(FD_LO + k);
    const int age = T.i(FI_AGE);
    T.d(FD_CONF) = det[4];
    T.d(FD_CLS) = det[5];
    if (sum5(lo) >= 0) {
        double prev[5];
        bool found = false;
        for (int i = 0; i < delta_t && !found; ++i) found = obs_lookup(T, age - (delta_t - i), prev);
        if (!found) {
#pragma unroll
            for (int k = 0; k < 5; ++k) prev[k] = lo[k];
        }
        double cx1 = (prev[0] + prev[2]) / 2.0, cy1 = (prev[1] + prev[3]) / 2.0;     // speed_direction, ocsort.py:49-54
        double cx2 = (det[0] + det[2]) / 2.0, cy2 = (det[1] + det[3]) / 2.0;
        double norm = sqrt((cy2 - cy1) * (cy2 - cy1) + (cx2 - cx1) * (cx2 - cx1)) + 1e-6;
        T.d(FD_VEL + 0) = (cy2 - cy1) / norm;
        T.d(FD_VEL + 1) = (cx2 - cx1) / norm;
        T.i(FI_HAS_VEL) = 1;
    }
    const int s = age % RINGN;
    T.i(FI_OBAGE + s) = age;
#pragma unroll
    for (int k = 0; k < 5; ++k) { T.d(FD_LO + k) = det[k]; T.d(FD_OB + s * 5 + k) = det[k]; }
    T.i(FI_NOBS) = T.i(FI_NOBS) + 1;
    T.i(FI_TSU) = 0;
    T.i(FI_HITS) = T.i(FI_HITS) + 1;
    T.i(FI_STREAK) = T.i(FI_STREAK) + 1;
    double z[4];
    bbox_to_z(det, z);
    kf_update_obs(T, z);
    T.d(FD_TID) = det[6];
}

__device__ __noinline__ void kbt_init(const Trk &T, const double *det, int id)   // ocsort.py:63-107
{
#pragma unroll
    for (int k = 0; k < 49; ++k) T.d(FD_P + k) = 0.0;
    double p4 = 1.0; p4 *= 1000.; p4 *= 10.;
    double p0 = 1.0; p0 *= 10.;
#pragma unroll
    for (int k = 0; k < 7; ++k) { T.d(FD_P + k * 8) = (k >= 4) ? p4 : p0; T.d(FD_X + k) = 0.0; }
    double z[4];
    bbox_to_z(det, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) T.d(FD_X + k) = z[k];
    T.i(FI_ID) = id; T.i(FI_TSU) = 0; T.i(FI_HITS) = 0; T.i(FI_STREAK) = 0; T.i(FI_AGE) = 0;
    T.i(FI_OBSERVED) = 0; T.i(FI_HAS_SAVED) = 0; T.i(FI_GAP) = 0; T.i(FI_HAS_VEL) = 0; T.i(FI_NOBS) = 0;
#pragma unroll
    for (int k = 0; k < RINGN; ++k) T.i(FI_OBAGE + k) = -1;
    T.d(FD_CONF) = det[4]; T.d(FD_CLS) = det[5]; T.d(FD_TID) = det[6];
#pragma unroll
    for (int k = 0; k < 5; ++k) T.d(FD_LO + k) = -1.0;
    T.d(FD_VEL) = 0.0; T.d(FD_VEL + 1) = 0.0;
}

// ------------------------------------------------------------------ the fused per-frame kernel
#ifdef TLK_LDS_CANARY
__device__ int g_canary_poke = 0;      // positive control of the guard-word build: 1 = write one word past L.rowcnt in the next frames
#endif
__global__ void __launch_bounds__(BLOCK, 1)
ocsort_frames_kernel(OcsDev D, OcsP P, const double *__restrict__ dets_all, const int *__restrict__ counts, int n_frames,
                     size_t det_stream_stride, size_t det_frame_stride, double *__restrict__ out_all, int out_cap,
                     int *__restrict__ out_counts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int MAXT = D.MAXT, MAXD = D.MAXD, S = D.S;
    Lds L;
    int cost_lds_entries = 0;
#ifdef TLK_LDS_CANARY
    bool can_live = false;
#endif
    int *hdr = D.hdr + (size_t)s * H_COUNT;
    int *order = D.order + (size_t)s * MAXT;
    int *freestk = D.freestk + (size_t)s * MAXT;
    double *lastb = D.lastb + (size_t)s * MAXT * 5;
    const size_t stride_d = (size_t)S * MAXT, stride_i = (size_t)S * MAXT;
    auto trk_at = [&](int slot) {
        Trk T; T.fd = D.fd + (size_t)s * MAXT + slot; T.fi = D.fi + (size_t)s * MAXT + slot;
        T.stride_d = stride_d; T.stride_i = stride_i; return T;
    };

    long long t_prev = 0;
#define PROF(i) do { if (D.prof && tid == 0) { const long long t_ = wall_clock64(); D.prof[(size_t)s * 16 + (i)] += t_ - t_prev; t_prev = t_; } } while (0)
    for (int f = 0; f < n_frames; ++f) {
        if (D.prof && tid == 0) t_prev = wall_clock64();
        const double *dets = dets_all + (size_t)s * det_stream_stride + (size_t)f * det_frame_stride;
        double *out = out_all + ((size_t)s * n_frames + f) * (size_t)out_cap * 8;
        int *out_count = out_counts + (size_t)s * n_frames + f;
        const int n_in = counts[(size_t)s * n_frames + f];
        __syncthreads();
#ifdef TLK_LDS_CANARY
        if (can_live) { const int bad = canary_check(L); can_live = false; if (bad && tid == 0) hdr[H_ERR] = -100 - bad; __syncthreads(); }
#endif
        if (hdr[H_ERR] != 0) { if (tid == 0) *out_count = hdr[H_ERR]; continue; }
        if (n_in > MAXD || n_in < 0) { if (tid == 0) { hdr[H_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } continue; }
        if (P.wrapper_mode && n_in == 0) { if (tid == 0) *out_count = 0; continue; }   // oc_sort_api.py:51-52
        // List / solver work area of THIS frame: the smallest tier that holds (tracks + detections, detections) -- 256 x 128 or 512 x 256 carved
        // out of LDS (the rest of the LDS is the cost matrix), or the bank's full capacity carved out of HBM for a scene beyond that (r04: the
        // reference's list of trackers just grows, oc_sort/ocsort.py:312-314; track state sits in HBM at capacity either way).
        {
            const int need_t = hdr[H_NTRK] + n_in;
            int ct = 0, cd = 0;
            const int tiers[2][2] = {{256, 128}, {512, 256}};
            for (int k = 0; k < 2 && ct == 0; ++k) {
                const int tt = MAXT < tiers[k][0] ? MAXT : tiers[k][0], td = MAXD < tiers[k][1] ? MAXD : tiers[k][1];
                if (need_t <= tt && n_in <= td && lds_fixed_bytes(tt, td) + 4096 <= (size_t)D.lds_bytes) { ct = tt; cd = td; }
            }
            if (ct) { carve(smem, ct, cd, L); cost_lds_entries = (int)(((size_t)D.lds_bytes - lds_fixed_bytes(ct, cd)) / sizeof(double)); }
            else { carve(D.big_ws + (size_t)s * D.big_stride, MAXT, MAXD, L); cost_lds_entries = 0; }
#ifdef TLK_LDS_CANARY
            if (cost_lds_entries > 16) cost_lds_entries -= 16;          // room for the guard after the cost area
            canary_fill(L, (unsigned char *)(L.cost + cost_lds_entries));
            can_live = true;
            if (g_canary_poke && tid == 0) L.rowcnt[ct ? cd : MAXD] = 1;      // one int past the array's MAXD entries: lands in its guard
#endif
        }

        // ---- split detections (ocsort.py:226-231), after the wrapper's conf filter (oc_sort_api.py:54)
        auto passes = [&](int i) { return !P.wrapper_mode || dets[(size_t)i * 7 + 4] > P.min_confidence; };
        const int N = block_compact(n_in, [&](int i) { return passes(i) && dets[(size_t)i * 7 + 4] > P.det_thresh; },
                                    [&](int i, int pos) { L.hi_idx[pos] = i; }, L.scan);
        int N2 = 0;
        if (P.use_byte)
            N2 = block_compact(n_in, [&](int i) { const double c = dets[(size_t)i * 7 + 4];
                                                  return passes(i) && c > 0.1 && c < P.det_thresh; },
                               [&](int i, int pos) { L.lo_idx[pos] = i; }, L.scan);
        int T = hdr[H_NTRK];
        __syncthreads();
        if (tid == 0) hdr[H_FRAME] = hdr[H_FRAME] + 1;

        PROF(0);
        // ---- predict (ocsort.py:234-244 -> :150-163)
        for (int p = tid; p < T; p += BLOCK) {
            const Trk K = trk_at(order[p]);
            double x[7], Pm[49];
            load_xp(K, x, Pm);
            if ((x[6] + x[2]) <= 0) x[6] *= 0.0;
            kf7_predict(x, Pm);
            store_xp(K, x, Pm);
            K.i(FI_AGE) = K.i(FI_AGE) + 1;
            const int tsu = K.i(FI_TSU);
            if (tsu > 0) K.i(FI_STREAK) = 0;
            K.i(FI_TSU) = tsu + 1;
            double b[4];
            x_to_bbox(x, b);
            const bool bad = (b[0] != b[0]) || (b[1] != b[1]) || (b[2] != b[2]) || (b[3] != b[3]);
            L.tmp_a[p] = bad ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) L.kobs[(size_t)p * 5 + k] = b[k];       // staging (compacted below)
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
                block_compact(T, [&](int p) { return L.tmp_a[p] != 0; },
                              [&](int p, int pos) { freestk[nfree + pos] = L.tmp_b[p]; }, L.scan);
                __syncthreads();
                if (tid == 0) { hdr[H_NFREE] = nfree + (T - kept); hdr[H_NTRK] = kept; }
                T = kept;
            }
        }
        __syncthreads();

        PROF(2);
        // ---- velocities / last_boxes / k_observations (ocsort.py:246-250, :10-18)
        for (int p = tid; p < T; p += BLOCK) {
            const Trk K = trk_at(order[p]);
            const bool hv = K.i(FI_HAS_VEL) != 0;
            L.velp[p * 2] = hv ? K.d(FD_VEL) : 0.0;
            L.velp[p * 2 + 1] = hv ? K.d(FD_VEL + 1) : 0.0;
            double lo[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) { lo[k] = K.d(FD_LO + k); lastb[(size_t)p * 5 + k] = lo[k]; }
            double ko[5];
            if (K.i(FI_NOBS) == 0) {
#pragma unroll
                for (int k = 0; k < 5; ++k) ko[k] = -1.0;
            } else {
                const int age = K.i(FI_AGE);
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
        __syncthreads();

        PROF(3);
        // ---- first association (association.py:242-298)
        double *cost = ((size_t)N * T <= (size_t)cost_lds_entries) ? L.cost : (D.cost_g + (size_t)s * MAXD * MAXT);
        int n_mi = 0;
        if (T > 0 && N > 0) {
            const double PI = 3.141592653589793;
#pragma unroll 4
            for (int e = tid; e < N * T; e += BLOCK) {
                const int d = e / T, t = e - d * T;
                const double *de = dets + (size_t)L.hi_idx[d] * 7;
                const double *ko = L.kobs + (size_t)t * 5;
                const double iou = box_similarity(TLK_IOU, de, L.trk_box + (size_t)t * 4);
                const double valid = ko[4] < 0 ? 0.0 : 1.0;
                double adc = 0.0;       // ((valid*ang)*w)*cls is an exact (signed) zero when any factor is zero
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
                    adc = ((valid * ang) * P.inertia) * de[5];              // "scores" = class column (dets[:, :-1][:, -1])
                }
                cost[e] = -(iou + adc);
                if (iou > P.iou_threshold) { atomicAdd(&L.rowcnt[d], 1); atomicAdd(&L.colcnt[t], 1); L.rowhit[d] = t; }
            }
            __syncthreads();
            // a.sum(1).max() == 1 and a.sum(0).max() == 1
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
            PROF(4);
            __syncthreads();
            if (one2one) {
                n_mi = block_compact(N, [&](int d) { return L.rowcnt[d] == 1; },
                                     [&](int d, int pos) { L.mi_r[pos] = d; L.mi_c[pos] = L.rowhit[d]; }, L.scan);
            } else {
                if (tid < WAVE) {
                    const int r = wave_lsa(cost, N, T, (size_t)T, (size_t)1, L.W, L.mi_r, L.mi_c);
                    if (tid == 0) { L.sc[SC_NMI] = r < 0 ? 0 : r; if (r == LSA_EINTERNAL) hdr[H_ERR] = TLK_EINTERNAL; }
                }
                __syncthreads();
                n_mi = L.sc[SC_NMI];
            }
        }
        __syncthreads();
        PROF(5);
        // unmatched lists + low-IoU rejection (association.py:276-296)
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
            nm = block_compact(n_mi, [&](int k) { return L.tmp_a[k] == 0; },
                               [&](int k, int pos) { L.m_d[pos] = L.mi_r[k]; L.m_t[pos] = L.mi_c[k]; }, L.scan);
            nud += nrej; nut += nrej;
        }
        __syncthreads();
        PROF(6);
        for (int k = tid; k < nm; k += BLOCK)                                  // ocsort.py:257-258
            kbt_update(trk_at(order[L.m_t[k]]), dets + (size_t)L.hi_idx[L.m_d[k]] * 7, P.delta_t);
        __syncthreads();

        PROF(7);
        // ---- second-round rounds share one routine: rows = candidate dets, cols = unmatched tracks
        auto second_round = [&](bool byte_round) {
            const int nrow = byte_round ? N2 : nud;
            const int ncol = nut;
            double *mat = ((size_t)nrow * ncol <= (size_t)cost_lds_entries) ? L.cost : (D.cost_g + (size_t)s * MAXD * MAXT);
            double lmax = -INFINITY; bool lnan = false;
            for (int e = tid; e < nrow * ncol; e += BLOCK) {
                const int r = e / ncol, c = e - r * ncol;
                const int di = byte_round ? L.lo_idx[r] : L.hi_idx[L.um_d[r]];
                const int tp = L.um_t[c];
                const double *tb = byte_round ? (L.trk_box + (size_t)tp * 4) : (lastb + (size_t)tp * 5);
                const double v = box_similarity(P.asso_func, dets + (size_t)di * 7, tb);
                mat[e] = v;
                lnan |= (v != v); lmax = v > lmax ? v : lmax;
            }
            double mx = block_max_nan(lnan ? NAN : lmax, true, L.red);
            if (P.asso_func == TLK_CT) {                                          // association.py:169-171
                lmax = -INFINITY; lnan = false;
                for (int e = tid; e < nrow * ncol; e += BLOCK) { double v = mat[e] / mx; mat[e] = v; lnan |= (v != v); lmax = v > lmax ? v : lmax; }
                const double m2 = block_max_nan(lnan ? NAN : lmax, true, L.red);
                lmax = -INFINITY; lnan = false;
                for (int e = tid; e < nrow * ncol; e += BLOCK) { double v = m2 - mat[e]; mat[e] = v; lnan |= (v != v); lmax = v > lmax ? v : lmax; }
                mx = block_max_nan(lnan ? NAN : lmax, true, L.red);
            }
            if (!(mx > P.iou_threshold)) return;                                  // uniform across the block
            for (int e = tid; e < nrow * ncol; e += BLOCK) mat[e] = -mat[e];
            __syncthreads();
            if (tid < WAVE) {
                const int r = wave_lsa(mat, nrow, ncol, (size_t)ncol, (size_t)1, L.W, L.mi_r, L.mi_c);
                if (tid == 0) { L.sc[SC_NL] = r < 0 ? 0 : r; if (r == LSA_EINTERNAL) hdr[H_ERR] = TLK_EINTERNAL; }
            }
            __syncthreads();
            const int nl = L.sc[SC_NL];
            // accepted pairs: iou_left >= threshold (mat holds the negated value)
            const int nacc = block_compact(nl, [&](int k) { return !((-mat[(size_t)L.mi_r[k] * ncol + L.mi_c[k]]) < P.iou_threshold); },
                                           [&](int k, int pos) {
                                               L.tmp_a[pos] = byte_round ? L.lo_idx[L.mi_r[k]] : L.hi_idx[L.um_d[L.mi_r[k]]];   // input det index
                                               L.tmp_b[pos] = L.um_t[L.mi_c[k]];                                               // track position
                                               L.m_d[pos] = byte_round ? -1 : L.um_d[L.mi_r[k]];                               // det index in hi list
                                           }, L.scan);
            __syncthreads();
            for (int k = tid; k < nacc; k += BLOCK)
                kbt_update(trk_at(order[L.tmp_b[k]]), dets + (size_t)L.tmp_a[k] * 7, P.delta_t);
            __syncthreads();
            // np.setdiff1d: sorted, unique (also when nothing was removed)
            nut = block_setdiff_sorted(L.um_t, nut, L.tmp_b, nacc, L.mi_c, L.scan);
            if (!byte_round) nud = block_setdiff_sorted(L.um_d, nud, L.m_d, nacc, L.mi_c, L.scan);
            __syncthreads();
        };
        if (P.use_byte && N2 > 0 && nut > 0) second_round(true);                   // ocsort.py:264-282
        if (nud > 0 && nut > 0) second_round(false);                               // ocsort.py:284-306
        __syncthreads();

        PROF(8);
        for (int k = tid; k < nut; k += BLOCK) kf_update_none(trk_at(order[L.um_t[k]]));   // ocsort.py:308-309
        // ---- births (ocsort.py:312-314)
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
        __syncthreads();
        nfree -= nud; nextid += nud; T += nud;
        PROF(9);
        // ---- emit rows in reversed list order + drop dead tracklets (ocsort.py:315-331)
        const int frame_count = hdr[H_FRAME];
        for (int q = tid; q < T; q += BLOCK) {
            const Trk K = trk_at(order[T - 1 - q]);
            const int tsu = K.i(FI_TSU);
            L.tmp_a[q] = (tsu < 1 && (K.i(FI_STREAK) >= P.min_hits || frame_count <= P.min_hits)) ? 1 : 0;
            L.tmp_b[T - 1 - q] = (tsu > P.max_age) ? 1 : 0;      // by position
            L.mi_r[T - 1 - q] = order[T - 1 - q];
        }
        __syncthreads();
        const int rows = block_compact(T, [&](int q) { return L.tmp_a[q] == 1; },
                                       [&](int q, int pos) {
                                           if (pos >= out_cap) return;
                                           const Trk K = trk_at(L.mi_r[T - 1 - q]);
                                           double lo[5], d4[4];
#pragma unroll
                                           for (int k = 0; k < 5; ++k) lo[k] = K.d(FD_LO + k);
                                           if (sum5(lo) < 0) {
                                               double x[4] = {K.d(FD_X), K.d(FD_X + 1), K.d(FD_X + 2), K.d(FD_X + 3)};
                                               x_to_bbox(x, d4);
                                           } else { d4[0] = lo[0]; d4[1] = lo[1]; d4[2] = lo[2]; d4[3] = lo[3]; }
                                           double *r = out + (size_t)pos * 8;
                                           r[0] = d4[0]; r[1] = d4[1]; r[2] = d4[2]; r[3] = d4[3];
                                           r[4] = (double)(K.i(FI_ID) + 1); r[5] = K.d(FD_CLS); r[6] = K.d(FD_CONF); r[7] = K.d(FD_TID);
                                       }, L.scan);
        const int kept = block_compact(T, [&](int p) { return L.tmp_b[p] == 0; },
                                       [&](int p, int pos) { order[pos] = L.mi_r[p]; }, L.scan);
        if (kept != T) {
            block_compact(T, [&](int p) { return L.tmp_b[p] != 0; },
                          [&](int p, int pos) { freestk[nfree + pos] = L.mi_r[p]; }, L.scan);
            nfree += T - kept;
        }
        __syncthreads();
        PROF(10);
        if (tid == 0) {
            hdr[H_NTRK] = kept; hdr[H_NFREE] = nfree; hdr[H_NEXTID] = nextid;
            *out_count = hdr[H_ERR] != 0 ? hdr[H_ERR] : (rows > out_cap ? TLK_ECAPACITY : rows);      // (H_ERR: TLK_EINTERNAL from a solver's loop bound)
        }
        __syncthreads();
    }
#ifdef TLK_LDS_CANARY
    if (can_live) {
        const int bad = canary_check(L);
        if (bad && tid == 0) { hdr[H_ERR] = -100 - bad; out_counts[(size_t)s * n_frames + n_frames - 1] = -100 - bad; }
    }
#endif
}

__global__ void ocsort_reset_kernel(OcsDev D, int stream)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        int *hdr = D.hdr + (size_t)s * H_COUNT;
        for (int k = threadIdx.x; k < D.MAXT; k += blockDim.x) D.freestk[(size_t)s * D.MAXT + k] = D.MAXT - 1 - k;   // pop order 0,1,2,...
        if (threadIdx.x == 0) { hdr[H_NTRK] = 0; hdr[H_FRAME] = 0; hdr[H_NEXTID] = 0; hdr[H_NFREE] = D.MAXT; hdr[H_ERR] = 0; }
    }
}

__global__ void ocsort_gather_kernel(OcsDev D, int stream, double *x, double *Pm, long long *ids, int cap, int *n_out)
{
    const int T = D.hdr[(size_t)stream * H_COUNT + H_NTRK];
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = T;
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < T && p < cap; p += gridDim.x * blockDim.x) {
        const int slot = D.order[(size_t)stream * D.MAXT + p];
        const double *fd = D.fd + (size_t)stream * D.MAXT + slot;
        for (int k = 0; k < 7; ++k) x[(size_t)p * 7 + k] = fd[(size_t)(FD_X + k) * stride];
        for (int k = 0; k < 49; ++k) Pm[(size_t)p * 49 + k] = fd[(size_t)(FD_P + k) * stride];
        ids[p] = D.fi[(size_t)FI_ID * stride + (size_t)stream * D.MAXT + slot];
    }
}

// ------------------------------------------------------------------ stateless KF7 entry points: one thread = one filter
__global__ void __launch_bounds__(BLOCK) kf7_predict_kernel(double *__restrict__ xs, double *__restrict__ Ps, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double x[7], P[49];
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = xs[(size_t)i * 7 + k];
#pragma unroll
    for (int k = 0; k < 49; ++k) P[k] = Ps[(size_t)i * 49 + k];
    kf7_predict(x, P);
#pragma unroll
    for (int k = 0; k < 7; ++k) xs[(size_t)i * 7 + k] = x[k];
#pragma unroll
    for (int k = 0; k < 49; ++k) Ps[(size_t)i * 49 + k] = P[k];
}
__global__ void __launch_bounds__(BLOCK) kf7_update_kernel(double *__restrict__ xs, double *__restrict__ Ps, const double *__restrict__ zs, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double x[7], P[49], z[4];
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = xs[(size_t)i * 7 + k];
#pragma unroll
    for (int k = 0; k < 49; ++k) P[k] = Ps[(size_t)i * 49 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = zs[(size_t)i * 4 + k];
    kf7_update_core(x, P, z);
#pragma unroll
    for (int k = 0; k < 7; ++k) xs[(size_t)i * 7 + k] = x[k];
#pragma unroll
    for (int k = 0; k < 49; ++k) Ps[(size_t)i * 49 + k] = P[k];
}

}  // namespace

// ------------------------------------------------------------------------------------ host side
struct tlk_ocsort {
    OcsDev D;
    OcsP P;
    int device;
    size_t smem;
    // staging for the host-buffer entry point
    double *d_dets, *d_out; int *d_cnt, *d_ocnt;
    double *h_pin; int *h_cnt;
    int out_cap;
};

static int ocs_free(tlk_ocsort *h)
{
    if (!h) return TLK_OK;
    hipSetDevice(h->device);
    hipFree(h->D.fd); hipFree(h->D.fi); hipFree(h->D.hdr); hipFree(h->D.order); hipFree(h->D.freestk);
    hipFree(h->D.lastb); hipFree(h->D.cost_g); hipFree(h->D.big_ws); if (h->D.prof) hipFree(h->D.prof); hipFree(h->d_dets); hipFree(h->d_out); hipFree(h->d_cnt); hipFree(h->d_ocnt);
    if (h->h_pin) hipHostFree(h->h_pin);
    if (h->h_cnt) hipHostFree(h->h_cnt);
    delete h;
    return TLK_OK;
}

extern "C" int tlk_ocsort_create(const tlk_ocsort_params *p, int n_streams, int device, tlk_ocsort **out)
{
    if (!p || !out) return fail(TLK_EINVAL, "tlk_ocsort_create: null pointer");
    if (n_streams < 1) return fail(TLK_EINVAL, "tlk_ocsort_create: n_streams must be >= 1");
    if (p->asso_func < TLK_IOU || p->asso_func > TLK_CT) return fail(TLK_EINVAL, "tlk_ocsort_create: unknown asso_func");
    if (p->delta_t < 0 || p->delta_t >= RINGN) return fail(TLK_EINVAL, "tlk_ocsort_create: delta_t must be in [0, 8)");
    const int MAXT = p->max_tracks > 0 ? p->max_tracks : 256, MAXD = p->max_dets > 0 ? p->max_dets : 128;
    // capacity = allocation size (r04): LDS tiers while the scene fits, HBM lists beyond (ocsort_frames_kernel)
    if (MAXT > 16384 || MAXD > 1024) return fail(TLK_ECAPACITY, "tlk_ocsort_create: max_tracks <= 16384 and max_dets <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(TLK_ENODEVICE, "tlk_ocsort_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_ocsort_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_ocsort *h = new tlk_ocsort();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->P = OcsP{p->det_thresh, p->iou_threshold, p->inertia, p->min_confidence, p->max_age, p->min_hits, p->delta_t,
                p->asso_func, p->use_byte, p->wrapper_mode};
    OcsDev &D = h->D;
    D.S = n_streams; D.MAXT = MAXT; D.MAXD = MAXD;
    const size_t budget = 160 * 1024 - 256;
    D.lds_bytes = (int)(budget & ~(size_t)15);
    h->smem = (size_t)D.lds_bytes;
    D.big_stride = (lds_fixed_bytes(MAXT, MAXD) + 255) & ~(size_t)255;
    const size_t slots = (size_t)n_streams * MAXT;
#define OCS_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { ocs_free(h); return fail(TLK_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    OCS_ALLOC(D.fd, sizeof(double) * FD_COUNT * slots);
    OCS_ALLOC(D.fi, sizeof(int) * FI_COUNT * slots);
    OCS_ALLOC(D.hdr, sizeof(int) * H_COUNT * n_streams);
    OCS_ALLOC(D.order, sizeof(int) * slots);
    OCS_ALLOC(D.freestk, sizeof(int) * slots);
    OCS_ALLOC(D.lastb, sizeof(double) * 5 * slots);
    OCS_ALLOC(D.cost_g, sizeof(double) * (size_t)n_streams * MAXD * MAXT);
    OCS_ALLOC(D.big_ws, D.big_stride * (size_t)n_streams);
    D.prof = nullptr;
    if (getenv("TLK_OCSORT_PROF")) { OCS_ALLOC(D.prof, sizeof(long long) * 16 * n_streams); hipMemset(D.prof, 0, sizeof(long long) * 16 * n_streams); }
    h->out_cap = MAXT + MAXD;
    OCS_ALLOC(h->d_dets, sizeof(double) * 7 * MAXD);
    OCS_ALLOC(h->d_out, sizeof(double) * 8 * h->out_cap);
    OCS_ALLOC(h->d_cnt, sizeof(int));
    OCS_ALLOC(h->d_ocnt, sizeof(int));
#undef OCS_ALLOC
    if (hipHostMalloc((void **)&h->h_pin, sizeof(double) * 8 * (size_t)(h->out_cap + MAXD)) != hipSuccess ||
        hipHostMalloc((void **)&h->h_cnt, sizeof(int) * 4) != hipSuccess) { ocs_free(h); return fail(TLK_EHIP, "hipHostMalloc failed"); }
    hipError_t e = hipMemset(D.fd, 0, sizeof(double) * FD_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.fi, 0, sizeof(int) * FI_COUNT * slots);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)ocsort_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) { ocs_free(h); return fail(TLK_EHIP, std::string("tlk_ocsort_create: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(ocsort_reset_kernel, dim3(n_streams < 256 ? n_streams : 256), dim3(BLOCK), 0, 0, D, -1);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { ocs_free(h); return fail(TLK_EHIP, std::string("tlk_ocsort_create: ") + hipGetErrorString(e)); }
    *out = h;
    return TLK_OK;
}

#ifdef TLK_LDS_CANARY
extern "C" int tlk_canary_selftest(int poke)
{
    TLK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_canary_poke), &poke, sizeof(int)));
    return TLK_OK;
}
#endif

extern "C" int tlk_ocsort_destroy(tlk_ocsort *h) { return ocs_free(h); }

extern "C" int tlk_ocsort_reset(tlk_ocsort *h, int stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_ocsort_reset: null handle");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ocsort_reset: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(ocsort_reset_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, 0, h->D, stream);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipStreamSynchronize(0));
    return TLK_OK;
}

extern "C" int tlk_ocsort_update_dev(tlk_ocsort *h, const double *dets_dev, const int32_t *counts_dev, int n_frames,
                                     double *out_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_ocsort_update_dev: null handle");
    if (n_frames < 0 || out_cap < 0) return fail(TLK_EINVAL, "tlk_ocsort_update_dev: negative size");
    if (n_frames == 0) return TLK_OK;
    if (!dets_dev || !counts_dev || !out_dev || !out_counts_dev) return fail(TLK_EINVAL, "tlk_ocsort_update_dev: null pointer");
    TLK_HIP(hipSetDevice(h->device));
    const size_t fstride = (size_t)h->D.MAXD * 7;
    hipLaunchKernelGGL(ocsort_frames_kernel, dim3(h->D.S), dim3(BLOCK), h->smem, (hipStream_t)hip_stream, h->D, h->P,
                       dets_dev, (const int *)counts_dev, n_frames, fstride * n_frames, fstride, out_dev, out_cap,
                       (int *)out_counts_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// single (stream, frame) with host buffers: run the same kernel on a 1-stream view of the bank
extern "C" int tlk_ocsort_update(tlk_ocsort *h, int stream, const double *dets, int n, double *out, int out_cap, int *n_out)
{
    if (!h || !n_out) return fail(TLK_EINVAL, "tlk_ocsort_update: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ocsort_update: stream out of range");
    if (n < 0 || (n > 0 && !dets)) return fail(TLK_EINVAL, "tlk_ocsort_update: bad detections");
    if (n > h->D.MAXD) return fail(TLK_ECAPACITY, "tlk_ocsort_update: more detections than max_dets");
    TLK_HIP(hipSetDevice(h->device));
    hipStream_t st = 0;
    double *pin_in = h->h_pin, *pin_out = h->h_pin + (size_t)8 * h->D.MAXD;
    if (n) memcpy(pin_in, dets, sizeof(double) * 7 * (size_t)n);
    h->h_cnt[0] = n;
    if (n) TLK_HIP(hipMemcpyAsync(h->d_dets, pin_in, sizeof(double) * 7 * (size_t)n, hipMemcpyHostToDevice, st));
    TLK_HIP(hipMemcpyAsync(h->d_cnt, h->h_cnt, sizeof(int), hipMemcpyHostToDevice, st));
    // view: shift every per-stream base by `stream`, keep the strides of the full bank
    OcsDev V = h->D;
    V.fd += (size_t)stream * V.MAXT; V.fi += (size_t)stream * V.MAXT;
    V.hdr += (size_t)stream * H_COUNT; V.order += (size_t)stream * V.MAXT; V.freestk += (size_t)stream * V.MAXT;
    V.lastb += (size_t)stream * V.MAXT * 5; V.cost_g += (size_t)stream * V.MAXD * V.MAXT; V.big_ws += (size_t)stream * V.big_stride;
    if (V.prof) V.prof += (size_t)stream * 16;
    hipLaunchKernelGGL(ocsort_frames_kernel, dim3(1), dim3(BLOCK), h->smem, st, V, h->P, (const double *)h->d_dets,
                       (const int *)h->d_cnt, 1, (size_t)0, (size_t)0, h->d_out, h->out_cap, h->d_ocnt);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipMemcpyAsync(h->h_cnt + 1, h->d_ocnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TLK_HIP(hipStreamSynchronize(st));
    const int rows = h->h_cnt[1];
    if (rows < 0) return fail_stream(rows, "tlk_ocsort_update");
    if (rows > out_cap) return fail(TLK_ECAPACITY, "tlk_ocsort_update: output buffer too small");
    if (rows) {
        TLK_HIP(hipMemcpyAsync(pin_out, h->d_out, sizeof(double) * 8 * (size_t)rows, hipMemcpyDeviceToHost, st));
        TLK_HIP(hipStreamSynchronize(st));
        memcpy(out, pin_out, sizeof(double) * 8 * (size_t)rows);
    }
    *n_out = rows;
    return TLK_OK;
}

extern "C" int tlk_ocsort_get_tracks(tlk_ocsort *h, int stream, double *x, double *P, int64_t *ids, int cap, int *n_tracks)
{
    if (!h || !n_tracks) return fail(TLK_EINVAL, "tlk_ocsort_get_tracks: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ocsort_get_tracks: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    double *dx = nullptr, *dP = nullptr; long long *di = nullptr; int *dn = nullptr;
    const size_t c = cap > 0 ? cap : 1;
    TLK_HIP(hipMalloc((void **)&dx, sizeof(double) * 7 * c));
    TLK_HIP(hipMalloc((void **)&dP, sizeof(double) * 49 * c));
    TLK_HIP(hipMalloc((void **)&di, sizeof(long long) * c));
    TLK_HIP(hipMalloc((void **)&dn, sizeof(int)));
    hipLaunchKernelGGL(ocsort_gather_kernel, dim3(4), dim3(BLOCK), 0, 0, h->D, stream, dx, dP, di, cap, dn);
    int n = 0;
    hipError_t e = hipMemcpy(&n, dn, sizeof(int), hipMemcpyDeviceToHost);
    const int k = n < cap ? n : cap;
    if (e == hipSuccess && k > 0) {
        e = hipMemcpy(x, dx, sizeof(double) * 7 * (size_t)k, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(P, dP, sizeof(double) * 49 * (size_t)k, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(ids, di, sizeof(long long) * (size_t)k, hipMemcpyDeviceToHost);
    }
    hipFree(dx); hipFree(dP); hipFree(di); hipFree(dn);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_ocsort_get_tracks: ") + hipGetErrorString(e));
    *n_tracks = n;
    return TLK_OK;
}

extern "C" int tlk_ocsort_get_profile(tlk_ocsort *h, int stream, long long *cycles16)
{
    if (!h || !cycles16) return fail(TLK_EINVAL, "tlk_ocsort_get_profile: null pointer");
    if (!h->D.prof) return fail(TLK_EINVAL, "tlk_ocsort_get_profile: create the bank with TLK_OCSORT_PROF=1 in the environment");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_ocsort_get_profile: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    TLK_HIP(hipMemcpy(cycles16, h->D.prof + (size_t)stream * 16, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    return TLK_OK;
}

extern "C" int tlk_kf7_predict_f64(double *x_dev, double *P_dev, int n, void *hip_stream)
{
    if (n < 0) return fail(TLK_EINVAL, "tlk_kf7_predict_f64: n < 0");
    if (n == 0) return TLK_OK;
    if (!x_dev || !P_dev) return fail(TLK_EINVAL, "tlk_kf7_predict_f64: null pointer");
    hipLaunchKernelGGL(kf7_predict_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)hip_stream, x_dev, P_dev, n);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_kf7_update_f64(double *x_dev, double *P_dev, const double *z_dev, int n, void *hip_stream)
{
    if (n < 0) return fail(TLK_EINVAL, "tlk_kf7_update_f64: n < 0");
    if (n == 0) return TLK_OK;
    if (!x_dev || !P_dev || !z_dev) return fail(TLK_EINVAL, "tlk_kf7_update_f64: null pointer");
    hipLaunchKernelGGL(kf7_update_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)hip_stream, x_dev, P_dev, z_dev, n);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
