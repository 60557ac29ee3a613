// tlk_bytetrack.hip -- ByteTrack (plugins/track/byte_track) on gfx950: ONE launch per frame for all streams of a bank.
//
// bytetrack_kernel: one 256-thread workgroup per stream walks BYTETracker.update (byte_tracker.py:167-320): split detections by
// score, predict the pool (tracked + lost), first association (IoU fused with the score) / second association (low scores) /
// unconfirmed tracks -- each a lap.lapjv(extend_cost=True, cost_limit) problem solved by the scipy-identical wavefront LSA in
// its reduced rows x cols form held in LDS (see lapjv_assign) --, KF updates, new tracks, time-outs, the tracked / lost list bookkeeping
// (including the frame a timed-out track lingers in the lost list, :296-298) and duplicate removal, output rows.
// Arithmetic follows the reference's dtype trail: boxes and IoU in float32 (+1 pixel convention, matching.py:181-217), a
// track's mean is float32 until its first predict/update, everything else float64 (-ffp-contract=off).
#include "tlk_common.hpp"
#include "tlk_strongsort_common.hpp"
#include "tlk_bytetrack_common.hpp"

using namespace tlk;

namespace {

enum : int { YD_MEAN = 0, YD_COV = 8, YD_SCORE = 72, YD_CLS = 73, YD_TLID = 74, YD_COUNT = 75 };
enum : int { YI_TID = 0, YI_STATE, YI_ACT, YI_TLEN, YI_FID, YI_SFID, YI_F32, YI_INREM, YI_COUNT };
enum : int { YH_NTRK = 0, YH_NLOST, YH_NFREE, YH_COUNT_ID, YH_FRAME, YH_ERR, YH_COUNT = 8 };
enum : int { BT_NEW = 0, BT_TRACKED = 1, BT_LOST = 2, BT_REMOVED = 3 };          // basetrack.py:5-9

struct ByDev {
    double *fd;              // YD_COUNT x S x MAXT
    int *fi;                 // YI_COUNT x S x MAXT
    int *hdr, *tracked, *lost, *freestk;      // lists hold slots, in list order
    double *ebuf;            // S x MAXT x MAXD   spill area of the assignment problem when it does not fit the LDS cost area
    int *alive;              // S x MAXT   slot-indexed marks of the end-of-frame free-slot sweep
    unsigned char *big_ws;   // S x big_stride: list / solver work area of the big-scene tier (by_carve_frame)
    size_t big_stride;
    int S, MAXT, MAXD, lds_bytes;
};
struct ByP { double track_thresh, match_thresh, det_thresh, min_conf; int max_time_lost, wrapper_mode; };
struct ByIn { const double *dets; const int *counts; size_t stream_stride_dets, count_stride; };

// multi_predict (kalman_filter.py:155-193): left = F cov first, then left F^T; noise from a float32 mean array stays float32
__device__ __forceinline__ void kfb_predict(double (&mean)[8], double (&cov)[64], bool all_f32)
{
    double q[8];
    if (all_f32) {
        const float h = (float)mean[3];
        const float sp = (float)W_POS * h, sv = (float)W_VEL * h, a = (float)1e-2 * 1.0f, b = (float)1e-5 * 1.0f;
        const float sd[8] = {sp, sp, a, sp, sv, sv, b, sv};
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float v = sd[i] * sd[i]; q[i] = (double)v; }
    } else {
        const double h = mean[3];
        const double sd[8] = {W_POS * h, W_POS * h, 1e-2, W_POS * h, W_VEL * h, W_VEL * h, 1e-5, W_VEL * h};
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

// STrack.tlbr (byte_tracker.py:97-117) in the mean's dtype, cast to float32 (matching.py:62-63)
__device__ __forceinline__ void trk_tlbr32(const BTrk &T, float *o)
{
    if (T.i(YI_F32)) {
        float r0 = (float)T.d(YD_MEAN), r1 = (float)T.d(YD_MEAN + 1), r2 = (float)T.d(YD_MEAN + 2), r3 = (float)T.d(YD_MEAN + 3);
        r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = r0; o[1] = r1; o[2] = r2 + r0; o[3] = r3 + r1;
    } else {
        double r0 = T.d(YD_MEAN), r1 = T.d(YD_MEAN + 1), r2 = T.d(YD_MEAN + 2), r3 = T.d(YD_MEAN + 3);
        r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = (float)r0; o[1] = (float)r1; o[2] = (float)(r2 + r0); o[3] = (float)(r3 + r1);
    }
}

__global__ void __launch_bounds__(BLOCK, 1)
bytetrack_kernel(ByDev Dv, ByP P, ByIn in, tlk_bytetrack_row *__restrict__ rows_all, size_t rows_stream_stride, int out_cap,
                 int *__restrict__ out_counts, size_t oc_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int MAXT = Dv.MAXT, MAXD = Dv.MAXD;
    ByLds L;
    int *hdr = Dv.hdr + (size_t)s * YH_COUNT;
    int *tracked = Dv.tracked + (size_t)s * MAXT, *lost = Dv.lost + (size_t)s * MAXT, *freestk = Dv.freestk + (size_t)s * MAXT;
    double *ebuf = Dv.ebuf + (size_t)s * MAXT * MAXD;
    const size_t stride = (size_t)Dv.S * MAXT;
    auto trk_at = [&](int slot) { BTrk T; T.fd = Dv.fd + (size_t)s * MAXT + slot; T.fi = Dv.fi + (size_t)s * MAXT + slot; T.stride = stride; return T; };
    tlk_bytetrack_row *rows = rows_all + (size_t)s * rows_stream_stride;
    int *out_count = out_counts + (size_t)s * oc_stride;
    const size_t dbase = (size_t)s * in.stream_stride_dets;
    const int n_in = in.counts[(size_t)s * in.count_stride];

    if (hdr[YH_ERR] != 0) { if (tid == 0) *out_count = hdr[YH_ERR]; return; }
    if (n_in > MAXD || n_in < 0) { if (tid == 0) { hdr[YH_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    if (P.wrapper_mode && n_in == 0) { if (tid == 0) *out_count = 0; return; }      // byte_track_api.py:55-56
    const int fid = hdr[YH_FRAME] + 1;                                                  // self.frame_id += 1
    int n_trk = hdr[YH_NTRK], n_lost = hdr[YH_NLOST], nfree = hdr[YH_NFREE], next_id = hdr[YH_COUNT_ID];
    const int cost_lds_entries = by_carve_frame(smem, Dv.lds_bytes, Dv.big_ws + (size_t)s * Dv.big_stride, MAXT, MAXD, n_trk + n_lost + n_in, n_in,
                                                Dv.alive + (size_t)s * MAXT, L);

    // wrapper filter inputs[:, 4] > min_confidence (byte_track_api.py:58), then the score split (:176-195)
    const int N = block_compact(n_in, [&](int i) { return in.dets[(dbase + i) * 7 + 4] > P.min_conf; }, [&](int i, int pos) { L.sel[pos] = i; }, L.scan);
    __syncthreads();
    for (int j = tid; j < N; j += BLOCK) {                  // STrack(xywh, ...): float32 "tlwh" = (cx, cy, w, h); tlbr / xyah in float32
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        const float t0 = (float)((d[0] + d[2]) / 2), t1 = (float)((d[1] + d[3]) / 2), t2 = (float)(d[2] - d[0]), t3 = (float)(d[3] - d[1]);
        L.dbox[j * 4] = t0; L.dbox[j * 4 + 1] = t1; L.dbox[j * 4 + 2] = t2 + t0; L.dbox[j * 4 + 3] = t3 + t1;
        L.dxyah[j * 4] = t0 + t2 / 2; L.dxyah[j * 4 + 1] = t1 + t3 / 2; L.dxyah[j * 4 + 2] = t2 / t3; L.dxyah[j * 4 + 3] = t3;
        L.dscore[j] = d[4];
    }
    __syncthreads();
    const int nhi = block_compact(N, [&](int j) { return L.dscore[j] > P.track_thresh; }, [&](int j, int pos) { L.hi[pos] = j; }, L.scan);
    const int nlo = block_compact(N, [&](int j) { const double c = L.dscore[j]; return !(c > P.track_thresh) && c > 0.1 && c < P.track_thresh; },
                                  [&](int j, int pos) { L.lo[pos] = j; }, L.scan);
    // unconfirmed / pool = joint(activated tracked, lost) (:201-212)
    const int n_unconf = block_compact(n_trk, [&](int p) { return trk_at(tracked[p]).i(YI_ACT) == 0; }, [&](int p, int pos) { L.unconf[pos] = tracked[p]; }, L.scan);
    const int n_act0 = block_compact(n_trk, [&](int p) { return trk_at(tracked[p]).i(YI_ACT) != 0; }, [&](int p, int pos) { L.pool[pos] = tracked[p]; }, L.scan);
    for (int q = tid; q < n_lost; q += BLOCK) L.pool[n_act0 + q] = lost[q];
    const int n_pool = n_act0 + n_lost;
    for (int p = tid; p < MAXT; p += BLOCK) L.alive[p] = 0;
    __syncthreads();
    // multi_predict (:26-40)
    {
        int f32all = 1;
        for (int p = tid; p < n_pool; p += BLOCK) f32all &= trk_at(L.pool[p]).i(YI_F32) != 0;
        const int all_f32 = __syncthreads_and(f32all);
        for (int p = tid; p < n_pool; p += BLOCK) {
            const BTrk Kt = trk_at(L.pool[p]);
            double mean[8], cov[64];
#pragma unroll
            for (int k = 0; k < 8; ++k) mean[k] = Kt.d(YD_MEAN + k);
#pragma unroll
            for (int k = 0; k < 64; ++k) cov[k] = Kt.d(YD_COV + k);
            if (Kt.i(YI_STATE) != BT_TRACKED) mean[7] = 0;
            kfb_predict(mean, cov, all_f32 != 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) Kt.d(YD_MEAN + k) = mean[k];
#pragma unroll
            for (int k = 0; k < 64; ++k) Kt.d(YD_COV + k) = cov[k];
            Kt.i(YI_F32) = 0;
        }
    }
    __syncthreads();
    // KF update of a matched track (STrack.update :76-94 / re_activate :60-74); one thread per match
    auto apply = [&](int slot, int j, bool reactivate) {
        const BTrk Kt = trk_at(slot);
        double mean[8], cov[64], z[4], sd[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) mean[k] = Kt.d(YD_MEAN + k);
#pragma unroll
        for (int k = 0; k < 64; ++k) cov[k] = Kt.d(YD_COV + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (double)L.dxyah[j * 4 + k];
        double sp;
        if (Kt.i(YI_F32)) { const float sf = (float)W_POS * (float)mean[3]; sp = (double)sf; } else sp = W_POS * mean[3];
        sd[0] = sp; sd[1] = sp; sd[2] = 1e-1; sd[3] = sp;          // project(): [h/20, h/20, 1e-1, h/20] (kalman_filter.py:141-145)
        kf8_update_sd(mean, cov, z, sd);
#pragma unroll
        for (int k = 0; k < 8; ++k) Kt.d(YD_MEAN + k) = mean[k];
#pragma unroll
        for (int k = 0; k < 64; ++k) Kt.d(YD_COV + k) = cov[k];
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        Kt.i(YI_F32) = 0;
        if (reactivate) { Kt.i(YI_TLEN) = 0; Kt.d(YD_CLS) = d[5]; } else Kt.i(YI_TLEN) = Kt.i(YI_TLEN) + 1;
        Kt.i(YI_FID) = fid; Kt.i(YI_STATE) = BT_TRACKED; Kt.i(YI_ACT) = 1;
        Kt.d(YD_SCORE) = d[4]; Kt.d(YD_TLID) = d[6];
    };

    // ---- step 2: first association, high-score detections (:214-227) ----
    for (int p = tid; p < n_pool; p += BLOCK) { trk_tlbr32(trk_at(L.pool[p]), L.tbox + p * 4); L.pre[p] = trk_at(L.pool[p]).i(YI_STATE); }
    __syncthreads();
    const AsgOut A1 = lapjv_assign(n_pool, nhi, P.match_thresh, [&](int r, int c) {
        const int j = L.hi[c];
        const float c32 = 1 - bbox_iou32(L.tbox + r * 4, L.dbox + j * 4);        // iou_distance (float32)
        const float sim = 1 - c32;                                               // fuse_score: (1 - cost) * score in float64
        return 1 - (double)sim * L.dscore[j];
    }, ebuf, L.cost, cost_lds_entries, L);
    for (int k = tid; k < A1.nm; k += BLOCK) apply(L.pool[L.m_r[k]], L.hi[L.m_c[k]], L.pre[L.m_r[k]] != BT_TRACKED);
    const int n_ref = block_compact(A1.nm, [&](int k) { return L.pre[L.m_r[k]] != BT_TRACKED; }, [&](int k, int pos) { L.refind[pos] = L.pool[L.m_r[k]]; }, L.scan);
    for (int k = tid; k < A1.n_uc; k += BLOCK) L.udet1[k] = L.hi[L.u_c[k]];     // remaining high-score detections (filtered index)
    const int n_udet1 = A1.n_uc;
    // ---- step 3: second association, low-score detections against the still-Tracked rest (:229-250) ----
    const int n_rtr = block_compact(A1.n_ur, [&](int q) { return L.pre[L.u_r[q]] == BT_TRACKED; }, [&](int q, int pos) { L.rtr[pos] = L.pool[L.u_r[q]]; }, L.scan);
    __syncthreads();
    for (int p = tid; p < n_rtr; p += BLOCK) trk_tlbr32(trk_at(L.rtr[p]), L.tbox + p * 4);
    __syncthreads();
    const AsgOut A2 = lapjv_assign(n_rtr, nlo, 0.5, [&](int r, int c) {
        return (double)(float)(1 - bbox_iou32(L.tbox + r * 4, L.dbox + L.lo[c] * 4));
    }, ebuf, L.cost, cost_lds_entries, L);
    for (int k = tid; k < A2.nm; k += BLOCK) apply(L.rtr[L.m_r[k]], L.lo[L.m_c[k]], false);
    for (int k = tid; k < A2.n_ur; k += BLOCK) { const int slot = L.rtr[L.u_r[k]]; trk_at(slot).i(YI_STATE) = BT_LOST; L.newlost[k] = slot; }   // mark_lost
    const int n_newlost = A2.n_ur;
    __syncthreads();
    // ---- unconfirmed tracks against the remaining high-score detections (:252-263) ----
    for (int p = tid; p < n_unconf; p += BLOCK) trk_tlbr32(trk_at(L.unconf[p]), L.tbox + p * 4);
    __syncthreads();
    const AsgOut A3 = lapjv_assign(n_unconf, n_udet1, 0.7, [&](int r, int c) {
        const int j = L.udet1[c];
        const float c32 = 1 - bbox_iou32(L.tbox + r * 4, L.dbox + j * 4);
        const float sim = 1 - c32;
        return 1 - (double)sim * L.dscore[j];
    }, ebuf, L.cost, cost_lds_entries, L);
    for (int k = tid; k < A3.nm; k += BLOCK) apply(L.unconf[L.m_r[k]], L.udet1[L.m_c[k]], false);
    for (int k = tid; k < A3.n_ur; k += BLOCK) { const int slot = L.unconf[L.u_r[k]]; trk_at(slot).i(YI_STATE) = BT_REMOVED; L.removed[k] = slot; }
    int n_removed = A3.n_ur;
    if (A1.err | A2.err | A3.err) { if (tid == 0) { hdr[YH_ERR] = TLK_EINTERNAL; *out_count = TLK_EINTERNAL; } return; }      // uniform: an assignment solver hit its loop bound
    __syncthreads();
    // ---- step 4: new tracks from the still unmatched detections with score >= det_thresh (:265-271) ----
    const int n_new = block_compact(A3.n_uc, [&](int q) { return !(L.dscore[L.udet1[L.u_c[q]]] < P.det_thresh); }, [&](int q, int pos) { L.rem[pos] = L.udet1[L.u_c[q]]; }, L.scan);
    if (n_new > nfree) { if (tid == 0) { hdr[YH_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    __syncthreads();
    for (int k = tid; k < n_new; k += BLOCK) {
        const int j = L.rem[k], slot = freestk[nfree - 1 - k];
        L.newtrk[k] = slot;
        const BTrk Kt = trk_at(slot);
        const double *d = in.dets + (dbase + L.sel[j]) * 7;
        // kalman_filter.py:55-87 on a float32 measurement: float32 products in a list with python floats -> float64 squares
        const float m3 = L.dxyah[j * 4 + 3];
        const float spf = (float)(2 * W_POS) * m3, svf = (float)(10 * W_VEL) * m3;
        const double sd[8] = {(double)spf, (double)spf, 1e-2, (double)spf, (double)svf, (double)svf, 1e-5, (double)svf};
#pragma unroll
        for (int q = 0; q < 64; ++q) Kt.d(YD_COV + q) = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { Kt.d(YD_MEAN + q) = (double)L.dxyah[j * 4 + q]; Kt.d(YD_MEAN + 4 + q) = 0.0; }
#pragma unroll
        for (int q = 0; q < 8; ++q) Kt.d(YD_COV + q * 9) = sd[q] * sd[q];
        Kt.i(YI_F32) = 1; Kt.i(YI_TID) = next_id + 1 + k;         // next_id(): _count += 1; return _count
        Kt.i(YI_TLEN) = 0; Kt.i(YI_STATE) = BT_TRACKED; Kt.i(YI_ACT) = fid == 1 ? 1 : 0; Kt.i(YI_FID) = fid; Kt.i(YI_SFID) = fid; Kt.i(YI_INREM) = 0;
        Kt.d(YD_SCORE) = d[4]; Kt.d(YD_CLS) = d[5]; Kt.d(YD_TLID) = d[6];
    }
    nfree -= n_new; next_id += n_new;
    __syncthreads();
    // ---- step 5: lost tracks that timed out (:273-277) ----
    {
        const int nto = block_compact(n_lost, [&](int q) { const BTrk Kt = trk_at(lost[q]); return fid - Kt.i(YI_FID) > P.max_time_lost; },
                                      [&](int q, int pos) { L.removed[n_removed + pos] = lost[q]; }, L.scan);
        __syncthreads();
        for (int k = tid; k < nto; k += BLOCK) trk_at(L.removed[n_removed + k]).i(YI_STATE) = BT_REMOVED;
        n_removed += nto;
    }
    __syncthreads();
    // ---- list bookkeeping (:279-290) ----
    // every slot that is live now: old tracked, old lost, new
    for (int p = tid; p < n_trk; p += BLOCK) L.alive[tracked[p]] = 1;
    for (int q = tid; q < n_lost; q += BLOCK) L.alive[lost[q]] = 1;
    for (int k = tid; k < n_new; k += BLOCK) L.alive[L.newtrk[k]] = 1;
    __syncthreads();
    // tracked' = [old tracked still Tracked] + new tracks + re-found (joint_stracks keeps the first occurrence)
    int nn = block_compact(n_trk, [&](int p) { return trk_at(tracked[p]).i(YI_STATE) == BT_TRACKED; }, [&](int p, int pos) { L.ntr[pos] = tracked[p]; }, L.scan);
    for (int k = tid; k < n_new; k += BLOCK) L.ntr[nn + k] = L.newtrk[k];
    for (int k = tid; k < n_ref; k += BLOCK) L.ntr[nn + n_new + k] = L.refind[k];
    nn += n_new + n_ref;
    // lost' = [old lost not re-found] + newly lost, minus the ids already in self.removed_stracks BEFORE this frame's are appended
    int nl = block_compact(n_lost, [&](int q) { const BTrk Kt = trk_at(lost[q]); return Kt.i(YI_STATE) != BT_TRACKED && Kt.i(YI_INREM) == 0; },
                           [&](int q, int pos) { L.nlost[pos] = lost[q]; }, L.scan);
    nl += block_compact(n_newlost, [&](int k) { return trk_at(L.newlost[k]).i(YI_INREM) == 0; }, [&](int k, int pos) { L.nlost[nl + pos] = L.newlost[k]; }, L.scan);
    __syncthreads();
    for (int k = tid; k < n_removed; k += BLOCK) trk_at(L.removed[k]).i(YI_INREM) = 1;
    if (nn > MAXT || nl > MAXT) { if (tid == 0) { hdr[YH_ERR] = TLK_ECAPACITY; *out_count = TLK_ECAPACITY; } return; }
    // remove_duplicate_stracks (:346-361)
    for (int p = tid; p < nn; p += BLOCK) { L.dupa[p] = 0; trk_tlbr32(trk_at(L.ntr[p]), L.tbox + p * 4); }
    for (int q = tid; q < nl; q += BLOCK) L.dupb[q] = 0;
    __syncthreads();
    for (int e = tid; e < nn * nl; e += BLOCK) {
        const int p = e / nl, q = e - p * nl;
        float lb[4];
        trk_tlbr32(trk_at(L.nlost[q]), lb);
        const float pd = 1 - bbox_iou32(L.tbox + p * 4, lb);
        if (pd < (float)0.15) {
            const BTrk Ka = trk_at(L.ntr[p]), Kb = trk_at(L.nlost[q]);
            const int timep = Ka.i(YI_FID) - Ka.i(YI_SFID), timeq = Kb.i(YI_FID) - Kb.i(YI_SFID);
            if (timep > timeq) L.dupb[q] = 1; else L.dupa[p] = 1;
        }
    }
    __syncthreads();
    const int nn2 = block_compact(nn, [&](int p) { return L.dupa[p] == 0; }, [&](int p, int pos) { tracked[pos] = L.ntr[p]; }, L.scan);
    const int nl2 = block_compact(nl, [&](int q) { return L.dupb[q] == 0; }, [&](int q, int pos) { lost[pos] = L.nlost[q]; }, L.scan);
    __syncthreads();
    // slots that are in neither list any more go back to the free stack
    for (int p = tid; p < nn2; p += BLOCK) L.alive[tracked[p]] = 0;
    for (int q = tid; q < nl2; q += BLOCK) L.alive[lost[q]] = 0;
    __syncthreads();
    const int ndead = block_compact(MAXT, [&](int slot) { return L.alive[slot] != 0; }, [&](int slot, int pos) { freestk[nfree + pos] = slot; }, L.scan);
    nfree += ndead;
    __syncthreads();
    if (tid == 0) { hdr[YH_NTRK] = nn2; hdr[YH_NLOST] = nl2; hdr[YH_NFREE] = nfree; hdr[YH_COUNT_ID] = next_id; hdr[YH_FRAME] = fid; }
    // ---- outputs (:292-309): activated tracks of tracked'', xywh2xyxy of the "tlwh" in the mean's dtype ----
    const int nrows = block_compact(nn2, [&](int p) { return trk_at(tracked[p]).i(YI_ACT) != 0; },
                                    [&](int p, int pos) {
                                        if (pos >= out_cap) return;
                                        const BTrk Kt = trk_at(tracked[p]);
                                        tlk_bytetrack_row r;
                                        if (Kt.i(YI_F32)) {
                                            float r0 = (float)Kt.d(YD_MEAN), r1 = (float)Kt.d(YD_MEAN + 1), r2 = (float)Kt.d(YD_MEAN + 2), r3 = (float)Kt.d(YD_MEAN + 3);
                                            r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
                                            const float hw = r2 / 2, hh = r3 / 2;
                                            r.ltrb[0] = r0 - hw; r.ltrb[1] = r1 - hh; r.ltrb[2] = r0 + hw; r.ltrb[3] = r1 + hh;
                                        } else {
                                            double r0 = Kt.d(YD_MEAN), r1 = Kt.d(YD_MEAN + 1), r2 = Kt.d(YD_MEAN + 2), r3 = Kt.d(YD_MEAN + 3);
                                            r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
                                            const double hw = r2 / 2, hh = r3 / 2;
                                            r.ltrb[0] = r0 - hw; r.ltrb[1] = r1 - hh; r.ltrb[2] = r0 + hw; r.ltrb[3] = r1 + hh;
                                        }
                                        r.det_id = (long long)Kt.d(YD_TLID); r.track_id = Kt.i(YI_TID);
                                        r.score = Kt.d(YD_SCORE); r.cls = Kt.d(YD_CLS);
                                        rows[pos] = r;
                                    }, L.scan);
    if (tid == 0) *out_count = nrows > out_cap ? TLK_ECAPACITY : nrows;
}

__global__ void bytetrack_reset_kernel(ByDev D, int stream, int keep_ids)
{
    const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? D.S : stream + 1;
    for (int s = s0 + blockIdx.x; s < s1; s += gridDim.x) {
        int *hdr = D.hdr + (size_t)s * YH_COUNT;
        for (int k = threadIdx.x; k < D.MAXT; k += blockDim.x) D.freestk[(size_t)s * D.MAXT + k] = D.MAXT - 1 - k;
        if (threadIdx.x == 0) { hdr[YH_NTRK] = 0; hdr[YH_NLOST] = 0; hdr[YH_NFREE] = D.MAXT; if (!keep_ids) hdr[YH_COUNT_ID] = 0; hdr[YH_FRAME] = 0; hdr[YH_ERR] = 0; }
    }
}

__global__ void bytetrack_gather_kernel(ByDev D, int stream, int which, long long *ids, double *mean, double *cov, long long *state5, int cap, int *n_out)
{
    const int n = D.hdr[(size_t)stream * YH_COUNT + (which ? YH_NLOST : YH_NTRK)];
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = n;
    const int *list = (which ? D.lost : D.tracked) + (size_t)stream * D.MAXT;
    const size_t stride = (size_t)D.S * D.MAXT;
    for (int p = blockIdx.x; p < n && p < cap; p += gridDim.x) {
        const int slot = list[p];
        const double *fd = D.fd + (size_t)stream * D.MAXT + slot;
        const int *fi = D.fi + (size_t)stream * D.MAXT + slot;
        if (ids && threadIdx.x == 0) ids[p] = fi[(size_t)YI_TID * stride];
        if (mean) for (int k = threadIdx.x; k < 8; k += blockDim.x) mean[(size_t)p * 8 + k] = fd[(size_t)(YD_MEAN + k) * stride];
        if (cov) for (int k = threadIdx.x; k < 64; k += blockDim.x) cov[(size_t)p * 64 + k] = fd[(size_t)(YD_COV + k) * stride];
        if (state5 && threadIdx.x == 0) {
            state5[(size_t)p * 5] = fi[(size_t)YI_STATE * stride]; state5[(size_t)p * 5 + 1] = fi[(size_t)YI_ACT * stride];
            state5[(size_t)p * 5 + 2] = fi[(size_t)YI_FID * stride]; state5[(size_t)p * 5 + 3] = fi[(size_t)YI_SFID * stride];
            state5[(size_t)p * 5 + 4] = fi[(size_t)YI_TLEN * stride];
        }
    }
}

}  // namespace

struct tlk_bytetrack {
    ByDev D; ByP P; int device; size_t smem;
    double *d_dets; int *d_cnt, *d_ocnt; tlk_bytetrack_row *d_rows;
    int out_cap;
};

static void by_free(tlk_bytetrack *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    void *ptrs[] = {h->D.fd, h->D.fi, h->D.hdr, h->D.tracked, h->D.lost, h->D.freestk, h->D.ebuf, h->D.alive, h->D.big_ws, h->d_dets, h->d_cnt, h->d_ocnt, h->d_rows};
    for (void *p : ptrs) if (p) hipFree(p);
    delete h;
}

extern "C" int tlk_bytetrack_create(const tlk_bytetrack_params *p, int n_streams, int device, tlk_bytetrack **out)
{
    if (!p || !out) return fail(TLK_EINVAL, "tlk_bytetrack_create: null pointer");
    if (n_streams < 1) return fail(TLK_EINVAL, "tlk_bytetrack_create: n_streams must be >= 1");
    const int MAXT = p->max_tracks > 0 ? p->max_tracks : 256, MAXD = p->max_dets > 0 ? p->max_dets : 128;
    // capacity = allocation size (r04): LDS tiers while the scene fits, HBM lists beyond (by_carve_frame)
    if (MAXT > 16384 || MAXD > 1024) return fail(TLK_ECAPACITY, "tlk_bytetrack_create: max_tracks <= 16384 and max_dets <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_bytetrack_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_bytetrack_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_bytetrack *h = new tlk_bytetrack();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->P = ByP{p->track_thresh, p->match_thresh, p->track_thresh + 0.1, p->min_confidence, (int)(p->frame_rate / 30.0 * p->track_buffer), p->wrapper_mode};
    ByDev &D = h->D;
    D.S = n_streams; D.MAXT = MAXT; D.MAXD = MAXD;
    const size_t budget = 160 * 1024 - 256;
    D.lds_bytes = (int)(budget & ~(size_t)15);
    h->smem = (size_t)D.lds_bytes;
    D.big_stride = (bylds_bytes(MAXT, MAXD, MAXT + MAXD) + 16 + 255) & ~(size_t)255;
    const size_t slots = (size_t)n_streams * MAXT;
    h->out_cap = MAXT;
#define BY_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { by_free(h); return fail(TLK_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    BY_ALLOC(D.fd, sizeof(double) * YD_COUNT * slots);
    BY_ALLOC(D.fi, sizeof(int) * YI_COUNT * slots);
    BY_ALLOC(D.hdr, sizeof(int) * YH_COUNT * n_streams);
    BY_ALLOC(D.tracked, sizeof(int) * slots);
    BY_ALLOC(D.lost, sizeof(int) * slots);
    BY_ALLOC(D.freestk, sizeof(int) * slots);
    BY_ALLOC(D.ebuf, sizeof(double) * slots * MAXD);
    BY_ALLOC(D.alive, sizeof(int) * slots);
    BY_ALLOC(D.big_ws, D.big_stride * (size_t)n_streams);
    BY_ALLOC(h->d_dets, sizeof(double) * 7 * MAXD);
    BY_ALLOC(h->d_cnt, sizeof(int));
    BY_ALLOC(h->d_ocnt, sizeof(int));
    BY_ALLOC(h->d_rows, sizeof(tlk_bytetrack_row) * h->out_cap);
#undef BY_ALLOC
    hipError_t e = hipMemset(D.fd, 0, sizeof(double) * YD_COUNT * slots);
    if (e == hipSuccess) e = hipMemset(D.fi, 0, sizeof(int) * YI_COUNT * slots);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)bytetrack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    if (e != hipSuccess) { by_free(h); return fail(TLK_EHIP, std::string("tlk_bytetrack_create: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(bytetrack_reset_kernel, dim3(n_streams < 256 ? n_streams : 256), dim3(BLOCK), 0, 0, D, -1, 0);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { by_free(h); return fail(TLK_EHIP, std::string("tlk_bytetrack_create: ") + hipGetErrorString(e)); }
    *out = h;
    return TLK_OK;
}

extern "C" int tlk_bytetrack_destroy(tlk_bytetrack *h) { by_free(h); return TLK_OK; }

static int bytetrack_reset_impl(tlk_bytetrack *h, int stream, int keep_ids)
{
    if (!h) return fail(TLK_EINVAL, "tlk_bytetrack_reset: null handle");
    if (stream >= h->D.S) return fail(TLK_EINVAL, "tlk_bytetrack_reset: stream out of range");
    TLK_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(bytetrack_reset_kernel, dim3(stream < 0 ? (h->D.S < 256 ? h->D.S : 256) : 1), dim3(BLOCK), 0, 0, h->D, stream, keep_ids);
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipStreamSynchronize(0));
    return TLK_OK;
}

extern "C" int tlk_bytetrack_reset(tlk_bytetrack *h, int stream) { return bytetrack_reset_impl(h, stream, 0); }
extern "C" int tlk_bytetrack_reset_keep_ids(tlk_bytetrack *h, int stream) { return bytetrack_reset_impl(h, stream, 1); }

extern "C" int tlk_bytetrack_update_dev(tlk_bytetrack *h, const double *dets_dev, const int32_t *counts_dev, int n_frames,
                                        tlk_bytetrack_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream)
{
    if (!h) return fail(TLK_EINVAL, "tlk_bytetrack_update_dev: null handle");
    if (n_frames < 0 || out_cap < 0) return fail(TLK_EINVAL, "tlk_bytetrack_update_dev: negative size");
    if (n_frames == 0) return TLK_OK;
    if (!dets_dev || !counts_dev || !rows_dev || !out_counts_dev) return fail(TLK_EINVAL, "tlk_bytetrack_update_dev: null pointer");
    TLK_HIP(hipSetDevice(h->device));
    const ByDev &D = h->D;
    for (int f = 0; f < n_frames; ++f) {
        ByIn in;
        in.dets = dets_dev + (size_t)f * D.MAXD * 7; in.counts = (const int *)counts_dev + f;
        in.stream_stride_dets = (size_t)n_frames * D.MAXD; in.count_stride = (size_t)n_frames;
        hipLaunchKernelGGL(bytetrack_kernel, dim3(D.S), dim3(BLOCK), h->smem, (hipStream_t)hip_stream, D, h->P, in, rows_dev + (size_t)f * out_cap,
                           (size_t)n_frames * out_cap, out_cap, (int *)out_counts_dev + f, (size_t)n_frames);
    }
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_bytetrack_update(tlk_bytetrack *h, int stream, const double *dets, int n, tlk_bytetrack_row *rows, int cap, int *n_out)
{
    if (!h || !n_out) return fail(TLK_EINVAL, "tlk_bytetrack_update: null pointer");
    if (stream < 0 || stream >= h->D.S) return fail(TLK_EINVAL, "tlk_bytetrack_update: stream out of range");
    if (n < 0 || (n > 0 && !dets)) return fail(TLK_EINVAL, "tlk_bytetrack_update: bad detections");
    if (n > h->D.MAXD) return fail(TLK_ECAPACITY, "tlk_bytetrack_update: more detections than max_dets");
    TLK_HIP(hipSetDevice(h->device));
    hipStream_t st = 0;
    if (n) TLK_HIP(hipMemcpyAsync(h->d_dets, dets, sizeof(double) * 7 * n, hipMemcpyHostToDevice, st));
    TLK_HIP(hipMemcpyAsync(h->d_cnt, &n, sizeof(int), hipMemcpyHostToDevice, st));
    ByDev V = h->D;
    const size_t sl = (size_t)stream * V.MAXT;
    V.fd += sl; V.fi += sl; V.hdr += (size_t)stream * YH_COUNT; V.tracked += sl; V.lost += sl; V.freestk += sl; V.ebuf += sl * V.MAXD; V.alive += sl; V.big_ws += (size_t)stream * V.big_stride;
    ByIn in;
    in.dets = h->d_dets; in.counts = h->d_cnt; in.stream_stride_dets = 0; in.count_stride = 0;
    hipLaunchKernelGGL(bytetrack_kernel, dim3(1), dim3(BLOCK), h->smem, st, V, h->P, in, h->d_rows, (size_t)0, h->out_cap, h->d_ocnt, (size_t)0);
    TLK_HIP(hipGetLastError());
    int rows_n = 0;
    TLK_HIP(hipMemcpyAsync(&rows_n, h->d_ocnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TLK_HIP(hipStreamSynchronize(st));
    if (rows_n < 0) return fail_stream(rows_n, "tlk_bytetrack_update");
    if (rows_n > cap) return fail(TLK_ECAPACITY, "tlk_bytetrack_update: output buffer too small");
    if (rows_n) TLK_HIP(hipMemcpy(rows, h->d_rows, sizeof(tlk_bytetrack_row) * rows_n, hipMemcpyDeviceToHost));
    *n_out = rows_n;
    return TLK_OK;
}

extern "C" int tlk_bytetrack_get_tracks(tlk_bytetrack *h, int stream, int which, int64_t *ids, double *mean, double *cov, int64_t *state5,
                                        int cap, int *n_tracks)
{
    if (!h || !n_tracks) return fail(TLK_EINVAL, "tlk_bytetrack_get_tracks: null pointer");
    if (stream < 0 || stream >= h->D.S || cap < 0 || which < 0 || which > 1) return fail(TLK_EINVAL, "tlk_bytetrack_get_tracks: bad argument");
    TLK_HIP(hipSetDevice(h->device));
    const size_t c = cap > 0 ? cap : 1;
    long long *d_ids = nullptr, *d_st = nullptr; double *d_mean = nullptr, *d_cov = nullptr; int *d_n = nullptr;
    TLK_HIP(hipMalloc((void **)&d_ids, sizeof(long long) * c)); TLK_HIP(hipMalloc((void **)&d_st, sizeof(long long) * 5 * c));
    TLK_HIP(hipMalloc((void **)&d_mean, sizeof(double) * 8 * c)); TLK_HIP(hipMalloc((void **)&d_cov, sizeof(double) * 64 * c));
    TLK_HIP(hipMalloc((void **)&d_n, sizeof(int)));
    hipLaunchKernelGGL(bytetrack_gather_kernel, dim3(64), dim3(64), 0, 0, h->D, stream, which, d_ids, d_mean, d_cov, d_st, cap, d_n);
    int n = 0;
    hipError_t e = hipMemcpy(&n, d_n, sizeof(int), hipMemcpyDeviceToHost);
    const int m = n < cap ? n : cap;
    if (e == hipSuccess && m > 0) {
        if (ids) e = hipMemcpy(ids, d_ids, sizeof(long long) * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && mean) e = hipMemcpy(mean, d_mean, sizeof(double) * 8 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && cov) e = hipMemcpy(cov, d_cov, sizeof(double) * 64 * m, hipMemcpyDeviceToHost);
        if (e == hipSuccess && state5) e = hipMemcpy(state5, d_st, sizeof(long long) * 5 * m, hipMemcpyDeviceToHost);
    }
    hipFree(d_ids); hipFree(d_st); hipFree(d_mean); hipFree(d_cov); hipFree(d_n);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_bytetrack_get_tracks: ") + hipGetErrorString(e));
    *n_tracks = n;
    return TLK_OK;
}
