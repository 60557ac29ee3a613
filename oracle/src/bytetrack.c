/*
 * oracle/src/bytetrack.c -- CPU ORACLE (test infrastructure, NOT product code).
 * ByteTrack: plugins/track/byte_track/{byte_tracker.py, matching.py, kalman_filter.py, basetrack.py} restated in C.
 *
 *   BYTETracker.update          byte_tracker.py:167-320   (steps 1-5, list bookkeeping, output rows)
 *   STrack.{activate,re_activate,update,tlwh,tlbr,tlwh_to_xyah,multi_predict}   byte_tracker.py:13-152
 *   joint_stracks / sub_stracks / remove_duplicate_stracks   byte_tracker.py:323-361
 *   iou_distance / ious / bbox_ious (+1 pixel convention, float32)   matching.py:50-90, :181-217
 *   fuse_score                  matching.py:171-179
 *   linear_assignment           matching.py:37-48  (lap.lapjv(extend_cost=True, cost_limit): orc_lapjv_limit)
 *   KalmanFilter                kalman_filter.py:55-270 (DeepSORT xyah filter: noise relative to h, 1e-2 / 1e-5 on the aspect ratio)
 *
 * Reproduced on purpose: detections are built from xyxy2xywh() rows, so the "tlwh" every STrack carries is really
 * (cx, cy, w, h) and the output xywh2xyxy() undoes it; a lost track that times out stays one more frame in the lost list
 * (self.removed_stracks is extended after it is subtracted, :296-298) and can be re-found in that frame; BaseTrack._count is a
 * class-level counter (per tracker handle here). dtype trail: STrack._tlwh is float32, so boxes / IoU are float32 arithmetic; a
 * track's mean stays float32 until its first predict or update, and noise terms taken from a float32 mean are float32 products.
 */
#include "orc.h"
#include "lapack_order.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { BT_NEW = 0, BT_TRACKED = 1, BT_LOST = 2, BT_REMOVED = 3 };
static const double WP = 1. / 20, WV = 1. / 160;

typedef struct {
    double mean[8], cov[64];
    int f32;                      /* mean is still the float32 array initiate() made */
    int is_activated, state, tracklet_len, frame_id, start_frame, in_removed;
    int64_t track_id;
    double score, cls, tracklab_id;
} btrk;

typedef struct { float tlwh[4]; double score, cls, tracklab_id; } bdet;

struct orc_bytetrack {
    double track_thresh, match_thresh, det_thresh;
    int max_time_lost, frame_id;
    btrk *T; int nT, capT;        /* all live track objects; lists hold indices */
    int *tracked, n_tracked, *lost, n_lost, cap_list;
    int64_t count;                /* BaseTrack._count */
};

orc_bytetrack *orc_bytetrack_create(double track_thresh, double match_thresh, int track_buffer, double frame_rate)
{
    orc_bytetrack *B = calloc(1, sizeof(*B));
    B->track_thresh = track_thresh; B->match_thresh = match_thresh; B->det_thresh = track_thresh + 0.1;
    B->max_time_lost = (int)(frame_rate / 30.0 * track_buffer);
    return B;
}
void orc_bytetrack_destroy(orc_bytetrack *B) { if (B) { free(B->T); free(B->tracked); free(B->lost); free(B); } }

/* ---- kalman_filter.py ---- */
static void kf_initiate(const float *m, btrk *k)        /* :55-87: float32 measurement -> float32 mean, std list with python floats -> float64 */
{
    const float sp = (float)(2 * WP) * m[3], sv = (float)(10 * WV) * m[3];
    const double std[8] = {sp, sp, 1e-2, sp, sv, sv, 1e-5, sv};
    memset(k->cov, 0, sizeof(k->cov));
    for (int i = 0; i < 4; ++i) { k->mean[i] = m[i]; k->mean[4 + i] = 0; }
    for (int i = 0; i < 8; ++i) k->cov[i * 9] = std[i] * std[i];
    k->f32 = 1;
}
/* multi_predict (:155-193) for one track; all_f32 = every mean of the pool is float32 (the stacked array stays float32) */
static void kf_predict(btrk *k, int all_f32)
{
    double q[8];
    if (all_f32) {
        const float h = (float)k->mean[3];
        const float sp = (float)WP * h, sv = (float)WV * h, a = (float)1e-2 * 1.0f, b = (float)1e-5 * 1.0f;
        const float std[8] = {sp, sp, a, sp, sv, sv, b, sv};
        for (int i = 0; i < 8; ++i) { const float s = std[i] * std[i]; q[i] = s; }
    } else {
        const double h = k->mean[3];
        const double std[8] = {WP * h, WP * h, 1e-2, WP * h, WV * h, WV * h, 1e-5, WV * h};
        for (int i = 0; i < 8; ++i) q[i] = std[i] * std[i];
    }
    double t[64], *cov = k->cov;
    /* left = F cov (rows 0..3 += rows 4..7), then left F^T (cols 0..3 += cols 4..7) */
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) t[i * 8 + j] = i < 4 ? cov[i * 8 + j] + cov[(i + 4) * 8 + j] : cov[i * 8 + j];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) cov[i * 8 + j] = j < 4 ? t[i * 8 + j] + t[i * 8 + j + 4] : t[i * 8 + j];
    for (int i = 0; i < 8; ++i) cov[i * 9] += q[i];
    for (int i = 0; i < 4; ++i) k->mean[i] = k->mean[i] + k->mean[i + 4];
    k->f32 = 0;
}
static void kf_update(btrk *k, const float *z32)         /* :195-224 with project :126-153 */
{
    double *mean = k->mean, *cov = k->cov;
    double sp;
    if (k->f32) { const float s = (float)WP * (float)mean[3]; sp = s; } else sp = WP * mean[3];
    const double std[4] = {sp, sp, 1e-1, sp};
    double pm[4], S[16];
    for (int i = 0; i < 4; ++i) { pm[i] = mean[i]; for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j] + (i == j ? std[i] * std[i] : 0.0); }
    const double z[4] = {(double)z32[0], (double)z32[1], (double)z32[2], (double)z32[3]};
    lo_kf8_update(mean, cov, z, pm, S);          /* cho_factor / cho_solve / np.dot / multi_dot in the libraries' operation order (lapack_order.h) */
    k->f32 = 0;
}

/* ---- boxes ---- */
static void trk_tlbr32(const btrk *k, float *o)          /* STrack.tlwh / tlbr (:97-117) in the mean's dtype, then float32 (matching.py:62-63) */
{
    if (k->f32) {
        float r0 = (float)k->mean[0], r1 = (float)k->mean[1], r2 = (float)k->mean[2], r3 = (float)k->mean[3];
        r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = r0; o[1] = r1; o[2] = r2 + r0; o[3] = r3 + r1;
    } else {
        double r0 = k->mean[0], r1 = k->mean[1], r2 = k->mean[2], r3 = k->mean[3];
        r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = (float)r0; o[1] = (float)r1; o[2] = (float)(r2 + r0); o[3] = (float)(r3 + r1);
    }
}
static void det_tlbr32(const bdet *d, float *o) { o[0] = d->tlwh[0]; o[1] = d->tlwh[1]; o[2] = d->tlwh[2] + d->tlwh[0]; o[3] = d->tlwh[3] + d->tlwh[1]; }
static void det_xyah32(const bdet *d, float *z) { z[0] = d->tlwh[0] + d->tlwh[2] / 2; z[1] = d->tlwh[1] + d->tlwh[3] / 2; z[2] = d->tlwh[2] / d->tlwh[3]; z[3] = d->tlwh[3]; }

/* matching.py:181-217 bbox_ious on float32 rows, result float32 */
static float bbox_iou32(const float *b, const float *q)
{
    const float box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    const float iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
    if (iw > 0) {
        const float ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
        if (ih > 0) {
            const float uaf = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
            const double ua = (double)uaf;
            return (float)((double)(iw * ih) / ua);
        }
    }
    return 0.f;
}

static int list_has(const int *l, int n, int v) { for (int i = 0; i < n; ++i) if (l[i] == v) return 1; return 0; }

/* linear_assignment (matching.py:37-48) on cost (nr, nc): matches ascending in row; unmatched rows / cols ascending */
static void assign(const double *cost, int nr, int nc, double thresh, int *m_r, int *m_c, int *nm, int *u_r, int *n_ur, int *u_c, int *n_uc)
{
    *nm = 0; *n_ur = 0; *n_uc = 0;
    if (nr == 0 || nc == 0) {
        for (int i = 0; i < nr; ++i) u_r[(*n_ur)++] = i;
        for (int j = 0; j < nc; ++j) u_c[(*n_uc)++] = j;
        return;
    }
    int32_t *x = malloc(sizeof(int32_t) * nr), *y = malloc(sizeof(int32_t) * nc);
    orc_lapjv_limit(cost, nr, nc, thresh, x, y);
    for (int i = 0; i < nr; ++i) { if (x[i] >= 0) { m_r[*nm] = i; m_c[*nm] = x[i]; (*nm)++; } else u_r[(*n_ur)++] = i; }
    for (int j = 0; j < nc; ++j) if (y[j] < 0) u_c[(*n_uc)++] = j;
    free(x); free(y);
}

static void trk_update(btrk *k, const bdet *d, int frame_id, int reactivate)
{
    float z[4];
    det_xyah32(d, z);
    kf_update(k, z);
    if (reactivate) { k->tracklet_len = 0; k->cls = d->cls; }       /* re_activate :60-74 */
    else k->tracklet_len++;                                        /* update :76-94 */
    k->frame_id = frame_id; k->state = BT_TRACKED; k->is_activated = 1;
    k->score = d->score; k->tracklab_id = d->tracklab_id;
}

int orc_bytetrack_update(orc_bytetrack *B, const double *dets, int N, double *rows_out, int out_cap)
{
    B->frame_id++;
    const int fid = B->frame_id;
    /* ---- split detections (:176-195); STrack(xywh, score, cls, id) keeps float32 "tlwh" = (cx, cy, w, h) ---- */
    bdet *hi = malloc(sizeof(bdet) * (N + 1)), *lo = malloc(sizeof(bdet) * (N + 1));
    int nhi = 0, nlo = 0;
    for (int i = 0; i < N; ++i) {
        const double *d = dets + 7 * (size_t)i;
        bdet b;
        b.tlwh[0] = (float)((d[0] + d[2]) / 2); b.tlwh[1] = (float)((d[1] + d[3]) / 2); b.tlwh[2] = (float)(d[2] - d[0]); b.tlwh[3] = (float)(d[3] - d[1]);
        b.score = d[4]; b.cls = d[5]; b.tracklab_id = d[6];
        if (d[4] > B->track_thresh) hi[nhi++] = b;
        else if (d[4] > 0.1 && d[4] < B->track_thresh) lo[nlo++] = b;
    }
    const int cap = B->n_tracked + B->n_lost + N + 4;
    int *unconf = malloc(sizeof(int) * cap), *pool = malloc(sizeof(int) * cap), n_unconf = 0, n_pool = 0;
    for (int i = 0; i < B->n_tracked; ++i) { const int t = B->tracked[i]; if (!B->T[t].is_activated) unconf[n_unconf++] = t; else pool[n_pool++] = t; }
    for (int i = 0; i < B->n_lost; ++i) if (!list_has(pool, n_pool, B->lost[i])) pool[n_pool++] = B->lost[i];       /* joint_stracks */
    /* ---- multi_predict (:26-40) ---- */
    if (n_pool > 0) {
        int all_f32 = 1;
        for (int i = 0; i < n_pool; ++i) all_f32 &= B->T[pool[i]].f32;
        for (int i = 0; i < n_pool; ++i) { btrk *k = &B->T[pool[i]]; if (k->state != BT_TRACKED) k->mean[7] = 0; kf_predict(k, all_f32); }
    }
    int *activated = malloc(sizeof(int) * cap), *refind = malloc(sizeof(int) * cap), *newlost = malloc(sizeof(int) * cap), *removed = malloc(sizeof(int) * cap);
    int n_act = 0, n_ref = 0, n_newlost = 0, n_removed = 0;
    int *m_r = malloc(sizeof(int) * cap), *m_c = malloc(sizeof(int) * cap), *u_r = malloc(sizeof(int) * cap), *u_c = malloc(sizeof(int) * cap);
    int nm, n_ur, n_uc;
    float *tb = malloc(sizeof(float) * 4 * (size_t)cap), *db = malloc(sizeof(float) * 4 * (size_t)(N + 1));
    double *cost = malloc(sizeof(double) * (size_t)cap * (N + 1));
    /* ---- step 2: first association, high-score detections (:214-227) ---- */
    for (int i = 0; i < n_pool; ++i) trk_tlbr32(&B->T[pool[i]], tb + 4 * i);
    for (int j = 0; j < nhi; ++j) det_tlbr32(&hi[j], db + 4 * j);
    for (int i = 0; i < n_pool; ++i)
        for (int j = 0; j < nhi; ++j) {
            const float c32 = 1 - bbox_iou32(tb + 4 * i, db + 4 * j);       /* iou_distance: float32 */
            const float sim = 1 - c32;                                      /* fuse_score: 1 - cost (float32), * score (float64) */
            cost[(size_t)i * nhi + j] = 1 - (double)sim * hi[j].score;
        }
    assign(cost, n_pool, nhi, B->match_thresh, m_r, m_c, &nm, u_r, &n_ur, u_c, &n_uc);
    for (int q = 0; q < nm; ++q) {
        btrk *k = &B->T[pool[m_r[q]]];
        if (k->state == BT_TRACKED) { trk_update(k, &hi[m_c[q]], fid, 0); activated[n_act++] = pool[m_r[q]]; }
        else { trk_update(k, &hi[m_c[q]], fid, 1); refind[n_ref++] = pool[m_r[q]]; }
    }
    int *u_det1 = malloc(sizeof(int) * (N + 1)); const int n_udet1 = n_uc;
    memcpy(u_det1, u_c, sizeof(int) * n_uc);
    /* ---- step 3: second association, low-score detections (:229-250) ---- */
    int *rtr = malloc(sizeof(int) * cap), n_rtr = 0;
    for (int q = 0; q < n_ur; ++q) if (B->T[pool[u_r[q]]].state == BT_TRACKED) rtr[n_rtr++] = pool[u_r[q]];
    for (int i = 0; i < n_rtr; ++i) trk_tlbr32(&B->T[rtr[i]], tb + 4 * i);
    for (int j = 0; j < nlo; ++j) det_tlbr32(&lo[j], db + 4 * j);
    for (int i = 0; i < n_rtr; ++i) for (int j = 0; j < nlo; ++j) cost[(size_t)i * nlo + j] = (double)(float)(1 - bbox_iou32(tb + 4 * i, db + 4 * j));
    assign(cost, n_rtr, nlo, 0.5, m_r, m_c, &nm, u_r, &n_ur, u_c, &n_uc);
    for (int q = 0; q < nm; ++q) {
        btrk *k = &B->T[rtr[m_r[q]]];
        if (k->state == BT_TRACKED) { trk_update(k, &lo[m_c[q]], fid, 0); activated[n_act++] = rtr[m_r[q]]; }
        else { trk_update(k, &lo[m_c[q]], fid, 1); refind[n_ref++] = rtr[m_r[q]]; }
    }
    for (int q = 0; q < n_ur; ++q) { btrk *k = &B->T[rtr[u_r[q]]]; if (k->state != BT_LOST) { k->state = BT_LOST; newlost[n_newlost++] = rtr[u_r[q]]; } }
    /* ---- unconfirmed tracks against the remaining high-score detections (:252-263) ---- */
    bdet *rem = malloc(sizeof(bdet) * (N + 1));
    for (int j = 0; j < n_udet1; ++j) rem[j] = hi[u_det1[j]];
    for (int i = 0; i < n_unconf; ++i) trk_tlbr32(&B->T[unconf[i]], tb + 4 * i);
    for (int j = 0; j < n_udet1; ++j) det_tlbr32(&rem[j], db + 4 * j);
    for (int i = 0; i < n_unconf; ++i)
        for (int j = 0; j < n_udet1; ++j) {
            const float c32 = 1 - bbox_iou32(tb + 4 * i, db + 4 * j);
            const float sim = 1 - c32;
            cost[(size_t)i * n_udet1 + j] = 1 - (double)sim * rem[j].score;
        }
    assign(cost, n_unconf, n_udet1, 0.7, m_r, m_c, &nm, u_r, &n_ur, u_c, &n_uc);
    for (int q = 0; q < nm; ++q) { trk_update(&B->T[unconf[m_r[q]]], &rem[m_c[q]], fid, 0); activated[n_act++] = unconf[m_r[q]]; }
    for (int q = 0; q < n_ur; ++q) { B->T[unconf[u_r[q]]].state = BT_REMOVED; removed[n_removed++] = unconf[u_r[q]]; }
    /* ---- step 4: new tracks (:265-271) ---- */
    for (int q = 0; q < n_uc; ++q) {
        const bdet *d = &rem[u_c[q]];
        if (d->score < B->det_thresh) continue;
        if (B->nT == B->capT) { B->capT = B->capT ? 2 * B->capT : 256; B->T = realloc(B->T, sizeof(btrk) * B->capT); }
        btrk *k = &B->T[B->nT];
        memset(k, 0, sizeof(*k));
        float z[4];
        det_xyah32(d, z);
        k->track_id = ++B->count;
        kf_initiate(z, k);
        k->tracklet_len = 0; k->state = BT_TRACKED; k->is_activated = fid == 1; k->frame_id = fid; k->start_frame = fid;
        k->score = d->score; k->cls = d->cls; k->tracklab_id = d->tracklab_id;
        activated[n_act++] = B->nT++;
    }
    /* ---- step 5: lost tracks that timed out (:273-277), then the list bookkeeping (:279-290) ---- */
    for (int i = 0; i < B->n_lost; ++i) { btrk *k = &B->T[B->lost[i]]; if (fid - k->frame_id > B->max_time_lost) { k->state = BT_REMOVED; removed[n_removed++] = B->lost[i]; } }
    const int ncap = B->n_tracked + n_act + n_ref + B->n_lost + n_newlost + 4;
    int *ntr = malloc(sizeof(int) * ncap), nn = 0, *nlost = malloc(sizeof(int) * ncap), nl = 0;
    for (int i = 0; i < B->n_tracked; ++i) if (B->T[B->tracked[i]].state == BT_TRACKED) ntr[nn++] = B->tracked[i];
    for (int i = 0; i < n_act; ++i) if (!list_has(ntr, nn, activated[i])) ntr[nn++] = activated[i];
    for (int i = 0; i < n_ref; ++i) if (!list_has(ntr, nn, refind[i])) ntr[nn++] = refind[i];
    for (int i = 0; i < B->n_lost; ++i) if (!list_has(ntr, nn, B->lost[i])) nlost[nl++] = B->lost[i];         /* sub_stracks(lost, tracked) */
    for (int i = 0; i < n_newlost; ++i) nlost[nl++] = newlost[i];
    { int k2 = 0; for (int i = 0; i < nl; ++i) if (!B->T[nlost[i]].in_removed) nlost[k2++] = nlost[i]; nl = k2; }   /* sub_stracks(lost, self.removed) BEFORE extend */
    for (int i = 0; i < n_removed; ++i) B->T[removed[i]].in_removed = 1;
    /* remove_duplicate_stracks (:346-361) */
    {
        char *dupa = calloc(nn + 1, 1), *dupb = calloc(nl + 1, 1);
        float *ta = malloc(sizeof(float) * 4 * (size_t)(nn + 1)), *tl = malloc(sizeof(float) * 4 * (size_t)(nl + 1));
        for (int i = 0; i < nn; ++i) trk_tlbr32(&B->T[ntr[i]], ta + 4 * i);
        for (int j = 0; j < nl; ++j) trk_tlbr32(&B->T[nlost[j]], tl + 4 * j);
        for (int p = 0; p < nn; ++p)
            for (int q = 0; q < nl; ++q) {
                const float pd = 1 - bbox_iou32(ta + 4 * p, tl + 4 * q);
                if (pd < (float)0.15) {
                    const int timep = B->T[ntr[p]].frame_id - B->T[ntr[p]].start_frame, timeq = B->T[nlost[q]].frame_id - B->T[nlost[q]].start_frame;
                    if (timep > timeq) dupb[q] = 1; else dupa[p] = 1;
                }
            }
        int k2 = 0; for (int i = 0; i < nn; ++i) if (!dupa[i]) ntr[k2++] = ntr[i]; nn = k2;
        k2 = 0; for (int j = 0; j < nl; ++j) if (!dupb[j]) nlost[k2++] = nlost[j]; nl = k2;
        free(dupa); free(dupb); free(ta); free(tl);
    }
    free(B->tracked); free(B->lost);
    B->tracked = ntr; B->n_tracked = nn; B->lost = nlost; B->n_lost = nl;
    /* ---- outputs (:292-309): activated tracked tracks; xywh2xyxy of the "tlwh" in the mean's dtype ---- */
    int n_out = 0;
    for (int i = 0; i < nn && n_out < out_cap; ++i) {
        const btrk *k = &B->T[ntr[i]];
        if (!k->is_activated) continue;
        double *o = rows_out + 8 * (size_t)n_out++;
        if (k->f32) {
            float r0 = (float)k->mean[0], r1 = (float)k->mean[1], r2 = (float)k->mean[2], r3 = (float)k->mean[3];
            r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
            const float hw = r2 / 2, hh = r3 / 2;
            o[0] = r0 - hw; o[1] = r1 - hh; o[2] = r0 + hw; o[3] = r1 + hh;
        } else {
            double r0 = k->mean[0], r1 = k->mean[1], r2 = k->mean[2], r3 = k->mean[3];
            r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
            const double hw = r2 / 2, hh = r3 / 2;
            o[0] = r0 - hw; o[1] = r1 - hh; o[2] = r0 + hw; o[3] = r1 + hh;
        }
        o[4] = (double)k->track_id; o[5] = k->cls; o[6] = k->score; o[7] = k->tracklab_id;
    }
    free(hi); free(lo); free(unconf); free(pool); free(activated); free(refind); free(newlost); free(removed);
    free(m_r); free(m_c); free(u_r); free(u_c); free(tb); free(db); free(cost); free(u_det1); free(rtr); free(rem);
    return n_out;
}

int orc_bytetrack_list_len(const orc_bytetrack *B, int which) { return which ? B->n_lost : B->n_tracked; }
/* debug: list `which` (0 tracked, 1 lost): ids, mean (.,8), cov (.,64), state5 (.,5) [state, is_activated, frame_id, start_frame, tracklet_len] */
int orc_bytetrack_list(const orc_bytetrack *B, int which, int64_t *ids, double *mean, double *cov, int64_t *state5, int cap)
{
    const int *l = which ? B->lost : B->tracked; int n = which ? B->n_lost : B->n_tracked;
    if (n > cap) n = cap;
    for (int i = 0; i < n; ++i) {
        const btrk *k = &B->T[l[i]];
        ids[i] = k->track_id; memcpy(mean + 8 * i, k->mean, sizeof(k->mean)); memcpy(cov + 64 * i, k->cov, sizeof(k->cov));
        state5[5 * i] = k->state; state5[5 * i + 1] = k->is_activated; state5[5 * i + 2] = k->frame_id; state5[5 * i + 3] = k->start_frame; state5[5 * i + 4] = k->tracklet_len;
    }
    return n;
}
