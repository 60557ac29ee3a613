/*
 * oracle/src/botsort.c -- CPU ORACLE (test infrastructure, NOT product code).
 * BoT-SORT: plugins/track/bot_sort/{bot_sort.py, matching.py, kalman_filter.py, basetrack.py} restated in C with cmc_method
 * "none" (gmc.py:75-78 returns the identity; the other camera-motion estimators are cv2: SURVEY 8f-3, out of scope).
 *
 *   BoTSORT.update               bot_sort.py:272-420  (score split, first association on the ReID embedding gated by the KF,
 *                                second association on IoU, unconfirmed tracks on min(IoU x score, embedding / 2), new tracks,
 *                                time-outs, tracked / lost bookkeeping, duplicate removal, output rows)
 *   STrack                       bot_sort.py:12-232   (update_features incl. the curr_feat / smooth_feat aliasing of a fresh
 *                                detection, update_cls vote, multi_predict, multi_gmc, activate / re_activate / update)
 *   embedding_distance, fuse_motion, iou_distance, fuse_score, linear_assignment   matching.py:37-48, :86-104, :127-171, :196-204
 *   KalmanFilter (x, y, w, h)    kalman_filter.py:55-270
 * scipy.spatial.distance.cdist(..., "cosine") is third-party (scipy 1.15.3): 1 - u.v / (|u| |v|) in float64, clipped to |cos| <= 1.
 * dtype trail: STrack._tlwh and the features are float32; a new track's mean AND covariance are float32 (every std entry is a
 * float32 product) until multi_predict / multi_gmc / update turn them into float64 arrays.
 */
#include "orc.h"
#include "lapack_order.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { BS_NEW = 0, BS_TRACKED = 1, BS_LOST = 2, BS_LONGLOST = 3, BS_REMOVED = 4 };      /* basetrack.py:5-10 */
static const double WP = 1. / 20, WV = 1. / 160;
#define MAXCLS 16

typedef struct {
    double mean[8], cov[64];
    int f32;
    int is_activated, state, tracklet_len, frame_id, start_frame, in_removed;
    int64_t track_id;
    double score, cls, tracklab_id;
    double hist_cls[MAXCLS], hist_freq[MAXCLS]; int nhist;
    float *smooth;                /* smooth_feat (D) */
} strk;

typedef struct { float tlwh[4]; double score, cls, tracklab_id; float *feat; } sdet;     /* feat = curr_feat or NULL */

struct orc_botsort {
    double track_high, new_track, match_thresh, proximity, appearance, lambda_;
    int max_time_lost, frame_id, D;
    strk *T; int nT, capT;
    int *tracked, n_tracked, *lost, n_lost;
    int64_t count;
};

orc_botsort *orc_botsort_create(double track_high_thresh, double new_track_thresh, int track_buffer, double match_thresh,
                                double proximity_thresh, double appearance_thresh, double frame_rate, double lambda_, int D)
{
    orc_botsort *B = calloc(1, sizeof(*B));
    B->track_high = track_high_thresh; B->new_track = new_track_thresh; B->match_thresh = match_thresh; B->proximity = proximity_thresh;
    B->appearance = appearance_thresh; B->lambda_ = lambda_; B->D = D;
    B->max_time_lost = (int)(frame_rate / 30.0 * track_buffer);
    return B;
}
void orc_botsort_destroy(orc_botsort *B)
{
    if (!B) return;
    for (int i = 0; i < B->nT; ++i) free(B->T[i].smooth);
    free(B->T); free(B->tracked); free(B->lost); free(B);
}

static float norm_f32(const float *x, int D) { float s = 0.f; for (int d = 0; d < D; ++d) s += x[d] * x[d]; return sqrtf(s); }

/* update_cls (bot_sort.py:48-66) */
static void update_cls(strk *k, double cls, double score)
{
    if (k->nhist > 0) {
        double max_freq = 0; int found = 0;
        for (int i = 0; i < k->nhist; ++i) {
            if (cls == k->hist_cls[i]) { k->hist_freq[i] += score; found = 1; }
            if (k->hist_freq[i] > max_freq) { max_freq = k->hist_freq[i]; k->cls = k->hist_cls[i]; }
        }
        if (!found) { if (k->nhist < MAXCLS) { k->hist_cls[k->nhist] = cls; k->hist_freq[k->nhist] = score; k->nhist++; } k->cls = cls; }
    } else { k->hist_cls[0] = cls; k->hist_freq[0] = score; k->nhist = 1; k->cls = cls; }
}

/* ---- kalman_filter.py (xywh) ---- */
static void kf_initiate(const float *m, strk *k)        /* :55-88, everything float32 */
{
    const float w = m[2], h = m[3];
    const float std[8] = {(float)(2 * WP) * w, (float)(2 * WP) * h, (float)(2 * WP) * w, (float)(2 * WP) * h,
                          (float)(10 * WV) * w, (float)(10 * WV) * h, (float)(10 * WV) * w, (float)(10 * WV) * h};
    memset(k->cov, 0, sizeof(k->cov));
    for (int i = 0; i < 4; ++i) { k->mean[i] = m[i]; k->mean[4 + i] = 0; }
    for (int i = 0; i < 8; ++i) { const float s = std[i] * std[i]; k->cov[i * 9] = s; }
    k->f32 = 1;
}
static void kf_predict(strk *k, int all_f32)            /* multi_predict :155-193 */
{
    double q[8];
    if (all_f32) {
        const float w = (float)k->mean[2], h = (float)k->mean[3];
        const float std[8] = {(float)WP * w, (float)WP * h, (float)WP * w, (float)WP * h, (float)WV * w, (float)WV * h, (float)WV * w, (float)WV * h};
        for (int i = 0; i < 8; ++i) { const float s = std[i] * std[i]; q[i] = s; }
    } else {
        const double w = k->mean[2], h = k->mean[3];
        const double std[8] = {WP * w, WP * h, WP * w, WP * h, WV * w, WV * h, WV * w, WV * h};
        for (int i = 0; i < 8; ++i) q[i] = std[i] * std[i];
    }
    double t[64], *cov = k->cov;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) t[i * 8 + j] = i < 4 ? cov[i * 8 + j] + cov[(i + 4) * 8 + j] : cov[i * 8 + j];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) cov[i * 8 + j] = j < 4 ? t[i * 8 + j] + t[i * 8 + j + 4] : t[i * 8 + j];
    for (int i = 0; i < 8; ++i) cov[i * 9] += q[i];
    for (int i = 0; i < 4; ++i) k->mean[i] = k->mean[i] + k->mean[i + 4];
    k->f32 = 0;
}
static void kf_project(const strk *k, double *pm, double *S)             /* :126-153 */
{
    double d[4];
    if (k->f32) {                 /* std is a list of float32 products; np.square keeps float32 */
        const float a = (float)WP * (float)k->mean[2], b = (float)WP * (float)k->mean[3];
        const float a2 = a * a, b2 = b * b;
        d[0] = a2; d[1] = b2; d[2] = a2; d[3] = b2;
    } else {
        const double a = WP * k->mean[2], b = WP * k->mean[3];
        d[0] = a * a; d[1] = b * b; d[2] = a * a; d[3] = b * b;
    }
    for (int i = 0; i < 4; ++i) { pm[i] = k->mean[i]; for (int j = 0; j < 4; ++j) S[i * 4 + j] = k->cov[i * 8 + j] + (i == j ? d[i] : 0.0); }
}
static void kf_update(strk *k, const float *z32)                         /* :195-224; library operation order: lapack_order.h */
{
    double pm[4], S[16];
    kf_project(k, pm, S);
    const double z[4] = {(double)z32[0], (double)z32[1], (double)z32[2], (double)z32[3]};
    lo_kf8_update(k->mean, k->cov, z, pm, S);
    k->f32 = 0;
}
/* gating_distance(..., metric="maha") :226-270 against float32 xywh measurements */
static void kf_gating(const strk *k, const float *meas, int n, double *out)
{
    double pm[4], S[16];
    kf_project(k, pm, S);
    double *m64 = malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < 4 * n; ++i) m64[i] = (double)meas[i];
    lo_kf8_gating(pm, S, 4, m64, n, out);
    free(m64);
}

/* ---- boxes ---- */
static void trk_tlbr32(const strk *k, float *o)          /* STrack.tlwh / tlbr (bot_sort.py:134-153) in the mean's dtype, then float32 */
{
    if (k->f32) {
        float r0 = (float)k->mean[0], r1 = (float)k->mean[1]; const float r2 = (float)k->mean[2], r3 = (float)k->mean[3];
        r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = r0; o[1] = r1; o[2] = r2 + r0; o[3] = r3 + r1;
    } else {
        double r0 = k->mean[0], r1 = k->mean[1]; const double r2 = k->mean[2], r3 = k->mean[3];
        r0 -= r2 / 2; r1 -= r3 / 2;
        o[0] = (float)r0; o[1] = (float)r1; o[2] = (float)(r2 + r0); o[3] = (float)(r3 + r1);
    }
}
static void det_tlbr32(const sdet *d, float *o) { o[0] = d->tlwh[0]; o[1] = d->tlwh[1]; o[2] = d->tlwh[2] + d->tlwh[0]; o[3] = d->tlwh[3] + d->tlwh[1]; }
static void det_xywh32(const sdet *d, float *z) { z[0] = d->tlwh[0] + d->tlwh[2] / 2; z[1] = d->tlwh[1] + d->tlwh[3] / 2; z[2] = d->tlwh[2]; z[3] = d->tlwh[3]; }

static float bbox_iou32(const float *b, const float *q)       /* matching.py bbox_ious */
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
/* max(0, cdist(u, v, "cosine")) of two float32 vectors (matching.py:141) */
static double cosine_dist(const float *u, const float *v, int D)
{
    double uv = 0, uu = 0, vv = 0;
    for (int d = 0; d < D; ++d) { uv += (double)u[d] * (double)v[d]; uu += (double)u[d] * (double)u[d]; vv += (double)v[d] * (double)v[d]; }
    double c = uv / (sqrt(uu) * sqrt(vv));
    if (fabs(c) > 1.) c = c < 0 ? -1. : 1.;
    const double r = 1. - c;
    return r > 0.0 ? r : 0.0;
}

static int list_has(const int *l, int n, int v) { for (int i = 0; i < n; ++i) if (l[i] == v) return 1; return 0; }

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

/* STrack.update / re_activate (bot_sort.py:96-131) */
static void trk_update(orc_botsort *B, strk *k, sdet *d, int frame_id, int reactivate)
{
    float z[4];
    det_xywh32(d, z);
    kf_update(k, z);
    if (d->feat) {                                              /* update_features(new_track.curr_feat), :37-46 */
        const int D = B->D;
        const float n1 = norm_f32(d->feat, D);
        for (int e = 0; e < D; ++e) d->feat[e] /= n1;           /* feat /= norm: in place on the detection's array */
        const float a = (float)0.9, b1 = (float)(1 - 0.9);
        for (int e = 0; e < D; ++e) k->smooth[e] = a * k->smooth[e] + b1 * d->feat[e];
        const float n2 = norm_f32(k->smooth, D);
        for (int e = 0; e < D; ++e) k->smooth[e] /= n2;
    }
    if (reactivate) k->tracklet_len = 0; else k->tracklet_len++;
    k->frame_id = frame_id; k->state = BS_TRACKED; k->is_activated = 1;
    k->score = d->score;
    update_cls(k, d->cls, d->score);
    k->tracklab_id = d->tracklab_id;
}

/* STrack.multi_gmc with a general (2,3) warp H (bot_sort.py:93-109): mean = kron(I4, R) mean, mean[:2] += t, cov = kron(I4, R) cov kron(I4, R)^T */
static void gmc_apply(strk *k, const double *H)
{
    const double R[4] = {H[0], H[1], H[3], H[4]}, t[2] = {H[2], H[5]};
    double m[8], t1[64], c2[64];
    for (int b = 0; b < 4; ++b) {
        m[2 * b] = R[0] * k->mean[2 * b] + R[1] * k->mean[2 * b + 1];
        m[2 * b + 1] = R[2] * k->mean[2 * b] + R[3] * k->mean[2 * b + 1];
    }
    m[0] += t[0]; m[1] += t[1];
    /* R8x8.dot(cov).dot(R8x8.T): two dgemm calls, every element an fma chain over k ascending from 0 (lapack_order.h); the zero blocks of
     * kron(I4, R) leave fma(r_b, c_b, r_a * c_a) with a < b */
    for (int i = 0; i < 8; ++i)                 /* t1 = R8 cov: row i mixes rows 2*(i/2), 2*(i/2)+1 */
        for (int j = 0; j < 8; ++j) { const int r0 = i & ~1; t1[i * 8 + j] = fma(R[(i & 1) * 2 + 1], k->cov[(r0 + 1) * 8 + j], R[(i & 1) * 2] * k->cov[r0 * 8 + j]); }
    for (int i = 0; i < 8; ++i)                 /* c2 = t1 R8^T: column j mixes columns 2*(j/2), 2*(j/2)+1 */
        for (int j = 0; j < 8; ++j) { const int c0 = j & ~1; c2[i * 8 + j] = fma(t1[i * 8 + c0 + 1], R[(j & 1) * 2 + 1], t1[i * 8 + c0] * R[(j & 1) * 2]); }
    memcpy(k->mean, m, sizeof(m)); memcpy(k->cov, c2, sizeof(c2));
}

int orc_botsort_update_gmc(orc_botsort *B, const double *dets, const float *feats, int N, const double *warp6, double *rows_out, int out_cap);
int orc_botsort_update(orc_botsort *B, const double *dets, const float *feats, int N, double *rows_out, int out_cap)
{ return orc_botsort_update_gmc(B, dets, feats, N, NULL, rows_out, out_cap); }

/* warp6: what GMC.apply returned for this frame ((2,3) float64, bot_sort.py:341), NULL = cmc_method "none" (the identity) */
int orc_botsort_update_gmc(orc_botsort *B, const double *dets, const float *feats, int N, const double *warp6, double *rows_out, int out_cap)
{
    const int D = B->D;
    B->frame_id++;
    const int fid = B->frame_id;
    sdet *hi = malloc(sizeof(sdet) * (N + 1)), *lo = malloc(sizeof(sdet) * (N + 1));
    float *fbuf = malloc(sizeof(float) * (size_t)(N + 1) * D);
    int nhi = 0, nlo = 0;
    for (int i = 0; i < N; ++i) {
        const double *d = dets + 7 * (size_t)i;
        const double cx = (d[0] + d[2]) / 2, cy = (d[1] + d[3]) / 2, w = d[2] - d[0], h = d[3] - d[1];
        sdet b;
        b.score = d[4]; b.cls = d[5]; b.tracklab_id = d[6]; b.feat = NULL;
        if (d[4] > B->track_high) {
            b.tlwh[0] = (float)cx; b.tlwh[1] = (float)cy; b.tlwh[2] = (float)w; b.tlwh[3] = (float)h;      /* STrack(xywh, ...) */
            /* update_features on a fresh STrack: curr_feat and smooth_feat are the SAME array, normalised twice in place */
            b.feat = fbuf + (size_t)nhi * D;
            memcpy(b.feat, feats + (size_t)i * D, sizeof(float) * D);
            const float n1 = norm_f32(b.feat, D);
            for (int e = 0; e < D; ++e) b.feat[e] /= n1;
            const float n2 = norm_f32(b.feat, D);
            for (int e = 0; e < D; ++e) b.feat[e] /= n2;
            hi[nhi++] = b;
        } else if (d[4] > 0.1 && d[4] < B->track_high) {
            /* STrack(STrack.tlbr_to_tlwh(xywh_row), ...): the xywh row is read as tlbr (bot_sort.py:339-340) */
            b.tlwh[0] = (float)cx; b.tlwh[1] = (float)cy; b.tlwh[2] = (float)(w - cx); b.tlwh[3] = (float)(h - cy);
            lo[nlo++] = b;
        }
    }
    const int cap = B->n_tracked + B->n_lost + N + 4;
    int *unconf = malloc(sizeof(int) * cap), *pool = malloc(sizeof(int) * cap), n_unconf = 0, n_pool = 0;
    for (int i = 0; i < B->n_tracked; ++i) { const int t = B->tracked[i]; if (!B->T[t].is_activated) unconf[n_unconf++] = t; else pool[n_pool++] = t; }
    for (int i = 0; i < B->n_lost; ++i) if (!list_has(pool, n_pool, B->lost[i])) pool[n_pool++] = B->lost[i];
    if (n_pool > 0) {                                           /* multi_predict :68-80 */
        int all_f32 = 1;
        for (int i = 0; i < n_pool; ++i) all_f32 &= B->T[pool[i]].f32;
        for (int i = 0; i < n_pool; ++i) { strk *k = &B->T[pool[i]]; if (k->state != BS_TRACKED) { k->mean[6] = 0; k->mean[7] = 0; } kf_predict(k, all_f32); }
    }
    /* multi_gmc (:93-109): the arrays become float64; with the identity warp the values are unchanged */
    for (int i = 0; i < n_pool; ++i) { B->T[pool[i]].f32 = 0; if (warp6) gmc_apply(&B->T[pool[i]], warp6); }
    for (int i = 0; i < n_unconf; ++i) { B->T[unconf[i]].f32 = 0; if (warp6) gmc_apply(&B->T[unconf[i]], warp6); }
    int *activated = malloc(sizeof(int) * cap), *refind = malloc(sizeof(int) * cap), *newlost = malloc(sizeof(int) * cap), *removed = malloc(sizeof(int) * cap);
    int n_act = 0, n_ref = 0, n_newlost = 0, n_removed = 0;
    int *m_r = malloc(sizeof(int) * cap), *m_c = malloc(sizeof(int) * cap), *u_r = malloc(sizeof(int) * cap), *u_c = malloc(sizeof(int) * cap);
    int nm, n_ur, n_uc;
    float *tb = malloc(sizeof(float) * 4 * (size_t)cap), *db = malloc(sizeof(float) * 4 * (size_t)(N + 1)), *dz = malloc(sizeof(float) * 4 * (size_t)(N + 1));
    double *cost = malloc(sizeof(double) * (size_t)cap * (N + 1)), *gd = malloc(sizeof(double) * (N + 1));
    /* ---- first association: embedding distance fused with the KF gate (:307-320) ---- */
    for (int j = 0; j < nhi; ++j) det_xywh32(&hi[j], dz + 4 * j);
    for (int i = 0; i < n_pool; ++i) {
        const strk *k = &B->T[pool[i]];
        kf_gating(k, dz, nhi, gd);
        for (int j = 0; j < nhi; ++j) {
            double c = cosine_dist(k->smooth, hi[j].feat, D);
            if (gd[j] > 9.4877) c = INFINITY;
            cost[(size_t)i * nhi + j] = B->lambda_ * c + (1 - B->lambda_) * gd[j];
        }
    }
    assign(cost, n_pool, nhi, B->match_thresh, m_r, m_c, &nm, u_r, &n_ur, u_c, &n_uc);
    for (int q = 0; q < nm; ++q) {
        strk *k = &B->T[pool[m_r[q]]];
        if (k->state == BS_TRACKED) { trk_update(B, k, &hi[m_c[q]], fid, 0); activated[n_act++] = pool[m_r[q]]; }
        else { trk_update(B, k, &hi[m_c[q]], fid, 1); refind[n_ref++] = pool[m_r[q]]; }
    }
    int *u_det1 = malloc(sizeof(int) * (N + 1)); const int n_udet1 = n_uc;
    memcpy(u_det1, u_c, sizeof(int) * n_uc);
    /* ---- second association: low-score detections by IoU (:322-345) ---- */
    int *rtr = malloc(sizeof(int) * cap), n_rtr = 0;
    for (int q = 0; q < n_ur; ++q) if (B->T[pool[u_r[q]]].state == BS_TRACKED) rtr[n_rtr++] = pool[u_r[q]];
    for (int i = 0; i < n_rtr; ++i) trk_tlbr32(&B->T[rtr[i]], tb + 4 * i);
    for (int j = 0; j < nlo; ++j) det_tlbr32(&lo[j], db + 4 * j);
    for (int i = 0; i < n_rtr; ++i) for (int j = 0; j < nlo; ++j) cost[(size_t)i * nlo + j] = (double)(float)(1 - bbox_iou32(tb + 4 * i, db + 4 * j));
    assign(cost, n_rtr, nlo, 0.5, m_r, m_c, &nm, u_r, &n_ur, u_c, &n_uc);
    for (int q = 0; q < nm; ++q) {
        strk *k = &B->T[rtr[m_r[q]]];
        if (k->state == BS_TRACKED) { trk_update(B, k, &lo[m_c[q]], fid, 0); activated[n_act++] = rtr[m_r[q]]; }
        else { trk_update(B, k, &lo[m_c[q]], fid, 1); refind[n_ref++] = rtr[m_r[q]]; }
    }
    for (int q = 0; q < n_ur; ++q) { strk *k = &B->T[rtr[u_r[q]]]; if (k->state != BS_LOST) { k->state = BS_LOST; newlost[n_newlost++] = rtr[u_r[q]]; } }
    /* ---- unconfirmed tracks: min(IoU distance fused with the score, embedding / 2 gated by both thresholds) (:347-365) ---- */
    sdet *rem = malloc(sizeof(sdet) * (N + 1));
    for (int j = 0; j < n_udet1; ++j) rem[j] = hi[u_det1[j]];
    for (int i = 0; i < n_unconf; ++i) trk_tlbr32(&B->T[unconf[i]], tb + 4 * i);
    for (int j = 0; j < n_udet1; ++j) det_tlbr32(&rem[j], db + 4 * j);
    for (int i = 0; i < n_unconf; ++i)
        for (int j = 0; j < n_udet1; ++j) {
            const float c32 = 1 - bbox_iou32(tb + 4 * i, db + 4 * j);
            const int far = c32 > (float)B->proximity;
            const float sim = 1 - c32;
            const double iou_d = 1 - (double)sim * rem[j].score;
            double emb = cosine_dist(B->T[unconf[i]].smooth, rem[j].feat, D) / 2.0;
            if (emb > B->appearance) emb = 1.0;
            if (far) emb = 1.0;
            cost[(size_t)i * n_udet1 + j] = iou_d < emb ? iou_d : emb;
        }
    assign(cost, n_unconf, n_udet1, 0.7, m_r, m_c, &nm, u_r, &n_ur, u_c, &n_uc);
    for (int q = 0; q < nm; ++q) { trk_update(B, &B->T[unconf[m_r[q]]], &rem[m_c[q]], fid, 0); activated[n_act++] = unconf[m_r[q]]; }
    for (int q = 0; q < n_ur; ++q) { B->T[unconf[u_r[q]]].state = BS_REMOVED; removed[n_removed++] = unconf[u_r[q]]; }
    /* ---- new tracks (:367-373) ---- */
    for (int q = 0; q < n_uc; ++q) {
        const sdet *d = &rem[u_c[q]];
        if (d->score < B->new_track) continue;
        if (B->nT == B->capT) { B->capT = B->capT ? 2 * B->capT : 256; B->T = realloc(B->T, sizeof(strk) * B->capT); }
        strk *k = &B->T[B->nT];
        memset(k, 0, sizeof(*k));
        float z[4];
        det_xywh32(d, z);                                       /* tlwh_to_xywh(self._tlwh) */
        k->track_id = ++B->count;
        kf_initiate(z, k);
        k->tracklet_len = 0; k->state = BS_TRACKED; k->is_activated = fid == 1; k->frame_id = fid; k->start_frame = fid;
        k->score = d->score; k->tracklab_id = d->tracklab_id;
        update_cls(k, d->cls, d->score);                        /* the detection's own one-entry history */
        k->smooth = malloc(sizeof(float) * D);
        memcpy(k->smooth, d->feat, sizeof(float) * D);
        activated[n_act++] = B->nT++;
    }
    /* ---- time-outs and list bookkeeping (:375-389), as in ByteTrack ---- */
    for (int i = 0; i < B->n_lost; ++i) { strk *k = &B->T[B->lost[i]]; if (fid - k->frame_id > B->max_time_lost) { k->state = BS_REMOVED; removed[n_removed++] = B->lost[i]; } }
    const int ncap = B->n_tracked + n_act + n_ref + B->n_lost + n_newlost + 4;
    int *ntr = malloc(sizeof(int) * ncap), nn = 0, *nlost = malloc(sizeof(int) * ncap), nl = 0;
    for (int i = 0; i < B->n_tracked; ++i) if (B->T[B->tracked[i]].state == BS_TRACKED) ntr[nn++] = B->tracked[i];
    for (int i = 0; i < n_act; ++i) if (!list_has(ntr, nn, activated[i])) ntr[nn++] = activated[i];
    for (int i = 0; i < n_ref; ++i) if (!list_has(ntr, nn, refind[i])) ntr[nn++] = refind[i];
    for (int i = 0; i < B->n_lost; ++i) if (!list_has(ntr, nn, B->lost[i])) nlost[nl++] = B->lost[i];
    for (int i = 0; i < n_newlost; ++i) nlost[nl++] = newlost[i];
    { int k2 = 0; for (int i = 0; i < nl; ++i) if (!B->T[nlost[i]].in_removed) nlost[k2++] = nlost[i]; nl = k2; }
    for (int i = 0; i < n_removed; ++i) B->T[removed[i]].in_removed = 1;
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
    /* ---- outputs (:391-407) ---- */
    int n_out = 0;
    for (int i = 0; i < nn && n_out < out_cap; ++i) {
        const strk *k = &B->T[ntr[i]];
        if (!k->is_activated) continue;
        double *o = rows_out + 8 * (size_t)n_out++;
        if (k->f32) {
            float r0 = (float)k->mean[0], r1 = (float)k->mean[1]; const float r2 = (float)k->mean[2], r3 = (float)k->mean[3];
            r0 -= r2 / 2; r1 -= r3 / 2;
            const float hw = r2 / 2, hh = r3 / 2;
            o[0] = r0 - hw; o[1] = r1 - hh; o[2] = r0 + hw; o[3] = r1 + hh;
        } else {
            double r0 = k->mean[0], r1 = k->mean[1]; const double r2 = k->mean[2], r3 = k->mean[3];
            r0 -= r2 / 2; r1 -= r3 / 2;
            const double hw = r2 / 2, hh = r3 / 2;
            o[0] = r0 - hw; o[1] = r1 - hh; o[2] = r0 + hw; o[3] = r1 + hh;
        }
        o[4] = (double)k->track_id; o[5] = k->cls; o[6] = k->score; o[7] = k->tracklab_id;
    }
    free(hi); free(lo); free(fbuf); free(unconf); free(pool); free(activated); free(refind); free(newlost); free(removed);
    free(m_r); free(m_c); free(u_r); free(u_c); free(tb); free(db); free(dz); free(cost); free(gd); free(u_det1); free(rtr); free(rem);
    return n_out;
}

int orc_botsort_list_len(const orc_botsort *B, int which) { return which ? B->n_lost : B->n_tracked; }
int orc_botsort_list(const orc_botsort *B, int which, int64_t *ids, double *mean, double *cov, int64_t *state5, float *feat, int cap)
{
    const int *l = which ? B->lost : B->tracked; int n = which ? B->n_lost : B->n_tracked;
    if (n > cap) n = cap;
    for (int i = 0; i < n; ++i) {
        const strk *k = &B->T[l[i]];
        ids[i] = k->track_id; memcpy(mean + 8 * i, k->mean, sizeof(k->mean)); memcpy(cov + 64 * i, k->cov, sizeof(k->cov));
        state5[5 * i] = k->state; state5[5 * i + 1] = k->is_activated; state5[5 * i + 2] = k->frame_id; state5[5 * i + 3] = k->start_frame; state5[5 * i + 4] = k->tracklet_len;
        memcpy(feat + (size_t)i * B->D, k->smooth, sizeof(float) * B->D);
    }
    return n;
}
