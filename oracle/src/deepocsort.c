/*
 * oracle/src/deepocsort.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Deep-OC-SORT: plugins/track/deep_oc_sort/{ocsort.py, association.py, kalmanfilter.py} restated in C with cmc_off (CMCComputer is
 * cv2 optical flow: SURVEY 8f-3) and the (x, y, w, h) "new" Kalman filter (new_kf_off: false, the yaml's value).
 *
 *   OCSort.update            ocsort.py:392-534   (NOTE: frame_count is never incremented there, so `frame_count <= min_hits`
 *                            always holds and min_hits has no effect; a track's conf is the one of its FIRST detection)
 *   KalmanBoxTracker         ocsort.py:94-330    (new_kf: process / measurement noise relative to w, h; frozen flag; ORU through
 *                            KalmanFilterNew.freeze / unfreeze, whose virtual boxes read the (x, y, w, h) measurement as
 *                            (x, y, s, r): kalmanfilter.py:433-478, and run with the filter's default R = I, Q = I)
 *   associate                association.py:291-364 (IoU + angle cost x the CLASS column + adaptive-weighted embedding cost)
 *   compute_aw_max_metric    association.py:263-288
 * dtype trail: the embeddings are float32 torch tensors throughout (a float64 numpy scalar times a float32 tensor stays float32 with
 * the scalar rounded to float32; tensor /= numpy float32 is a true division); dets_embs @ trk_embs.T is a float32 GEMM, and
 * compute_aw_max_metric's row / column weights are float32 arithmetic on it. Everything else is float64.
 */
#include "orc.h"
#include "lapack_order.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RING 64

typedef struct {
    double x[8], P[64];
    int observed, has_saved;
    double sx[8], sP[64];
    double last_z[4];
    int gap;
} kfn;

static void noise8(double w, double h, double *q)          /* new_kf_process_noise (ocsort.py:77-82) */
{
    const double p = 1. / 20, v = 1. / 160;
    q[0] = (p * w) * (p * w); q[1] = (p * h) * (p * h); q[2] = q[0]; q[3] = q[1];
    q[4] = (v * w) * (v * w); q[5] = (v * h) * (v * h); q[6] = q[4]; q[7] = q[5];
}
static void noise4(double w, double h, double *r)          /* new_kf_measurement_noise (:85-89) */
{
    const double m = 1. / 20;
    r[0] = (m * w) * (m * w); r[1] = (m * h) * (m * h); r[2] = r[0]; r[3] = r[1];
}

static void kfn_predict(kfn *k, const double *qdiag)       /* kalmanfilter.py:340-379, F of ocsort.py:118-130 */
{
    double F[64], Ft[64], t1[64], t2[64], nx[8];
    memset(F, 0, sizeof(F));
    for (int i = 0; i < 8; ++i) F[i * 9] = 1;
    for (int i = 0; i < 4; ++i) F[i * 8 + i + 4] = 1;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) Ft[i * 8 + j] = F[j * 8 + i];
    lo_gemm(F, k->x, nx, 8, 8, 1);
    memcpy(k->x, nx, sizeof(nx));
    lo_gemm(F, k->P, t1, 8, 8, 8);
    lo_gemm(t1, Ft, t2, 8, 8, 8);
    for (int i = 0; i < 64; ++i) k->P[i] = 1.0 * t2[i];
    for (int i = 0; i < 8; ++i) k->P[i * 9] += qdiag[i];
}
static void kfn_update_core(kfn *k, const double *z, const double *rdiag)        /* kalmanfilter.py:522-564 */
{
    double y[4], PHT[32], S[16], SI[16], K[32], IKH[64], t1[64], t2[64], IKHt[64], KR[32], Kt[32], t3[64];
    for (int i = 0; i < 4; ++i) y[i] = z[i] - k->x[i];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) PHT[i * 4 + j] = k->P[i * 8 + j];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) S[i * 4 + j] = PHT[i * 4 + j] + (i == j ? rdiag[i] : 0.0);
    lo_inv4(S, SI);                /* np.linalg.inv in LAPACK's operation order (lapack_order.h) */
    lo_gemm(PHT, SI, K, 8, 4, 4);
    for (int i = 0; i < 8; ++i) {
        k->x[i] = k->x[i] + lo_dot4_h2(K + i * 4, y);          /* x + dot(K, y): dgemv order */
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) IKH[i * 8 + j] = (i == j ? 1.0 : 0.0) - (j < 4 ? K[i * 4 + j] : 0.0);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) IKHt[i * 8 + j] = IKH[j * 8 + i];
    lo_gemm(IKH, k->P, t1, 8, 8, 8);
    lo_gemm(t1, IKHt, t2, 8, 8, 8);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) { KR[i * 4 + j] = K[i * 4 + j] * rdiag[j]; Kt[j * 8 + i] = K[i * 4 + j]; }
    lo_gemm(KR, Kt, t3, 8, 4, 8);
    for (int i = 0; i < 64; ++i) k->P[i] = t2[i] + t3[i];
}
static void kfn_update(kfn *k, const double *z /* or NULL */, const double *rdiag)
{
    static const double ONE4[4] = {1, 1, 1, 1}, ONE8[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    if (!z) {                                   /* kalmanfilter.py:507-520 */
        if (k->observed) { memcpy(k->sx, k->x, sizeof(k->x)); memcpy(k->sP, k->P, sizeof(k->P)); k->has_saved = 1; }
        k->observed = 0;
        k->gap += 1;
        return;
    }
    if (!k->observed && k->has_saved) {          /* unfreeze(): kalmanfilter.py:433-478 */
        const double x1 = k->last_z[0], y1 = k->last_z[1], s1 = k->last_z[2], r1 = k->last_z[3];
        const double w1 = sqrt(s1 * r1), h1 = sqrt(s1 / r1);
        const double x2 = z[0], y2 = z[1], s2 = z[2], r2 = z[3];
        const double w2 = sqrt(s2 * r2), h2 = sqrt(s2 / r2);
        const int time_gap = k->gap + 1;
        const double dx = (x2 - x1) / time_gap, dy = (y2 - y1) / time_gap, dw = (w2 - w1) / time_gap, dh = (h2 - h1) / time_gap;
        memcpy(k->x, k->sx, sizeof(k->x));
        memcpy(k->P, k->sP, sizeof(k->P));
        k->has_saved = 0;
        double nb[4];
        for (int i = 0; i < time_gap; ++i) {
            const double x = x1 + (i + 1) * dx, y = y1 + (i + 1) * dy, w = w1 + (i + 1) * dw, h = h1 + (i + 1) * dh;
            nb[0] = x; nb[1] = y; nb[2] = w * h; nb[3] = w / h;
            kfn_update_core(k, nb, ONE4);                    /* self.update(new_box): self.R = eye(4) */
            if (i != time_gap - 1) kfn_predict(k, ONE8);     /* self.predict(): self.Q = eye(8) */
        }
        memcpy(k->last_z, nb, sizeof(nb));
        k->gap = 0;
        k->observed = 1;
        kfn_update_core(k, z, rdiag);
        return;
    }
    k->observed = 1;
    memcpy(k->last_z, z, 4 * sizeof(double));
    k->gap = 0;
    kfn_update_core(k, z, rdiag);
}

/* ------------------------------------------------------------------ KalmanBoxTracker (new_kf) */
typedef struct {
    kfn kf;
    int64_t id;
    int tsu, hits, hit_streak, age, delta_t, frozen, last_upd_age;
    double conf, cls, tracklab_id;
    double last_obs[5];
    int has_vel; double vel[2];
    int n_obs;
    int obs_age[RING]; double obs_box[RING][5];
    float *emb;
} dkbt;

struct orc_deepocsort {
    double det_thresh, iou_threshold, inertia, w_emb, alpha_fixed, aw_param;
    int max_age, min_hits, delta_t, asso_func, aw_off, D;
    int64_t count;
    dkbt **trk; int n, cap;
};

static void bbox_to_z(const double *b, double *z) { const double w = b[2] - b[0], h = b[3] - b[1]; z[0] = b[0] + w / 2.0; z[1] = b[1] + h / 2.0; z[2] = w; z[3] = h; }
static void x_to_bbox(const double *x, double *b) { b[0] = x[0] - x[2] / 2; b[1] = x[1] - x[3] / 2; b[2] = x[0] + x[2] / 2; b[3] = x[1] + x[3] / 2; }
static void speed_direction(const double *b1, const double *b2, double *out)
{
    const double cx1 = (b1[0] + b1[2]) / 2.0, cy1 = (b1[1] + b1[3]) / 2.0, cx2 = (b2[0] + b2[2]) / 2.0, cy2 = (b2[1] + b2[3]) / 2.0;
    const double norm = sqrt((cy2 - cy1) * (cy2 - cy1) + (cx2 - cx1) * (cx2 - cx1)) + 1e-6;
    out[0] = (cy2 - cy1) / norm; out[1] = (cx2 - cx1) / norm;
}
static double sum5(const double *a) { return (((a[0] + a[1]) + a[2]) + a[3]) + a[4]; }
static double sum4of5(const double *a) { return sum5(a); }

static const double *obs_lookup(const dkbt *t, int age)
{
    if (age < 0) return NULL;
    const int s = age % RING;
    return t->obs_age[s] == age ? t->obs_box[s] : NULL;
}

static dkbt *kbt_new(const double *bbox5, double cls, int delta_t, const float *emb, int D, double tracklab_id, int64_t id)
{
    dkbt *t = calloc(1, sizeof(*t));
    double z[4], q[8];
    bbox_to_z(bbox5, z);
    noise8(z[2], z[3], q);
    for (int i = 0; i < 4; ++i) t->kf.P[i * 9] = q[i] * 4;
    for (int i = 4; i < 8; ++i) t->kf.P[i * 9] = q[i] * 100;
    for (int i = 0; i < 4; ++i) t->kf.x[i] = z[i];
    t->id = id; t->conf = bbox5[4]; t->cls = cls; t->delta_t = delta_t; t->tracklab_id = tracklab_id;
    for (int i = 0; i < 5; ++i) t->last_obs[i] = -1;
    for (int i = 0; i < RING; ++i) t->obs_age[i] = -1;
    t->emb = malloc(sizeof(float) * D);
    memcpy(t->emb, emb, sizeof(float) * D);
    return t;
}
static void kbt_free(dkbt *t) { free(t->emb); free(t); }

static void kbt_update(dkbt *t, const double *bbox5, double cls, double tid)       /* ocsort.py:208-252 */
{
    if (bbox5) {
        t->frozen = 0;
        t->cls = cls;
        if (sum5(t->last_obs) >= 0) {
            const double *prev = NULL;
            for (int dt = t->delta_t; dt > 0; --dt) { prev = obs_lookup(t, t->age - dt); if (prev) break; }
            if (!prev) prev = t->last_obs;
            speed_direction(prev, bbox5, t->vel);
            t->has_vel = 1;
        }
        memcpy(t->last_obs, bbox5, 5 * sizeof(double));
        t->last_upd_age = t->age;
        const int s = t->age % RING;
        t->obs_age[s] = t->age; memcpy(t->obs_box[s], bbox5, 5 * sizeof(double));
        t->n_obs += 1;
        t->tsu = 0; t->hits += 1; t->hit_streak += 1;
        double z[4], r[4];
        noise4(t->kf.x[2], t->kf.x[3], r);            /* R from the state BEFORE kf.update (and so before unfreeze) */
        bbox_to_z(bbox5, z);
        kfn_update(&t->kf, z, r);
        t->tracklab_id = tid;
    } else {
        kfn_update(&t->kf, NULL, NULL);
        t->frozen = 1;
    }
}
static void kbt_update_emb(dkbt *t, const float *emb, double alpha, int D)         /* ocsort.py:254-256, float32 */
{
    const float a = (float)alpha, b = (float)(1 - alpha);
    float ss = 0.f;
    for (int d = 0; d < D; ++d) { const float u = a * t->emb[d], v = b * emb[d]; t->emb[d] = u + v; ss += t->emb[d] * t->emb[d]; }
    const float n = sqrtf(ss);
    for (int d = 0; d < D; ++d) t->emb[d] = t->emb[d] / n;
}
static void kbt_predict(dkbt *t, double *pos)                                      /* ocsort.py:283-309 */
{
    double *x = t->kf.x, q[8];
    if (x[2] + x[6] <= 0) x[6] = 0;
    if (x[3] + x[7] <= 0) x[7] = 0;
    if (t->frozen) { x[6] = 0; x[7] = 0; }
    noise8(x[2], x[3], q);
    kfn_predict(&t->kf, q);
    t->age += 1;
    if (t->tsu > 0) t->hit_streak = 0;
    t->tsu += 1;
    x_to_bbox(t->kf.x, pos);
}

/* KalmanBoxTracker.apply_affine_correction (ocsort.py:261-281) + KalmanFilterNew.apply_affine_correction (kalmanfilter.py:387-405,
 * new_kf branch). last_observation and observations[age of the last update] are the SAME numpy array in the reference (update()
 * stores one view of the detection row in both, ocsort.py:232-234), so a box inside the delta_t window is warped twice. */
static void affine_pts(const double *A, double *b4)
{
    const double x1 = A[0] * b4[0] + A[1] * b4[1], y1 = A[3] * b4[0] + A[4] * b4[1];
    const double x2 = A[0] * b4[2] + A[1] * b4[3], y2 = A[3] * b4[2] + A[4] * b4[3];
    b4[0] = x1 + A[2]; b4[1] = y1 + A[5]; b4[2] = x2 + A[2]; b4[3] = y2 + A[5];
}
static void affine_state(const double *A, double *x, double *P)
{
    const double R[4] = {A[0], A[1], A[3], A[4]};
    double m[8], t1[64], c2[64];
    for (int b = 0; b < 4; ++b) { m[2 * b] = R[0] * x[2 * b] + R[1] * x[2 * b + 1]; m[2 * b + 1] = R[2] * x[2 * b] + R[3] * x[2 * b + 1]; }
    m[0] += A[2]; m[1] += A[5];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) { const int r0 = i & ~1; t1[i * 8 + j] = R[(i & 1) * 2] * P[r0 * 8 + j] + R[(i & 1) * 2 + 1] * P[(r0 + 1) * 8 + j]; }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) { const int c0 = j & ~1; c2[i * 8 + j] = t1[i * 8 + c0] * R[(j & 1) * 2] + t1[i * 8 + c0 + 1] * R[(j & 1) * 2 + 1]; }
    memcpy(x, m, sizeof(m)); memcpy(P, c2, sizeof(c2));
}
static void kbt_affine(dkbt *t, const double *A)
{
    const int alias_slot = (t->n_obs > 0) ? t->last_upd_age % RING : -1;
    if (sum5(t->last_obs) > 0) {
        affine_pts(A, t->last_obs);
        if (alias_slot >= 0 && t->obs_age[alias_slot] == t->last_upd_age) memcpy(t->obs_box[alias_slot], t->last_obs, 4 * sizeof(double));
    }
    for (int dt = t->delta_t; dt >= 0; --dt) {
        const int age = t->age - dt;
        if (age < 0) continue;
        const int s = age % RING;
        if (t->obs_age[s] != age) continue;
        affine_pts(A, t->obs_box[s]);
        if (s == alias_slot && age == t->last_upd_age) memcpy(t->last_obs, t->obs_box[s], 4 * sizeof(double));
    }
    affine_state(A, t->kf.x, t->kf.P);
    if (!t->kf.observed && t->kf.has_saved) {
        affine_state(A, t->kf.sx, t->kf.sP);
        double *lm = t->kf.last_z;
        const double p0 = A[0] * lm[0] + A[1] * lm[1] + A[2], p1 = A[3] * lm[0] + A[4] * lm[1] + A[5];
        const double s0 = A[0] * lm[2] + A[1] * lm[3], s1 = A[3] * lm[2] + A[4] * lm[3];
        lm[0] = p0; lm[1] = p1; lm[2] = s0; lm[3] = s1;
    }
}

orc_deepocsort *orc_deepocsort_create(double det_thresh, int max_age, int min_hits, double iou_threshold, int delta_t, int asso_func,
                                      double inertia, double w_association_emb, double alpha_fixed_emb, double aw_param, int aw_off, int D)
{
    orc_deepocsort *o = calloc(1, sizeof(*o));
    o->det_thresh = det_thresh; o->max_age = max_age; o->min_hits = min_hits; o->iou_threshold = iou_threshold; o->delta_t = delta_t;
    o->asso_func = asso_func; o->inertia = inertia; o->w_emb = w_association_emb; o->alpha_fixed = alpha_fixed_emb; o->aw_param = aw_param;
    o->aw_off = aw_off; o->D = D;
    return o;
}
void orc_deepocsort_destroy(orc_deepocsort *o)
{ if (!o) return; for (int i = 0; i < o->n; ++i) kbt_free(o->trk[i]); free(o->trk); free(o); }
int orc_deepocsort_num_tracks(const orc_deepocsort *o) { return o->n; }
int orc_deepocsort_get_tracks(const orc_deepocsort *o, int64_t *ids, double *x, double *P, float *emb, int64_t *state6, double *vel, double *last, int cap)
{
    const int n = o->n < cap ? o->n : cap;
    for (int i = 0; i < n; ++i) {
        const dkbt *k = o->trk[i];
        ids[i] = k->id; memcpy(x + 8 * i, k->kf.x, 64); memcpy(P + 64 * i, k->kf.P, 512); memcpy(emb + (size_t)i * o->D, k->emb, sizeof(float) * o->D);
        state6[6 * i] = k->tsu; state6[6 * i + 1] = k->hits; state6[6 * i + 2] = k->hit_streak; state6[6 * i + 3] = k->age;
        state6[6 * i + 4] = k->frozen; state6[6 * i + 5] = k->kf.observed;
        vel[2 * i] = k->has_vel ? k->vel[0] : 0; vel[2 * i + 1] = k->has_vel ? k->vel[1] : 0;
        memcpy(last + 5 * i, k->last_obs, 40);
    }
    return n;
}

static void trk_pop(orc_deepocsort *o, int i)
{ kbt_free(o->trk[i]); memmove(o->trk + i, o->trk + i + 1, sizeof(*o->trk) * (size_t)(o->n - i - 1)); o->n--; }
static void trk_push(orc_deepocsort *o, dkbt *k)
{ if (o->n == o->cap) { o->cap = o->cap ? 2 * o->cap : 64; o->trk = realloc(o->trk, sizeof(*o->trk) * (size_t)o->cap); } o->trk[o->n++] = k; }
static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
static int setdiff_sorted(int *a, int na, const int *rem, int nrem)
{
    qsort(a, (size_t)na, sizeof(int), cmp_int);
    int k = 0;
    for (int i = 0; i < na; ++i) {
        if (i > 0 && a[i] == a[i - 1]) continue;
        int drop = 0;
        for (int j = 0; j < nrem; ++j) if (rem[j] == a[i]) { drop = 1; break; }
        if (!drop) a[k++] = a[i];
    }
    return k;
}
static double mat_max(const double *a, size_t n) { double m = a[0]; for (size_t i = 1; i < n; ++i) m = (a[i] > m || isnan(a[i])) ? a[i] : m; return m; }

/* row / column weight of compute_aw_max_metric from the two largest entries (float32 arithmetic, association.py:266-286) */
static float aw_weight(float top1, float top2, double bottom)
{
    if (top1 == 0.f) return 0.f;
    const float r = top2 / top1;
    const float m = r - (float)bottom;
    if (!(m > 0.f)) return 1.f;                   /* max(m, 0) -> int 0 -> weight exactly 1.0 */
    return 1.f - m / (float)(1 - bottom);
}
static void top2(const float *v, int n, int stride, float *t1, float *t2)
{
    float a = -INFINITY, b = -INFINITY;
    for (int i = 0; i < n; ++i) { const float x = v[(size_t)i * stride]; if (x > a) { b = a; a = x; } else if (x > b) b = x; }
    *t1 = a; *t2 = b;
}

/* association.py:291-364. dets (N,7) rows (the reference passes dets[:, :-1]: "scores = detections[:, -1]" is the CLASS column) */
static void associate(const orc_deepocsort *o, const double *dets, const float *demb, int N, const double *trks, const float *temb, int T,
                      const double *vel, const double *kobs, int *matches, int *n_matches, int *um_d, int *n_um_d, int *um_t, int *n_um_t)
{
    const int D = o->D;
    const double iou_thr = o->iou_threshold;
    *n_matches = 0; *n_um_d = 0; *n_um_t = 0;
    if (T == 0) { for (int d = 0; d < N; ++d) um_d[(*n_um_d)++] = d; return; }
    int64_t *mi_r = malloc(sizeof(int64_t) * (size_t)(N + T + 1)), *mi_c = malloc(sizeof(int64_t) * (size_t)(N + T + 1));
    int n_mi = 0;
    double *iou = malloc(sizeof(double) * (size_t)N * T + 8);
    if (N > 0) {
        orc_iou_matrix(ORC_IOU, dets, N, 7, trks, T, 4, iou);
        int rmax = 0, cmax = 0;
        int *csum = calloc((size_t)T, sizeof(int));
        for (int d = 0; d < N; ++d) {
            int rs = 0;
            for (int t = 0; t < T; ++t) if (iou[(size_t)d * T + t] > iou_thr) { rs++; csum[t]++; }
            if (rs > rmax) rmax = rs;
        }
        for (int t = 0; t < T; ++t) if (csum[t] > cmax) cmax = csum[t];
        free(csum);
        if (rmax == 1 && cmax == 1) {
            for (int d = 0; d < N; ++d) for (int t = 0; t < T; ++t)
                if (iou[(size_t)d * T + t] > iou_thr) { mi_r[n_mi] = d; mi_c[n_mi] = t; n_mi++; }
        } else {
            double *cost = malloc(sizeof(double) * (size_t)N * T);
            float *ec = malloc(sizeof(float) * (size_t)N * T);
            for (int d = 0; d < N; ++d)                              /* dets_embs @ trk_embs.T (float32), zeroed where the IoU is 0 */
                for (int t = 0; t < T; ++t) {
                    float s = 0.f;
                    for (int k = 0; k < D; ++k) s += demb[(size_t)d * D + k] * temb[(size_t)t * D + k];
                    ec[(size_t)d * T + t] = iou[(size_t)d * T + t] <= 0 ? 0.f : s;
                }
            if (!o->aw_off) {
                float *wm = malloc(sizeof(float) * (size_t)N * T);
                for (size_t e = 0; e < (size_t)N * T; ++e) wm[e] = (float)o->w_emb;
                if (T >= 2)
                    for (int d = 0; d < N; ++d) {
                        float a, b;
                        top2(ec + (size_t)d * T, T, 1, &a, &b);
                        const float w = aw_weight(a, b, o->aw_param);
                        for (int t = 0; t < T; ++t) wm[(size_t)d * T + t] *= w;
                    }
                if (N >= 2)
                    for (int t = 0; t < T; ++t) {
                        float a, b;
                        top2(ec + t, N, T, &a, &b);
                        const float w = aw_weight(a, b, o->aw_param);
                        for (int d = 0; d < N; ++d) wm[(size_t)d * T + t] *= w;
                    }
                for (size_t e = 0; e < (size_t)N * T; ++e) ec[e] = wm[e] * ec[e];
                free(wm);
            } else
                for (size_t e = 0; e < (size_t)N * T; ++e) ec[e] = ec[e] * (float)o->w_emb;
            for (int t = 0; t < T; ++t) {
                const double *ko = kobs + t * 5;
                const double cx2 = (ko[0] + ko[2]) / 2.0, cy2 = (ko[1] + ko[3]) / 2.0;
                const double valid = ko[4] < 0 ? 0.0 : 1.0;
                for (int d = 0; d < N; ++d) {
                    const double *de = dets + d * 7;
                    const double cx1 = (de[0] + de[2]) / 2.0, cy1 = (de[1] + de[3]) / 2.0;
                    double dx = cx1 - cx2, dy = cy1 - cy2;
                    const double norm = sqrt(dx * dx + dy * dy) + 1e-6;
                    dx = dx / norm; dy = dy / norm;
                    double c = vel[t * 2 + 1] * dx + vel[t * 2 + 0] * dy;
                    c = c < -1 ? -1 : (c > 1 ? 1 : c);
                    double ang = acos(c);
                    ang = (M_PI / 2.0 - fabs(ang)) / M_PI;
                    const double adc = ((valid * ang) * o->inertia) * de[5];
                    cost[(size_t)d * T + t] = -((iou[(size_t)d * T + t] + adc) + (double)ec[(size_t)d * T + t]);
                }
            }
            n_mi = orc_lsa(cost, N, T, mi_r, mi_c);
            if (n_mi < 0) n_mi = 0;
            free(cost); free(ec);
        }
    }
    for (int d = 0; d < N; ++d) { int f = 0; for (int k = 0; k < n_mi; ++k) if (mi_r[k] == d) { f = 1; break; } if (!f) um_d[(*n_um_d)++] = d; }
    for (int t = 0; t < T; ++t) { int f = 0; for (int k = 0; k < n_mi; ++k) if (mi_c[k] == t) { f = 1; break; } if (!f) um_t[(*n_um_t)++] = t; }
    for (int k = 0; k < n_mi; ++k) {
        if (iou[(size_t)mi_r[k] * T + mi_c[k]] < iou_thr) { um_d[(*n_um_d)++] = (int)mi_r[k]; um_t[(*n_um_t)++] = (int)mi_c[k]; }
        else { matches[2 * *n_matches] = (int)mi_r[k]; matches[2 * *n_matches + 1] = (int)mi_c[k]; (*n_matches)++; }
    }
    free(mi_r); free(mi_c); free(iou);
}

int orc_deepocsort_update_cmc(orc_deepocsort *o, const double *dets_in, const float *embs_in, int n_in, const double *warp6, double *out, int out_cap);
int orc_deepocsort_update(orc_deepocsort *o, const double *dets_in, const float *embs_in, int n_in, double *out, int out_cap)
{ return orc_deepocsort_update_cmc(o, dets_in, embs_in, n_in, NULL, out, out_cap); }

/* warp6: what CMCComputer.compute_affine returned for this frame ((2,3) float64, ocsort.py:425-428), NULL = cmc_off */
int orc_deepocsort_update_cmc(orc_deepocsort *o, const double *dets_in, const float *embs_in, int n_in, const double *warp6, double *out, int out_cap)
{
    const int D = o->D;
    if (warp6) for (int t = 0; t < o->n; ++t) kbt_affine(o->trk[t], warp6);
    double *dets = malloc(sizeof(double) * 7 * (size_t)(n_in + 1)), *alpha = malloc(sizeof(double) * (size_t)(n_in + 1));
    float *demb = malloc(sizeof(float) * (size_t)(n_in + 1) * D);
    int N = 0;
    for (int i = 0; i < n_in; ++i)
        if (dets_in[i * 7 + 4] > o->det_thresh) {                                   /* ocsort.py:407-408 */
            memcpy(dets + 7 * N, dets_in + 7 * i, 56);
            memcpy(demb + (size_t)N * D, embs_in + (size_t)i * D, sizeof(float) * D);
            const double trust = (dets_in[i * 7 + 4] - o->det_thresh) / (1 - o->det_thresh);
            alpha[N] = o->alpha_fixed + (1 - o->alpha_fixed) * (1 - trust);          /* :433-436 */
            N++;
        }
    const int T0 = o->n;
    double *trks = malloc(sizeof(double) * 4 * (size_t)(T0 + 1));
    int T = 0;
    for (int t = 0; t < o->n;) {                                                     /* :439-456 */
        double pos[4];
        kbt_predict(o->trk[t], pos);
        if (isnan(pos[0]) || isnan(pos[1]) || isnan(pos[2]) || isnan(pos[3])) { trk_pop(o, t); continue; }
        memcpy(trks + 4 * T, pos, 32); T++; t++;
    }
    double *vel = malloc(sizeof(double) * 2 * (size_t)(T + 1)), *last_boxes = malloc(sizeof(double) * 5 * (size_t)(T + 1));
    double *kobs = malloc(sizeof(double) * 5 * (size_t)(T + 1));
    float *temb = malloc(sizeof(float) * (size_t)(T + 1) * D);
    for (int t = 0; t < T; ++t) {
        const dkbt *k = o->trk[t];
        vel[2 * t] = k->has_vel ? k->vel[0] : 0; vel[2 * t + 1] = k->has_vel ? k->vel[1] : 0;
        memcpy(last_boxes + 5 * t, k->last_obs, 40);
        memcpy(temb + (size_t)t * D, k->emb, sizeof(float) * D);                     /* embeddings as of before this frame's updates */
        const double *p = NULL;
        if (k->n_obs == 0) { for (int i = 0; i < 5; ++i) kobs[5 * t + i] = -1; }
        else {
            for (int i = 0; i < o->delta_t; ++i) { p = obs_lookup(k, k->age - (o->delta_t - i)); if (p) break; }
            if (!p) p = k->last_obs;
            memcpy(kobs + 5 * t, p, 40);
        }
    }
    const int cap = N + T + 4;
    int *matches = malloc(sizeof(int) * 2 * (size_t)cap), *um_d = malloc(sizeof(int) * (size_t)cap), *um_t = malloc(sizeof(int) * (size_t)cap);
    int nm, nud, nut;
    associate(o, dets, demb, N, trks, temb, T, vel, kobs, matches, &nm, um_d, &nud, um_t, &nut);
    for (int k = 0; k < nm; ++k) {
        const int di = matches[2 * k];
        const double *d = dets + 7 * di;
        kbt_update(o->trk[matches[2 * k + 1]], d, d[5], d[6]);
        kbt_update_emb(o->trk[matches[2 * k + 1]], demb + (size_t)di * D, alpha[di], D);
    }
    /* second round by OCR on the last observations (:480-513) */
    if (nud > 0 && nut > 0) {
        int64_t *lr = malloc(sizeof(int64_t) * (size_t)cap), *lc = malloc(sizeof(int64_t) * (size_t)cap);
        double *ld = malloc(sizeof(double) * 7 * (size_t)nud), *lt = malloc(sizeof(double) * 5 * (size_t)nut);
        double *il = malloc(sizeof(double) * (size_t)nud * nut);
        for (int k = 0; k < nud; ++k) memcpy(ld + 7 * k, dets + 7 * um_d[k], 56);
        for (int k = 0; k < nut; ++k) memcpy(lt + 5 * k, last_boxes + 5 * um_t[k], 40);
        orc_iou_matrix(o->asso_func, ld, nud, 7, lt, nut, 5, il);
        if (mat_max(il, (size_t)nud * nut) > o->iou_threshold) {
            double *neg = malloc(sizeof(double) * (size_t)nud * nut);
            for (size_t k = 0; k < (size_t)nud * nut; ++k) neg[k] = -il[k];
            const int nl = orc_lsa(neg, nud, nut, lr, lc);
            int *remd = malloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1)), *remt = malloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1)), nr = 0;
            for (int k = 0; k < nl; ++k) {
                const int di = um_d[lr[k]], ti = um_t[lc[k]];
                if (il[(size_t)lr[k] * nut + lc[k]] < o->iou_threshold) continue;
                const double *d = dets + 7 * di;
                kbt_update(o->trk[ti], d, d[5], d[6]);
                kbt_update_emb(o->trk[ti], demb + (size_t)di * D, alpha[di], D);
                remd[nr] = di; remt[nr] = ti; nr++;
            }
            nud = setdiff_sorted(um_d, nud, remd, nr);
            nut = setdiff_sorted(um_t, nut, remt, nr);
            free(remd); free(remt); free(neg);
        }
        free(ld); free(lt); free(il); free(lr); free(lc);
    }
    for (int k = 0; k < nut; ++k) kbt_update(o->trk[um_t[k]], NULL, 0, 0);
    for (int k = 0; k < nud; ++k) {
        const double *d = dets + 7 * um_d[k];
        trk_push(o, kbt_new(d, d[5], o->delta_t, demb + (size_t)um_d[k] * D, D, d[6], o->count++));
    }
    int rows = 0;
    for (int i = o->n - 1; i >= 0; --i) {                                            /* :521-531; frame_count stays 0 */
        const dkbt *k = o->trk[i];
        double d[4];
        if (sum4of5(k->last_obs) < 0) x_to_bbox(k->kf.x, d); else memcpy(d, k->last_obs, 32);
        if (k->tsu < 1 && (k->hit_streak >= o->min_hits || 0 <= o->min_hits) && rows < out_cap) {
            double *r = out + 8 * rows++;
            r[0] = d[0]; r[1] = d[1]; r[2] = d[2]; r[3] = d[3];
            r[4] = (double)(k->id + 1); r[5] = k->cls; r[6] = k->conf; r[7] = k->tracklab_id;
        }
        if (k->tsu > o->max_age) trk_pop(o, i);
    }
    free(dets); free(alpha); free(demb); free(trks); free(vel); free(last_boxes); free(kobs); free(temb);
    free(matches); free(um_d); free(um_t);
    return rows;
}
