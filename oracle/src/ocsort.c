/* CPU ORACLE (test infrastructure) -- see orc.h.
 * OC-SORT restated from plugins/track/oc_sort/{ocsort,association,kalmanfilter}.py. */
#include "orc.h"
#include "lapack_order.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RING 64            /* observations ring; delta_t must be < RING */

/* ------------------------------------------------------------------ KalmanFilterNew (dim_x=7, dim_z=4)
 * oc_sort/kalmanfilter.py:283-337 (__init__), :339-379 (predict), :383-434 (freeze/unfreeze), :437-526 (update) */
typedef struct {
    double x[7], P[49];
    int observed, has_saved;
    double sx[7], sP[49];      /* attr_saved: x,P at freeze time (kalmanfilter.py:383-387) */
    double last_z[4];          /* last non-None entry of history_obs */
    int gap;                   /* number of trailing None entries in history_obs */
} kf7;

static const double R_DIAG[4] = {1., 1., 10., 10.};       /* ocsort.py:80 */

static void kf7_init(kf7 *k)
{
    memset(k, 0, sizeof(*k));
    for (int i = 0; i < 7; ++i) k->P[i * 7 + i] = 1.0;
    for (int i = 4; i < 7; ++i) k->P[i * 7 + i] *= 1000.;   /* ocsort.py:81 */
    for (int i = 0; i < 49; ++i) k->P[i] *= 10.;            /* ocsort.py:82 */
}

static void kf7_Q(double *q)   /* ocsort.py:83-84 */
{
    for (int i = 0; i < 7; ++i) q[i] = 1.0;
    q[6] *= 0.01;
    for (int i = 4; i < 7; ++i) q[i] *= 0.01;
}

static void kf7_predict(kf7 *k)   /* kalmanfilter.py:368-379 with F of ocsort.py:75-76 */
{
    static const double F[49] = {1,0,0,0,1,0,0, 0,1,0,0,0,1,0, 0,0,1,0,0,0,1, 0,0,0,1,0,0,0,
                                 0,0,0,0,1,0,0, 0,0,0,0,0,1,0, 0,0,0,0,0,0,1};
    double Ft[49], t1[49], t2[49], nx[7], q[7];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) Ft[i * 7 + j] = F[j * 7 + i];
    lo_gemm(F, k->x, nx, 7, 7, 1);
    memcpy(k->x, nx, sizeof(nx));
    lo_gemm(F, k->P, t1, 7, 7, 7);
    lo_gemm(t1, Ft, t2, 7, 7, 7);
    kf7_Q(q);
    for (int i = 0; i < 49; ++i) k->P[i] = 1.0 * t2[i];      /* _alpha_sq = 1 */
    for (int i = 0; i < 7; ++i) k->P[i * 7 + i] += q[i];
}

static void kf7_update_core(kf7 *k, const double *z)    /* kalmanfilter.py:480-526, H of ocsort.py:77-78 */
{
    double y[4], PHT[28], S[16], SI[16], K[28], IKH[49], t1[49], t2[49], IKHt[49], KR[28], Kt[28], t3[49];
    for (int i = 0; i < 4; ++i) y[i] = z[i] - k->x[i];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) PHT[i * 4 + j] = k->P[i * 7 + j];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) S[i * 4 + j] = PHT[i * 4 + j] + (i == j ? R_DIAG[i] : 0.0);
    lo_inv4(S, SI);                /* np.linalg.inv in LAPACK's operation order (lapack_order.h) */
    lo_gemm(PHT, SI, K, 7, 4, 4);
    for (int i = 0; i < 7; ++i) {
        k->x[i] = k->x[i] + lo_dot4_h2(K + i * 4, y);          /* x + dot(K, y): dgemv order */
    }
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j)
        IKH[i * 7 + j] = (i == j ? 1.0 : 0.0) - (j < 4 ? K[i * 4 + j] : 0.0);
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) IKHt[i * 7 + j] = IKH[j * 7 + i];
    lo_gemm(IKH, k->P, t1, 7, 7, 7);
    lo_gemm(t1, IKHt, t2, 7, 7, 7);
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) { KR[i * 4 + j] = K[i * 4 + j] * R_DIAG[j]; Kt[j * 7 + i] = K[i * 4 + j]; }
    lo_gemm(KR, Kt, t3, 7, 4, 7);
    for (int i = 0; i < 49; ++i) k->P[i] = t2[i] + t3[i];
}

/* stateless forms (for per-method parity tests): x (7), P (7,7) in place */
void orc_kf7_predict(double *x, double *P)
{
    kf7 k; memset(&k, 0, sizeof(k)); memcpy(k.x, x, sizeof(k.x)); memcpy(k.P, P, sizeof(k.P));
    kf7_predict(&k);
    memcpy(x, k.x, sizeof(k.x)); memcpy(P, k.P, sizeof(k.P));
}
void orc_kf7_update(double *x, double *P, const double *z)
{
    kf7 k; memset(&k, 0, sizeof(k)); memcpy(k.x, x, sizeof(k.x)); memcpy(k.P, P, sizeof(k.P));
    kf7_update_core(&k, z);
    memcpy(x, k.x, sizeof(k.x)); memcpy(P, k.P, sizeof(k.P));
}

static void kf7_update(kf7 *k, const double *z /* or NULL */)
{
    if (!z) {                                   /* kalmanfilter.py:465-477 */
        if (k->observed) {                      /* freeze(): deepcopy(__dict__) */
            memcpy(k->sx, k->x, sizeof(k->x));
            memcpy(k->sP, k->P, sizeof(k->P));
            k->has_saved = 1;
        }
        k->observed = 0;
        k->gap += 1;
        return;
    }
    if (!k->observed && k->has_saved) {          /* unfreeze(): kalmanfilter.py:390-434 */
        double x1 = k->last_z[0], y1 = k->last_z[1], s1 = k->last_z[2], r1 = k->last_z[3];
        double w1 = sqrt(s1 * r1), h1 = sqrt(s1 / r1);
        double x2 = z[0], y2 = z[1], s2 = z[2], r2 = z[3];
        double w2 = sqrt(s2 * r2), h2 = sqrt(s2 / r2);
        int time_gap = k->gap + 1;
        double dx = (x2 - x1) / time_gap, dy = (y2 - y1) / time_gap;
        double dw = (w2 - w1) / time_gap, dh = (h2 - h1) / time_gap;
        memcpy(k->x, k->sx, sizeof(k->x));
        memcpy(k->P, k->sP, sizeof(k->P));
        k->has_saved = 0;                        /* restored dict carries attr_saved=None */
        double nb[4];
        for (int i = 0; i < time_gap; ++i) {
            double x = x1 + (i + 1) * dx, y = y1 + (i + 1) * dy;
            double w = w1 + (i + 1) * dw, h = h1 + (i + 1) * dh;
            nb[0] = x; nb[1] = y; nb[2] = w * h; nb[3] = w / h;
            kf7_update_core(k, nb);
            if (i != time_gap - 1) kf7_predict(k);
        }
        memcpy(k->last_z, nb, sizeof(nb));       /* history_obs ends with the last virtual box; z's append was lost */
        k->gap = 0;
        k->observed = 1;
        kf7_update_core(k, z);
        return;
    }
    k->observed = 1;
    memcpy(k->last_z, z, 4 * sizeof(double));
    k->gap = 0;
    kf7_update_core(k, z);
}

/* ------------------------------------------------------------------ KalmanBoxTracker, ocsort.py:57-169 */
struct orc_kbt {
    kf7 kf;
    int64_t id;
    int tsu, hits, hit_streak, age, delta_t;
    double conf, cls, tracklab_id;
    double last_obs[5];
    int has_vel; double vel[2];
    int n_obs;
    int obs_age[RING]; double obs_box[RING][5];
};

static void bbox_to_z(const double *b, double *z)   /* ocsort.py:21-33 */
{
    double w = b[2] - b[0], h = b[3] - b[1];
    z[0] = b[0] + w / 2.; z[1] = b[1] + h / 2.; z[2] = w * h; z[3] = w / (h + 1e-6);
}
static void x_to_bbox(const double *x, double *b)   /* ocsort.py:36-46 */
{
    double w = sqrt(x[2] * x[3]), h = x[2] / w;
    b[0] = x[0] - w / 2.; b[1] = x[1] - h / 2.; b[2] = x[0] + w / 2.; b[3] = x[1] + h / 2.;
}
static void speed_direction(const double *b1, const double *b2, double *out)   /* ocsort.py:49-54 */
{
    double cx1 = (b1[0] + b1[2]) / 2.0, cy1 = (b1[1] + b1[3]) / 2.0;
    double cx2 = (b2[0] + b2[2]) / 2.0, cy2 = (b2[1] + b2[3]) / 2.0;
    double norm = sqrt((cy2 - cy1) * (cy2 - cy1) + (cx2 - cx1) * (cx2 - cx1)) + 1e-6;
    out[0] = (cy2 - cy1) / norm; out[1] = (cx2 - cx1) / norm;
}
static double sum5(const double *a) { return (((a[0] + a[1]) + a[2]) + a[3]) + a[4]; }

static void kbt_init(orc_kbt *t, const double *bbox5, double cls, int delta_t, double tracklab_id, int64_t id)
{
    memset(t, 0, sizeof(*t));
    kf7_init(&t->kf);
    double z[4];
    bbox_to_z(bbox5, z);
    for (int i = 0; i < 4; ++i) t->kf.x[i] = z[i];
    t->id = id; t->conf = bbox5[4]; t->cls = cls; t->delta_t = delta_t; t->tracklab_id = tracklab_id;
    for (int i = 0; i < 5; ++i) t->last_obs[i] = -1;
    for (int i = 0; i < RING; ++i) t->obs_age[i] = -1;
}

static const double *obs_lookup(const orc_kbt *t, int age)
{
    if (age < 0) return NULL;
    int s = age % RING;
    return t->obs_age[s] == age ? t->obs_box[s] : NULL;
}

static void kbt_update(orc_kbt *t, const double *bbox5, double cls, int has_tid, double tid)   /* ocsort.py:109-148 */
{
    if (bbox5) {
        t->conf = bbox5[4]; t->cls = cls;
        if (sum5(t->last_obs) >= 0) {
            const double *prev = NULL;
            for (int i = 0; i < t->delta_t; ++i) {
                int dt = t->delta_t - i;
                prev = obs_lookup(t, t->age - dt);
                if (prev) break;
            }
            if (!prev) prev = t->last_obs;
            speed_direction(prev, bbox5, t->vel);
            t->has_vel = 1;
        }
        memcpy(t->last_obs, bbox5, 5 * sizeof(double));
        int s = t->age % RING;
        t->obs_age[s] = t->age; memcpy(t->obs_box[s], bbox5, 5 * sizeof(double));
        t->n_obs += 1;
        t->tsu = 0; t->hits += 1; t->hit_streak += 1;
        double z[4]; bbox_to_z(bbox5, z);
        kf7_update(&t->kf, z);
    } else {
        kf7_update(&t->kf, NULL);
    }
    if (has_tid) t->tracklab_id = tid;
}

static void kbt_predict(orc_kbt *t, double *pos)   /* ocsort.py:150-163 */
{
    if ((t->kf.x[6] + t->kf.x[2]) <= 0) t->kf.x[6] *= 0.0;
    kf7_predict(&t->kf);
    t->age += 1;
    if (t->tsu > 0) t->hit_streak = 0;
    t->tsu += 1;
    x_to_bbox(t->kf.x, pos);
}

orc_kbt *orc_kbt_create(const double *bbox5, double cls, int delta_t)
{ orc_kbt *k = malloc(sizeof(*k)); kbt_init(k, bbox5, cls, delta_t, 0, 0); return k; }
void orc_kbt_destroy(orc_kbt *k) { free(k); }
void orc_kbt_predict(orc_kbt *k, double *pos4) { kbt_predict(k, pos4); }
void orc_kbt_update(orc_kbt *k, const double *b, double cls) { kbt_update(k, b, cls, 0, 0); }
void orc_kbt_state(const orc_kbt *k, double *x7, double *P49, double *vel2)
{ memcpy(x7, k->kf.x, 56); memcpy(P49, k->kf.P, 392); vel2[0] = k->has_vel ? k->vel[0] : 0; vel2[1] = k->has_vel ? k->vel[1] : 0; }

/* ------------------------------------------------------------------ OCSort, ocsort.py:185-334 */
struct orc_ocsort {
    double det_thresh, iou_threshold, inertia;
    int max_age, min_hits, delta_t, asso_func, use_byte;
    int64_t frame_count, count;     /* KalmanBoxTracker.count, reset in __init__ (ocsort.py:201) */
    orc_kbt **trk; int n, cap;
};

orc_ocsort *orc_ocsort_create(double det_thresh, int max_age, int min_hits, double iou_threshold,
                              int delta_t, int asso_func, double inertia, int use_byte)
{
    orc_ocsort *t = calloc(1, sizeof(*t));
    t->det_thresh = det_thresh; t->max_age = max_age; t->min_hits = min_hits; t->iou_threshold = iou_threshold;
    t->delta_t = delta_t; t->asso_func = asso_func; t->inertia = inertia; t->use_byte = use_byte;
    return t;
}
void orc_ocsort_destroy(orc_ocsort *t)
{ if (!t) return; for (int i = 0; i < t->n; ++i) free(t->trk[i]); free(t->trk); free(t); }
int orc_ocsort_num_tracks(const orc_ocsort *t) { return t->n; }
int orc_ocsort_get_track(const orc_ocsort *t, int i, double *x7, double *P49, int64_t *id)
{ if (i < 0 || i >= t->n) return 0; memcpy(x7, t->trk[i]->kf.x, 56); memcpy(P49, t->trk[i]->kf.P, 392); *id = t->trk[i]->id; return 1; }

static void trk_pop(orc_ocsort *t, int i)
{ free(t->trk[i]); memmove(t->trk + i, t->trk + i + 1, sizeof(*t->trk) * (size_t)(t->n - i - 1)); t->n--; }
static void trk_push(orc_ocsort *t, orc_kbt *k)
{ if (t->n == t->cap) { t->cap = t->cap ? 2 * t->cap : 64; t->trk = realloc(t->trk, sizeof(*t->trk) * (size_t)t->cap); } t->trk[t->n++] = k; }

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
/* np.setdiff1d(a, rem): sorted unique of a minus rem */
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
static double mat_max(const double *a, size_t n)     /* ndarray.max(): NaN propagates */
{ double m = a[0]; for (size_t i = 1; i < n; ++i) m = (a[i] > m || isnan(a[i])) ? a[i] : m; return m; }

/* association.py:242-298. dets (N,7) rows (cols 0..5 used: the reference passes dets[:, :-1], so
 * "scores = detections[:,-1]" is the CLASS column), trks (T,4), vel (T,2), kobs (T,5). */
static void associate(const double *dets, int N, const double *trks, int T, double iou_thr,
                      const double *vel, const double *kobs, double vdc_weight,
                      int *matches, int *n_matches, int *um_d, int *n_um_d, int *um_t, int *n_um_t)
{
    *n_matches = 0; *n_um_d = 0; *n_um_t = 0;
    if (T == 0) { for (int d = 0; d < N; ++d) um_d[(*n_um_d)++] = d; return; }
    int64_t *mi_r = malloc(sizeof(int64_t) * (size_t)(N + T + 1)), *mi_c = malloc(sizeof(int64_t) * (size_t)(N + T + 1));
    int n_mi = 0;
    double *iou = malloc(sizeof(double) * (size_t)N * T + 8);
    if (N > 0) {
        double *cost = malloc(sizeof(double) * (size_t)N * T);
        orc_iou_matrix(ORC_IOU, dets, N, 7, trks, T, 4, iou);
        for (int t = 0; t < T; ++t) {
            const double *ko = kobs + t * 5;
            double cx2 = (ko[0] + ko[2]) / 2.0, cy2 = (ko[1] + ko[3]) / 2.0;
            double valid = ko[4] < 0 ? 0.0 : 1.0;
            for (int d = 0; d < N; ++d) {
                const double *de = dets + d * 7;
                double cx1 = (de[0] + de[2]) / 2.0, cy1 = (de[1] + de[3]) / 2.0;
                double dx = cx1 - cx2, dy = cy1 - cy2;
                double norm = sqrt(dx * dx + dy * dy) + 1e-6;
                dx = dx / norm; dy = dy / norm;
                double c = vel[t * 2 + 1] * dx + vel[t * 2 + 0] * dy;
                c = c < -1 ? -1 : (c > 1 ? 1 : c);
                double ang = acos(c);
                ang = (M_PI / 2.0 - fabs(ang)) / M_PI;
                double adc = ((valid * ang) * vdc_weight) * de[5];
                cost[(size_t)d * T + t] = -(iou[(size_t)d * T + t] + adc);
            }
        }
        /* association.py:267-272 */
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
            n_mi = orc_lsa(cost, N, T, mi_r, mi_c);
            if (n_mi < 0) n_mi = 0;
        }
        free(cost);
    }
    for (int d = 0; d < N; ++d) { int f = 0; for (int k = 0; k < n_mi; ++k) if (mi_r[k] == d) { f = 1; break; } if (!f) um_d[(*n_um_d)++] = d; }
    for (int t = 0; t < T; ++t) { int f = 0; for (int k = 0; k < n_mi; ++k) if (mi_c[k] == t) { f = 1; break; } if (!f) um_t[(*n_um_t)++] = t; }
    for (int k = 0; k < n_mi; ++k) {
        if (iou[(size_t)mi_r[k] * T + mi_c[k]] < iou_thr) { um_d[(*n_um_d)++] = (int)mi_r[k]; um_t[(*n_um_t)++] = (int)mi_c[k]; }
        else { matches[2 * *n_matches] = (int)mi_r[k]; matches[2 * *n_matches + 1] = (int)mi_c[k]; (*n_matches)++; }
    }
    free(mi_r); free(mi_c); free(iou);
}

int orc_ocsort_update(orc_ocsort *o, const double *dets_in, int n_in, double *out, int out_cap)
{
    o->frame_count += 1;
    /* ocsort.py:226-231 */
    double *dets = malloc(sizeof(double) * 7 * (size_t)(n_in + 1)), *dets2 = malloc(sizeof(double) * 7 * (size_t)(n_in + 1));
    int N = 0, N2 = 0;
    for (int i = 0; i < n_in; ++i) {
        double c = dets_in[i * 7 + 4];
        if (c > 0.1 && c < o->det_thresh) memcpy(dets2 + 7 * N2++, dets_in + 7 * i, 56);
        if (c > o->det_thresh) memcpy(dets + 7 * N++, dets_in + 7 * i, 56);
    }
    /* ocsort.py:234-244 */
    int T0 = o->n;
    double *trks = malloc(sizeof(double) * 4 * (size_t)(T0 + 1));
    int T = 0;
    for (int t = 0; t < o->n;) {
        double pos[4];
        kbt_predict(o->trk[t], pos);
        if (isnan(pos[0]) || isnan(pos[1]) || isnan(pos[2]) || isnan(pos[3])) { trk_pop(o, t); continue; }
        memcpy(trks + 4 * T, pos, 32); T++; t++;
    }
    /* ocsort.py:246-250 */
    double *vel = malloc(sizeof(double) * 2 * (size_t)(T + 1)), *last_boxes = malloc(sizeof(double) * 5 * (size_t)(T + 1));
    double *kobs = malloc(sizeof(double) * 5 * (size_t)(T + 1));
    for (int t = 0; t < T; ++t) {
        orc_kbt *k = o->trk[t];
        vel[2 * t] = k->has_vel ? k->vel[0] : 0; vel[2 * t + 1] = k->has_vel ? k->vel[1] : 0;
        memcpy(last_boxes + 5 * t, k->last_obs, 40);
        const double *p = NULL;                         /* k_previous_obs, ocsort.py:10-18 */
        if (k->n_obs == 0) { for (int i = 0; i < 5; ++i) kobs[5 * t + i] = -1; }
        else {
            for (int i = 0; i < o->delta_t; ++i) { p = obs_lookup(k, k->age - (o->delta_t - i)); if (p) break; }
            if (!p) p = k->last_obs;
            memcpy(kobs + 5 * t, p, 40);
        }
    }
    int cap = N + T + N2 + 4;
    int *matches = malloc(sizeof(int) * 2 * (size_t)cap), *um_d = malloc(sizeof(int) * (size_t)cap), *um_t = malloc(sizeof(int) * (size_t)cap);
    int nm, nud, nut;
    associate(dets, N, trks, T, o->iou_threshold, vel, kobs, o->inertia, matches, &nm, um_d, &nud, um_t, &nut);
    for (int k = 0; k < nm; ++k) {
        const double *d = dets + 7 * matches[2 * k];
        kbt_update(o->trk[matches[2 * k + 1]], d, d[5], 1, d[6]);
    }
    int64_t *lr = malloc(sizeof(int64_t) * (size_t)cap), *lc = malloc(sizeof(int64_t) * (size_t)cap);
    /* BYTE, ocsort.py:264-282 */
    if (o->use_byte && N2 > 0 && nut > 0) {
        double *ut = malloc(sizeof(double) * 4 * (size_t)nut), *il = malloc(sizeof(double) * (size_t)N2 * nut);
        for (int k = 0; k < nut; ++k) memcpy(ut + 4 * k, trks + 4 * um_t[k], 32);
        orc_iou_matrix(o->asso_func, dets2, N2, 7, ut, nut, 4, il);
        if (mat_max(il, (size_t)N2 * nut) > o->iou_threshold) {
            double *neg = malloc(sizeof(double) * (size_t)N2 * nut);
            for (size_t k = 0; k < (size_t)N2 * nut; ++k) neg[k] = -il[k];
            int nl = orc_lsa(neg, N2, nut, lr, lc);
            int *rem = malloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1)), nrem = 0;
            for (int k = 0; k < nl; ++k) {
                int di = (int)lr[k], ti = um_t[lc[k]];
                if (il[(size_t)lr[k] * nut + lc[k]] < o->iou_threshold) continue;
                const double *d = dets2 + 7 * di;
                kbt_update(o->trk[ti], d, d[5], 1, d[6]);
                rem[nrem++] = ti;
            }
            nut = setdiff_sorted(um_t, nut, rem, nrem);
            free(rem); free(neg);
        }
        free(ut); free(il);
    }
    /* OCR, ocsort.py:284-306 */
    if (nud > 0 && nut > 0) {
        double *ld = malloc(sizeof(double) * 7 * (size_t)nud), *lt = malloc(sizeof(double) * 5 * (size_t)nut);
        double *il = malloc(sizeof(double) * (size_t)nud * nut);
        for (int k = 0; k < nud; ++k) memcpy(ld + 7 * k, dets + 7 * um_d[k], 56);
        for (int k = 0; k < nut; ++k) memcpy(lt + 5 * k, last_boxes + 5 * um_t[k], 40);
        orc_iou_matrix(o->asso_func, ld, nud, 7, lt, nut, 5, il);
        if (mat_max(il, (size_t)nud * nut) > o->iou_threshold) {
            double *neg = malloc(sizeof(double) * (size_t)nud * nut);
            for (size_t k = 0; k < (size_t)nud * nut; ++k) neg[k] = -il[k];
            int nl = orc_lsa(neg, nud, nut, lr, lc);
            int *remd = malloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1)), *remt = malloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1)), nr = 0;
            for (int k = 0; k < nl; ++k) {
                int di = um_d[lr[k]], ti = um_t[lc[k]];
                if (il[(size_t)lr[k] * nut + lc[k]] < o->iou_threshold) continue;
                const double *d = dets + 7 * di;
                kbt_update(o->trk[ti], d, d[5], 1, d[6]);
                remd[nr] = di; remt[nr] = ti; nr++;
            }
            nud = setdiff_sorted(um_d, nud, remd, nr);
            nut = setdiff_sorted(um_t, nut, remt, nr);
            free(remd); free(remt); free(neg);
        }
        free(ld); free(lt); free(il);
    }
    for (int k = 0; k < nut; ++k) kbt_update(o->trk[um_t[k]], NULL, 0, 0, 0);     /* ocsort.py:308-309 */
    for (int k = 0; k < nud; ++k) {                                                /* ocsort.py:312-314 */
        const double *d = dets + 7 * um_d[k];
        orc_kbt *nk = malloc(sizeof(*nk));
        kbt_init(nk, d, d[5], o->delta_t, d[6], o->count++);
        trk_push(o, nk);
    }
    /* ocsort.py:315-331 */
    int rows = 0;
    for (int i = o->n - 1; i >= 0; --i) {
        orc_kbt *k = o->trk[i];
        double d[4];
        if (sum5(k->last_obs) < 0) x_to_bbox(k->kf.x, d); else memcpy(d, k->last_obs, 32);
        if (k->tsu < 1 && (k->hit_streak >= o->min_hits || o->frame_count <= o->min_hits) && rows < out_cap) {
            double *r = out + 8 * rows++;
            r[0] = d[0]; r[1] = d[1]; r[2] = d[2]; r[3] = d[3];
            r[4] = (double)(k->id + 1); r[5] = k->cls; r[6] = k->conf; r[7] = k->tracklab_id;
        }
        if (k->tsu > o->max_age) trk_pop(o, i);
    }
    free(dets); free(dets2); free(trks); free(vel); free(last_boxes); free(kobs);
    free(matches); free(um_d); free(um_t); free(lr); free(lc);
    return rows;
}
