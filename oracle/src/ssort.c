/*
 * oracle/src/ssort.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Plain StrongSORT: plugins/track/strong_sort/(strong_sort.py and the sort package) restated in C.
 *
 *   update()                      strong_sort.py:41-84      (xyxy -> xywh -> tlwh(float32) Detections, predict, update, outputs)
 *   Tracker.predict/update/_match sort/tracker.py:53-58, :81-114, :152-188
 *   matching_cascade              sort/linear_assignment.py:75-128 (single level: the per-age filter is commented out, :116-119)
 *   min_cost_matching             sort/linear_assignment.py:11-72
 *   gate_cost_matrix              sort/linear_assignment.py:131-174
 *   iou / iou_cost                sort/iou_matching.py:7-39, :42-82 (time_since_update > 1 -> INFTY_COST row)
 *   Track.{__init__,predict,update,update_kf,mark_missed}   sort/track.py:69-98, :255-301
 *   KalmanFilter                  sort/kalman_filter.py:50-214 (position/velocity noise relative to x, y, h; aspect-ratio noise 1*a / 0.1*a;
 *                                 NSA measurement noise (1-conf) * [h/20, h/20, 1e-1, h/20])
 *   NearestNeighborDistanceMetric sort/nn_matching.py:94-161 (cosine, budget, partial_fit keeps confirmed targets only)
 *
 * dtype trail that the arithmetic follows (numpy 2 / NEP 50 promotion, the version the golden vectors were made with):
 *   Detection.tlwh and .feature are float32 (sort/detection.py:33-36); to_xyah() therefore works in float32;
 *   Track.__init__ -> kf.initiate(float32 xyah): mean, std and covariance stay float32 until the first predict(), whose
 *   process noise is consequently squared in float32 as well; after that everything is float64;
 *   iou(): candidates are float32 -> their bottom-right corner and area are float32 sums/products.
 */
#include "orc.h"
#include "lapack_order.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define INFTY_COST 1e5
static const double CHI2INV95_4 = 9.4877;
static const double WP = 1. / 20, WV = 1. / 160;
enum { ST_TENTATIVE = 1, ST_CONFIRMED = 2, ST_DELETED = 3 };

typedef struct {
    double mean[8], cov[64];
    int f32_state;                /* state still as produced by initiate(float32 measurement) */
    int64_t id;
    int class_id; double conf, tracklab_id;
    int hits, age, tsu, state, uwa;
    float *feat;                  /* features[-1] (D) */
    float *gal; int glen;         /* metric.samples[id]: glen rows of D, oldest first */
} strk;

struct orc_ssort {
    orc_ssort_cfg c; int D;
    strk *trk; int n, cap;
    int64_t next_id;
};

orc_ssort *orc_ssort_create(const orc_ssort_cfg *c, int D)
{
    orc_ssort *t = calloc(1, sizeof(*t));
    t->c = *c; t->D = D; t->next_id = 1;
    return t;
}
static void trk_free(strk *k) { free(k->feat); free(k->gal); }
void orc_ssort_destroy(orc_ssort *t)
{
    if (!t) return;
    for (int i = 0; i < t->n; ++i) trk_free(&t->trk[i]);
    free(t->trk); free(t);
}

/* ------------------------------------------------------------------ Kalman filter (sort/kalman_filter.py) */
static void kf_initiate_f32(const float *m, double *mean, double *cov)      /* :55-83, float32 arithmetic */
{
    const float std[8] = {(float)(2 * WP) * m[0], (float)(2 * WP) * m[1], m[2], (float)(2 * WP) * m[3],
                          (float)(10 * WV) * m[0], (float)(10 * WV) * m[1], (float)0.1 * m[2], (float)(10 * WV) * m[3]};
    memset(cov, 0, 64 * sizeof(double));
    for (int i = 0; i < 4; ++i) { mean[i] = m[i]; mean[4 + i] = 0; }
    for (int i = 0; i < 8; ++i) { float s = std[i] * std[i]; cov[i * 9] = s; }
}
static void kf_predict(strk *k)                                             /* :85-119 */
{
    double q[8];
    const double *m = k->mean;
    if (k->f32_state) {
        const float f[4] = {(float)m[0], (float)m[1], (float)m[2], (float)m[3]};
        const float std[8] = {(float)WP * f[0], (float)WP * f[1], f[2], (float)WP * f[3],
                              (float)WV * f[0], (float)WV * f[1], (float)0.1 * f[2], (float)WV * f[3]};
        for (int i = 0; i < 8; ++i) { float s = std[i] * std[i]; q[i] = s; }
    } else {
        const double std[8] = {WP * m[0], WP * m[1], 1 * m[2], WP * m[3], WV * m[0], WV * m[1], 0.1 * m[2], WV * m[3]};
        for (int i = 0; i < 8; ++i) q[i] = std[i] * std[i];
    }
    /* multi_dot(F, cov, F^T) evaluates F (cov F^T) on the equal-cost tie */
    double t[64], *cov = k->cov;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) t[i * 8 + j] = j < 4 ? cov[i * 8 + j] + cov[i * 8 + j + 4] : cov[i * 8 + j];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) cov[i * 8 + j] = i < 4 ? t[i * 8 + j] + t[(i + 4) * 8 + j] : t[i * 8 + j];
    for (int i = 0; i < 8; ++i) cov[i * 9] += q[i];
    for (int i = 0; i < 4; ++i) k->mean[i] = k->mean[i] + k->mean[i + 4];
    k->f32_state = 0;
}
static void kf_project(const double *mean, const double *cov, double conf, double *pm, double *pc)   /* :121-152 */
{
    const double std[4] = {WP * mean[3], WP * mean[3], 1e-1, WP * mean[3]};
    for (int i = 0; i < 4; ++i) {
        double s = (1 - conf) * std[i];
        pm[i] = mean[i];
        for (int j = 0; j < 4; ++j) pc[i * 4 + j] = cov[i * 8 + j] + (i == j ? s * s : 0.0);
    }
}
static void kf_update(double *mean, double *cov, const double *z, double conf)                    /* :154-187; library operation order: lapack_order.h */
{
    double pm[4], S[16];
    kf_project(mean, cov, conf, pm, S);
    lo_kf8_update(mean, cov, z, pm, S);
}
static void kf_gating(const double *mean, const double *cov, const double *meas, int n, double *out)   /* :189-214, 4 dof */
{
    double pm[4], S[16];
    kf_project(mean, cov, 0.0, pm, S);
    lo_kf8_gating(pm, S, 4, meas, n, out);
}

static void trk_tlwh(const strk *k, double *o)          /* track.py:99-111 */
{ double w = k->mean[2] * k->mean[3]; o[2] = w; o[3] = k->mean[3]; o[0] = k->mean[0] - w / 2; o[1] = k->mean[1] - k->mean[3] / 2; }

/* iou_matching.py:7-39: bbox float64 tlwh, candidate float32 tlwh */
static double iou_f32cand(const double *b, const float *c)
{
    const double bbr0 = b[0] + b[2], bbr1 = b[1] + b[3];
    const float cbr0 = c[0] + c[2], cbr1 = c[1] + c[3];
    const double tl0 = b[0] > (double)c[0] ? b[0] : (double)c[0], tl1 = b[1] > (double)c[1] ? b[1] : (double)c[1];
    const double br0 = bbr0 < (double)cbr0 ? bbr0 : (double)cbr0, br1 = bbr1 < (double)cbr1 ? bbr1 : (double)cbr1;
    double w = br0 - tl0, h = br1 - tl1;
    w = w > 0. ? w : 0.; h = h > 0. ? h : 0.;
    const double ai = w * h;
    const float ac = c[2] * c[3];
    return ai / (b[2] * b[3] + (double)ac - ai);
}

static float norm_f32(const float *x, int D)            /* np.linalg.norm of a float32 vector (float32 accumulation) */
{
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += x[d] * x[d];
    return sqrtf(s);
}

/* linear_assignment.py:11-72. cost (nt, nd), rows = trk_idx entries, cols = det_idx entries */
static void min_cost_matching(const double *cost, int nt, int nd, double max_distance, const int *trk_idx, const int *det_idx,
                              int *m_t, int *m_d, int *nm, int *um_t, int *n_um_t, int *um_d, int *n_um_d)
{
    *nm = 0; *n_um_t = 0; *n_um_d = 0;
    if (nd == 0 || nt == 0) {
        for (int i = 0; i < nt; ++i) um_t[(*n_um_t)++] = trk_idx[i];
        for (int j = 0; j < nd; ++j) um_d[(*n_um_d)++] = det_idx[j];
        return;
    }
    double *cm = malloc(sizeof(double) * (size_t)nt * nd);
    for (size_t k = 0; k < (size_t)nt * nd; ++k) cm[k] = cost[k] > max_distance ? max_distance + 1e-5 : cost[k];
    int64_t *rows = malloc(sizeof(int64_t) * (nt < nd ? nt : nd)), *cols = malloc(sizeof(int64_t) * (nt < nd ? nt : nd));
    int np = orc_lsa(cm, nt, nd, rows, cols);
    if (np < 0) np = 0;
    char *rowm = calloc(nt, 1), *colm = calloc(nd, 1);
    for (int k = 0; k < np; ++k) { rowm[rows[k]] = 1; colm[cols[k]] = 1; }
    for (int j = 0; j < nd; ++j) if (!colm[j]) um_d[(*n_um_d)++] = det_idx[j];
    for (int i = 0; i < nt; ++i) if (!rowm[i]) um_t[(*n_um_t)++] = trk_idx[i];
    for (int k = 0; k < np; ++k) {
        int r = (int)rows[k], c = (int)cols[k];
        if (cm[(size_t)r * nd + c] > max_distance) { um_t[(*n_um_t)++] = trk_idx[r]; um_d[(*n_um_d)++] = det_idx[c]; }
        else { m_t[*nm] = trk_idx[r]; m_d[*nm] = det_idx[c]; (*nm)++; }
    }
    free(cm); free(rows); free(cols); free(rowm); free(colm);
}

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

int orc_ssort_update(orc_ssort *T, const double *dets, const float *feat_in, int N, double *rows_out, int out_cap)
{
    const int D = T->D;
    /* ---- strong_sort.py:43-60: xyxy -> xywh (float64) -> tlwh -> Detection (float32 tlwh, float32 feature) ---- */
    float *tlwh = malloc(sizeof(float) * 4 * (size_t)(N + 1)), *xyah32 = malloc(sizeof(float) * 4 * (size_t)(N + 1));
    double *xyah = malloc(sizeof(double) * 4 * (size_t)(N + 1));
    float *feat = malloc(sizeof(float) * (size_t)(N + 1) * D);
    memcpy(feat, feat_in, sizeof(float) * (size_t)N * D);
    for (int j = 0; j < N; ++j) {
        const double *d = dets + 7 * (size_t)j;
        const double cx = (d[0] + d[2]) / 2, cy = (d[1] + d[3]) / 2, w = d[2] - d[0], h = d[3] - d[1];
        float *b = tlwh + 4 * j;
        b[0] = (float)(cx - w / 2.); b[1] = (float)(cy - h / 2.); b[2] = (float)w; b[3] = (float)h;
        float *z = xyah32 + 4 * j;                              /* detection.py:46-53 in float32 */
        z[0] = b[0] + b[2] / 2; z[1] = b[1] + b[3] / 2; z[2] = b[2] / b[3]; z[3] = b[3];
        for (int q = 0; q < 4; ++q) xyah[4 * j + q] = z[q];
    }
    /* ---- tracker.predict ---- */
    for (int t = 0; t < T->n; ++t) { kf_predict(&T->trk[t]); T->trk[t].age++; T->trk[t].tsu++; }

    const int NT = T->n;
    int *conf_idx = malloc(sizeof(int) * (NT + 1)), *unconf_idx = malloc(sizeof(int) * (NT + 1));
    int nc = 0, nu = 0;
    for (int t = 0; t < NT; ++t) { if (T->trk[t].state == ST_CONFIRMED) conf_idx[nc++] = t; else unconf_idx[nu++] = t; }
    int *alld = malloc(sizeof(int) * (N + 1));
    for (int j = 0; j < N; ++j) alld[j] = j;
    int cap = NT + N + 1;
    int *m_t = malloc(sizeof(int) * cap), *m_d = malloc(sizeof(int) * cap), *um_ta = malloc(sizeof(int) * cap), *um_da = malloc(sizeof(int) * cap);
    int *m_t2 = malloc(sizeof(int) * cap), *m_d2 = malloc(sizeof(int) * cap), *um_tb = malloc(sizeof(int) * cap), *um_db = malloc(sizeof(int) * cap);
    int nma = 0, n_uta = 0, n_uda = 0;
    /* ---- appearance stage over the confirmed tracks (tracker.py:154-171) ---- */
    {
        double *cm = NULL;
        if (nc > 0 && N > 0) {
            /* metric.distance: cosine gallery minimum (nn_matching.py:144-161) */
            size_t grows = 0;
            for (int r = 0; r < nc; ++r) grows += T->trk[conf_idx[r]].glen;
            float *gal = malloc(sizeof(float) * (grows + 1) * D);
            int32_t *offs = malloc(sizeof(int32_t) * (nc + 1));
            offs[0] = 0;
            for (int r = 0; r < nc; ++r) {
                const strk *k = &T->trk[conf_idx[r]];
                memcpy(gal + (size_t)offs[r] * D, k->gal, sizeof(float) * (size_t)k->glen * D);
                offs[r + 1] = offs[r] + k->glen;
            }
            cm = malloc(sizeof(double) * (size_t)nc * N);
            orc_cosine_gallery_min_f32(gal, offs, nc, feat, N, D, cm);
            free(gal); free(offs);
            double *gd = malloc(sizeof(double) * N);
            for (int r = 0; r < nc; ++r) {                     /* gate_cost_matrix, linear_assignment.py:167-174 */
                const strk *k = &T->trk[conf_idx[r]];
                kf_gating(k->mean, k->cov, xyah, N, gd);
                for (int j = 0; j < N; ++j) {
                    double c = cm[(size_t)r * N + j];
                    if (gd[j] > CHI2INV95_4) c = INFTY_COST;
                    cm[(size_t)r * N + j] = T->c.mc_lambda * c + (1 - T->c.mc_lambda) * gd[j];
                }
            }
            free(gd);
        }
        min_cost_matching(cm, nc, N, T->c.max_dist, conf_idx, alld, m_t, m_d, &nma, um_ta, &n_uta, um_da, &n_uda);
        free(cm);
        if (orc_get_python_set_order()) n_uta = orc_pyset_difference_order(conf_idx, nc, m_t, nma, um_ta);      /* linear_assignment.py:126 */
        else qsort(um_ta, n_uta, sizeof(int), cmp_int);        /* list(set(track_indices) - matched): ascending while indices < table size */
    }
    /* ---- IoU stage (tracker.py:173-184) ---- */
    int *cand = malloc(sizeof(int) * cap), ncand = 0, *left = malloc(sizeof(int) * cap), nleft = 0;
    for (int i = 0; i < nu; ++i) cand[ncand++] = unconf_idx[i];
    for (int i = 0; i < n_uta; ++i) { if (T->trk[um_ta[i]].tsu == 1) cand[ncand++] = um_ta[i]; else left[nleft++] = um_ta[i]; }
    int nmb = 0, n_utb = 0, n_udb = 0;
    {
        double *cm = NULL;
        if (ncand > 0 && n_uda > 0) {
            cm = malloc(sizeof(double) * (size_t)ncand * n_uda);
            for (int r = 0; r < ncand; ++r) {
                const strk *k = &T->trk[cand[r]];
                double b[4]; trk_tlwh(k, b);
                for (int j = 0; j < n_uda; ++j)
                    cm[(size_t)r * n_uda + j] = k->tsu > 1 ? INFTY_COST : 1. - iou_f32cand(b, tlwh + 4 * um_da[j]);
            }
        }
        min_cost_matching(cm, ncand, n_uda, T->c.max_iou_dist, cand, um_da, m_t2, m_d2, &nmb, um_tb, &n_utb, um_db, &n_udb);
        free(cm);
    }
    /* ---- Tracker.update (tracker.py:90-104) ---- */
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass ? nmb : nma; const int *mt = pass ? m_t2 : m_t, *md = pass ? m_d2 : m_d;
        for (int q = 0; q < n; ++q) {                           /* Track.update, track.py:267-296 */
            strk *k = &T->trk[mt[q]]; const int j = md[q];
            const double *d = dets + 7 * (size_t)j;
            k->conf = d[4]; k->class_id = (int)d[5];
            kf_update(k->mean, k->cov, xyah + 4 * j, d[4]);
            const float *f = feat + (size_t)j * D;
            const float nf = norm_f32(f, D);
            const float a = (float)T->c.ema_alpha, b1 = (float)(1 - T->c.ema_alpha);
            for (int e = 0; e < D; ++e) k->feat[e] = a * k->feat[e] + b1 * (f[e] / nf);
            const float ns = norm_f32(k->feat, D);
            for (int e = 0; e < D; ++e) k->feat[e] /= ns;
            k->hits++; k->tsu = 0;
            if (k->state == ST_TENTATIVE && k->hits >= T->c.n_init) k->state = ST_CONFIRMED;
            k->tracklab_id = d[6];
        }
    }
    for (int pass = 0; pass < 2; ++pass) {                      /* unmatched tracks: left over from stage a (tsu != 1) + stage b */
        const int n = pass ? n_utb : nleft; const int *ut = pass ? um_tb : left;
        for (int q = 0; q < n; ++q) {
            strk *k = &T->trk[ut[q]];
            if (k->state == ST_TENTATIVE) k->state = ST_DELETED;           /* mark_missed, track.py:298-303 */
            else if (k->tsu > T->c.max_age) k->state = ST_DELETED;
            if (T->c.max_unmatched_preds != 0 && k->uwa < 7) {             /* update_kf, track.py:258-265, detection.py:55-62 */
                double b[4], z[4]; trk_tlwh(k, b);
                z[0] = b[0] + b[2] / 2; z[1] = b[1] + b[3] / 2; z[2] = b[2] / b[3]; z[3] = b[3];
                k->uwa++;
                kf_update(k->mean, k->cov, z, 0.5);
            }
        }
    }
    for (int q = 0; q < n_udb; ++q) {                           /* _initiate_track, tracker.py:190-193; Track.__init__ */
        const int j = um_db[q];
        if (T->n == T->cap) { T->cap = T->cap ? 2 * T->cap : 64; T->trk = realloc(T->trk, sizeof(strk) * T->cap); }
        strk *k = &T->trk[T->n++];
        memset(k, 0, sizeof(*k));
        const double *d = dets + 7 * (size_t)j;
        k->id = T->next_id++; k->class_id = (int)d[5]; k->conf = d[4]; k->tracklab_id = d[6];
        k->hits = 1; k->age = 1; k->tsu = 0; k->uwa = 0; k->state = ST_TENTATIVE;
        k->feat = malloc(sizeof(float) * D);
        const float *f = feat + (size_t)j * D; const float nf = norm_f32(f, D);
        for (int e = 0; e < D; ++e) k->feat[e] = f[e] / nf;
        kf_initiate_f32(xyah32 + 4 * j, k->mean, k->cov); k->f32_state = 1;
    }
    { int k2 = 0;
      for (int t = 0; t < T->n; ++t) { if (T->trk[t].state == ST_DELETED) trk_free(&T->trk[t]); else { if (k2 != t) T->trk[k2] = T->trk[t]; k2++; } }
      T->n = k2; }
    /* ---- metric.partial_fit (tracker.py:106-114, nn_matching.py:124-142) ---- */
    for (int t = 0; t < T->n; ++t) {
        strk *k = &T->trk[t];
        if (k->state != ST_CONFIRMED) continue;
        const int B = T->c.nn_budget;
        if (B > 0 && k->glen == B) { memmove(k->gal, k->gal + D, sizeof(float) * (size_t)(B - 1) * D); k->glen--; }
        k->gal = realloc(k->gal, sizeof(float) * (size_t)(k->glen + 1) * D);
        memcpy(k->gal + (size_t)k->glen * D, k->feat, sizeof(float) * D);
        k->glen++;
    }
    /* ---- outputs (strong_sort.py:62-79, _tlwh_to_xyxy :111-122) ---- */
    int n_out = 0;
    for (int t = 0; t < T->n; ++t) {
        const strk *k = &T->trk[t];
        if (k->state != ST_CONFIRMED || k->tsu > 1) continue;
        if (n_out >= out_cap) break;
        double b[4]; trk_tlwh(k, b);
        int x1 = (int)b[0], x2 = (int)(b[0] + b[2]), y1 = (int)b[1], y2 = (int)(b[1] + b[3]);
        if (x1 < 0) x1 = 0;
        if (x2 > T->c.img_w - 1) x2 = T->c.img_w - 1;
        if (y1 < 0) y1 = 0;
        if (y2 > T->c.img_h - 1) y2 = T->c.img_h - 1;
        double *o = rows_out + 8 * (size_t)n_out++;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = (double)k->id; o[5] = k->class_id; o[6] = k->conf; o[7] = k->tracklab_id;
    }
    free(tlwh); free(xyah32); free(xyah); free(feat); free(conf_idx); free(unconf_idx); free(alld);
    free(m_t); free(m_d); free(um_ta); free(um_da); free(m_t2); free(m_d2); free(um_tb); free(um_db); free(cand); free(left);
    return n_out;
}

int orc_ssort_num_tracks(const orc_ssort *T) { return T->n; }
/* debug dump: ids (n), mean (n,8), cov (n,64), feat (n,D), state (n,5) [hits, age, tsu, state, updates_wo_assignment], gallery rows (n) */
int orc_ssort_tracks(const orc_ssort *T, int64_t *ids, double *mean, double *cov, float *feat, int64_t *state5, int64_t *glen, int cap)
{
    int n = T->n < cap ? T->n : cap;
    for (int t = 0; t < n; ++t) {
        const strk *k = &T->trk[t];
        ids[t] = k->id;
        memcpy(mean + 8 * t, k->mean, sizeof(k->mean)); memcpy(cov + 64 * t, k->cov, sizeof(k->cov));
        memcpy(feat + (size_t)t * T->D, k->feat, sizeof(float) * T->D);
        state5[5 * t] = k->hits; state5[5 * t + 1] = k->age; state5[5 * t + 2] = k->tsu; state5[5 * t + 3] = k->state; state5[5 * t + 4] = k->uwa;
        glen[t] = k->glen;
    }
    return n;
}

/* Tracker.camera_update -> Track.camera_update (sort/tracker.py:66-68, sort/track.py:221-239) with the ECC estimate passed in:
 * warp (2,3) as cv2.findTransformECC returned it (float32 values) after the translation rescaling of Track.ECC (:204-206); a third row
 * [0, 0, 1] is appended, get_matrix (:213-219) replaces it by the identity when ||I - M||_F >= 100, every track's (x1, y1, x2, y2)
 * corners go through it and the mean becomes [cx, cy, w / h, h] in the mean's own dtype. The estimator itself is cv2. */
void orc_ssort_camera_update(orc_ssort *t, const double *warp6)
{
    double M[9] = {warp6[0], warp6[1], warp6[2], warp6[3], warp6[4], warp6[5], 0, 0, 1};
    double d2 = 0;
    for (int i = 0; i < 9; ++i) { const double e = (i % 4 == 0 ? 1.0 : 0.0) - M[i]; d2 += e * e; }
    if (!(sqrt(d2) < 100)) { memset(M, 0, sizeof(M)); M[0] = M[4] = M[8] = 1; }
    for (int i = 0; i < t->n; ++i) {
        strk *k = &t->trk[i];
        double x1, y1, x2, y2;
        if (k->f32_state) {                                   /* to_tlwh / to_tlbr on a float32 mean (track.py:94-113) */
            float r0 = (float)k->mean[0], r1 = (float)k->mean[1], r2 = (float)k->mean[2]; const float r3 = (float)k->mean[3];
            r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
            x1 = r0; y1 = r1; x2 = (float)(r0 + r2); y2 = (float)(r1 + r3);
        } else {
            double r0 = k->mean[0], r1 = k->mean[1], r2 = k->mean[2]; const double r3 = k->mean[3];
            r2 *= r3; r0 -= r2 / 2; r1 -= r3 / 2;
            x1 = r0; y1 = r1; x2 = r0 + r2; y2 = r1 + r3;
        }
        /* matrix @ [x, y, 1]: numpy's 3 x 3 matrix-vector product evaluates fma(m0, x, m1 * y) + m2 * 1 (identified against numpy itself,
         * tools/blas_order_probe.py) */
        const double x1_ = fma(M[0], x1, M[1] * y1) + M[2] * 1.0, y1_ = fma(M[3], x1, M[4] * y1) + M[5] * 1.0;
        const double x2_ = fma(M[0], x2, M[1] * y2) + M[2] * 1.0, y2_ = fma(M[3], x2, M[4] * y2) + M[5] * 1.0;
        const double w = x2_ - x1_, h = y2_ - y1_, cx = x1_ + w / 2, cy = y1_ + h / 2;
        const double nm[4] = {cx, cy, w / h, h};
        for (int q = 0; q < 4; ++q) k->mean[q] = k->f32_state ? (double)(float)nm[q] : nm[q];
    }
}
