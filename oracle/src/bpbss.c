/* CPU ORACLE (test infrastructure) -- see orc.h.
 * BPBReID-StrongSORT restated from plugins/track/bpbreid_strong_sort/(strong_sort.py and the sort package). */
#include "orc.h"
#include "lapack_order.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define INFTY_COST 1e+5                       /* sort/linear_assignment.py:8 */
static const double CHI2INV95[10] = {0, 3.8415, 5.9915, 7.8147, 9.4877, 11.070, 12.592, 14.067, 15.507, 16.919};  /* kalman_filter.py:9-18 */
static const double W_POS = 1. / 20, W_VEL = 1. / 160;            /* kalman_filter.py:50-51 */

/* ------------------------------------------------------------------ kalman_filter.py */
void orc_kf8_initiate(const double *m, double *mean, double *cov)    /* :53-83 */
{
    for (int i = 0; i < 4; ++i) { mean[i] = m[i]; mean[4 + i] = 0; }
    memset(cov, 0, 64 * sizeof(double));
    double sp = 2 * W_POS * m[3], sv = 10 * W_VEL * m[3];
    for (int i = 0; i < 4; ++i) { cov[i * 9] = sp * sp; cov[(4 + i) * 9] = sv * sv; }
}

void orc_kf8_predict(double *mean, double *cov)    /* :85-119; multi_dot(F, cov, F^T) evaluates F (cov F^T) on an equal-cost tie */
{
    double sp = W_POS * mean[3], sv = W_VEL * mean[3];
    double t[64];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) t[i * 8 + j] = j < 4 ? cov[i * 8 + j] + cov[i * 8 + j + 4] : cov[i * 8 + j];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) cov[i * 8 + j] = i < 4 ? t[i * 8 + j] + t[(i + 4) * 8 + j] : t[i * 8 + j];
    for (int i = 0; i < 4; ++i) { cov[i * 9] += sp * sp; cov[(4 + i) * 9] += sv * sv; mean[i] = mean[i] + mean[i + 4]; }
}

void orc_kf8_project(const double *mean, const double *cov, double conf, double *pm, double *pc)   /* :121-152 */
{
    double s = (1 - conf) * (W_POS * mean[3]);
    for (int i = 0; i < 4; ++i) {
        pm[i] = mean[i];
        for (int j = 0; j < 4; ++j) pc[i * 4 + j] = cov[i * 8 + j] + (i == j ? s * s : 0.0);
    }
}

void orc_kf8_update(double *mean, double *cov, const double *z, double conf)    /* :154-187; library operation order: lapack_order.h */
{
    double pm[4], S[16];
    orc_kf8_project(mean, cov, conf, pm, S);
    lo_kf8_update(mean, cov, z, pm, S);
}

void orc_kf8_gating(const double *mean, const double *cov, const double *meas, int n, int only_position, double *out)   /* :189-227 */
{
    double pm[4], S[16];
    orc_kf8_project(mean, cov, 0.0, pm, S);
    lo_kf8_gating(pm, S, only_position ? 2 : 4, meas, n, out);
}

/* ------------------------------------------------------------------ nn_matching.py:99-135 (+ restated torchreid fn) */
void orc_partdist_f32(const float *q, const uint8_t *qvis, int T, const float *g, const uint8_t *gvis,
                      int N, int K, int D, double *out)
{
    float *qn = malloc(sizeof(float) * (size_t)T * K * D), *gn = malloc(sizeof(float) * (size_t)N * K * D);
    float *qs = malloc(sizeof(float) * (size_t)T * K), *gs = malloc(sizeof(float) * (size_t)N * K);
    for (int pass = 0; pass < 2; ++pass) {
        const float *src = pass ? g : q; float *dst = pass ? gn : qn; float *sq = pass ? gs : qs; int R = (pass ? N : T) * K;
        for (int r = 0; r < R; ++r) {
            float ss = 0;
            for (int d = 0; d < D; ++d) ss += src[(size_t)r * D + d] * src[(size_t)r * D + d];
            float nrm = sqrtf(ss); if (nrm < 1e-12f) nrm = 1e-12f;        /* F.normalize eps */
            float s2 = 0;
            for (int d = 0; d < D; ++d) { float v = src[(size_t)r * D + d] / nrm; dst[(size_t)r * D + d] = v; s2 += v * v; }
            sq[r] = s2;
        }
    }
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < N; ++n) {
            float sum = 0; int cnt = 0;
            for (int k = 0; k < K; ++k) {
                const float *a = qn + ((size_t)t * K + k) * D, *b = gn + ((size_t)n * K + k) * D;
                float dot = 0;
                for (int d = 0; d < D; ++d) dot += a[d] * b[d];
                float d2 = qs[t * K + k] - 2 * dot + gs[n * K + k];
                if (d2 < 0) d2 = 0;
                float dist = sqrtf(d2);
                if (qvis[t * K + k] && gvis[n * K + k]) { sum += dist; cnt++; }
            }
            float pair = cnt ? sum / (float)cnt : -1.0f;
            out[(size_t)t * N + n] = (double)(pair / 2);
        }
    free(qn); free(gn); free(qs); free(gs);
}

/* ------------------------------------------------------------------ tracker */
enum { ST_TENTATIVE = 0, ST_CONFIRMED = 1, ST_DELETED = 2 };
typedef struct {
    int64_t track_id;
    int hits, age, tsu, state;
    double mean[8], cov[64];
    float *feat; uint8_t *fvis;
    /* last_detection */
    int64_t det_id; int matched_name; double matched_dist;
    int pred_valid; double pred_ltwh[4];
    double kp[51];             /* last_detection.keypoints */
} trk_t;

struct orc_bpbss {
    orc_bpbss_cfg c; int K, D;
    trk_t *trk; int n, cap;
    int64_t next_id;
};

orc_bpbss *orc_bpbss_create(const orc_bpbss_cfg *cfg, int K, int D)
{ orc_bpbss *t = calloc(1, sizeof(*t)); t->c = *cfg; t->K = K; t->D = D; t->next_id = 1; return t; }
void orc_bpbss_destroy(orc_bpbss *t)
{ if (!t) return; for (int i = 0; i < t->n; ++i) { free(t->trk[i].feat); free(t->trk[i].fvis); } free(t->trk); free(t); }
int orc_bpbss_num_tracks(const orc_bpbss *t) { return t->n; }
int orc_bpbss_get_track(const orc_bpbss *t, int i, int64_t *track_id, double *mean8, double *cov64, float *feat, uint8_t *fvis)
{
    if (i < 0 || i >= t->n) return 0;
    const trk_t *k = &t->trk[i];
    *track_id = k->track_id; memcpy(mean8, k->mean, 64); memcpy(cov64, k->cov, 512);
    memcpy(feat, k->feat, sizeof(float) * (size_t)t->K * t->D); memcpy(fvis, k->fvis, (size_t)t->K);
    return 1;
}

static void to_xyah(const double *ltwh, double *o)        /* detection.py:51-58 */
{ o[0] = ltwh[0] + ltwh[2] / 2; o[1] = ltwh[1] + ltwh[3] / 2; o[2] = ltwh[2] / ltwh[3]; o[3] = ltwh[3]; }
static void trk_to_ltwh(const trk_t *k, double *o)         /* track.py:97-100 */
{ double w = k->mean[2] * k->mean[3]; o[0] = k->mean[0] - w / 2; o[1] = k->mean[1] - k->mean[3] / 2; o[2] = w; o[3] = k->mean[3]; }

/* sort/iou_matching.py:7-39 */
static double iou_ltwh(const double *b, const double *c)
{
    double btl0 = b[0], btl1 = b[1], bbr0 = b[0] + b[2], bbr1 = b[1] + b[3];
    double ctl0 = c[0], ctl1 = c[1], cbr0 = c[0] + c[2], cbr1 = c[1] + c[3];
    double tl0 = btl0 > ctl0 ? btl0 : ctl0, tl1 = btl1 > ctl1 ? btl1 : ctl1;
    double br0 = bbr0 < cbr0 ? bbr0 : cbr0, br1 = bbr1 < cbr1 ? bbr1 : cbr1;
    double w = br0 - tl0, h = br1 - tl1;
    w = w > 0. ? w : 0.; h = h > 0. ? h : 0.;
    double ai = w * h;
    return ai / (b[2] * b[3] + c[2] * c[3] - ai);
}

/* sort/oks_matching.py:7-92: similarity of track keypoints `kp` (17,3) to candidate `c` (17,3) */
static const double KAPPA[17] = {0.026, 0.025, 0.025, 0.035, 0.035, 0.079, 0.079, 0.072, 0.072, 0.062, 0.062, 0.107, 0.107, 0.087, 0.087, 0.089, 0.089};
static double oks_scale(const double *kp, int *nvis_out)
{
    const double c45 = 0.7071067811865476, s45 = 0.7071067811865475;     /* np.cos/np.sin(np.deg2rad(45)) */
    double tl[2] = {INFINITY, INFINITY}, br[2] = {-INFINITY, -INFINITY}, ttl[2] = {INFINITY, INFINITY}, tbr[2] = {-INFINITY, -INFINITY};
    double tl4[2] = {INFINITY, INFINITY}, br4[2] = {-INFINITY, -INFINITY}, ttl4[2] = {INFINITY, INFINITY}, tbr4[2] = {-INFINITY, -INFINITY};
    int nvis = 0;
    for (int k = 0; k < 17; ++k) {
        double x = kp[3 * k], y = kp[3 * k + 1];
        double r[2] = {c45 * x + (-s45) * y, s45 * x + c45 * y};
        double p[2] = {x, y};
        int v = kp[3 * k + 2] > 0.0;
        nvis += v;
        for (int a = 0; a < 2; ++a) {
            if (p[a] < ttl[a]) ttl[a] = p[a]; if (p[a] > tbr[a]) tbr[a] = p[a];
            if (r[a] < ttl4[a]) ttl4[a] = r[a]; if (r[a] > tbr4[a]) tbr4[a] = r[a];
            if (v) { if (p[a] < tl[a]) tl[a] = p[a]; if (p[a] > br[a]) br[a] = p[a];
                     if (r[a] < tl4[a]) tl4[a] = r[a]; if (r[a] > br4[a]) br4[a] = r[a]; }
        }
    }
    *nvis_out = nvis;
    double area = (br[0] - tl[0]) * (br[1] - tl[1]), total_area = (tbr[0] - ttl[0]) * (tbr[1] - ttl[1]);
    double area45 = (br4[0] - tl4[0]) * (br4[1] - tl4[1]), total45 = (tbr4[0] - ttl4[0]) * (tbr4[1] - ttl4[1]);
    double f1 = area > 0.1 ? total_area / area : INFINITY, f2 = area45 > 0.1 ? total45 / area45 : INFINITY;
    double factor = sqrt(f1 < f2 ? f1 : f2);
    double fc = factor < 5.0 ? factor : 5.0;
    double scale = sqrt(area) * fc;
    if (scale < 0.1) scale = NAN;
    return scale;
}
static double oks_one(const double *kp, double scale, int nvis, const double *c)
{
    double sum = 0;
    for (int k = 0; k < 17; ++k) {
        double dx = kp[3 * k] - c[3 * k], dy = kp[3 * k + 1] - c[3 * k + 1];
        double d = sqrt(dx * dx + dy * dy);
        double e = exp(-(d * d) / (2 * (scale * scale) * (KAPPA[k] * KAPPA[k])));
        sum += e * (kp[3 * k + 2] > 0.0 ? 1.0 : 0.0);
    }
    return sum / (double)nvis;
}

/* iou_cost / oks_cost as (T x N) matrices (iou_matching.py:42-78 without the time_since_update row rule, oks_matching.py:95-128) */
void orc_iou_ltwh_cost(const double *trk, int T, const double *det, int N, double *out)
{
    for (int t = 0; t < T; ++t) for (int j = 0; j < N; ++j) out[(size_t)t * N + j] = 1. - iou_ltwh(trk + 4 * t, det + 4 * j);
}
void orc_oks_cost(const double *tkp, int T, const double *dkp, int N, double *out)
{
    for (int t = 0; t < T; ++t) {
        int nvis; double sc = oks_scale(tkp + 51 * (size_t)t, &nvis);
        for (int j = 0; j < N; ++j) out[(size_t)t * N + j] = 1.0 - oks_one(tkp + 51 * (size_t)t, sc, nvis, dkp + 51 * (size_t)j);
    }
}

/* sort/linear_assignment.py:11-73. cost (nt, nd) un-thresholded. outputs index lists into trk_idx/det_idx */
static void min_cost_matching(const double *cost, int nt, int nd, double max_distance, const int *trk_idx, const int *det_idx,
                              int *m_t, int *m_d, int *m_row, int *m_col, int *nm, int *um_t, int *n_um_t, int *um_d, int *n_um_d)
{
    *nm = 0; *n_um_t = 0; *n_um_d = 0;
    if (nd == 0 || nt == 0) {
        for (int i = 0; i < nt; ++i) um_t[(*n_um_t)++] = trk_idx[i];
        for (int j = 0; j < nd; ++j) um_d[(*n_um_d)++] = det_idx[j];
        return;
    }
    double *th = malloc(sizeof(double) * (size_t)nt * nd);
    for (size_t k = 0; k < (size_t)nt * nd; ++k) th[k] = cost[k] > max_distance ? max_distance + 1e-5 : cost[k];
    int kmax = nt < nd ? nt : nd;
    int64_t *r = malloc(sizeof(int64_t) * (size_t)(kmax + 1)), *c = malloc(sizeof(int64_t) * (size_t)(kmax + 1));
    int np = orc_lsa(th, nt, nd, r, c);
    if (np < 0) np = 0;
    for (int col = 0; col < nd; ++col) { int f = 0; for (int k = 0; k < np; ++k) if (c[k] == col) { f = 1; break; } if (!f) um_d[(*n_um_d)++] = det_idx[col]; }
    for (int row = 0; row < nt; ++row) { int f = 0; for (int k = 0; k < np; ++k) if (r[k] == row) { f = 1; break; } if (!f) um_t[(*n_um_t)++] = trk_idx[row]; }
    for (int k = 0; k < np; ++k) {
        if (th[(size_t)r[k] * nd + c[k]] > max_distance) { um_t[(*n_um_t)++] = trk_idx[r[k]]; um_d[(*n_um_d)++] = det_idx[c[k]]; }
        else { m_t[*nm] = trk_idx[r[k]]; m_d[*nm] = det_idx[c[k]]; m_row[*nm] = (int)r[k]; m_col[*nm] = (int)c[k]; (*nm)++; }
    }
    free(th); free(r); free(c);
}

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

int orc_bpbss_update(orc_bpbss *T, const int64_t *ids_in, const double *ltwh_in, const float *emb_in, const uint8_t *vis_in,
                     const double *conf_in, const double *classes, int n_in, orc_bpbss_row *out)
{ return orc_bpbss_update_kp(T, ids_in, ltwh_in, emb_in, vis_in, conf_in, classes, NULL, n_in, out); }

int orc_bpbss_update_kp(orc_bpbss *T, const int64_t *ids_in, const double *ltwh_in, const float *emb_in, const uint8_t *vis_in,
                        const double *conf_in, const double *classes, const double *kps_in, int n_in, orc_bpbss_row *out)
{
    const int K = T->K, D = T->D;
    const size_t FD = (size_t)K * D;
    /* filter_detections, strong_sort.py:143-147 */
    int *sel = malloc(sizeof(int) * (size_t)(n_in + 1)); int N = 0;
    for (int i = 0; i < n_in; ++i) if (conf_in[i] > T->c.min_bbox_confidence) sel[N++] = i;
    /* Tracker.predict, tracker.py:92-99 -> track.py:128-135 */
    for (int t = 0; t < T->n; ++t) {
        trk_t *k = &T->trk[t];
        if (k->tsu < T->c.max_kalman_prediction_without_update) orc_kf8_predict(k->mean, k->cov);
        k->age += 1; k->tsu += 1;
    }
    int rows = 0;
    if (N > 0) {
        const int NT = T->n;
        double *xyah = malloc(sizeof(double) * 4 * (size_t)N), *dltwh = malloc(sizeof(double) * 4 * (size_t)N);
        float *demb = malloc(sizeof(float) * FD * (size_t)N); uint8_t *dvis = malloc((size_t)K * N);
        int *d_mname = calloc((size_t)N, sizeof(int)); double *d_mdist = calloc((size_t)N, sizeof(double));
        for (int j = 0; j < N; ++j) {
            memcpy(dltwh + 4 * j, ltwh_in + 4 * sel[j], 32);
            to_xyah(dltwh + 4 * j, xyah + 4 * j);
            memcpy(demb + FD * j, emb_in + FD * sel[j], sizeof(float) * FD);
            memcpy(dvis + (size_t)K * j, vis_in + (size_t)K * sel[j], (size_t)K);
        }
        int cap = NT + N + 2;
        int *cand = malloc(sizeof(int) * (size_t)cap), *alld = malloc(sizeof(int) * (size_t)cap);
        int *m_t = malloc(sizeof(int) * 2 * (size_t)cap), *m_d = malloc(sizeof(int) * 2 * (size_t)cap);
        int *m_row = malloc(sizeof(int) * (size_t)cap), *m_col = malloc(sizeof(int) * (size_t)cap);
        int *um_ta = malloc(sizeof(int) * (size_t)cap), *um_da = malloc(sizeof(int) * (size_t)cap);
        int *um_tb = malloc(sizeof(int) * (size_t)cap), *um_db = malloc(sizeof(int) * (size_t)cap);
        int *um_t = malloc(sizeof(int) * 2 * (size_t)cap);
        int nm = 0, n_umt = 0, n_umd = 0;
        for (int j = 0; j < N; ++j) alld[j] = j;
        /* full T x N reid matrix once (rows of metric.distance are independent per track) */
        float *tf = malloc(sizeof(float) * FD * (size_t)(NT + 1)); uint8_t *tv = malloc((size_t)K * (NT + 1));
        for (int t = 0; t < NT; ++t) { memcpy(tf + FD * t, T->trk[t].feat, sizeof(float) * FD); memcpy(tv + (size_t)K * t, T->trk[t].fvis, (size_t)K); }
        double *reid = malloc(sizeof(double) * (size_t)(NT + 1) * N), *gate = malloc(sizeof(double) * (size_t)(NT + 1) * N);
        double *stc = malloc(sizeof(double) * (size_t)(NT + 1) * N);
        orc_partdist_f32(tf, tv, NT, demb, dvis, N, K, D, reid);
        const int gdim = T->c.only_position ? 2 : 4;
        for (int t = 0; t < NT; ++t) orc_kf8_gating(T->trk[t].mean, T->trk[t].cov, xyah, N, T->c.only_position, gate + (size_t)t * N);
        if (T->c.motion_criterium == 1 && kps_in) {      /* oks_cost, oks_matching.py:95-128 */
            for (int t = 0; t < NT; ++t) {
                int nvis; double sc = oks_scale(T->trk[t].kp, &nvis);
                for (int j = 0; j < N; ++j) stc[(size_t)t * N + j] = 1.0 - oks_one(T->trk[t].kp, sc, nvis, kps_in + 51 * (size_t)sel[j]);
            }
        } else
        for (int t = 0; t < NT; ++t) { double tl[4]; trk_to_ltwh(&T->trk[t], tl); for (int j = 0; j < N; ++j) stc[(size_t)t * N + j] = 1. - iou_ltwh(tl, dltwh + 4 * j); }

        if (T->c.matching_strategy == 0) {
            /* ---- strong_sort_matching, tracker.py:242-333 ---- */
            int nc = 0, nu = 0; int *unconf = malloc(sizeof(int) * (size_t)cap);
            for (int t = 0; t < NT; ++t) { if (T->trk[t].state == ST_CONFIRMED) cand[nc++] = t; else unconf[nu++] = t; }
            double *cm = malloc(sizeof(double) * (size_t)(nc + 1) * N);
            for (int r = 0; r < nc; ++r)        /* gate_cost_matrix, linear_assignment.py:132-175 */
                for (int j = 0; j < N; ++j) {
                    double c = reid[(size_t)cand[r] * N + j], g = gate[(size_t)cand[r] * N + j];
                    if (g > CHI2INV95[gdim]) c = INFTY_COST;
                    cm[(size_t)r * N + j] = T->c.mc_lambda * c + (1 - T->c.mc_lambda) * g;
                }
            int nma = 0, n_uta = 0, n_uda = 0;
            min_cost_matching(cm, nc, N, T->c.max_dist, cand, alld, m_t, m_d, m_row, m_col, &nma, um_ta, &n_uta, um_da, &n_uda);
            if (nc > 0) for (int k = 0; k < nma; ++k) { d_mname[m_d[k]] = 1; d_mdist[m_d[k]] = cm[(size_t)m_row[k] * N + m_col[k]]; }
            /* matching_cascade: unmatched_tracks = list(set(track_indices) - matched) -> ascending */
            n_uta = 0;
            if (orc_get_python_set_order()) n_uta = orc_pyset_difference_order(cand, nc, m_t, nma, um_ta);
            else for (int r = 0; r < nc; ++r) { int f = 0; for (int k = 0; k < nma; ++k) if (m_t[k] == cand[r]) { f = 1; break; } if (!f) um_ta[n_uta++] = cand[r]; }
            /* stage B candidates */
            int nb = 0; int *bc = malloc(sizeof(int) * (size_t)cap);
            for (int k = 0; k < nu; ++k) bc[nb++] = unconf[k];
            int keep = 0;
            for (int k = 0; k < n_uta; ++k) { if (T->trk[um_ta[k]].tsu == 1) bc[nb++] = um_ta[k]; else um_ta[keep++] = um_ta[k]; }
            n_uta = keep;
            double *cb = malloc(sizeof(double) * (size_t)(nb + 1) * (n_uda + 1));
            for (int r = 0; r < nb; ++r) for (int j = 0; j < n_uda; ++j) cb[(size_t)r * n_uda + j] = stc[(size_t)bc[r] * N + um_da[j]];
            int nmb = 0, n_utb = 0, n_udb = 0;
            min_cost_matching(cb, nb, n_uda, (T->c.motion_criterium == 1 ? T->c.max_oks_distance : T->c.max_iou_distance), bc, um_da, m_t + nma, m_d + nma, m_row, m_col, &nmb, um_tb, &n_utb, um_db, &n_udb);
            if (nb > 0 && n_uda > 0) for (int k = 0; k < nmb; ++k) { d_mname[m_d[nma + k]] = 2; d_mdist[m_d[nma + k]] = cb[(size_t)m_row[k] * n_uda + m_col[k]]; }
            nm = nma + nmb;
            for (int k = 0; k < n_uta; ++k) um_t[n_umt++] = um_ta[k];
            for (int k = 0; k < n_utb; ++k) um_t[n_umt++] = um_tb[k];
            n_umd = n_udb; memcpy(um_da, um_db, sizeof(int) * (size_t)n_udb);
            free(unconf); free(cm); free(bc); free(cb);
        } else {
            /* ---- bot_sort_matching, tracker.py:335-363 with _full_cost_metric :169-240 ---- */
            for (int t = 0; t < NT; ++t) cand[t] = t;
            double *cm = malloc(sizeof(double) * (size_t)(NT + 1) * N);
            const double GT = sqrt(CHI2INV95[gdim]);
            const double wsum = T->c.w_kfgd + T->c.w_reid + T->c.w_st;
            for (int t = 0; t < NT; ++t) for (int j = 0; j < N; ++j) {
                size_t e = (size_t)t * N + j;
                double pos = sqrt(gate[e]) / (GT * T->c.gating_thres_factor);
                int pos_gate = T->c.w_kfgd > 0 ? pos > 1.0 : 0;
                int app_gate = T->c.w_reid > 0 ? reid[e] > T->c.max_dist : 0;
                int st_gate = T->c.w_st > 0 ? stc[e] > (T->c.motion_criterium == 1 ? T->c.max_oks_distance : T->c.max_iou_distance) : 0;
                double c = (T->c.w_kfgd * pos + T->c.w_reid * reid[e] + stc[e] * T->c.w_st) / wsum;
                int m = T->c.w_kfgd > 0 ? (pos_gate || app_gate)          /* np.logical_or(a, b, out=st_gate): st_gate is the out= array */
                                        : (T->c.w_st > 0 ? (app_gate || st_gate) : app_gate);
                cm[e] = m ? INFTY_COST : c;
            }
            int n_uda = 0, n_uta = 0;
            min_cost_matching(cm, NT, N, T->c.max_dist, cand, alld, m_t, m_d, m_row, m_col, &nm, um_ta, &n_uta, um_da, &n_uda);
            if (NT > 0) for (int k = 0; k < nm; ++k) { d_mname[m_d[k]] = 1; d_mdist[m_d[k]] = cm[(size_t)m_row[k] * N + m_col[k]]; }
            for (int t = 0; t < NT; ++t) { int f = 0; for (int k = 0; k < nm; ++k) if (m_t[k] == t) { f = 1; break; } if (!f) um_t[n_umt++] = t; }
            n_umd = n_uda;
            free(cm);
        }
        /* list(set(...)) of small ints iterates ascending */
        qsort(um_t, (size_t)n_umt, sizeof(int), cmp_int);
        { int k2 = 0; for (int k = 0; k < n_umt; ++k) if (k == 0 || um_t[k] != um_t[k - 1]) um_t[k2++] = um_t[k]; n_umt = k2; }

        /* ---- Tracker.update, tracker.py:134-167 ---- */
        for (int k = 0; k < nm; ++k) {          /* Track.update, track.py:137-174 */
            trk_t *tr = &T->trk[m_t[k]]; int j = m_d[k];
            tr->det_id = ids_in[sel[j]]; tr->matched_name = d_mname[j]; tr->matched_dist = d_mdist[j];
            trk_to_ltwh(tr, tr->pred_ltwh); tr->pred_valid = 1;
            if (kps_in) memcpy(tr->kp, kps_in + 51 * (size_t)sel[j], sizeof(tr->kp));
            orc_kf8_update(tr->mean, tr->cov, xyah + 4 * j, conf_in[sel[j]]);
            const float a_t = (float)T->c.ema_alpha, a_d = (float)(1 - T->c.ema_alpha);
            for (int p = 0; p < K; ++p) {
                int tvv = tr->fvis[p] != 0, dvv = dvis[(size_t)K * j + p] != 0;
                int both = tvv && dvv, x = tvv != dvv;
                float et = (float)both * a_t + (float)(x && tvv);
                float ed = (float)both * a_d + (float)(x && dvv);
                float *f = tr->feat + (size_t)p * D; const float *df = demb + FD * j + (size_t)p * D;
                if (et == 0.f && ed == 0.f) { for (int d = 0; d < D; ++d) f[d] = 1.f; }
                else for (int d = 0; d < D; ++d) { float a = et * f[d]; float b = ed * df[d]; f[d] = a + b; }
                tr->fvis[p] = (uint8_t)(tvv || dvv);
            }
            tr->hits += 1; tr->tsu = 0;
            if (tr->state == ST_TENTATIVE && tr->hits >= T->c.n_init) tr->state = ST_CONFIRMED;
        }
        for (int k = 0; k < n_umt; ++k) {       /* mark_missed, track.py:181-187 */
            trk_t *tr = &T->trk[um_t[k]];
            if (tr->state == ST_TENTATIVE) tr->state = ST_DELETED;
            else if (tr->tsu > T->c.max_age) tr->state = ST_DELETED;
        }
        for (int k = 0; k < n_umd; ++k) {       /* _initiate_track, tracker.py:427-441 */
            int j = um_da[k];
            if (T->n == T->cap) { T->cap = T->cap ? 2 * T->cap : 64; T->trk = realloc(T->trk, sizeof(trk_t) * (size_t)T->cap); }
            trk_t *tr = &T->trk[T->n++];
            memset(tr, 0, sizeof(*tr));
            tr->track_id = T->next_id++; tr->hits = 1; tr->age = 1; tr->tsu = 0; tr->state = ST_TENTATIVE;
            tr->feat = malloc(sizeof(float) * FD); tr->fvis = malloc((size_t)K);
            memcpy(tr->feat, demb + FD * j, sizeof(float) * FD); memcpy(tr->fvis, dvis + (size_t)K * j, (size_t)K);
            orc_kf8_initiate(xyah + 4 * j, tr->mean, tr->cov);
            tr->det_id = ids_in[sel[j]]; tr->matched_name = d_mname[j]; tr->matched_dist = d_mdist[j]; tr->pred_valid = 0;
            if (kps_in) memcpy(tr->kp, kps_in + 51 * (size_t)sel[j], sizeof(tr->kp));
            if (tr->hits >= T->c.n_init) tr->state = ST_CONFIRMED;
        }
        { int k2 = 0;                            /* drop deleted (stable) */
          for (int t = 0; t < T->n; ++t) { if (T->trk[t].state == ST_DELETED) { free(T->trk[t].feat); free(T->trk[t].fvis); } else { if (k2 != t) T->trk[k2] = T->trk[t]; k2++; } }
          T->n = k2; }
        free(xyah); free(dltwh); free(demb); free(dvis); free(d_mname); free(d_mdist); free(cand); free(alld);
        free(m_t); free(m_d); free(m_row); free(m_col); free(um_ta); free(um_da); free(um_tb); free(um_db); free(um_t);
        free(tf); free(tv); free(reid); free(gate); free(stc);
    }
    /* outputs, strong_sort.py:93-141 */
    for (int t = 0; t < T->n; ++t) {
        trk_t *k = &T->trk[t];
        if (k->state != ST_CONFIRMED || k->tsu > 0) continue;
        orc_bpbss_row *r = &out[rows++];
        r->det_id = k->det_id; r->track_id = k->track_id;
        trk_to_ltwh(k, r->kf_ltwh);
        r->pred_valid = k->pred_valid;
        for (int i = 0; i < 4; ++i) r->pred_ltwh[i] = k->pred_valid ? k->pred_ltwh[i] : NAN;
        r->matched_name = k->matched_name; r->matched_dist = k->matched_name ? k->matched_dist : NAN;
        r->hits = k->hits; r->age = k->age; r->tsu = k->tsu; r->state = k->state;
    }
    free(sel);
    return rows;
}
