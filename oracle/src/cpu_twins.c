/* cpu_twins.c -- TEST INFRASTRUCTURE, part of the CPU oracle (oracle/_build/liborc.so), never linked into libtlk.so.
 *
 * SURVEY.md section 8(b) lists a minimum C-ABI export list "each with a `_cpu` twin used as the on-box CPU baseline".  The twins are these:
 * every function of that list under its libtlk name + `_cpu`, with libtlk's signature (include/tlk_cpu.h), HOST pointers where libtlk takes
 * device pointers, the trailing `hip_stream` argument present and ignored -- so that a harness can call either side through one function
 * pointer type.  Each twin is a thin adapter over the orc_* restatement of the reference (the file:line citations are on those functions, orc.h):
 * single-threaded fp64 / fp32 C, the arithmetic bench.py's `cpu_baseline` times.  tests/test_cpu_twins.py pins every twin to its orc_* function,
 * tests/test_gpu_cpu_twins.py runs libtlk and the twins side by side through the shared signatures. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tlk_cpu.h"
#include "orc.h"

#define BAD(c) do { if (c) return TLK_EINVAL; } while (0)

int tlk_iou_matrix_f64_cpu(int variant, const double *b1, int n, const double *b2, int m, double *out, void *hip_stream)
{
    (void)hip_stream;
    BAD(variant < TLK_IOU || variant > TLK_CT || n < 0 || m < 0 || ((n && m) && (!b1 || !b2 || !out)));
    if (n && m) orc_iou_matrix(variant, b1, n, 4, b2, m, 4, out);
    return TLK_OK;
}

int tlk_lsa_f64_cpu(const double *cost, int batch, int nr, int nc, int32_t *rows, int32_t *cols, int32_t *n_pairs, void *hip_stream)
{
    (void)hip_stream;
    BAD(batch < 0 || nr < 0 || nc < 0 || (batch && (!rows || !cols || !n_pairs)) || (batch && nr && nc && !cost));
    const int k = nr < nc ? nr : nc;
    int64_t *r64 = malloc(sizeof(int64_t) * (size_t)(2 * (k > 0 ? k : 1)));
    if (!r64) return TLK_EINVAL;
    int64_t *c64 = r64 + (k > 0 ? k : 1);
    for (int b = 0; b < batch; ++b) {
        if (k == 0) { n_pairs[b] = 0; continue; }                                  /* an empty problem has an empty assignment */
        const int n = orc_lsa(cost + (size_t)b * nr * nc, nr, nc, r64, c64);      /* pairs sorted by row; -1 infeasible, -2 NaN / -inf */
        n_pairs[b] = n;
        for (int i = 0; i < k; ++i) {
            rows[(size_t)b * k + i] = i < n ? (int32_t)r64[i] : -1;
            cols[(size_t)b * k + i] = i < n ? (int32_t)c64[i] : -1;
        }
    }
    free(r64);
    return TLK_OK;
}

int tlk_lsa_lapjv_limit_f64_cpu(const double *cost, int batch, int nr, int nc, double cost_limit, int32_t *x, int32_t *y, void *hip_stream)
{
    (void)hip_stream;
    BAD(batch < 0 || nr < 0 || nc < 0 || nr + nc > 512 || (batch && (!x || !y)) || (batch && nr && nc && !cost));
    for (int b = 0; b < batch; ++b) orc_lapjv_limit(cost + (size_t)b * nr * nc, nr, nc, cost_limit, x + (size_t)b * nr, y + (size_t)b * nc);
    return TLK_OK;
}

int tlk_kf7_predict_f64_cpu(double *x, double *P, int n, void *hip_stream)
{
    (void)hip_stream;
    BAD(n < 0 || (n && (!x || !P)));
    for (int i = 0; i < n; ++i) orc_kf7_predict(x + 7 * (size_t)i, P + 49 * (size_t)i);
    return TLK_OK;
}

int tlk_kf7_update_f64_cpu(double *x, double *P, const double *z, int n, void *hip_stream)
{
    (void)hip_stream;
    BAD(n < 0 || (n && (!x || !P || !z)));
    for (int i = 0; i < n; ++i) orc_kf7_update(x + 7 * (size_t)i, P + 49 * (size_t)i, z + 4 * (size_t)i);
    return TLK_OK;
}

int tlk_kf8_initiate_f64_cpu(const double *meas_xyah, double *mean, double *cov, int n, void *hip_stream)
{
    (void)hip_stream;
    BAD(n < 0 || (n && (!meas_xyah || !mean || !cov)));
    for (int i = 0; i < n; ++i) orc_kf8_initiate(meas_xyah + 4 * (size_t)i, mean + 8 * (size_t)i, cov + 64 * (size_t)i);
    return TLK_OK;
}

int tlk_kf8_predict_f64_cpu(double *mean, double *cov, int n, void *hip_stream)
{
    (void)hip_stream;
    BAD(n < 0 || (n && (!mean || !cov)));
    for (int i = 0; i < n; ++i) orc_kf8_predict(mean + 8 * (size_t)i, cov + 64 * (size_t)i);
    return TLK_OK;
}

int tlk_kf8_project_f64_cpu(const double *mean, const double *cov, const double *conf, double *pmean, double *pcov, int n, void *hip_stream)
{
    (void)hip_stream;
    BAD(n < 0 || (n && (!mean || !cov || !pmean || !pcov)));
    for (int i = 0; i < n; ++i) orc_kf8_project(mean + 8 * (size_t)i, cov + 64 * (size_t)i, conf ? conf[i] : 0.0, pmean + 4 * (size_t)i, pcov + 16 * (size_t)i);
    return TLK_OK;
}

int tlk_kf8_update_f64_cpu(double *mean, double *cov, const double *meas_xyah, const double *conf, int n, void *hip_stream)
{
    (void)hip_stream;
    BAD(n < 0 || (n && (!mean || !cov || !meas_xyah)));
    for (int i = 0; i < n; ++i) orc_kf8_update(mean + 8 * (size_t)i, cov + 64 * (size_t)i, meas_xyah + 4 * (size_t)i, conf ? conf[i] : 0.0);
    return TLK_OK;
}

int tlk_kf8_gate_f64_cpu(const double *mean, const double *cov, int n_tracks, const double *meas_xyah, int n_meas, int only_position, double *out,
                         void *hip_stream)
{
    (void)hip_stream;
    BAD(n_tracks < 0 || n_meas < 0 || ((n_tracks && n_meas) && (!mean || !cov || !meas_xyah || !out)));
    for (int t = 0; t < n_tracks && n_meas; ++t)
        orc_kf8_gating(mean + 8 * (size_t)t, cov + 64 * (size_t)t, meas_xyah, n_meas, only_position, out + (size_t)t * n_meas);
    return TLK_OK;
}

int tlk_iou_ltwh_cost_f64_cpu(const double *tracks_ltwh, int n_tracks, const double *dets_ltwh, int n_dets, double *out, void *hip_stream)
{
    (void)hip_stream;
    BAD(n_tracks < 0 || n_dets < 0 || ((n_tracks && n_dets) && (!tracks_ltwh || !dets_ltwh || !out)));
    if (n_tracks && n_dets) orc_iou_ltwh_cost(tracks_ltwh, n_tracks, dets_ltwh, n_dets, out);
    return TLK_OK;
}

int tlk_oks_cost_f64_cpu(const double *track_kps, int n_tracks, const double *det_kps, int n_dets, double *out, void *hip_stream)
{
    (void)hip_stream;
    BAD(n_tracks < 0 || n_dets < 0 || ((n_tracks && n_dets) && (!track_kps || !det_kps || !out)));
    if (n_tracks && n_dets) orc_oks_cost(track_kps, n_tracks, det_kps, n_dets, out);
    return TLK_OK;
}

int tlk_partdist_f32_cpu(const float *q, const uint8_t *qvis, int T, const float *g, const uint8_t *gvis, int N, int K, int D, double *out,
                         void *hip_stream)
{
    (void)hip_stream;
    BAD(T < 0 || N < 0 || K <= 0 || K > 8 || D <= 0 || D % 16 || ((T && N) && (!q || !qvis || !g || !gvis || !out)));
    if (T && N) orc_partdist_f32(q, qvis, T, g, gvis, N, K, D, out);
    return TLK_OK;
}

int tlk_cosine_gallery_min_f32_cpu(const float *gallery, const int32_t *offsets, int T, int gallery_rows, const float *dets, int N, int D,
                                   double *out, void *hip_stream)
{
    (void)hip_stream;
    BAD(T < 0 || N < 0 || gallery_rows < 0 || D <= 0 || D % 16 || (T && !offsets) || ((T && N) && (!dets || !out)) || (gallery_rows && !gallery));
    BAD(T && (offsets[0] < 0 || offsets[T] > gallery_rows));
    if (T && N) orc_cosine_gallery_min_f32(gallery, offsets, T, dets, N, D, out);
    return TLK_OK;
}

/* frames (batch, h, w, 3) u8 -> out (batch, 3, size, size) fp32.  The twin implements what the reference's detector wrapper asks for
 * (rtmlib: BGR planes, fp32): layout TLK_NCHW, dtype TLK_F32; libtlk's other layouts / storage types are its own and have no CPU form. */
int tlk_letterbox_u8_cpu(const uint8_t *frames, int batch, int h, int w, int size, int layout, int dtype, void *out, double *ratio_out,
                         void *hip_stream)
{
    (void)hip_stream;
    BAD(batch < 0 || h <= 0 || w <= 0 || size <= 0 || size % 16 || (batch && (!frames || !out)));
    if (layout != TLK_NCHW || dtype != TLK_F32) return TLK_EUNSUPPORTED;
    double ratio = fmin((double)size / h, (double)size / w);
    for (int b = 0; b < batch; ++b) ratio = orc_letterbox(frames + (size_t)b * h * w * 3, h, w, size, (float *)out + (size_t)b * 3 * size * size);
    if (ratio_out) *ratio_out = ratio;
    return TLK_OK;
}

/* boxes (batch, max_n, 4) fp32 ltwh + counts (batch) -> out (batch * max_n, 3, out_h, out_w) fp32 NCHW; slots >= counts[b] are not written */
int tlk_roi_crop_resize_norm_cpu(const uint8_t *frames, int batch, int h, int w, const float *boxes_ltwh, const int32_t *counts, int max_n,
                                 int out_h, int out_w, const float *mean3, const float *std3, int layout, int dtype, void *out, void *hip_stream)
{
    (void)hip_stream;
    BAD(batch < 0 || h <= 0 || w <= 0 || max_n < 0 || out_h <= 0 || out_w <= 0 || out_w % 8 || !mean3 || !std3);
    BAD(batch && max_n && (!frames || !boxes_ltwh || !counts || !out));
    if (layout != TLK_NCHW || dtype != TLK_F32) return TLK_EUNSUPPORTED;
    for (int b = 0; b < batch; ++b) {
        const int n = counts[b] < 0 ? 0 : (counts[b] > max_n ? max_n : counts[b]);
        for (int i = 0; i < n; ++i) {
            double ltwh[4];
            int32_t ltrb[4];
            for (int c = 0; c < 4; ++c) ltwh[c] = boxes_ltwh[((size_t)b * max_n + i) * 4 + c];
            orc_ltwh_to_crop_ltrb(ltwh, 1, w, h, ltrb);
            orc_crop_resize_norm(frames + (size_t)b * h * w * 3, h, w, ltrb, out_h, out_w, mean3, std3,
                                 (float *)out + ((size_t)b * max_n + i) * 3 * out_h * out_w);
        }
    }
    return TLK_OK;
}

/* pred (batch, A, 5 + num_classes) fp32 -> per frame up to max_out detections in rtmlib order; ltwh = sanitize_bbox_ltrb + ltrb_to_ltwh in
 * float32 (tracklab/utils/coordinates.py:270-295, :318-328) as RTMLibDetector applies them (rtmlib_api.py:36-41); trk_in rows as
 * OCSORT.preprocess builds them (oc_sort_api.py:37-45) */
int tlk_yolox_decode_nms_cpu(const float *pred, int batch, int size, int num_classes, float ratio, float nms_thr, float score_thr, int img_w,
                             int img_h, int max_out, float *ltwh, float *xyxy, float *scores, int32_t *cls, int32_t *counts, double *trk_in,
                             int64_t det_id_base, double category_id, void *hip_stream)
{
    (void)hip_stream;
    BAD(batch < 0 || size <= 0 || size % 32 || num_classes <= 0 || max_out <= 0 || img_w <= 1 || img_h <= 1);
    BAD(batch && (!pred || !ltwh || !xyxy || !scores || !cls || !counts));
    const size_t A = (size_t)(size / 8) * (size / 8) + (size_t)(size / 16) * (size / 16) + (size_t)(size / 32) * (size / 32);
    float *bx = malloc(sizeof(float) * 6 * A);
    if (!bx) return TLK_EINVAL;
    float *sc = bx + 4 * A;
    int32_t *cl = (int32_t *)(bx + 5 * A);
    for (int b = 0; b < batch; ++b) {
        const int n = orc_yolox_postprocess(pred + (size_t)b * A * (5 + num_classes), size, num_classes, ratio, nms_thr, score_thr, bx, sc, cl, (int)A);
        if (n > max_out) { counts[b] = TLK_ECAPACITY; continue; }
        counts[b] = n;
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)b * max_out + i;
            const float *e = bx + 4 * (size_t)i;
            const float l = fmaxf(0.f, fminf(e[0], (float)(img_w - 2))), t = fmaxf(0.f, fminf(e[1], (float)(img_h - 2)));
            const float r = fmaxf(1.f, fminf(e[2], (float)(img_w - 1))), bt = fmaxf(1.f, fminf(e[3], (float)(img_h - 1)));
            for (int c = 0; c < 4; ++c) xyxy[o * 4 + c] = e[c];
            ltwh[o * 4 + 0] = l; ltwh[o * 4 + 1] = t; ltwh[o * 4 + 2] = r - l; ltwh[o * 4 + 3] = bt - t;
            scores[o] = sc[i];
            cls[o] = cl[i];
            if (trk_in) {
                double *row = trk_in + o * 7;
                const float bw = r - l, bh = bt - t;        /* the wrapper's ltwh -> ltrb stays in float32 (oc_sort_api.py:37-45) */
                row[0] = l; row[1] = t; row[2] = (double)(float)(l + bw); row[3] = (double)(float)(t + bh);
                row[4] = 1.0; row[5] = category_id; row[6] = (double)(det_id_base + (int64_t)b * max_out + i);
            }
        }
    }
    free(bx);
    return TLK_OK;
}

/* ---- OC-SORT bank: n_streams independent orc_ocsort trackers behind libtlk's bank signatures ---- */
struct tlk_ocsort_cpu {
    tlk_ocsort_params p;
    int n_streams;
    orc_ocsort **trk;
};

static orc_ocsort *ocsort_new(const tlk_ocsort_params *p)
{
    return orc_ocsort_create(p->det_thresh, p->max_age, p->min_hits, p->iou_threshold, p->delta_t, p->asso_func, p->inertia, p->use_byte);
}

int tlk_ocsort_create_cpu(const tlk_ocsort_params *p, int n_streams, int device, tlk_ocsort_cpu **out)
{
    (void)device;
    BAD(!p || !out || n_streams <= 0 || p->asso_func < TLK_IOU || p->asso_func > TLK_CT || p->max_tracks <= 0 || p->max_dets <= 0);
    tlk_ocsort_cpu *h = calloc(1, sizeof(*h));
    if (!h) return TLK_EINVAL;
    h->p = *p;
    h->n_streams = n_streams;
    h->trk = calloc((size_t)n_streams, sizeof(*h->trk));
    for (int s = 0; h->trk && s < n_streams; ++s) h->trk[s] = ocsort_new(p);
    *out = h;
    return TLK_OK;
}

int tlk_ocsort_destroy_cpu(tlk_ocsort_cpu *h)
{
    if (!h) return TLK_OK;
    for (int s = 0; h->trk && s < h->n_streams; ++s) orc_ocsort_destroy(h->trk[s]);
    free(h->trk);
    free(h);
    return TLK_OK;
}

int tlk_ocsort_reset_cpu(tlk_ocsort_cpu *h, int stream)
{
    BAD(!h || stream >= h->n_streams);
    for (int s = 0; s < h->n_streams; ++s)
        if (stream < 0 || s == stream) {
            orc_ocsort_destroy(h->trk[s]);
            h->trk[s] = ocsort_new(&h->p);
        }
    return TLK_OK;
}

/* one frame of one stream: dets (n,7) [l,t,r,b,conf,cls,tracklab_id] -> rows (<= out_cap, 8) [l,t,r,b,track_id,cls,conf,tracklab_id];
 * wrapper_mode 1 = OCSORT.process (tracklab/wrappers/track/oc_sort_api.py:50-56): nothing happens on an empty frame, detections are
 * filtered by conf > min_confidence before the tracker sees them */
int tlk_ocsort_update_cpu(tlk_ocsort_cpu *h, int stream, const double *dets, int n, double *out, int out_cap, int *n_out)
{
    BAD(!h || stream < 0 || stream >= h->n_streams || n < 0 || (n && !dets) || out_cap < 0 || (out_cap && !out) || !n_out);
    if (n > h->p.max_dets) return TLK_ECAPACITY;
    *n_out = 0;
    const double *use = dets;
    double *kept = NULL;
    if (h->p.wrapper_mode) {
        if (n == 0) return TLK_OK;
        kept = malloc(sizeof(double) * 7 * (size_t)n);
        if (!kept) return TLK_EINVAL;
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (dets[7 * (size_t)i + 4] > h->p.min_confidence) memcpy(kept + 7 * (size_t)m++, dets + 7 * (size_t)i, sizeof(double) * 7);
        use = kept;
        n = m;
    }
    const int cap = n + orc_ocsort_num_tracks(h->trk[stream]) + 1;
    double *rows = malloc(sizeof(double) * 8 * (size_t)cap);
    int rc = TLK_OK;
    if (!rows) rc = TLK_EINVAL;
    else {
        const int m = orc_ocsort_update(h->trk[stream], use, n, rows, cap);
        if (m > out_cap || orc_ocsort_num_tracks(h->trk[stream]) > h->p.max_tracks) rc = TLK_ECAPACITY;
        else {
            memcpy(out, rows, sizeof(double) * 8 * (size_t)m);
            *n_out = m;
        }
    }
    free(rows);
    free(kept);
    return rc;
}

/* ---- BPBReID-StrongSORT bank ---- */
struct tlk_bpbss_cpu {
    tlk_bpbss_params p;
    int n_streams;
    orc_bpbss **trk;
};

static orc_bpbss *bpbss_new(const tlk_bpbss_params *p)
{
    orc_bpbss_cfg c;
    memset(&c, 0, sizeof(c));
    c.ema_alpha = p->ema_alpha; c.mc_lambda = p->mc_lambda; c.max_dist = p->max_dist; c.max_iou_distance = p->max_iou_distance;
    c.min_bbox_confidence = p->min_bbox_confidence; c.gating_thres_factor = p->gating_thres_factor;
    c.w_kfgd = p->w_kfgd; c.w_reid = p->w_reid; c.w_st = p->w_st;
    c.max_age = p->max_age; c.n_init = p->n_init; c.only_position = p->only_position_for_kf_gating;
    c.max_kalman_prediction_without_update = p->max_kalman_prediction_without_update;
    c.matching_strategy = p->matching_strategy; c.motion_criterium = p->motion_criterium; c.max_oks_distance = p->max_oks_distance;
    return orc_bpbss_create(&c, p->parts, p->dim);
}

int tlk_bpbss_create_cpu(const tlk_bpbss_params *p, int n_streams, int device, tlk_bpbss_cpu **out)
{
    (void)device;
    BAD(!p || !out || n_streams <= 0 || p->parts <= 0 || p->parts > 8 || p->dim <= 0 || p->dim % 16 || p->max_tracks <= 0 || p->max_dets <= 0);
    tlk_bpbss_cpu *h = calloc(1, sizeof(*h));
    if (!h) return TLK_EINVAL;
    h->p = *p;
    h->n_streams = n_streams;
    h->trk = calloc((size_t)n_streams, sizeof(*h->trk));
    for (int s = 0; h->trk && s < n_streams; ++s) h->trk[s] = bpbss_new(p);
    *out = h;
    return TLK_OK;
}

int tlk_bpbss_destroy_cpu(tlk_bpbss_cpu *h)
{
    if (!h) return TLK_OK;
    for (int s = 0; h->trk && s < h->n_streams; ++s) orc_bpbss_destroy(h->trk[s]);
    free(h->trk);
    free(h);
    return TLK_OK;
}

int tlk_bpbss_reset_cpu(tlk_bpbss_cpu *h, int stream)
{
    BAD(!h || stream >= h->n_streams);
    for (int s = 0; s < h->n_streams; ++s)
        if (stream < 0 || s == stream) {
            orc_bpbss_destroy(h->trk[s]);
            h->trk[s] = bpbss_new(&h->p);
        }
    return TLK_OK;
}

/* one frame of one stream (wrapper_mode 1: the tracker is skipped on a frame with no detections,
 * tracklab/wrappers/track/bpbreid_strong_sort_api.py:103-104); classes are all 0, as the wrapper passes them */
int tlk_bpbss_update_cpu(tlk_bpbss_cpu *h, int stream, const int64_t *ids, const double *ltwh, const float *emb, const uint8_t *vis,
                         const double *conf, const double *kps, int n, tlk_bpbss_row *rows, int cap, int *n_out)
{
    BAD(!h || stream < 0 || stream >= h->n_streams || n < 0 || (n && (!ids || !ltwh || !emb || !vis || !conf)) || cap < 0 || (cap && !rows) || !n_out);
    BAD(h->p.motion_criterium == 1 && n && !kps);
    if (n > h->p.max_dets) return TLK_ECAPACITY;
    *n_out = 0;
    if (h->p.wrapper_mode && n == 0) return TLK_OK;
    double *classes = calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    orc_bpbss_row *tmp = malloc(sizeof(orc_bpbss_row) * (size_t)(n > 0 ? n : 1));
    int rc = TLK_OK;
    if (!classes || !tmp) rc = TLK_EINVAL;
    else {
        const int m = orc_bpbss_update_kp(h->trk[stream], ids, ltwh, emb, vis, conf, classes, kps, n, tmp);
        if (m > cap || orc_bpbss_num_tracks(h->trk[stream]) > h->p.max_tracks) rc = TLK_ECAPACITY;
        else {
            for (int i = 0; i < m; ++i) {
                tlk_bpbss_row *r = rows + i;
                memset(r, 0, sizeof(*r));
                r->det_id = tmp[i].det_id; r->track_id = tmp[i].track_id;
                memcpy(r->kf_ltwh, tmp[i].kf_ltwh, sizeof(r->kf_ltwh));
                memcpy(r->pred_ltwh, tmp[i].pred_ltwh, sizeof(r->pred_ltwh));
                r->pred_valid = tmp[i].pred_valid; r->matched_name = tmp[i].matched_name; r->matched_dist = tmp[i].matched_dist;
                r->hits = tmp[i].hits; r->age = tmp[i].age; r->time_since_update = tmp[i].tsu; r->state = tmp[i].state;
            }
            *n_out = m;
        }
    }
    free(classes);
    free(tmp);
    return rc;
}
