/* tlk_cpu.h -- the `_cpu` twins of libtlk's minimum export list ("each with a `_cpu` twin used as the on-box CPU
 * baseline").
 *
 * They are NOT part of libtlk.so: they live in the CPU oracle, oracle/_build/liborc.so (oracle/src/cpu_twins.c, built by oracle/Makefile), which
 * is test infrastructure -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; libtlk has no CPU path and fails
 * loudly without a GPU.  Every twin has its libtlk counterpart's name + `_cpu` and its signature, with HOST pointers where libtlk takes device
 * pointers and the trailing `hip_stream` argument present but ignored, so that a harness can call either side through one function-pointer type.
 * Each is a thin adapter over the orc_* restatement of the reference code named on the libtlk declaration (tlk.h) -- single-threaded C, fp64
 * where the reference is fp64.  Return values as in tlk.h (TLK_OK / TLK_EINVAL / TLK_ECAPACITY; TLK_EUNSUPPORTED for libtlk's own output
 * layouts and storage types that the reference's CPU path does not have: the image twins implement TLK_NCHW + TLK_F32).
 *
 *   libtlk function (tlk.h)            twin                                   restatement (oracle/src/orc.h)
 *   tlk_iou_matrix_f64                 tlk_iou_matrix_f64_cpu                 orc_iou_matrix
 *   tlk_lsa_f64 / _lapjv_limit_f64     tlk_lsa_f64_cpu / ..._cpu              orc_lsa / orc_lapjv_limit
 *   tlk_kf7_{predict,update}_f64       ..._cpu                                orc_kf7_{predict,update}
 *   tlk_kf8_{initiate,predict,project,update,gate}_f64   ..._cpu              orc_kf8_{initiate,predict,project,update,gating}
 *   tlk_iou_ltwh_cost_f64, tlk_oks_cost_f64              ..._cpu              orc_iou_ltwh_cost, orc_oks_cost
 *   tlk_partdist_f32, tlk_cosine_gallery_min_f32         ..._cpu              orc_partdist_f32, orc_cosine_gallery_min_f32
 *   tlk_letterbox_u8                   tlk_letterbox_u8_cpu                   orc_letterbox
 *   tlk_roi_crop_resize_norm           tlk_roi_crop_resize_norm_cpu           orc_ltwh_to_crop_ltrb + orc_crop_resize_norm
 *   tlk_yolox_decode_nms               tlk_yolox_decode_nms_cpu               orc_yolox_postprocess (+ the wrapper's float32 box clipping)
 *   tlk_ocsort_{create,destroy,reset,update}             ..._cpu              orc_ocsort_* (one tracker per stream)
 *   tlk_bpbss_{create,destroy,reset,update}              ..._cpu              orc_bpbss_*  (one tracker per stream)
 */
#ifndef TLK_CPU_H
#define TLK_CPU_H

#include "tlk.h"

#ifdef __cplusplus
extern "C" {
#endif

int tlk_iou_matrix_f64_cpu(int variant, const double *b1, int n, const double *b2, int m, double *out, void *hip_stream);
int tlk_lsa_f64_cpu(const double *cost, int batch, int nr, int nc, int32_t *rows, int32_t *cols, int32_t *n_pairs, void *hip_stream);
int tlk_lsa_lapjv_limit_f64_cpu(const double *cost, int batch, int nr, int nc, double cost_limit, int32_t *x, int32_t *y, void *hip_stream);

int tlk_kf7_predict_f64_cpu(double *x, double *P, int n, void *hip_stream);
int tlk_kf7_update_f64_cpu(double *x, double *P, const double *z, int n, void *hip_stream);
int tlk_kf8_initiate_f64_cpu(const double *meas_xyah, double *mean, double *cov, int n, void *hip_stream);
int tlk_kf8_predict_f64_cpu(double *mean, double *cov, int n, void *hip_stream);
int tlk_kf8_project_f64_cpu(const double *mean, const double *cov, const double *conf, double *pmean, double *pcov, int n, void *hip_stream);
int tlk_kf8_update_f64_cpu(double *mean, double *cov, const double *meas_xyah, const double *conf, int n, void *hip_stream);
int tlk_kf8_gate_f64_cpu(const double *mean, const double *cov, int n_tracks, const double *meas_xyah, int n_meas, int only_position, double *out,
                         void *hip_stream);

int tlk_iou_ltwh_cost_f64_cpu(const double *tracks_ltwh, int n_tracks, const double *dets_ltwh, int n_dets, double *out, void *hip_stream);
int tlk_oks_cost_f64_cpu(const double *track_kps, int n_tracks, const double *det_kps, int n_dets, double *out, void *hip_stream);
int tlk_partdist_f32_cpu(const float *q, const uint8_t *qvis, int T, const float *g, const uint8_t *gvis, int N, int K, int D, double *out,
                         void *hip_stream);
int tlk_cosine_gallery_min_f32_cpu(const float *gallery, const int32_t *offsets, int T, int gallery_rows, const float *dets, int N, int D,
                                   double *out, void *hip_stream);

int tlk_letterbox_u8_cpu(const uint8_t *frames, int batch, int h, int w, int size, int layout, int dtype, void *out, double *ratio_out,
                         void *hip_stream);
int tlk_roi_crop_resize_norm_cpu(const uint8_t *frames, int batch, int h, int w, const float *boxes_ltwh, const int32_t *counts, int max_n,
                                 int out_h, int out_w, const float *mean3, const float *std3, int layout, int dtype, void *out, void *hip_stream);
int tlk_yolox_decode_nms_cpu(const float *pred, int batch, int size, int num_classes, float ratio, float nms_thr, float score_thr, int img_w,
                             int img_h, int max_out, float *ltwh, float *xyxy, float *scores, int32_t *cls, int32_t *counts, double *trk_in,
                             int64_t det_id_base, double category_id, void *hip_stream);

typedef struct tlk_ocsort_cpu tlk_ocsort_cpu;
int tlk_ocsort_create_cpu(const tlk_ocsort_params *p, int n_streams, int device, tlk_ocsort_cpu **out);
int tlk_ocsort_destroy_cpu(tlk_ocsort_cpu *h);
int tlk_ocsort_reset_cpu(tlk_ocsort_cpu *h, int stream);
int tlk_ocsort_update_cpu(tlk_ocsort_cpu *h, int stream, const double *dets, int n, double *out, int out_cap, int *n_out);

typedef struct tlk_bpbss_cpu tlk_bpbss_cpu;
int tlk_bpbss_create_cpu(const tlk_bpbss_params *p, int n_streams, int device, tlk_bpbss_cpu **out);
int tlk_bpbss_destroy_cpu(tlk_bpbss_cpu *h);
int tlk_bpbss_reset_cpu(tlk_bpbss_cpu *h, int stream);
int tlk_bpbss_update_cpu(tlk_bpbss_cpu *h, int stream, const int64_t *ids, const double *ltwh, const float *emb, const uint8_t *vis,
                         const double *conf, const double *kps, int n, tlk_bpbss_row *rows, int cap, int *n_out);

#ifdef __cplusplus
}
#endif
#endif
