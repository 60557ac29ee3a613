/*
 * tlk.h -- C ABI of libtlk.so: hand-written HIP (gfx950 / MI355X) kernels for TrackLab's
 * per-frame detector -> ReID -> association hot path.
 *
 * Plain pointers and sizes only; no torch types. Every entry point names the reference
 * interface it replaces (paths relative to the TrackLab tree, v1.3.24). The reference is 100 %
 * Python, so "the FFI a maintainer would bind" is ctypes: see INTEGRATION.md for the stub.
 *
 * Conventions
 *  - every function returns TLK_OK (0) or a negative TLK_E* code; tlk_last_error() gives the
 *    message for the calling thread. Nothing aborts the process.
 *  - "_dev" pointers are device (HBM) addresses on the handle's device; everything else is host.
 *  - `hip_stream` is a hipStream_t passed as void* (NULL = the default stream); calls that take
 *    one are asynchronous with respect to the host.
 *  - a handle is used by one host thread at a time; different handles are independent.
 *  - arrays are contiguous row-major, float64 unless the name says otherwise.
 */
#ifndef TLK_H
#define TLK_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TLK_OK 0
#define TLK_EINVAL (-1)     /* bad argument */
#define TLK_EHIP (-2)       /* HIP runtime error (message has hipGetErrorString) */
#define TLK_ECAPACITY (-3)  /* more tracks / detections than the handle was created for */
#define TLK_ENODEVICE (-4)  /* no usable gfx950 device */
#define TLK_EUNSUPPORTED (-5) /* an optional library route is not available (callers keep their other GPU route) */
#define TLK_EINTERNAL (-6)  /* a bounded device loop hit its bound: solver / tracker state no consistent run can produce (r05) */

const char *tlk_last_error(void);
int tlk_version(void);               /* 10000*major + 100*minor + patch */
int tlk_device_count(int *count);    /* number of HIP devices visible to this process */

/* ------------------------------------------------------------------------------------------
 * Similarity matrices.  Replaces plugins/track/oc_sort/association.py:5-171
 * (iou_batch / giou_batch / diou_batch / ciou_batch / ct_dist).
 * b1 (n, 4) xyxy, b2 (m, 4) xyxy -> out (n, m). All device pointers.
 * ------------------------------------------------------------------------------------------ */
enum { TLK_IOU = 0, TLK_GIOU = 1, TLK_DIOU = 2, TLK_CIOU = 3, TLK_CT = 4 };
int tlk_iou_matrix_f64(int variant, const double *b1_dev, int n, const double *b2_dev, int m,
                       double *out_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Linear sum assignment, scipy-identical (rectangular Jonker-Volgenant; same scan order and
 * tie-breaks as scipy.optimize.linear_sum_assignment).  Replaces the call sites
 * plugins/track/oc_sort/association.py:193-195 and
 * plugins/track/bpbreid_strong_sort/sort/linear_assignment.py:56.
 * Batched: `batch` independent problems, cost_dev (batch, nr, nc). rows/cols_dev
 * (batch, min(nr,nc)) int32, pairs sorted by row; n_pairs_dev (batch) int32 (= min(nr,nc), or
 * -1 infeasible / -2 NaN or -inf in the input / -3 the solver hit a loop bound no consistent state reaches, see below).
 * One wavefront solves one problem.
 * ------------------------------------------------------------------------------------------ */
int tlk_lsa_f64(const double *cost_dev, int batch, int nr, int nc, int32_t *rows_dev, int32_t *cols_dev,
                int32_t *n_pairs_dev, void *hip_stream);

/* Every data-dependent loop of the device-side assignment solver is bounded (r05): the column scan by the number of columns, the
 * augmenting-path walk by the number of rows, every index it follows by its range.  A bound that trips means the solver's work area
 * was corrupted under it; tlk_lsa_f64 then reports n_pairs = -3 and the tracker banks poison the stream: tlk_*_update returns
 * TLK_EINTERNAL (tlk_*_update_dev writes it to out_counts) until the stream is reset.  This TEST HOOK lowers the walk's cap to
 * `hops` steps for tlk_lsa_f64 (0 restores the natural bound) so the exit can be exercised on purpose. */
int tlk_debug_lsa_hop_limit(int hops);

/* `list(set(a) - set(b))` in the order CPython 3.10 iterates the result set -- how both StrongSORT plugins build the
 * unmatched-track list of their matching cascade (plugins/track/strong_sort/sort/linear_assignment.py:126-127,
 * plugins/track/bpbreid_strong_sort/sort/linear_assignment.py:127-128); that list is the row order of the IoU / OKS stage.
 * HOST buffers: a (na) ascending distinct non-negative keys (the cascade's track_indices), b (nb) distinct keys out of a
 * (the matched tracks), out (na) / n_out the ordered difference. tlk_ssort_* / tlk_bpbss_* run the same device code inside
 * their association kernels; force_table != 0 skips the "no key can wrap -> ascending" shortcut (test hook). */
int tlk_pyset_difference_order(const int32_t *a, int na, const int32_t *b, int nb, int32_t *out, int32_t *n_out, int force_table);

/* lap.lapjv(cost, extend_cost=True, cost_limit=L) as ByteTrack's linear_assignment uses it
 * (plugins/track/byte_track/matching.py:37-48; third-party lap, not installed -> restated from its documented
 * embedding: (nr+nc)^2 problem, padding entries L/2, lower-right block 0). x_dev (batch, nr): column of row i or -1;
 * y_dev (batch, nc): row of column j or -1. The set of matched pairs is the unique optimum whenever no two
 * real costs tie; a pair is only kept when it beats leaving both ends unmatched (cost < L). nr + nc <= 512. */
int tlk_lsa_lapjv_limit_f64(const double *cost_dev, int batch, int nr, int nc, double cost_limit, int32_t *x_dev,
                            int32_t *y_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * OC-SORT tracker bank: `n_streams` independent trackers whose whole state lives in HBM.
 * Replaces plugins/track/oc_sort/ocsort.py:185-334 (OCSort.__init__/update, KalmanBoxTracker)
 * + oc_sort/kalmanfilter.py:339-526 (KalmanFilterNew predict/update/freeze/unfreeze)
 * + oc_sort/association.py:242-298 (associate) and the wrapper's per-frame filtering,
 * tracklab/wrappers/track/oc_sort_api.py:50-56.
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_ocsort tlk_ocsort;
typedef struct {
    double det_thresh;       /* ocsort.py:186  */
    double iou_threshold;
    double inertia;
    double min_confidence;   /* oc_sort_api.py:54; only used when wrapper_mode != 0 */
    int32_t max_age, min_hits, delta_t;
    int32_t asso_func;       /* TLK_IOU..TLK_CT (ASSO_FUNCS, ocsort.py:178-182) */
    int32_t use_byte;
    int32_t wrapper_mode;    /* 1: OCSORT.process semantics (skip the tracker on an empty frame,
                                filter conf > min_confidence); 0: bare OCSort.update */
    int32_t max_tracks;      /* allocation per stream (live + coasting tracks), <= 16384: the reference's list grows (oc_sort/ocsort.py:312-314); per-frame lists in LDS while tracks + detections <= 512, HBM beyond */
    int32_t max_dets;        /* detections per frame, <= 1024 */
} tlk_ocsort_params;

int tlk_ocsort_create(const tlk_ocsort_params *p, int n_streams, int device, tlk_ocsort **out);
int tlk_ocsort_destroy(tlk_ocsort *h);
/* Module.reset(): stream < 0 resets every stream of the bank */
int tlk_ocsort_reset(tlk_ocsort *h, int stream);
/* One frame of one stream, host buffers (the drop-in path used by the TrackLab Module):
 * dets (n,7) [l,t,r,b,conf,cls,tracklab_id] -> out rows (<=out_cap, 8)
 * [l,t,r,b,track_id,cls,conf,tracklab_id]; synchronous. */
int tlk_ocsort_update(tlk_ocsort *h, int stream, const double *dets, int n, double *out, int out_cap,
                      int *n_out);
/* All streams x n_frames frames in ONE launch, device buffers, asynchronous:
 * dets_dev (n_streams, n_frames, max_dets, 7); counts_dev (n_streams, n_frames) int32;
 * out_dev (n_streams, n_frames, out_cap, 8); out_counts_dev (n_streams, n_frames) int32.
 * out_counts < 0 reports TLK_ECAPACITY for that stream/frame. */
int tlk_ocsort_update_dev(tlk_ocsort *h, const double *dets_dev, const int32_t *counts_dev, int n_frames,
                          double *out_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream);
/* Debug / parity: copy the KF state of stream's live tracks in list order. x (cap,7), P (cap,7,7),
 * ids (cap) int64; returns count via n_tracks. Synchronous. */
int tlk_ocsort_get_tracks(tlk_ocsort *h, int stream, double *x, double *P, int64_t *ids, int cap, int *n_tracks);
/* Diagnostics: per-phase 100 MHz wall-clock ticks accumulated since create (only when the bank was created
 * with TLK_OCSORT_PROF set in the environment). cycles16: 16 int64. */
int tlk_ocsort_get_profile(tlk_ocsort *h, int stream, long long *cycles16);

/* ------------------------------------------------------------------------------------------
 * Stateless Kalman-filter steps, batched: entry i of every array is one independent filter
 * (one GPU thread per filter, state in registers). They are the same device functions the fused
 * tracker kernels call; exported so each reference method has its own parity test.
 *
 * KF7 = OC-SORT's KalmanFilterNew(dim_x=7, dim_z=4) as configured by KalmanBoxTracker
 * (plugins/track/oc_sort/ocsort.py:63-107: F, H, R[2:,2:]*=10, P, Q):
 *   tlk_kf7_predict_f64  <- KalmanFilterNew.predict   (oc_sort/kalmanfilter.py:368-379), x (n,7), P (n,7,7) in place
 *   tlk_kf7_update_f64   <- KalmanFilterNew.update(z) (oc_sort/kalmanfilter.py:480-526, z is not None branch), z (n,4)
 * KF8 = StrongSORT's xyah filter with NSA noise scaling (plugins/track/bpbreid_strong_sort/sort/kalman_filter.py):
 *   tlk_kf8_initiate_f64 <- initiate :53-83     meas (n,4) xyah -> mean (n,8), cov (n,8,8)
 *   tlk_kf8_predict_f64  <- predict  :85-119    in place
 *   tlk_kf8_project_f64  <- project  :121-152   conf_dev (n) or NULL (= 0) -> pmean (n,4), pcov (n,4,4)
 *   tlk_kf8_update_f64   <- update   :154-187   in place; conf_dev (n) or NULL
 *   tlk_kf8_gate_f64     <- gating_distance :189-227  (T filters) x (N measurements) -> out (T,N) squared Mahalanobis
 * ------------------------------------------------------------------------------------------ */
int tlk_kf7_predict_f64(double *x_dev, double *P_dev, int n, void *hip_stream);
int tlk_kf7_update_f64(double *x_dev, double *P_dev, const double *z_dev, int n, void *hip_stream);
int tlk_kf8_initiate_f64(const double *meas_xyah_dev, double *mean_dev, double *cov_dev, int n, void *hip_stream);
int tlk_kf8_predict_f64(double *mean_dev, double *cov_dev, int n, void *hip_stream);
int tlk_kf8_project_f64(const double *mean_dev, const double *cov_dev, const double *conf_dev, double *pmean_dev,
                        double *pcov_dev, int n, void *hip_stream);
int tlk_kf8_update_f64(double *mean_dev, double *cov_dev, const double *meas_xyah_dev, const double *conf_dev, int n,
                       void *hip_stream);
int tlk_kf8_gate_f64(const double *mean_dev, const double *cov_dev, int n_tracks, const double *meas_xyah_dev, int n_meas,
                     int only_position, double *out_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Motion costs of the StrongSORT family, (T tracks) x (N detections) -> out (T,N) f64:
 *   tlk_iou_ltwh_cost_f64 <- iou / iou_cost (plugins/track/bpbreid_strong_sort/sort/iou_matching.py:7-39, :42-78):
 *                            1 - IoU of ltwh boxes (the time_since_update > 1 -> INFTY_COST row rule of :68-70 is the
 *                            caller's: it depends on track state, not on geometry)
 *   tlk_oks_cost_f64      <- oks / oks_cost (sort/oks_matching.py:30-92, :95-128; kappa :7-27): keypoints (.,17,3) f64
 *                            [x, y, visibility]; scale from the track's visible-keypoint extent, 45-degree fallback,
 *                            factor clip 5, scale < 0.1 -> NaN
 * ------------------------------------------------------------------------------------------ */
int tlk_iou_ltwh_cost_f64(const double *tracks_ltwh_dev, int n_tracks, const double *dets_ltwh_dev, int n_dets,
                          double *out_dev, void *hip_stream);
int tlk_oks_cost_f64(const double *track_kps_dev, int n_tracks, const double *det_kps_dev, int n_dets, double *out_dev,
                     void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Part-based ReID distance (the one embedding x embedding contraction of the path; f32 MFMA).
 * Replaces NearestNeighborDistanceMetric.distance -> _nn_part_based
 * (plugins/track/bpbreid_strong_sort/sort/nn_matching.py:191-200, :99-135) including the third-party
 * torchreid compute_distance_matrix_using_bp_features it calls (:127-134):
 * L2-normalise every part embedding, per part d = sqrt(relu(|q|^2 - 2 q.g + |g|^2)), mean over the parts
 * visible in both (-1 if none), / 2.  q_dev (T,K,D) f32, qvis_dev (T,K) u8, g_dev (N,K,D) f32,
 * gvis_dev (N,K) u8 -> out_dev (T,N) f64. K <= 8, D % 16 == 0.
 * ------------------------------------------------------------------------------------------ */
int tlk_partdist_f32(const float *q_dev, const uint8_t *qvis_dev, int T, const float *g_dev, const uint8_t *gvis_dev,
                     int N, int K, int D, double *out_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Cosine gallery distance (f32 MFMA): cost[t][n] = min over track t's gallery of 1 - cos(gallery row, det n).
 * Replaces NearestNeighborDistanceMetric("cosine").distance of plain StrongSORT
 * (plugins/track/strong_sort/sort/nn_matching.py:30-50 _cosine_distance, :73-91 _nn_cosine_distance, :144-161).
 * gallery_dev (gallery_rows, D) f32, rows of track t = [offsets[t], offsets[t+1]); offsets_dev (T+1) int32;
 * dets_dev (N, D) f32 -> out_dev (T, N) f64. D % 16 == 0.
 * ------------------------------------------------------------------------------------------ */
int tlk_cosine_gallery_min_f32(const float *gallery_dev, const int32_t *offsets_dev, int T, int gallery_rows,
                               const float *dets_dev, int N, int D, double *out_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * BPBReID-StrongSORT tracker bank (n_streams independent trackers, state in HBM).
 * Replaces plugins/track/bpbreid_strong_sort/strong_sort.py:11-147 (StrongSORT.update),
 * sort/tracker.py:92-441 (predict, update, strong_sort_matching / bot_sort_matching, _initiate_track),
 * sort/track.py:68-187, sort/kalman_filter.py:53-227, sort/linear_assignment.py:11-175,
 * sort/iou_matching.py:7-78, sort/nn_matching.py:173-200 and the wrapper's empty-frame rule
 * (tracklab/wrappers/track/bpbreid_strong_sort_api.py:103-104).
 * Also sort/oks_matching.py:7-128 (motion_criterium "oks" on COCO-17 keypoints).
 * Not covered: ECC camera compensation (ecc: False in the yaml), the per-detection debug `costs` dictionaries
 * (tracker.py:365-407).
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_bpbss tlk_bpbss;
typedef struct {
    double ema_alpha, mc_lambda, max_dist, max_iou_distance, min_bbox_confidence, gating_thres_factor;
    double w_kfgd, w_reid, w_st;
    int32_t max_age, n_init, only_position_for_kf_gating, max_kalman_prediction_without_update;
    int32_t matching_strategy;   /* 0 strong_sort_matching, 1 bot_sort_matching */
    int32_t wrapper_mode;        /* 1: skip the tracker entirely on a frame with 0 detections */
    int32_t parts, dim;          /* K, D of the embeddings */
    int32_t max_tracks;          /* allocation per stream, <= 16384 (live + coasting tracks; the reference's lists grow: sort/tracker.py:427-441).  Track state lives in
                                  * HBM at this capacity; a frame's lists and Hungarian work area use LDS while tracks + detections <= 1024 and detections <= 256
                                  * (tiers 256 x 128 and 1024 x 256), HBM beyond.  TLK_ECAPACITY only past the allocation */
    int32_t max_dets;            /* detections per frame, <= 1024 (same tiers) */
    int32_t motion_criterium;    /* 0 "iou" (sort/iou_matching.py), 1 "oks" (sort/oks_matching.py; needs keypoints) */
    int32_t reserved_;
    double max_oks_distance;
} tlk_bpbss_params;

typedef struct {                 /* one output row = one confirmed track updated in this frame (strong_sort.py:93-141) */
    int64_t det_id, track_id;
    double kf_ltwh[4];           /* track_bbox_kf_ltwh      */
    double pred_ltwh[4];         /* track_bbox_pred_kf_ltwh (NaN when None) */
    int32_t pred_valid;
    int32_t matched_name;        /* matched_with: 0 None, 1 "R" (ReID stage), 2 "S" (spatio-temporal stage) */
    double matched_dist;
    int32_t hits, age, time_since_update;
    int32_t state;               /* 0 't', 1 'c', 2 'd' (TrackState, track.py:16-18) */
} tlk_bpbss_row;

int tlk_bpbss_create(const tlk_bpbss_params *p, int n_streams, int device, tlk_bpbss **out);
int tlk_bpbss_destroy(tlk_bpbss *h);
/* diagnostics: per-phase time accumulators of the association kernel in 100 MHz ticks (bank created with TLK_BPBSS_PROF set
 * in the environment; the extra barriers make the kernel slower). ticks16: 16 int64: filter+predict, preparation, appearance cost
 * fill, LSA A, set order + motion cost fill, LSA B, Kalman updates, embedding EMA, misses + births, deaths + rows. */
int tlk_bpbss_get_profile(tlk_bpbss *h, int stream, long long *ticks16);
int tlk_bpbss_reset(tlk_bpbss *h, int stream);
/* one frame of one stream, host buffers, synchronous: ids (n) int64, ltwh (n,4) f64, emb (n,K,D) f32,
 * vis (n,K) u8, conf (n) f64, kps (n,17,3) f64 [x,y,conf] or NULL -> rows (<= cap) */
int tlk_bpbss_update(tlk_bpbss *h, int stream, const int64_t *ids, const double *ltwh, const float *emb,
                     const uint8_t *vis, const double *conf, const double *kps, int n, tlk_bpbss_row *rows, int cap,
                     int *n_out);
/* n_frames consecutive frames of every stream, device buffers, asynchronous (3 launches per frame):
 * ids_dev (S,F,max_dets) int64; ltwh_dev (S,F,max_dets,4) f64; emb_dev (S,F,max_dets,K,D) f32;
 * vis_dev (S,F,max_dets,K) u8; conf_dev (S,F,max_dets) f64; kps_dev (S,F,max_dets,17,3) f64 or NULL;
 * counts_dev (S,F) int32;
 * rows_dev (S,F,out_cap) tlk_bpbss_row; out_counts_dev (S,F) int32 (<0 = error code). */
int tlk_bpbss_update_dev(tlk_bpbss *h, const int64_t *ids_dev, const double *ltwh_dev, const float *emb_dev,
                         const uint8_t *vis_dev, const double *conf_dev, const double *kps_dev, const int32_t *counts_dev,
                         int n_frames, tlk_bpbss_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream);
/* Debug / parity: live tracks of a stream in list order. ids (cap) int64, mean (cap,8), cov (cap,8,8),
 * feat (cap,K,D) f32, fvis (cap,K) u8 (any may be NULL). Synchronous. */
int tlk_bpbss_get_tracks(tlk_bpbss *h, int stream, int64_t *ids, double *mean, double *cov, float *feat,
                         uint8_t *fvis, int cap, int *n_tracks);

/* ------------------------------------------------------------------------------------------
 * ByteTrack tracker bank (n_streams independent trackers, state in HBM). Replaces BYTETracker.update
 * (plugins/track/byte_track/byte_tracker.py:167-320) with STrack (:13-152), joint/sub/remove_duplicate_stracks
 * (:323-361), matching.{iou_distance, ious, bbox_ious, fuse_score, linear_assignment} (byte_track/matching.py:37-90,
 * :171-217) and KalmanFilter (byte_track/kalman_filter.py:55-270). Hyper-parameter names =
 * configs/modules/track/byte_track.yaml `hyperparams` (+ the wrapper's min_confidence, byte_track_api.py:58).
 * The id counter (class-level BaseTrack._count in the reference) is per stream and restarts at 1 on reset.
 * max_tracks <= 16384 (tracked + lost), max_dets <= 1024: allocation sizes -- the per-frame lists and the assignment problem sit in LDS
 * while tracked + lost + detections <= 384 (128 detections), in HBM beyond (the reference's lists simply grow).
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_bytetrack_params {
    double track_thresh, match_thresh, frame_rate;
    double min_confidence;              /* -inf disables */
    int32_t track_buffer;
    int32_t wrapper_mode;               /* 1: a frame without detections leaves the tracker untouched (byte_track_api.py:55-56) */
    int32_t max_tracks, max_dets;       /* capacities per stream (0 = 256 / 128) */
} tlk_bytetrack_params;

typedef struct tlk_bytetrack_row {      /* one row of BYTETracker.update's output (byte_tracker.py:296-308) */
    int64_t det_id;                     /* tracklab_id of the detection last matched to the track */
    int64_t track_id;
    double ltrb[4];
    double score, cls;
} tlk_bytetrack_row;

typedef struct tlk_bytetrack tlk_bytetrack;
int tlk_bytetrack_create(const tlk_bytetrack_params *p, int n_streams, int device, tlk_bytetrack **out);
int tlk_bytetrack_destroy(tlk_bytetrack *h);
int tlk_bytetrack_reset(tlk_bytetrack *h, int stream);       /* stream < 0: all */
/* as tlk_bytetrack_reset but the id counter keeps counting: the reference's BaseTrack._count is class-level and never reset
 * (plugins/track/byte_track/basetrack.py:13,35-37), so the tracks of a second video continue the numbering of the first */
int tlk_bytetrack_reset_keep_ids(tlk_bytetrack *h, int stream);
/* host buffers: dets (n,7) f64 [x1,y1,x2,y2,conf,cls,tracklab_id] -> rows (cap) */
int tlk_bytetrack_update(tlk_bytetrack *h, int stream, const double *dets, int n, tlk_bytetrack_row *rows, int cap, int *n_out);
/* device buffers, all streams, n_frames consecutive frames per stream, asynchronous on hip_stream:
 * dets_dev (S, n_frames, max_dets, 7), counts_dev (S, n_frames) -> rows_dev (S, n_frames, out_cap), out_counts_dev (S, n_frames) */
int tlk_bytetrack_update_dev(tlk_bytetrack *h, const double *dets_dev, const int32_t *counts_dev, int n_frames,
                             tlk_bytetrack_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream);
/* debug/test: which = 0 tracked_stracks, 1 lost_stracks, in list order: ids, mean (.,8), cov (.,8,8), state5 (.,5)
 * [state (1 tracked, 2 lost, 3 removed), is_activated, frame_id, start_frame, tracklet_len]; any may be NULL */
int tlk_bytetrack_get_tracks(tlk_bytetrack *h, int stream, int which, int64_t *ids, double *mean, double *cov, int64_t *state5,
                             int cap, int *n_tracks);

/* ------------------------------------------------------------------------------------------
 * Deep-OC-SORT tracker bank (n_streams independent trackers, state + embeddings in HBM).
 * Replaces deep_oc_sort.ocsort.OCSort.update (plugins/track/deep_oc_sort/ocsort.py:392-534) from the point where the ReID
 * embeddings of the detections exist: KalmanBoxTracker with the (x, y, w, h) filter (ocsort.py:94-330,
 * kalmanfilter.py:340-569 incl. freeze / unfreeze), associate + compute_aw_max_metric (association.py:263-364), the OCR
 * second round, update_emb, births / deaths and the output rows.
 * Hyper-parameter names = configs/modules/track/deep_oc_sort.yaml `hyperparams` (+ the wrapper's min_confidence,
 * deep_oc_sort_api.py:62). cmc_off must be 1 (cmc.py is cv2), embedding_off and new_kf_off must be 0:
 * tlk_deepocsort_create answers TLK_EUNSUPPORTED otherwise. delta_t < 8.
 * max_tracks <= 16384, max_dets <= 1024: allocation sizes -- per-frame lists in LDS while tracks + detections <= 512 (256 detections), HBM beyond.
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_deepocsort_params {
    double det_thresh, iou_threshold, inertia, w_association_emb, alpha_fixed_emb, aw_param;
    double min_confidence;              /* -inf disables */
    int32_t max_age, min_hits, delta_t;
    int32_t asso_func;                  /* TLK_IOU .. TLK_CT */
    int32_t embedding_off, cmc_off, aw_off, new_kf_off;
    int32_t wrapper_mode;               /* 1: a frame without detections leaves the tracker untouched (deep_oc_sort_api.py:59-60) */
    int32_t dim;                        /* embedding length */
    int32_t max_tracks, max_dets;       /* capacities per stream (0 = 256 / 128) */
} tlk_deepocsort_params;

typedef struct tlk_deepocsort tlk_deepocsort;
int tlk_deepocsort_create(const tlk_deepocsort_params *p, int n_streams, int device, tlk_deepocsort **out);
int tlk_deepocsort_destroy(tlk_deepocsort *h);
int tlk_deepocsort_reset(tlk_deepocsort *h, int stream);     /* stream < 0: all */
/* camera-motion compensation with the estimate passed in (host pointer to the (2,3) affine CMCComputer.compute_affine returns):
 * KalmanBoxTracker.apply_affine_correction for every tracker (ocsort.py:261-281, :425-428), to be called before
 * tlk_deepocsort_update; the estimator (cmc.py, cv2 optical flow) is not part of libtlk. stream < 0: all. Asynchronous. */
int tlk_deepocsort_affine_correction(tlk_deepocsort *h, int stream, const double *warp6, void *hip_stream);
/* host buffers: dets (n,7) f64 [x1,y1,x2,y2,conf,cls,tracklab_id], embs (n,dim) f32 -> out (rows,8) f64
 * [x1,y1,x2,y2,track_id(+1),cls,conf,tracklab_id] (ocsort.py:527-529) */
int tlk_deepocsort_update(tlk_deepocsort *h, int stream, const double *dets, const float *embs, int n, double *out, int out_cap, int *n_out);
/* device buffers, all streams, n_frames consecutive frames per stream, ONE launch, asynchronous on hip_stream:
 * dets_dev (S, n_frames, max_dets, 7), embs_dev (S, n_frames, max_dets, dim), counts_dev (S, n_frames)
 * -> out_dev (S, n_frames, out_cap, 8), out_counts_dev (S, n_frames) */
int tlk_deepocsort_update_dev(tlk_deepocsort *h, const double *dets_dev, const float *embs_dev, const int32_t *counts_dev, int n_frames,
                              double *out_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream);
/* debug/test, trackers in list order: ids, x (.,8), P (.,8,8), emb (.,dim), state6 (.,6) [time_since_update, hits, hit_streak, age,
 * frozen, kf.observed], velocity (.,2), last_observation (.,5); none may be NULL */
int tlk_deepocsort_get_tracks(tlk_deepocsort *h, int stream, int64_t *ids, double *x, double *P, float *emb, int64_t *state6, double *vel,
                              double *last, int cap, int *n_tracks);
/* diagnostics: 16 accumulated wall-clock counters (100 MHz) per kernel phase; the bank must be created with TLK_DEEPOCSORT_PROF=1 */
int tlk_deepocsort_get_profile(tlk_deepocsort *h, int stream, long long *cycles16);

/* ------------------------------------------------------------------------------------------
 * BoT-SORT tracker bank (n_streams independent trackers, state + smoothed features in HBM).
 * Replaces bot_sort.BoTSORT.update (plugins/track/bot_sort/bot_sort.py:275-485) from the point where the ReID features of
 * the high-score detections exist: STrack (bot_sort.py:15-232: update_features, update_cls, multi_predict, multi_gmc with
 * the identity warp, activate / re_activate / update), matching.{embedding_distance, fuse_motion, iou_distance, fuse_score,
 * linear_assignment} (matching.py:37-48, :85-103, :127-142, :159-171, :188-196), the (x, y, w, h) KalmanFilter
 * (kalman_filter.py:55-269), joint / sub / remove_duplicate_stracks (bot_sort.py:507-545) and the output rows (:468-485).
 * Hyper-parameter names = configs/modules/track/bot_sort.yaml `hyperparams` (+ the wrapper's min_confidence,
 * bot_sort_api.py:62). cmc_method must be 0 ("none", gmc.py:75-78): the other estimators are cv2 (out of scope) and
 * tlk_botsort_create answers TLK_EUNSUPPORTED for them. A track may collect at most 16 distinct classes (cls_hist).
 * max_tracks <= 16384 (tracked + lost), max_dets <= 1024: allocation sizes -- the per-frame lists and the assignment problem sit in LDS
 * while tracked + lost + detections <= 384 (128 detections), in HBM beyond (the reference's lists simply grow). dim % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_botsort_params {
    double track_high_thresh, new_track_thresh, match_thresh, proximity_thresh, appearance_thresh, frame_rate, lambda_;
    double min_confidence;              /* -inf disables */
    int32_t track_buffer;
    int32_t cmc_method;                 /* gmc.py:18-78: 0 none, 1 orb, 2 sift, 3 ecc, 4 sparseOptFlow, 5 file. The bank APPLIES a warp
                                         * (STrack.multi_gmc); it does not estimate one: with a method other than 0 every update must
                                         * come through the *_gmc entry points with the frame's (2,3) warp (tlk_cmc_* estimate it) */
    int32_t wrapper_mode;               /* 1: a frame without detections leaves the tracker untouched (bot_sort_api.py:59-60) */
    int32_t dim;                        /* ReID embedding length */
    int32_t max_tracks, max_dets;       /* capacities per stream (0 = 256 / 128) */
} tlk_botsort_params;

typedef tlk_bytetrack_row tlk_botsort_row;      /* same columns: bot_sort.py:471-483 */

typedef struct tlk_botsort tlk_botsort;
int tlk_botsort_create(const tlk_botsort_params *p, int n_streams, int device, tlk_botsort **out);
int tlk_botsort_destroy(tlk_botsort *h);
int tlk_botsort_reset(tlk_botsort *h, int stream);           /* stream < 0: all */
/* as tlk_botsort_reset but the id counter keeps counting: the reference's BaseTrack._count is class-level and never reset
 * (plugins/track/bot_sort/basetrack.py), so the tracks of a second video continue the numbering of the first */
int tlk_botsort_reset_keep_ids(tlk_botsort *h, int stream);
/* host buffers: dets (n,7) f64 [x1,y1,x2,y2,conf,cls,tracklab_id], feats (n,dim) f32 (rows of detections with
 * conf <= track_high_thresh are not read) -> rows (cap) */
int tlk_botsort_update(tlk_botsort *h, int stream, const double *dets, const float *feats, int n, tlk_botsort_row *rows, int cap, int *n_out);
/* device buffers, all streams, n_frames consecutive frames per stream, asynchronous on hip_stream:
 * dets_dev (S, n_frames, max_dets, 7), feats_dev (S, n_frames, max_dets, dim), counts_dev (S, n_frames)
 * -> rows_dev (S, n_frames, out_cap), out_counts_dev (S, n_frames) */
int tlk_botsort_update_dev(tlk_botsort *h, const double *dets_dev, const float *feats_dev, const int32_t *counts_dev, int n_frames,
                           tlk_botsort_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream);
/* The same with camera-motion compensation: warp6 / warps_dev = what GMC.apply returned for the frame, a (2,3) float64 matrix
 * [[a, b, tx], [c, d, ty]] (plugins/track/bot_sort/bot_sort.py:341-343), applied to the predicted pool tracks and to the unconfirmed
 * tracks between multi_predict and the association (STrack.multi_gmc, bot_sort.py:93-109). NULL = identity (cmc_method "none").
 * warps_dev: (S, n_frames, 6). */
int tlk_botsort_update_gmc(tlk_botsort *h, int stream, const double *dets, const float *feats, int n, const double *warp6,
                           tlk_botsort_row *rows, int cap, int *n_out);
int tlk_botsort_update_dev_gmc(tlk_botsort *h, const double *dets_dev, const float *feats_dev, const int32_t *counts_dev,
                               const double *warps_dev, int n_frames, tlk_botsort_row *rows_dev, int out_cap, int32_t *out_counts_dev,
                               void *hip_stream);
/* debug/test: as tlk_bytetrack_get_tracks plus smooth_feat (., dim) f32; state: 1 tracked, 2 lost, 4 removed */
int tlk_botsort_get_tracks(tlk_botsort *h, int stream, int which, int64_t *ids, double *mean, double *cov, int64_t *state5,
                           float *smooth_feat, int cap, int *n_tracks);

/* ------------------------------------------------------------------------------------------
 * Plain StrongSORT tracker bank (n_streams independent trackers, state + feature galleries in HBM).
 * Replaces strong_sort.StrongSORT.update (plugins/track/strong_sort/strong_sort.py:41-84) from the point where the
 * ReID features exist, i.e. Tracker.predict / Tracker.update (sort/tracker.py:53-114, _match :152-188),
 * linear_assignment.{min_cost_matching,matching_cascade,gate_cost_matrix} (sort/linear_assignment.py:11-174),
 * iou_matching.{iou,iou_cost} (sort/iou_matching.py:7-82), Track (sort/track.py:69-301), KalmanFilter
 * (sort/kalman_filter.py:50-214), NearestNeighborDistanceMetric("cosine") with budget (sort/nn_matching.py:94-161),
 * and the output loop with _tlwh_to_xyxy (strong_sort.py:62-79, :111-122).
 * Hyper-parameter names = configs/modules/track/strong_sort.yaml `hyperparams` (+ the wrapper's min_confidence).
 * nn_budget in [1, 1024] = the reference's budget (ring of that many rows per track in HBM); nn_budget = -rows (rows <= 65536) = the
 * reference's budget=None: every sample is kept, with room for `rows` per track -- outgrowing it is TLK_ECAPACITY, never a dropped sample
 * (288 GB of HBM: 256 tracks x 8192 rows x 512 floats = 4.3 GB per stream). ECC (cfg.ecc, sort/track.py:130-239): tlk_ecc_* estimates the
 * warp, tlk_ssort_camera_update applies it.
 * max_tracks <= 16384, max_dets <= 1024: allocation sizes (gallery = max_tracks x |nn_budget| x dim floats per stream) -- per-frame lists and
 * the Hungarian work area in LDS while tracks + detections <= 1024 (256 detections), HBM beyond (the reference's track list grows).
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_ssort_params {
    double max_dist, max_iou_dist;      /* strong_sort.py:24-25 */
    int32_t max_age, max_unmatched_preds, n_init, nn_budget;
    double mc_lambda, ema_alpha;
    double min_confidence;              /* wrapper filter `inputs[:, 4] > min_confidence` (strong_sort_api.py:71); -inf disables */
    int32_t wrapper_mode;               /* 1: a frame without detections leaves the tracker untouched (strong_sort_api.py:68-69) */
    int32_t img_w, img_h;               /* ori_img.shape: output boxes are int-truncated and clipped to it */
    int32_t dim;                        /* feature length: 32, 64, 128, 256 or 512 */
    int32_t max_tracks, max_dets;       /* capacities per stream (0 = 256 / 128) */
} tlk_ssort_params;

typedef struct tlk_ssort_row {          /* one row of StrongSORT.update's output (strong_sort.py:70-77) */
    int64_t det_id;                     /* tracklab_id of the detection last matched to the track */
    int64_t track_id;
    double ltrb[4];                     /* integer-valued, clipped to the image */
    double conf;
    int32_t class_id, time_since_update;
} tlk_ssort_row;

typedef struct tlk_ssort tlk_ssort;
int tlk_ssort_create(const tlk_ssort_params *p, int n_streams, int device, tlk_ssort **out);
int tlk_ssort_destroy(tlk_ssort *h);
int tlk_ssort_reset(tlk_ssort *h, int stream);       /* stream < 0: all */
/* camera compensation with the ECC estimate passed in (host pointer to the (2,3) warp Track.ECC returns): Tracker.camera_update ->
 * Track.camera_update (sort/tracker.py:66-68, sort/track.py:221-239), to be called before tlk_ssort_update like
 * strong_sort_api.py:62-65 does; the estimator (cv2.findTransformECC) is not part of libtlk. stream < 0: all. Asynchronous. */
int tlk_ssort_camera_update(tlk_ssort *h, int stream, const double *warp6, void *hip_stream);
/* host buffers: dets (n,7) f64 [x1,y1,x2,y2,conf,cls,tracklab_id], feat (n,dim) f32 -> rows (cap) */
int tlk_ssort_update(tlk_ssort *h, int stream, const double *dets, const float *feat, int n, tlk_ssort_row *rows, int cap,
                     int *n_out);
/* device buffers, all streams, n_frames consecutive frames per stream, asynchronous on hip_stream:
 * dets_dev (S, n_frames, max_dets, 7), feat_dev (S, n_frames, max_dets, dim), counts_dev (S, n_frames) ->
 * rows_dev (S, n_frames, out_cap), out_counts_dev (S, n_frames) (negative = TLK_E* for that stream) */
int tlk_ssort_update_dev(tlk_ssort *h, const double *dets_dev, const float *feat_dev, const int32_t *counts_dev, int n_frames,
                         tlk_ssort_row *rows_dev, int out_cap, int32_t *out_counts_dev, void *hip_stream);
/* debug/test: track list in list order: ids, mean (.,8), cov (.,8,8), feat (.,dim), state5 (.,5) [hits, age,
 * time_since_update, state (1 tentative, 2 confirmed), updates_wo_assignment], gallery_rows (.); any may be NULL */
int tlk_ssort_get_tracks(tlk_ssort *h, int stream, int64_t *ids, double *mean, double *cov, float *feat, int64_t *state5,
                         int64_t *gallery_rows, int cap, int *n_tracks);

/* ------------------------------------------------------------------------------------------
 * Detector / ReID pre- and post-processing. In the reference this arithmetic sits in third-party
 * packages behind the adapters tracklab/wrappers/bbox_detector/rtmlib_api.py:27-46 (rtmlib
 * YOLOX.preprocess / postprocess, cv2.resize) and tracklab/wrappers/reid/kpreid_api.py:115-144
 * (crop image[t:b,l:r] of the rounded ltrb + albumentations Resize/Normalize).
 * ------------------------------------------------------------------------------------------ */
enum { TLK_NCHW = 0, TLK_NHWC = 1, TLK_FOCUS_NHWC = 2 };   /* FOCUS: YOLOX space-to-depth fused, (B,S/2,S/2,12) */
/* OR-ed into a `layout` argument: output channel c is SOURCE channel 2 - c (mean3/std3 stay indexed by OUTPUT channel).
 * TrackLab keeps one RGB frame (cv2_load_image) for the ReID crops while its detector / pose estimator re-read the file
 * as BGR (rtmlib on cv2.imread, wrappers/bbox_detector/rtmlib_api.py:30, wrappers/pose_estimator/rtmlib_api.py:30):
 * with this flag the frames stay RGB in HBM and the letterbox / pose crops read them as BGR without a flip pass. */
enum { TLK_SWAP_RB = 0x100 };
enum { TLK_F32 = 0, TLK_F16 = 1, TLK_BF16 = 2 };

/* frames_dev (batch, h, w, 3) uint8 -> out_dev (batch, 3, size, size) in `layout`/`dtype`, values 0..255,
 * pad 114, ratio = min(size/h, size/w) returned through ratio_out (host). size % 16 == 0. */
int tlk_letterbox_u8(const uint8_t *frames_dev, int batch, int h, int w, int size, int layout, int dtype,
                     void *out_dev, double *ratio_out, void *hip_stream);

/* boxes_ltwh_dev (batch, max_n, 4) float32 + counts_dev (batch) -> out_dev (batch*max_n, 3, out_h, out_w);
 * slot b*max_n+i holds crop i of frame b; slots >= counts[b] are NOT written (zero the buffer once if the consumer
 * needs zeros there; a box that is empty after clipping gives a zero crop). mean3/std3 are HOST float[3]
 * (ImageNet statistics in kpreid); value = (u8 - 255*mean) * (1/(255*std)). out_w % 8 == 0.
 * layout TLK_NCHW or TLK_NHWC. */
int tlk_roi_crop_resize_norm(const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev,
                             const int32_t *counts_dev, int max_n, int out_h, int out_w, const float *mean3,
                             const float *std3, int layout, int dtype, void *out_dev, void *hip_stream);

/* The same crops written as a DENSE batch (r05; the reference's ReID wrapper batches real detections only,
 * tracklab/wrappers/reid/kpreid_api.py:147-182): crop i of frame b lands at slot slot_base_dev[b] + i, padding slots are not written anywhere.
 * tlk_crop_slot_bases fills base_dev (batch) = exclusive prefix sums of the counts (clamped to [0, max_n]), total_dev (1) = their sum, and --
 * when slot_of_dev is not NULL -- slot_of_dev (batch * max_n) int64 = position of (b, i) in the dense batch (a valid position for padding
 * slots too: consumers gather rows through it and ignore i >= counts[b]). The ReID convolutions then run on total_dev[0] images
 * (tlk_conv_set_dynamic_batch) instead of batch * max_n slots. */
int tlk_crop_slot_bases(const int32_t *counts_dev, int batch, int max_n, int32_t *base_dev, int32_t *total_dev, int64_t *slot_of_dev, void *hip_stream);
int tlk_roi_crop_resize_norm_compact(const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev,
                                     const int32_t *counts_dev, const int32_t *slot_base_dev, int max_n, int out_h, int out_w, const float *mean3,
                                     const float *std3, int layout, int dtype, void *out_dev, void *hip_stream);

/* Plain StrongSORT's ReID input. Replaces StrongSORT._get_features' crop loop
 * (plugins/track/strong_sort/strong_sort.py:135-141 with _xywh_to_xyxy :102-108: int-truncated box clipped to
 * [0, w-1] x [0, h-1], crop ori_img[y1:y2, x1:x2]) and ReIDDetectMultiBackend._preprocess
 * (plugins/track/strong_sort/reid_multibackend.py:44-52, :184-195: ToPILImage -> Resize((256,128)) -> ToTensor ->
 * Normalize). The resize is Pillow's Image.resize(BILINEAR): separable, antialiased (support = max(1, scale)),
 * 22-bit fixed-point weights, uint8 intermediate -- bit-exact; value = ((u8 / 255) - mean) / std in float32.
 * boxes_xyxy_dev: (batch, max_n) rows of `box_stride` doubles whose first four are x1,y1,x2,y2 (box_stride = 7 reads
 * the tracker's (n,7) detection rows in place). out as tlk_roi_crop_resize_norm. mean3/std3 HOST float[3]. */
int tlk_roi_crop_pil_resize_norm(const uint8_t *frames_dev, int batch, int h, int w, const double *boxes_xyxy_dev, int box_stride,
                                 const int32_t *counts_dev, int max_n, int out_h, int out_w, const float *mean3,
                                 const float *std3, int layout, int dtype, void *out_dev, void *hip_stream);

/* Pose-estimator pre/post-processing (config 4). Replaces what rtmlib.RTMPose(image, bboxes) does around its network, behind
 * tracklab/wrappers/pose_estimator/rtmlib_api.py:27-33 (third-party rtmlib 0.0.13: RTMPose.preprocess / postprocess,
 * bbox_xyxy2cs(padding 1.25), _fix_aspect_ratio, get_warp_matrix, top_down_affine, get_simcc_maximum; OpenCV
 * getAffineTransform + warpAffine INTER_LINEAR, constant-0 border).
 *   tlk_pose_crop_warp_norm: boxes (batch, max_n) rows of `box_stride` doubles starting with x1,y1,x2,y2 + counts ->
 *     out (batch*max_n, 3, in_h, in_w) [layout/dtype as the other crops], value = (float)((u8 - mean) / std) with HOST double
 *     mean3/std3 (rtmlib: 123.675,116.28,103.53 / 58.395,57.12,57.375), and meta (batch*max_n, 10) f64 =
 *     [centre x,y, scale w,h, inverse warp matrix (6)] which tlk_simcc_decode needs. in_w % 8 == 0.
 *   tlk_simcc_decode: simcc_x (n, K, wx), simcc_y (n, K, wy) f32 -> kps_xyc (n, K, 3) f64 [x, y, score] in image coordinates
 *     (= keypoints_xyc, directly usable as tlk_bpbss_update*'s keypoints), scores (n, K) f32, conf (n) f32 = mean score
 *     (= keypoints_conf; may be NULL). split_ratio = 2.0 for RTMPose. */
int tlk_pose_crop_warp_norm(const uint8_t *frames_dev, int batch, int h, int w, const double *boxes_xyxy_dev, int box_stride,
                            const int32_t *counts_dev, int max_n, int in_w, int in_h, const double *mean3, const double *std3,
                            int layout, int dtype, void *out_dev, double *meta_dev, void *hip_stream);
int tlk_simcc_decode(const float *simcc_x_dev, const float *simcc_y_dev, int n, int n_keypoints, int wx, int wy, double split_ratio,
                     const double *meta_dev, int in_w, int in_h, double *kps_xyc_dev, float *scores_dev, float *conf_dev,
                     void *hip_stream);

/* pred_dev (batch, A, 5+num_classes) float32 raw YOLOX head (A = (s/8)^2+(s/16)^2+(s/32)^2) ->
 * per frame up to max_out detections in rtmlib order (class-major, score-descending):
 * ltwh_dev (batch,max_out,4) clipped to the image like RTMLibDetector (rtmlib_api.py:36-41),
 * xyxy_dev unclipped, scores_dev, cls_dev, counts_dev (batch; <0 = TLK_ECAPACITY).
 * trk_in_dev (optional, may be NULL): (batch, max_out, 7) float64 rows [l,t,r,b,1.0,category_id,det_id]
 * exactly as OCSORT.preprocess builds them from the detector's float32 ltwh (oc_sort_api.py:37-45), with
 * det_id = det_id_base + frame*max_out + i, ready for tlk_ocsort_update_dev. */
int tlk_yolox_decode_nms(const float *pred_dev, int batch, int size, int num_classes, float ratio, float nms_thr,
                         float score_thr, int img_w, int img_h, int max_out, float *ltwh_dev, float *xyxy_dev,
                         float *scores_dev, int32_t *cls_dev, int32_t *counts_dev, double *trk_in_dev,
                         int64_t det_id_base, double category_id, void *hip_stream);

/* non_max_suppression(boxes, max_bbox_overlap, scores=None) of plugins/track/strong_sort/sort/preprocessing.py:6-73 (same file in
 * bpbreid_strong_sort/sort/; dead code in the reference: never called). boxes (n, 4) float64 (x, y, w, h) and scores (n) float64 or NULL
 * (then the order is by the bottom edge) in device memory; pick (n) int32 receives the kept indices in the order the reference appends
 * them, *n_pick their number. n <= 1024. Equal keys keep ascending index order (np.argsort's default sort is not stable). */
int tlk_deepsort_nms_f64(const double *boxes_xywh_dev, const double *scores_dev, int n, double max_bbox_overlap, int32_t *pick_dev,
                         int32_t *n_pick_dev, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Camera-motion estimation on the device. Replaces GMC(method="sparseOptFlow", downscale).apply(frame) of
 * plugins/track/bot_sort/gmc.py:239-303 (cv2.cvtColor -> resize -> goodFeaturesToTrack(1000, 0.01, 1, 3) ->
 * calcOpticalFlowPyrLK -> estimateAffinePartial2D(RANSAC)); the same chain is deep_oc_sort/cmc.py:136-166. All of it is
 * third-party OpenCV in the reference -- PARITY UNPINNED; the kernels follow oracle/src/cmc.c operation for operation.
 * One handle = one video stream (it keeps the previous frame's pyramid and corners). The first frame returns the identity.
 *   tlk_cmc_apply_dev: frame_dev (h, w, 3) uint8 on the device, as the tracker receives it (the reference hands its RGB frame
 *     to COLOR_BGR2GRAY; so does this); warp6_dev: 6 doubles [[a, b, tx], [c, d, ty]] in device memory, ready for
 *     tlk_botsort_update_dev_gmc. Asynchronous on hip_stream, no host synchronisation.
 *   tlk_cmc_apply: the same with a host frame and a host result (n_inliers may be NULL).
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_cmc tlk_cmc;
int tlk_cmc_create(int h, int w, int downscale, int max_corners, int device, tlk_cmc **out);
int tlk_cmc_destroy(tlk_cmc *c);
int tlk_cmc_reset(tlk_cmc *c);                        /* new video: forget the previous frame */
int tlk_cmc_apply_dev(tlk_cmc *c, const uint8_t *frame_dev, double *warp6_dev, void *hip_stream);
/* The same for a caller whose "does this frame reach the tracker?" lives in device memory: the reference returns BEFORE GMC.apply on a frame
 * without detections (wrappers/track/bot_sort_api.py:59-60), so its next warp spans the frames either side of it. If *count_dev == 0 when the
 * chain has run, the estimator's state (previous pyramid, derivatives, corners) is put back to what it was before the call, on the device and
 * in stream order -- no host synchronisation; warp6_dev of such a frame is not meaningful (the tracker banks skip the frame). */
int tlk_cmc_apply_dev_gated(tlk_cmc *c, const uint8_t *frame_dev, double *warp6_dev, const int32_t *count_dev, void *hip_stream);
int tlk_cmc_apply(tlk_cmc *c, const uint8_t *frame_host, double *warp6_host, int *n_inliers);
/* debug / test: stage outputs of the last tlk_cmc_apply*: what = 0 downscaled grey image, 1 eigenvalue image (float32), 2 corners
 * (n, 2) float32, 3 tracked positions of the previous corners, 4 their status bytes, 10 + l pyramid image l, 20 + l its int16
 * (dx, dy) Scharr derivatives. n_out: element / point count (row width for the pyramid items). */
int tlk_cmc_debug_get(tlk_cmc *c, int what, void *host_buf, size_t cap_bytes, int *n_out);

/* ------------------------------------------------------------------------------------------
 * StrongSORT's camera-motion estimator on the device. Replaces Track.ECC(previous_frame, next_frame) of
 * plugins/track/strong_sort/sort/track.py:129-211 (= plugins/track/bpbreid_strong_sort/ecc.py:4-99): BGR2GRAY, cv2.resize(0.1),
 * cv2.findTransformECC(MOTION_EUCLIDEAN, 100 iterations, eps 1e-5, gaussFiltSize 1), translation / 0.1. Third-party OpenCV in the
 * reference -- PARITY UNPINNED; the kernel follows oracle/src/ecc.c operation for operation (incl. its summation order).
 * One handle = one video stream (it keeps the previous frame's 0.1-scaled grey image).
 *   warp6: [[cos, -sin, tx], [sin, cos, ty]] as doubles of the float32 values the reference's numpy array holds, ready for
 *     tlk_ssort_camera_update. status: 0 on the first frame (identity; the reference has no previous frame and skips the update),
 *     n >= 1 = Gauss-Newton iterations run, -1 where cv2 raises (NaN correlation / non-positive lambda denominator): the reference
 *     catches it and skips the camera update -- so must the caller.
 *   tlk_ecc_apply_dev: frame (h, w, 3) uint8, warp6 and status in device memory; asynchronous on hip_stream.
 *   tlk_ecc_apply: host frame, host results (rho = final correlation coefficient, may be NULL).
 *   tlk_ecc_find_transform (tests): findTransformECC alone on two (h, w) uint8 host images, translation not rescaled.
 * ------------------------------------------------------------------------------------------ */
typedef struct tlk_ecc tlk_ecc;
int tlk_ecc_create(int h, int w, int device, tlk_ecc **out);
int tlk_ecc_destroy(tlk_ecc *c);
int tlk_ecc_reset(tlk_ecc *c);                        /* new video: forget the previous frame */
int tlk_ecc_apply_dev(tlk_ecc *c, const uint8_t *frame_dev, double *warp6_dev, int *status_dev, void *hip_stream);
int tlk_ecc_apply(tlk_ecc *c, const uint8_t *frame_host, double *warp6_host, int *status, double *rho);
int tlk_ecc_find_transform(const uint8_t *templ_host, const uint8_t *image_host, int h, int w, int max_iter, double eps,
                           double *warp6_host, int *status, double *rho, int device);

/* ------------------------------------------------------------------------------------------
 * HOTA of one sequence on the device. Replaces HOTA.eval_sequence of the TrackEval copy the reference vendors
 * (plugins/eval/PoseTrack21/posetrack21/posetrack21/trackeval/metrics/hota.py:30-155; official path: pip trackeval behind
 * tracklab/wrappers/eval/trackeval_evaluator.py:28-110). HOST buffers: gt_ids / tr_ids (ids re-labelled 0..n-1 like TrackEval's
 * preprocessing, frames concatenated), *_ltrb (., 4) float64 x0 y0 x1 y1, *_off (n_frames + 1) frame offsets, alphas19 the 19
 * thresholds (np.arange(0.05, 0.99, 0.05)). stats (7, 19) float64: HOTA_TP, HOTA_FN, HOTA_FP, LocA sum, AssA, AssRe, AssPr per
 * threshold -- what tracklab_amd.hota.pack() turns into the vector the ranks SUM all-reduce. At most 512 boxes per frame and side.
 * ------------------------------------------------------------------------------------------ */
int tlk_hota_sequence_f64(const int32_t *gt_ids, const double *gt_ltrb, const int64_t *gt_off, const int32_t *tr_ids, const double *tr_ltrb,
                          const int64_t *tr_off, int n_frames, int n_gt, int n_tr, const double *alphas19, double *stats);
/* The same with all six arrays already in DEVICE memory -- the tracker side straight from the engine's HBM-resident per-video table
 * (tracklab_amd.evaluate.evaluate_device_log: no host round trip of the table), the ground truth uploaded once by the caller. n_gt_boxes /
 * n_tr_boxes (the totals) and n_match (>= sum over the frames of gt boxes x tracker boxes: it sizes the similarity matrices) are host
 * scalars; alphas19 and stats stay host buffers (stats is valid on return: the call synchronises hip_stream). */
int tlk_hota_sequence_dev_f64(const int32_t *gt_ids_dev, const double *gt_ltrb_dev, const int64_t *gt_off_dev, const int32_t *tr_ids_dev,
                              const double *tr_ltrb_dev, const int64_t *tr_off_dev, int n_frames, int n_gt, int n_tr, int64_t n_gt_boxes,
                              int64_t n_tr_boxes, int64_t n_match, const double *alphas19, double *stats, void *hip_stream);

/* CLEAR-MOT and ID counts of one sequence on the device. Replaces MOTAccumulator.update per frame + the measures of the
 * py-motmetrics copy the reference vendors for its PoseTrack21 MOT evaluator (plugins/eval/PoseTrack21/posetrack21_mot/posetrack21_mot/
 * motmetrics: mot.py:134-345, metrics.py:342-728 incl. id_global_assignment :610-653, distances.py:83-129 iou_matrix(max_iou),
 * lap.py:79-130). HOST buffers like tlk_hota_sequence_f64, except: ids dense 0..n-1 in the SORTED order of the original ids, boxes
 * (x, y, w, h). counts19: num_frames, num_matches, num_switches, num_transfer, num_ascend, num_migrate, num_false_positives, num_misses,
 * num_objects, num_predictions, num_unique_objects, mostly_tracked, partially_tracked, mostly_lost, num_fragmentations, sum_distance,
 * idtp, idfp, idfn -- the summable fields (tracklab_amd.clearmot.SUM_FIELDS) the ranks SUM all-reduce; MOTA / MOTP / IDF1 ... are
 * ratios of them (clearmot.finalize). At most 512 boxes per frame and side. */
int tlk_clear_sequence_f64(const int32_t *gt_ids, const double *gt_ltwh, const int64_t *gt_off, const int32_t *tr_ids, const double *tr_ltwh,
                           const int64_t *tr_off, int n_frames, int n_gt, int n_tr, double max_iou, double *counts19);
/* The same with all six arrays already in DEVICE memory (see tlk_hota_sequence_dev_f64); counts19 is a host buffer, valid on return. */
int tlk_clear_sequence_dev_f64(const int32_t *gt_ids_dev, const double *gt_ltwh_dev, const int64_t *gt_off_dev, const int32_t *tr_ids_dev,
                               const double *tr_ltwh_dev, const int64_t *tr_off_dev, int n_frames, int n_gt, int n_tr, double max_iou,
                               double *counts19, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Fused convolution epilogue for the PyTorch-ROCm backbones (not a reference function: the reference's
 * backbones run inside third-party ONNXRuntime / torchreid): x = act(x + bias[c] (+ residual)) in place on a
 * channels-last activation viewed as (rows, channels), channels % 8 == 0. act: 0 none, 1 ReLU, 2 SiLU.
 * dtype TLK_F16 or TLK_BF16; bias has `channels` elements of the same dtype; residual may be NULL.
 * ------------------------------------------------------------------------------------------ */
int tlk_bias_act_nhwc(void *x_dev, const void *bias_dev, const void *residual_dev, long long rows, int channels,
                      int act_kind, int dtype, void *hip_stream);

/* act_kind of the convolution entry points below: 0 none / 1 ReLU / 2 SiLU, optionally OR-ed with TLK_ACT_RES_AFTER: the residual is then added
 * AFTER the activation, y = act(conv + bias) + residual (the identity add of a CSPNeXt block, mmdet CSPNeXtBlock.forward), instead of before it
 * (y = act(conv + bias + residual): the ResNet bottleneck). */
#define TLK_ACT_RES_AFTER 0x100

/* fp32 convolution of a channels-last activation with the convolution epilogue inside -- the backbones at the REFERENCE's precision
 * (the reference runs them in fp32: configs/modules/track/strong_sort.yaml:10 `fp16: false`; ONNXRuntime fp32 behind
 * wrappers/bbox_detector/rtmlib_api.py:21 and wrappers/pose_estimator/rtmlib_api.py:21; torchreid fp32 behind wrappers/reid/kpreid_api.py:147-182):
 *   y[n,ho,wo,co] = act( sum_{kh,kw,ci} x[n, ho*stride+kh-pad, wo*stride+kw-pad, ci] * w[co,kh,kw,ci] + bias[co] (+ residual[n,ho,wo,co]) )
 * x (n,h,w,cin) and y (n,ho,wo,cout) are NHWC (torch channels_last), w is (cout,kh,kw,cin) (torch's channels_last weight), cin % 4 == 0,
 * x and w 16-byte aligned; bias / residual may be NULL; act 0 none / 1 ReLU / 2 SiLU.  *_pix_stride = floats between two pixels of x / y /
 * residual (0 = densely packed): a call may read or write a channel slice of a wider tensor (e.g. its part of a concatenation).
 * Hand-written implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, tlk_conv.hip).  Each output element is ONE fmaf chain over
 * k = (kh,kw,ci) in the order 0,4,1,5,2,6,3,7 within every group of 8 (groups ascending, zero terms where the tap is outside the image),
 * then + bias, + residual, activation: bit-identical for every tile configuration and to oracle/src/conv.c. */
int tlk_conv2d_nhwc_f32(const float *x_dev, const float *w_dev, const float *bias_dev, const float *residual_dev, float *y_dev,
                        int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                        int x_pix_stride, int y_pix_stride, int res_pix_stride, void *hip_stream);
/* Probes / tests: force one of the kernel's tile configurations (0: 128x128, 1: 256x64, 2: 128x64, 3: 256x96, 4: 256x32, 5: 64x128, 6: 128x64 (waves stacked)
 * pixels x output channels; r05: 7 / 8 / 9 = 128x128 / 128x64 / 64x128 with ONE LDS stage and the epilogue in passes; 21..33 = the direct-to-LDS
 * kernels of tlk_conv16x.hip on fp32 tensors (need cin % 32 == 0), of which 30..33 are the PATCH-resident 3 x 3 kernels: stride 1, pad 1,
 * cin == 32 exactly, tiles of whole image rows (8 <= wo <= 64, tile rows a multiple of wo, ho * wo a multiple of the tile) -- TLK_EINVAL
 * names the violated constraint; the heuristic routes 32-wide layers to 27 / 31 by itself; r06: 34..37 the same on 64 channels, 38 / 39 = 128x128
 * and 256x128 two-stage tiles with the residual read in the epilogue); -1 = the heuristic (r06: re-derived from a sweep of every layer of six
 * networks under every configuration, tools/sweep_conv_f32.py).  Results do not depend on it: one fmaf chain, as above. */
int tlk_conv2d_set_config(int cfg);
/* Tile configuration the most recent tlk_conv2d_nhwc_f32 call of this process launched (as above; 15 = the direct RGB stem kernel), -1 before the first. */
int tlk_conv2d_last_config(void);
/* r05.  cin == 3 (an RGB stem: 7x7 or 3x3, stride 2, cout <= 64, no residual) is accepted too and goes to a direct kernel (tlk_conv_stem.hip) that
 * reads the 3-channel image as it is -- its chain is the one above on the image padded to 4 channels, minus the zero terms: bit-identical.
 * Dynamic batch: every convolution launched (or captured into a hipGraph) while n_images_dev is set reads its image count from
 * n_images_dev[0] WHEN THE KERNEL RUNS and treats the `n` of the call as the capacity; rows beyond are neither read nor written and the
 * workgroups beyond them leave at once.  This is how the ReID batch of a step is convolved on its real crops only (the reference batches real
 * detections only, tracklab/wrappers/reid/kpreid_api.py:147-182) from ONE captured graph.  NULL switches it off.  Applies to
 * tlk_conv2d_nhwc_f32, tlk_conv2d_nhwc_16, tlk_conv_stem16_nhwc and tlk_maxpool2d_nhwc.  The setting belongs to the CALLING HOST THREAD
 * (thread-local): launches of other threads are not affected; set it, launch (or capture), clear it -- in one thread. */
int tlk_conv_set_dynamic_batch(const int32_t *n_images_dev);

/* The same convolution on the 16-bit MFMA (v_mfma_f32_32x32x16_f16; tlk_conv16.hip), two modes selected by the pointers given:
 *   f16 mode   (x_lo_dev == NULL): x, w, residual, y are f16 NHWC / (cout,kh,kw,cin); fp32 accumulation and epilogue; bias fp32.
 *   split mode (x_lo_dev != NULL): every tensor is a PAIR of f16 planes (hi, lo) with value = hi + lo * 2^-11 -- an fp32 number to a relative
 *              2^-22; three MFMAs per operand pair (hi*hi, hi*lo, lo*hi), two fp32 accumulators: fp32-class results (|err| <= ~3 * 2^-22 *
 *              sum|a||b| + fp32 accumulation round-off; tests/test_gpu_conv16.py holds it to the SAME fp64 bound as tlk_conv2d_nhwc_f32)
 *              at ~5x the fp32-input MFMA peak.  Range |x| <= 65504.  tlk_split_f32_planes / tlk_merge_planes_f32 convert.
 * y_f32_dev != NULL: the output is written as plain fp32 there instead of y_dev (/ y_lo_dev).  cin % 8 == 0, cout % 8 == 0, all pointers
 * 16-byte aligned; pixel strides in ELEMENTS (0 = dense).  Same reference role as tlk_conv2d_nhwc_f32 (the backbones of
 * wrappers/bbox_detector/rtmlib_api.py:21, wrappers/reid/kpreid_api.py:147-182, wrappers/pose_estimator/rtmlib_api.py:21). */
int tlk_conv2d_nhwc_16(const void *x_dev, const void *x_lo_dev, const void *w_dev, const void *w_lo_dev, const float *bias_dev,
                       const void *res_dev, const void *res_lo_dev, void *y_dev, void *y_lo_dev, float *y_f32_dev,
                       int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                       int x_pix_stride, int y_pix_stride, int res_pix_stride, void *hip_stream);
/* r06 -- SCALED split planes: the value of a plane pair is scale * (hi + lo * 2^-11) with `scale` a power of two >= 1 held in device memory
 * beside the tensor, so that activations beyond float16's range no longer saturate the split-precision networks (the
 * reference's fp32 networks have fp32's range: tracklab/wrappers/reid/kpreid_api.py:147-182).  Multiplying by a power of two is exact, so a
 * tensor that fits float16 keeps scale 1 and the planes -- and every result -- of the unscaled call, bit for bit.
 *   tlk_conv2d_nhwc_16s = tlk_conv2d_nhwc_16 (split mode only) + in_scale (1 float: the scale of the x planes, NULL = 1), res_scale (the residual
 *     planes'), out_state (2 floats {scale of the planes this call writes, largest |output| recorded by atomic max}; NULL = scale 1, nothing
 *     recorded; must be NULL with y_f32).  The scale in use is read when the kernel RUNS (a captured hipGraph follows it).
 *   tlk_split_f32_planes_s / tlk_merge_planes_f32_s: the conversions with a state / scale; pixels_per_image > 0 makes the split honour
 *     tlk_conv_set_dynamic_batch (live images only are converted and recorded).
 *   tlk_split_scale_update: n states {scale, recorded maximum} -> scale for the NEXT forward: the smallest power of two >= 1 with
 *     maximum / scale <= 2^14 when that is larger than the current one (growth at once), twice that value when the maximum fell a factor 8 below
 *     (hysteresis), else unchanged; maxima cleared.  *changed_dev (nullable) += number of states that grew or recorded a non-finite maximum --
 *     the calibration loop (run the network, update, repeat while it is non-zero) reads it; in steady state the update is the forward's last
 *     node and nothing is read. */
int tlk_conv2d_nhwc_16s(const void *x_dev, const void *x_lo_dev, const void *w_dev, const void *w_lo_dev, const float *bias_dev,
                        const void *res_dev, const void *res_lo_dev, void *y_dev, void *y_lo_dev, float *y_f32_dev,
                        int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                        int x_pix_stride, int y_pix_stride, int res_pix_stride, const float *in_scale_dev, const float *res_scale_dev,
                        float *out_state_dev, void *hip_stream);
int tlk_split_f32_planes_s(const float *x_dev, long long pixels, int c_in, int x_pix_stride, int c_out, void *hi_dev, void *lo_dev, float *state_dev,
                           long long pixels_per_image, void *hip_stream);
int tlk_merge_planes_f32_s(const void *hi_dev, const void *lo_dev, long long n, const float *scale_dev, float *y_dev, void *hip_stream);
int tlk_split_scale_update(float *states_dev, int n_states, int *changed_dev, void *hip_stream);
/* r06 -- the element-wise joints of a split-precision network in ONE pass (tlk_split_fuse.hip): y = [relu]( sum of 1..4 terms ), written as
 * scaled (hi, lo) planes (out_state_dev as in tlk_conv2d_nhwc_16s: {scale in use, recorded maximum}; NULL = scale 1).  Term t is a plane pair
 * (hi_dev[t], lo_dev[t], scale_dev[t] nullable = 1) or one fp32 tensor (f32_dev[t]; hi / lo NULL), NHWC with c channels at resolution
 * (h >> shift[t], w >> shift[t]) -- nearest up-sampling by 2^shift -- and pixel stride pix_stride[t] elements (0 = c).  The tables are HOST arrays of
 * n_terms entries holding device pointers.  y_pix_stride > c writes a channel slice of a wider tensor (one term, no ReLU: the concatenation of
 * branches with different scales onto one).  dynamic_batch != 0 honours tlk_conv_set_dynamic_batch.  c % 8 == 0, 16-byte aligned pointers.
 * The sum is taken in term order in fp32, as torch does for HRNet's exchange units (the ReID backbone the reference's yaml selects:
 * tracklab/configs/modules/reid/bpbreid.yaml:53, run through tracklab/wrappers/reid/kpreid_api.py:147-182). */
int tlk_split_fuse_sum(int n_terms, const void *const *hi_dev, const void *const *lo_dev, const float *const *f32_dev,
                       const float *const *scale_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                       void *y_hi_dev, void *y_lo_dev, int y_pix_stride, float *out_state_dev, int dynamic_batch, void *hip_stream);
/* The same joint for the EXACT fp32 route (same kernel, fp32 terms and fp32 output): y = [relu](((x_0 + x_1) + x_2) + x_3), term t an fp32 NHWC tensor at
 * (h >> shift[t], w >> shift[t]) -- the association and rounding of torch's `y = y + t` chain over nearest-up-sampled tensors followed by relu, so the
 * result equals that composition's bit for bit, in one pass instead of 2 n (HRNet's exchange units at the reference's precision,
 * tracklab/configs/modules/reid/bpbreid.yaml:53 through tracklab/wrappers/reid/kpreid_api.py:147-182).  y_pix_stride > c: a channel slice of a wider
 * tensor (one term: up-sampling + concatenation in one pass). */
int tlk_fuse_sum_f32(int n_terms, const float *const *x_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                     float *y_dev, int y_pix_stride, int dynamic_batch, void *hip_stream);
/* ... and for the f16 route: f16 terms, f16 output, every partial sum rounded to f16 -- what torch's half-precision `y = y + t` chain computes (each
 * add in fp32, rounded to half), bit for bit. */
int tlk_fuse_sum_f16(int n_terms, const void *const *x_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                     void *y_dev, int y_pix_stride, int dynamic_batch, void *hip_stream);
/* Probes / tests: 0 = the register-staged kernel for every shape, 1 (default; env TLK_CONV16_GLDS) = the direct-to-LDS kernel where it applies
 * (Cout > 64, Cin a multiple of the K step).  Results do not depend on it beyond fp32 summation order inside a 16-wide slice (none: same order). */
int tlk_conv16_set_glds(int on);
/* r05: the large-tile / one-stage / ring kernels of tlk_conv16x.hip (direct-to-LDS buffer loads with hardware zero fill, XOR-swizzled LDS rows,
 * one to four LDS stages with counted waits, residual prefetched into registers).  cfg 0 (default) = they take the shapes their launch-size
 * heuristic claims (cin a multiple of the K step: 64 in f16 mode, 32 in split mode) and the r04 kernels the rest; -1 = r04 kernels only;
 * 1..26 (f16; 23..26 = 32-column tiles, 23 / 24 half-step; 17 / 18 = the patch-resident 3 x 3 kernel: stride 1, exactly 64 channels, whole image rows per tile; r06: 19..22 = the HALF-STEP
 * tiles, K step 32 halfs, for cin a multiple of 32 but not of 64 -- the only ones such a layer accepts) / 1..12 (split; 8..12: 32-column tiles) = force one
 * tile configuration (probes / tests).  Same arithmetic contract as above in every configuration.  r06: in split mode the heuristic also
 * takes the 1 x 1 expansions WITH residual (128 x 128 tiles of eight 32 x 64 wavefronts, configurations 6 / 7). */
int tlk_conv16_set_config(int cfg);
/* Tile configuration the most recent tlk_conv2d_nhwc_16 / _16s call of this process launched (as above), -1 = the r04 kernels (tests of the heuristic). */
int tlk_conv16_last_config(void);
/* fp32 NHWC pixels (c_in channels, x_pix_stride floats apart, 0 = dense) -> (hi, lo) f16 planes with c_out >= c_in channels, zero padded. */
int tlk_split_f32_planes(const float *x_dev, long long pixels, int c_in, int x_pix_stride, int c_out, void *hi_dev, void *lo_dev, void *hip_stream);
/* y[i] = hi[i] + lo[i] * 2^-11 for n elements. */
int tlk_merge_planes_f32(const void *hi_dev, const void *lo_dev, long long n, float *y_dev, void *hip_stream);

/* Depthwise k x k convolution (k = 3 or 5, stride 1, pad k/2) of a channels-last activation with bias + activation inside -- the depthwise
 * halves of RTMPose's CSPNeXt blocks (mmdet DepthwiseSeparableConvModule; the reference runs the network in ONNXRuntime behind
 * tracklab/wrappers/pose_estimator/rtmlib_api.py:21-36, configs/modules/pose_estimator/rtmpose_rtmlib.yaml):
 *   y[n,oy,ox,ch] = act( sum_{ky,kx} x[n, oy+ky-k/2, ox+kx-k/2, ch] * w[ky,kx,ch] + bias[ch] )
 * x, y (n,h,w,c) NHWC of `dtype` (TLK_F32 or TLK_F16), w (k,k,c) of `dtype` (torch's (c,1,k,k) weight permuted to taps-major), bias fp32 or
 * NULL, c a multiple of 16 bytes of elements, x / w / y 16-byte aligned, pixel strides in elements (0 = dense; a call may read or write a
 * channel slice of a wider tensor).  fp32 accumulation for both types; each output element is ONE fmaf chain over ky then kx ascending
 * (rows outside the image skipped, columns outside as zero terms): bit-identical to oracle/src/conv.c orc_dwconv2d_nhwc_f32 in fp32.
 * Every input element read once from HBM, every output written once (tlk_dwconv.hip).  r06: the arithmetic is packed fp32 FMAs (two channels
 * per instruction, each half an ordinary fused multiply-add: same results), f16 inputs are widened once per loaded pixel, a lane owns two
 * columns; SiLU is v * rcp(1 + exp(-v)) with the device reciprocal (~3e-7 relative; the pointwise kernels keep the IEEE division). */
int tlk_dwconv2d_nhwc(const void *x_dev, const void *w_dev, const float *bias_dev, void *y_dev, int n, int h, int w, int c, int k,
                      int act_kind, int dtype, int x_pix_stride, int y_pix_stride, void *hip_stream);
/* Probes: 0 = the default lane shape (two columns per lane, input rows loaded two ahead), 1..4 = (columns per lane, rows ahead) = (1, 1),
 * (2, 1), (1, 2), (2, 2).  Same results in every configuration. */
int tlk_dwconv_set_config(int cfg);

/* The pooling half of an SPPBottleneck (YOLOX CSPDarknet / RTMPose CSPNeXt; kernel sizes 5, 9, 13, stride 1, -inf padding; the reference runs
 * both networks in ONNXRuntime behind tracklab/wrappers/bbox_detector/rtmlib_api.py:21 and wrappers/pose_estimator/rtmlib_api.py:21):
 *   y[n,y,x, 0:c] = x,  y[.., c:2c] = max over the 5 x 5 window,  y[.., 2c:3c] = 9 x 9,  y[.., 3c:4c] = 13 x 13
 * x (n,h,w,c) and y (n,h,w,4c) NHWC of `dtype` (TLK_F32 or TLK_F16), c a multiple of 16 bytes of elements, 16-byte aligned, pixel strides in
 * elements (0 = dense).  One pass: the map is read once and the concatenation written once (tlk_spp.hip); max is exact, the result equals
 * torch's max_pool2d + cat bit for bit on NaN-free input (oracle/src/conv.c orc_spp_maxpool_nhwc_f32). */
int tlk_spp_maxpool_nhwc(const void *x_dev, void *y_dev, int n, int h, int w, int c, int dtype, int x_pix_stride, int y_pix_stride,
                         void *hip_stream);

/* r05: plain k x k max pooling with stride and -inf padding (pad <= k / 2) of a channels-last map, NHWC in and out: the pool behind ResNet-50's
 * stem in the ReID networks (3 x 3, stride 2, pad 1; torchreid behind tracklab/wrappers/reid/kpreid_api.py:147-182).  Same tensor conventions as
 * tlk_spp_maxpool_nhwc; max is exact, so the result equals torch's max_pool2d bit for bit on NaN-free input.  Honours
 * tlk_conv_set_dynamic_batch (images beyond the live count are not written). */
int tlk_maxpool2d_nhwc(const void *x_dev, void *y_dev, int n, int h, int w, int c, int k, int stride, int pad, int dtype, int x_pix_stride,
                       int y_pix_stride, void *hip_stream);

/* r05: the RGB stem of a backbone in f16 -- Cin = 3, 7 x 7 (ResNet-50) or 3 x 3 (HRNet-W32, RTMPose), stride 2, Cout <= 64 and a multiple of 8 --
 * as ONE direct kernel on v_mfma_f32_32x32x16_f16 with bias + activation inside and, with `pool` != 0, the 3 x 3 / stride 2 / pad 1 max-pool
 * that follows ResNet-50's stem fused behind it (torchreid's resnet.py conv1 -> bn1 -> relu -> maxpool, behind
 * tracklab/wrappers/reid/kpreid_api.py:147-182): the full-resolution map is never written.  Same arithmetic as tlk_conv2d_nhwc_16's f16 mode
 * (f16 operands, fp32 accumulation, fp32 bias, result rounded to f16; max commutes with the rounding, so the fused pool equals the two-pass one).
 *   x  (n, h, w) pixels of x_pix_stride halfs (0 = 3), channels 0..2 used;
 *   y  `pool` ? (n, (ho - 1) / 2 + 1, (wo - 1) / 2 + 1, cout) : (n, ho, wo, cout), y_pix_stride halfs per pixel (0 = cout), 16-byte aligned;
 *   packed_w: the (cout, kh, kw, 3) f16 weight in MFMA fragment order, tlk_conv_stem16_packed_halfs(...) halfs long, written by
 *      tlk_conv_stem16_pack (once per weight version: the caller caches it).
 * `pool` needs wo <= 64 (one column strip).  Honours tlk_conv_set_dynamic_batch.  TLK_EINVAL names the violated constraint. */
long long tlk_conv_stem16_packed_halfs(int cout, int kh, int kw, int stride);
int tlk_conv_stem16_pack(const void *w_dev, void *packed_dev, int cout, int kh, int kw, int stride, void *hip_stream);
int tlk_conv_stem16_nhwc(const void *x_dev, const void *packed_w_dev, const float *bias_dev, void *y_dev, int n, int h, int w, int cout, int kh, int kw,
                         int stride, int pad, int act, int pool, int x_pix_stride, int y_pix_stride, void *hip_stream);

/* 1x1 convolution of a channels-last tensor as ONE GEMM with the convolution epilogue inside:
 *   out[M,N] = act(x[M,K] . w[N,K]^T + bias[N] (+ residual[M,N])),  act 0 none / 1 ReLU / 2 SiLU, dtype TLK_F16 or TLK_BF16.
 * hipBLASLt (library GEMM, taken from the process with dlopen) with its BIAS / RELU_BIAS / SWISH_BIAS epilogue and beta*C for
 * the residual; the algorithm is tuned per (M,N,K,epilogue) over the heuristic's candidates on first use and cached.
 * Replaces conv + tlk_bias_act_nhwc for the bottleneck 1x1 convolutions of the backbones. Returns TLK_EUNSUPPORTED when
 * hipBLASLt or a matching algorithm is missing (callers fall back to GEMM + tlk_bias_act_nhwc, still on the GPU). */
int tlk_gemm_bias_act(const void *x_dev, const void *w_dev, const void *bias_dev, const void *residual_dev, void *out_dev,
                      long long M, int N, int K, int act, int dtype, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * r06: the prediction heads of the two networks of the path, ONE launch each (tlk_heads.hip).  Not reference functions (the reference's heads
 * run inside third-party graphs: rtmlib's YOLOX ONNX model behind tracklab/wrappers/bbox_detector/rtmlib_api.py:21,30; the torchreid fork's
 * BPBReID / KPR model behind tracklab/wrappers/reid/kpreid_api.py:147-182) -- they replace the 1 / 4 / 6-channel library convolutions and the
 * element-wise torch passes that followed them (~40 + ~14 launches per step).
 * Arithmetic contract (oracle/src/heads.c): every dot product is ONE fmaf chain over the channels ascending from 0, then + bias;
 * sigmoid(v) = 1 / (1 + exp(-v)); softmax over k = exp(l_k - max_k l) / (sum over k ascending); pooled sums are fmaf chains over the pixels
 * ascending.  Bit-exact against the oracle except for the device's exp (the tests allow 2e-6 relative behind an exp).
 *
 * tlk_yolox_head_nhwc: YOLOX's decoupled head outputs for `levels` (<= 4) pyramid levels.  Per level l: cls_feat[l] / reg_feat[l] are the
 *   (batch, hw[l], channels) NHWC outputs of the classification / regression branches (dtype TLK_F32 or TLK_F16, pixel strides in elements,
 *   0 = dense, 16-byte aligned), w[l] is (5 + num_classes, channels) fp32 -- rows 0..3 the reg prediction, 4 the obj prediction (both read
 *   reg_feat), 5.. the cls predictions (read cls_feat) -- and b[l] the 5 + num_classes fp32 biases.  out (batch, A, 5 + num_classes) fp32,
 *   A = sum hw[l], level l at anchors [sum_{i<l} hw[i], +hw[l]): [reg0..3 raw, sigmoid(obj), sigmoid(cls...)] -- the tensor
 *   tlk_yolox_decode_nms consumes.  The pointer arrays are HOST arrays of device pointers.
 * tlk_reid_part_head: feat (rows_dense, hw, dim) NHWC feature map (TLK_F32 / TLK_F16), w (parts, dim) / b (parts) fp32 pixel-wise part
 *   classifier; for every output row r: att = softmax_k(w . feat[p] + b) per pixel p, emb[r,k,:] = sum_p att[p,k] feat[p,:] / max(sum_p att[p,k], 1e-6),
 *   vis[r,k] = (k == 0) || (max_p att[p,k] > vis_threshold)  (pass the model's threshold / parts).  With counts (frames, i32) and max_dets:
 *   rows = frames * max_dets, row r = (frame b, slot j) is live when j < counts[b] and reads dense row slot_base[b] + j (slot_base NULL:
 *   row r itself); padding rows are ZERO-FILLED (emb and vis).  nonfinite_flag (1 byte, nullable) is set to 1 when a live embedding
 *   value is NaN or infinite (never cleared here).  dim % (16 bytes of elements) == 0, dim <= 512, parts <= 8. */
int tlk_yolox_head_nhwc(const void *const *cls_feat_dev, const void *const *reg_feat_dev, const int *hw, const int *cls_pix_stride,
                        const int *reg_pix_stride, const float *const *w_dev, const float *const *b_dev, int levels, int batch, int channels,
                        int num_classes, int dtype, float *out_dev, void *hip_stream);
int tlk_reid_part_head(const void *feat_dev, int feat_pix_stride, int dtype, int hw, int dim, int parts, const float *w_dev, const float *b_dev,
                       const int32_t *counts_dev, const int32_t *slot_base_dev, int rows, int max_dets, float vis_threshold,
                       float *emb_dev, unsigned char *vis_dev, unsigned char *nonfinite_flag_dev, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* TLK_H */
