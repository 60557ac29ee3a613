/* CPU ORACLE (test infrastructure) -- see orc.h.
 *
 * Detector / ReID pre- and post-processing. In the reference this arithmetic lives in THIRD-PARTY
 * packages that are not vendored and not installed here (rtmlib 0.0.13, opencv-python 4.11,
 * torchreid fork + albumentations): call sites tracklab/wrappers/bbox_detector/rtmlib_api.py:27-46 and
 * tracklab/wrappers/reid/kpreid_api.py:115-144. No reference test pins them -> PARITY UNPINNED; the
 * functions below restate the published algorithms:
 *   - cv2.resize(INTER_LINEAR) on uint8: OpenCV imgproc/resize.cpp fixed-point path
 *     (INTER_RESIZE_COEF_BITS = 11; horizontal pass in int32, vertical pass
 *     ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).
 *   - rtmlib YOLOX.preprocess: ratio = min(H_in/h, W_in/w), resize to (int(w*ratio), int(h*ratio)),
 *     paste top-left on a 114-filled canvas, float32 CHW, no normalisation.
 *   - rtmlib YOLOX.postprocess: (xy + grid) * stride, exp(wh) * stride, score = obj * cls,
 *     xyxy / ratio, per-class greedy NMS with the "+1" pixel convention, nms_thr 0.45, score_thr 0.7.
 *   - KPReId.preprocess crop image[t:b, l:r] of the rounded/clipped ltrb (coordinates.py:216-267),
 *     albumentations Resize (cv2 INTER_LINEAR) + Normalize ((x - 255*mean) * (1/(255*std)), float32).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cv2 fixed-point coefficient for destination index d: source index (unclamped for rows, clamped for
 * columns exactly as resize.cpp does) and the two int16 weights. */
static void cv_coef(int d, int ssize, int dsize, int is_col, int *s0, short *w0, short *w1)
{
    double scale = (double)ssize / dsize;
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (is_col) {
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    }
    float c0 = 1.f - f, c1 = f;
    *w0 = (short)lrintf(c0 * 2048.f);
    *w1 = (short)lrintf(c1 * 2048.f);
    *s0 = s;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* src: (sh, sw, 3) uint8 with row stride `sstride` bytes; dst (dh, dw, 3) uint8 contiguous */
void orc_cv_resize_linear_u8(const uint8_t *src, int sh, int sw, int sstride, uint8_t *dst, int dh, int dw)
{
    for (int dy = 0; dy < dh; ++dy) {
        int sy; short b0, b1;
        cv_coef(dy, sh, dh, 0, &sy, &b0, &b1);
        const uint8_t *r0 = src + (size_t)clampi(sy, 0, sh - 1) * sstride;
        const uint8_t *r1 = src + (size_t)clampi(sy + 1, 0, sh - 1) * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx; short a0, a1;
            cv_coef(dx, sw, dw, 1, &sx, &a0, &a1);
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            for (int c = 0; c < 3; ++c) {
                int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
                int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
                int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)dy * dw + dx) * 3 + c] = (uint8_t)clampi(v, 0, 255);
            }
        }
    }
}

/* rtmlib YOLOX.preprocess. img (h, w, 3) u8 -> out (3, S, S) float32 (CHW), returns ratio */
double orc_letterbox(const uint8_t *img, int h, int w, int S, float *out)
{
    double ratio = fmin((double)S / h, (double)S / w);
    int rw = (int)(w * ratio), rh = (int)(h * ratio);
    uint8_t *tmp = malloc((size_t)rw * rh * 3);
    orc_cv_resize_linear_u8(img, h, w, w * 3, tmp, rh, rw);
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x)
                out[((size_t)c * S + y) * S + x] = (y < rh && x < rw) ? (float)tmp[((size_t)y * rw + x) * 3 + c] : 114.f;
    free(tmp);
    return ratio;
}

/* KPReId crop + albumentations Resize + Normalize. ltrb int (exclusive r, b), out (3, oh, ow) float32 CHW.
 * channel c of the source is used as-is (the caller supplies RGB like cv2_load_image does). */
void orc_crop_resize_norm(const uint8_t *img, int h, int w, const int32_t *ltrb, int oh, int ow,
                          const float *mean3, const float *std3, float *out)
{
    int l = ltrb[0], t = ltrb[1], r = ltrb[2], b = ltrb[3];
    int cw = r - l, ch = b - t;
    if (cw <= 0 || ch <= 0) { memset(out, 0, sizeof(float) * 3 * (size_t)oh * ow); return; }
    uint8_t *tmp = malloc((size_t)oh * ow * 3);
    orc_cv_resize_linear_u8(img + ((size_t)t * w + l) * 3, ch, cw, w * 3, tmp, oh, ow);
    for (int c = 0; c < 3; ++c) {
        float m = mean3[c] * 255.f;
        float den = 1.0f / (std3[c] * 255.f);
        for (int i = 0; i < oh * ow; ++i) {
            float v = (float)tmp[(size_t)i * 3 + c];
            v -= m; v *= den;
            out[(size_t)c * oh * ow + i] = v;
        }
    }
    free(tmp);
}

/* rtmlib YOLOX.postprocess on a raw head tensor pred (A, 5+C) float32 for input size S (A = sum (S/s)^2,
 * s in 8,16,32). Writes kept boxes xyxy (original image scale), scores, class ids in rtmlib order
 * (class-major, descending score inside a class; ties -> larger anchor index first). Returns count. */
typedef struct { float score; int idx; } sc_t;
static int cmp_desc(const void *a, const void *b)
{
    const sc_t *x = a, *y = b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (y->idx > x->idx) - (y->idx < x->idx);
}
int orc_yolox_postprocess(const float *pred, int S, int C, float ratio, float nms_thr, float score_thr,
                          float *boxes_out, float *scores_out, int32_t *cls_out, int cap)
{
    static const int strides[3] = {8, 16, 32};
    int A = 0;
    for (int k = 0; k < 3; ++k) A += (S / strides[k]) * (S / strides[k]);
    float *xyxy = malloc(sizeof(float) * 4 * (size_t)A), *area = malloc(sizeof(float) * (size_t)A);
    int a = 0;
    for (int k = 0; k < 3; ++k) {
        int hs = S / strides[k], ws = S / strides[k];
        float st = (float)strides[k];
        for (int gy = 0; gy < hs; ++gy)
            for (int gx = 0; gx < ws; ++gx, ++a) {
                const float *p = pred + (size_t)a * (5 + C);
                float cx = (p[0] + (float)gx) * st, cy = (p[1] + (float)gy) * st;
                float w = expf(p[2]) * st, h = expf(p[3]) * st;
                float x1 = cx - w / 2.f, y1 = cy - h / 2.f, x2 = cx + w / 2.f, y2 = cy + h / 2.f;
                x1 /= ratio; y1 /= ratio; x2 /= ratio; y2 /= ratio;
                xyxy[4 * a] = x1; xyxy[4 * a + 1] = y1; xyxy[4 * a + 2] = x2; xyxy[4 * a + 3] = y2;
                area[a] = (x2 - x1 + 1) * (y2 - y1 + 1);
            }
    }
    sc_t *cand = malloc(sizeof(sc_t) * (size_t)A);
    char *dead = malloc((size_t)A);
    int n_out = 0;
    for (int c = 0; c < C; ++c) {
        int n = 0;
        for (int i = 0; i < A; ++i) {
            const float *p = pred + (size_t)i * (5 + C);
            float s = p[4] * p[5 + c];
            if (s > score_thr) { cand[n].score = s; cand[n].idx = i; n++; }
        }
        qsort(cand, (size_t)n, sizeof(sc_t), cmp_desc);
        memset(dead, 0, (size_t)n);
        for (int i = 0; i < n; ++i) {
            if (dead[i]) continue;
            int bi = cand[i].idx;
            if (n_out < cap && cand[i].score > 0.3f) {
                memcpy(boxes_out + 4 * n_out, xyxy + 4 * bi, 16);
                scores_out[n_out] = cand[i].score; cls_out[n_out] = c; n_out++;
            }
            for (int j = i + 1; j < n; ++j) {
                if (dead[j]) continue;
                int bj = cand[j].idx;
                float xx1 = fmaxf(xyxy[4 * bi], xyxy[4 * bj]), yy1 = fmaxf(xyxy[4 * bi + 1], xyxy[4 * bj + 1]);
                float xx2 = fminf(xyxy[4 * bi + 2], xyxy[4 * bj + 2]), yy2 = fminf(xyxy[4 * bi + 3], xyxy[4 * bj + 3]);
                float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
                float inter = w * h;
                float ovr = inter / (area[bi] + area[bj] - inter);
                if (!(ovr <= nms_thr)) dead[j] = 1;
            }
        }
    }
    free(xyxy); free(area); free(cand); free(dead);
    return n_out;
}

/* coordinates.py:216-267: ltwh -> clipped, rounded (half to even) integer ltrb used for the ReID crop */
void orc_ltwh_to_crop_ltrb(const double *ltwh, int n, int img_w, int img_h, int32_t *ltrb)
{
    for (int i = 0; i < n; ++i) {
        double b0 = ltwh[4 * i], b1 = ltwh[4 * i + 1], b2 = ltwh[4 * i + 2], b3 = ltwh[4 * i + 3];
        b0 = fmax(0, fmin(b0, img_w - 2));
        b1 = fmax(0, fmin(b1, img_h - 2));
        b2 = fmax(1, fmin(b2, img_w - 1 - b0));
        b3 = fmax(1, fmin(b3, img_h - 1 - b1));
        ltrb[4 * i] = (int32_t)nearbyint(b0);
        ltrb[4 * i + 1] = (int32_t)nearbyint(b1);
        ltrb[4 * i + 2] = (int32_t)nearbyint(b0 + b2);
        ltrb[4 * i + 3] = (int32_t)nearbyint(b1 + b3);
    }
}
