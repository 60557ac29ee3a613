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

/* ------------------------------------------------------------------------------------------------------------------
 * Plain StrongSORT's ReID input. strong_sort.py:102-108 + :135-141 crop ori_img[y1:y2, x1:x2];
 * reid_multibackend.py:44-52, :184-195: ToPILImage -> Resize((256,128)) -> ToTensor -> Normalize(ImageNet).
 * The resize arithmetic is third-party Pillow (12.2.0 in this image), src/libImaging/Resample.c: bilinear filter with
 * support 1.0 scaled by the downscale factor (antialias), precompute_coeffs, normalize_coeffs_8bpc (22 fractional bits),
 * horizontal pass into an 8-bit intermediate, then vertical pass. Pinned on tests/golden/pil_preprocess.npz (made by Pillow).
 * ------------------------------------------------------------------------------------------------------------------ */
#define PIL_PRECISION_BITS (32 - 8 - 2)

static int pil_coeffs(int inSize, int outSize, int **bounds_out, int32_t **kk_out)
{
    double scale = (double)inSize / outSize, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    int *bounds = malloc(sizeof(int) * 2 * (size_t)outSize);
    int32_t *kk = malloc(sizeof(int32_t) * (size_t)outSize * ksize);
    double *k = malloc(sizeof(double) * ksize);
    for (int xx = 0; xx < outSize; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w; ww += w;
        }
        for (x = 0; x < xmax; ++x) if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0;
        for (x = 0; x < ksize; ++x)
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PIL_PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PIL_PRECISION_BITS));
        bounds[xx * 2] = xmin; bounds[xx * 2 + 1] = xmax;
    }
    free(k);
    *bounds_out = bounds; *kk_out = kk;
    return ksize;
}
static inline uint8_t pil_clip8(int v) { v >>= PIL_PRECISION_BITS; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* self.im.resize(size, BILINEAR, box) (ImagingResample): horizontal pass into an 8-bit intermediate, then the vertical pass */
static void pil_resample_core(const uint8_t *src, int sh, int sw, int sstride, uint8_t *dst, int dh, int dw)
{
    if (sh == dh && sw == dw) {                              /* Image.resize returns self.copy() when nothing changes */
        for (int y = 0; y < sh; ++y) memcpy(dst + (size_t)y * dw * 3, src + (size_t)y * sstride, (size_t)sw * 3);
        return;
    }
    int *bh, *bv; int32_t *kh, *kv;
    const int ksh = pil_coeffs(sw, dw, &bh, &kh), ksv = pil_coeffs(sh, dh, &bv, &kv);
    const int need_h = dw != sw, need_v = dh != sh;
    const int y_first = bv[0], y_last = bv[dh * 2 - 2] + bv[dh * 2 - 1];
    const uint8_t *in = src; int in_stride = sstride, in_w = sw;
    uint8_t *tmp = NULL;
    int voff = 0;                                            /* first source row of the vertical pass inside `in` */
    if (need_h) {
        tmp = malloc((size_t)(y_last - y_first) * dw * 3);
        for (int yy = 0; yy < y_last - y_first; ++yy)
            for (int xx = 0; xx < dw; ++xx) {
                const int xmin = bh[xx * 2], xmax = bh[xx * 2 + 1];
                const int32_t *k = kh + (size_t)xx * ksh;
                int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
                const uint8_t *row = src + (size_t)(yy + y_first) * sstride;
                for (int x = 0; x < xmax; ++x) { s0 += row[(x + xmin) * 3] * k[x]; s1 += row[(x + xmin) * 3 + 1] * k[x]; s2 += row[(x + xmin) * 3 + 2] * k[x]; }
                uint8_t *o = tmp + ((size_t)yy * dw + xx) * 3;
                o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
            }
        in = tmp; in_stride = dw * 3; in_w = dw; voff = y_first;    /* bounds_vert shifted by ybox_first */
    }
    if (need_v) {
        for (int yy = 0; yy < dh; ++yy) {
            const int ymin = bv[yy * 2] - voff, ymax = bv[yy * 2 + 1];
            const int32_t *k = kv + (size_t)yy * ksv;
            for (int xx = 0; xx < in_w; ++xx) {
                int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
                for (int y = 0; y < ymax; ++y) {
                    const uint8_t *p = in + (size_t)(y + ymin) * in_stride + xx * 3;
                    s0 += p[0] * k[y]; s1 += p[1] * k[y]; s2 += p[2] * k[y];
                }
                uint8_t *o = dst + ((size_t)yy * dw + xx) * 3;
                o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
            }
        }
    } else {
        for (int y = 0; y < dh; ++y) memcpy(dst + (size_t)y * dw * 3, in + (size_t)y * in_stride, (size_t)dw * 3);
    }
    free(tmp); free(bh); free(bv); free(kh); free(kv);
}

/* Image.resize((dw, dh), BILINEAR) of an RGB image src (sh, sw, 3) with row stride sstride bytes -> dst (dh, dw, 3).
 * Image.resize itself (PIL/Image.py, Pillow 12.2.0): `if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]` the image is resized
 * VERTICALLY first (to (sw, dh)) and horizontally afterwards -- two core calls, i.e. the uint8 rounding between the passes happens in the other order.
 * Only crops more than 100 times taller than wide that are being reduced in height: a 3 x 301 px box. Found by the r03 sweep fixture (pil_sweep.npz). */
void orc_pil_resize_bilinear_rgb(const uint8_t *src, int sh, int sw, int sstride, uint8_t *dst, int dh, int dw)
{
    if (sh == dh && sw == dw) { pil_resample_core(src, sh, sw, sstride, dst, dh, dw); return; }      /* Image.resize returns self.copy() */
    if (sh > sw * 100 && dh < sh) {
        uint8_t *mid = malloc((size_t)dh * sw * 3);
        pil_resample_core(src, sh, sw, sstride, mid, dh, sw);               /* width unchanged: vertical pass only */
        pil_resample_core(mid, dh, sw, sw * 3, dst, dh, dw);                /* height unchanged: horizontal pass only */
        free(mid);
        return;
    }
    pil_resample_core(src, sh, sw, sstride, dst, dh, dw);
}

/* strong_sort.py:43-51 + :102-108: xyxy (float64) -> xywh -> int-truncated, clipped x1,y1,x2,y2 */
void orc_ssort_crop_box(const double *xyxy, int img_w, int img_h, int32_t *out4)
{
    const double x = (xyxy[0] + xyxy[2]) / 2, y = (xyxy[1] + xyxy[3]) / 2, w = xyxy[2] - xyxy[0], h = xyxy[3] - xyxy[1];
    int x1 = (int)(x - w / 2), x2 = (int)(x + w / 2), y1 = (int)(y - h / 2), y2 = (int)(y + h / 2);
    out4[0] = x1 > 0 ? x1 : 0; out4[2] = x2 < img_w - 1 ? x2 : img_w - 1;
    out4[1] = y1 > 0 ? y1 : 0; out4[3] = y2 < img_h - 1 ? y2 : img_h - 1;
}

/* crop + Resize + ToTensor + Normalize: out (3, oh, ow) float32 = ((u8 / 255) - mean[c]) / std[c] in float32; also the resized u8 (oh, ow, 3) if u8_out */
void orc_ssort_reid_preprocess(const uint8_t *img, int h, int w, const double *xyxy, int oh, int ow, const float *mean3, const float *std3,
                               float *out, uint8_t *u8_out)
{
    int32_t b[4];
    orc_ssort_crop_box(xyxy, w, h, b);
    const int cw = b[2] - b[0], ch = b[3] - b[1];
    if (cw <= 0 || ch <= 0) {                                /* the reference raises on an empty crop; the HIP path emits a zero slot */
        memset(out, 0, sizeof(float) * 3 * (size_t)oh * ow);
        if (u8_out) memset(u8_out, 0, (size_t)oh * ow * 3);
        return;
    }
    uint8_t *r = malloc((size_t)oh * ow * 3);
    orc_pil_resize_bilinear_rgb(img + ((size_t)b[1] * w + b[0]) * 3, ch, cw, w * 3, r, oh, ow);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < oh * ow; ++i) {
            float v = (float)r[i * 3 + c] / 255.0f;
            v = v - mean3[c];
            out[(size_t)c * oh * ow + i] = v / std3[c];
        }
    if (u8_out) memcpy(u8_out, r, (size_t)oh * ow * 3);
    free(r);
}
