/*
 * oracle/src/pose.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Pose-estimator pre/post-processing behind tracklab/wrappers/pose_estimator/rtmlib_api.py:27-33 (RTMPose.process calls
 * rtmlib.RTMPose(image, bboxes)). The arithmetic is THIRD-PARTY and not in the reference tree: rtmlib==0.0.13 (uv.lock:3019)
 *   rtmlib/tools/pose_estimation/rtmpose.py        RTMPose.preprocess / postprocess
 *   rtmlib/tools/pose_estimation/pre_processings.py  bbox_xyxy2cs, _fix_aspect_ratio, get_warp_matrix, top_down_affine
 *   rtmlib/tools/pose_estimation/post_processings.py get_simcc_maximum
 * and OpenCV (cv2.getAffineTransform, cv2.warpAffine INTER_LINEAR / BORDER_CONSTANT: modules/imgproc/src/imgwarp.cpp,
 * fixed point: 10-bit coordinates, 5-bit sub-pixel positions, 15-bit bilinear weights). Neither is installed here and no
 * reference test pins them: PARITY UNPINNED -- restated from the published algorithms; the HIP kernels are bit-exact to THIS.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* bbox_xyxy2cs(padding) + _fix_aspect_ratio(w / h): center (2), scale (2) */
void orc_rtmpose_center_scale(const double *xyxy, double padding, int in_w, int in_h, double *center, double *scale)
{
    center[0] = (xyxy[0] + xyxy[2]) * 0.5; center[1] = (xyxy[1] + xyxy[3]) * 0.5;
    const double w = (xyxy[2] - xyxy[0]) * padding, h = (xyxy[3] - xyxy[1]) * padding;
    const double ar = (double)in_w / (double)in_h;
    if (w > h * ar) { scale[0] = w; scale[1] = w / ar; }
    else { scale[0] = h * ar; scale[1] = h; }
}

/* cv2.getAffineTransform: 6x6 system in double, Gaussian elimination with partial pivoting (cv::solve DECOMP_LU) */
static void get_affine(const float src[6], const float dst[6], double M[6])
{
    double a[36], b[6];
    for (int i = 0; i < 3; ++i) {
        const int j = i * 12, k = i * 12 + 6;
        a[j] = a[k + 3] = src[i * 2]; a[j + 1] = a[k + 4] = src[i * 2 + 1]; a[j + 2] = a[k + 5] = 1;
        a[j + 3] = a[j + 4] = a[j + 5] = 0; a[k] = a[k + 1] = a[k + 2] = 0;
        b[i * 2] = dst[i * 2]; b[i * 2 + 1] = dst[i * 2 + 1];
    }
    const int n = 6;
    for (int i = 0; i < n; ++i) {                            /* cv::hal::LU64f */
        int k = i;
        for (int j = i + 1; j < n; ++j) if (fabs(a[j * n + i]) > fabs(a[k * n + i])) k = j;
        if (fabs(a[k * n + i]) < 2.220446049250313e-16 * 100) { memset(M, 0, 6 * sizeof(double)); return; }
        if (k != i) {
            for (int j = i; j < n; ++j) { double t = a[i * n + j]; a[i * n + j] = a[k * n + j]; a[k * n + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        const double d = -1 / a[i * n + i];
        for (int j = i + 1; j < n; ++j) {
            const double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; ++c) a[j * n + c] += alpha * a[i * n + c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * b[k];
        b[i] = s / a[i * n + i];
    }
    memcpy(M, b, 6 * sizeof(double));
}

/* get_warp_matrix(center, scale, rot=0, output_size=(w, h)) -> forward 2x3 matrix (source -> crop) */
void orc_rtmpose_warp_matrix(const double *center, const double *scale, int out_w, int out_h, double *M)
{
    float src[6], dst[6];
    const double src_w = scale[0];
    /* _rotate_point([0, -src_w/2], 0) = [0*cos - y*sin, 0*sin + y*cos] with sin 0 = 0, cos 0 = 1 */
    const double sdx = 0.0 * 1.0 - (src_w * -0.5) * 0.0, sdy = 0.0 * 0.0 + (src_w * -0.5) * 1.0;
    src[0] = (float)center[0]; src[1] = (float)center[1];
    src[2] = (float)(center[0] + sdx); src[3] = (float)(center[1] + sdy);
    /* _get_3rd_point(a, b): direction = a - b; c = b + [-direction[1], direction[0]]  (float32 rows) */
    { const float d0 = src[0] - src[2], d1 = src[1] - src[3]; src[4] = src[2] + (-d1); src[5] = src[3] + d0; }
    dst[0] = (float)(out_w * 0.5); dst[1] = (float)(out_h * 0.5);
    dst[2] = (float)(out_w * 0.5 + 0.0); dst[3] = (float)(out_h * 0.5 + out_w * -0.5);
    { const float d0 = dst[0] - dst[2], d1 = dst[1] - dst[3]; dst[4] = dst[2] + (-d1); dst[5] = dst[3] + d0; }
    get_affine(src, dst, M);
}

static inline int cv_round(double v) { return (int)lrint(v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* BilinearTab_i[ay*32 + ax][4] of imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt): products of (1 - i/32, i/32) scaled by
 * 32768 and stored as short; the (0,0) entry saturates to 32767 and the unit sum is restored on the largest of the entries the
 * fix-up loop scans (which, for the 2x2 kernel, starts at the last weight) */
void orc_cv_bilinear_tab(int ay, int ax, int w[4])
{
    const float vy[2] = {1.f - ay * (1.f / 32), ay * (1.f / 32)}, vx[2] = {1.f - ax * (1.f / 32), ax * (1.f / 32)};
    int isum = 0;
    for (int k1 = 0; k1 < 2; ++k1) for (int k2 = 0; k2 < 2; ++k2) { w[k1 * 2 + k2] = sat_short(cv_round((double)(vy[k1] * vx[k2] * 32768.f))); isum += w[k1 * 2 + k2]; }
    if (isum != 32768) w[3] = (short)(w[3] - (isum - 32768));
}

/* cv2.warpAffine(src (sh, sw, 3) u8, M forward 2x3, (dw, dh), INTER_LINEAR, BORDER_CONSTANT 0) -> dst (dh, dw, 3) */
void orc_cv_warp_affine_linear_u8(const uint8_t *src, int sh, int sw, const double *Mf, uint8_t *dst, int dh, int dw)
{
    double M[6];
    memcpy(M, Mf, sizeof(M));
    {   /* invert: dst -> src */
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, round_delta = AB_SCALE / 32 / 2;
    for (int y = 0; y < dh; ++y) {
        const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta, Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dw; ++x) {
            const int X = (X0 + cv_round(M[0] * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
            const int Y = (Y0 + cv_round(M[3] * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
            const int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
            int w[4];
            orc_cv_bilinear_tab(Y & 31, X & 31, w);
            uint8_t *o = dst + ((size_t)y * dw + x) * 3;
            if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) { o[0] = o[1] = o[2] = 0; continue; }
            for (int c = 0; c < 3; ++c) {
                int v[4];
                for (int k = 0; k < 4; ++k) {
                    const int xx = sx + (k & 1), yy = sy + (k >> 1);
                    v[k] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[((size_t)yy * sw + xx) * 3 + c] : 0;
                }
                int r = (v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3] + (1 << 14)) >> 15;
                o[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
        }
    }
}

/* RTMPose.preprocess + the transpose/cast of inference(): out (3, in_h, in_w) float32 = (float)((u8 - mean) / std), mean/std doubles */
void orc_rtmpose_preprocess(const uint8_t *img, int h, int w, const double *xyxy, int in_w, int in_h, const double *mean3, const double *std3,
                            float *out, double *center, double *scale)
{
    double M[6];
    orc_rtmpose_center_scale(xyxy, 1.25, in_w, in_h, center, scale);
    orc_rtmpose_warp_matrix(center, scale, in_w, in_h, M);
    uint8_t *crop = malloc((size_t)in_w * in_h * 3);
    orc_cv_warp_affine_linear_u8(img, h, w, M, crop, in_h, in_w);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < in_w * in_h; ++i) out[(size_t)c * in_w * in_h + i] = (float)(((double)crop[i * 3 + c] - mean3[c]) / std3[c]);
    free(crop);
}

/* get_simcc_maximum + RTMPose.postprocess for one box: simcc_x (K, Wx), simcc_y (K, Wy) f32 -> keypoints (K,2) f64, scores (K) f32 */
void orc_simcc_decode(const float *sx, const float *sy, int K, int Wx, int Wy, double split_ratio, const double *center, const double *scale,
                      int in_w, int in_h, double *kps, float *scores)
{
    for (int k = 0; k < K; ++k) {
        int ax = 0, ay = 0;
        for (int i = 1; i < Wx; ++i) if (sx[(size_t)k * Wx + i] > sx[(size_t)k * Wx + ax]) ax = i;       /* np.argmax: first maximum */
        for (int i = 1; i < Wy; ++i) if (sy[(size_t)k * Wy + i] > sy[(size_t)k * Wy + ay]) ay = i;
        const float mx = sx[(size_t)k * Wx + ax], my = sy[(size_t)k * Wy + ay];
        const float val = mx > my ? my : mx;
        float lx = (float)ax, ly = (float)ay;
        if (val <= 0.f) { lx = -1.f; ly = -1.f; }
        const float kx = lx / (float)split_ratio, ky = ly / (float)split_ratio;          /* float32 / python float */
        kps[k * 2] = (double)kx / (double)in_w * scale[0] + center[0] - scale[0] / 2;
        kps[k * 2 + 1] = (double)ky / (double)in_h * scale[1] + center[1] - scale[1] / 2;
        scores[k] = val;
    }
}

/* keypoints_conf = np.mean(scores, axis=1) on float32 (rtmlib_api.py:31): numpy's pairwise sum -- 8 running accumulators over the
 * multiples of 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail -- divided by K in float32 */
float orc_mean_f32_numpy(const float *a, int K)
{
    float res;
    if (K < 8) { res = 0.f; for (int k = 0; k < K; ++k) res += a[k]; }
    else {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int k = 8;
        for (; k < K - (K % 8); k += 8) for (int j = 0; j < 8; ++j) r[j] += a[k + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; k < K; ++k) res += a[k];
    }
    return res / (float)K;
}
