/*
 * oracle/src/ecc.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Camera-motion estimation of StrongSORT: `ECC(src, dst)` of plugins/track/strong_sort/sort/track.py:129-211 (the same function is
 * plugins/track/bpbreid_strong_sort/ecc.py:4-99): cvtColor(BGR2GRAY) of both frames -> cv2.resize(fx = fy = 0.1, INTER_LINEAR) ->
 * cv2.findTransformECC(template = previous, input = current, eye(2,3) float32, MOTION_EUCLIDEAN, (COUNT | EPS, 100, 1e-5), no mask,
 * gaussFiltSize 1) -> translation divided by the scale. A cv2.error (NaN correlation, non-positive lambda denominator) is caught by
 * the reference and means "no camera update for this frame".
 *
 * PARITY UNPINNED: findTransformECC is third-party OpenCV (video/src/ecc.cpp; not installed, not vendored, no fixture in the
 * reference). This file restates the published algorithm (Evangelidis & Psarakis, PAMI 2008, as implemented there): per iteration
 *   warpAffine(INTER_LINEAR | WARP_INVERSE_MAP) of the float image and of its central-difference gradients, warpAffine(INTER_NEAREST)
 *   of the all-ones mask; masked mean / std of warped image and template; zero-mean both under the mask; Euclidean Jacobian
 *   [gx * hatX + gy * hatY, gx, gy]; Hessian = J^T J (float), inverse by the 3 x 3 adjugate in double; rho = <tz, iw> / (|iw| |tz|);
 *   lambda = (|iw|^2 - ip^T H^-1 ip) / (<tz, iw> - tp^T H^-1 ip); dp = H^-1 J^T (lambda tz - iw); theta += dp0, t += (dp1, dp2);
 *   stop when |rho - rho_prev| < eps or after 100 iterations
 * with OpenCV's warpAffine coordinate arithmetic (10-bit fixed-point coordinates, 5-bit interpolation fractions, float tap weights,
 * constant-zero border). gaussFiltSize 1 makes every GaussianBlur the identity and the mask threshold step leaves the all-ones mask.
 * What IS pinned: the HIP kernel (tracklab_amd/csrc/tlk_cmc.hip, ecc_kernel) against this file, and the recovered warp against the
 * known motion of synthetic frame pairs.
 *
 * Sums over the image are taken in double in a FIXED order that a 1024-thread workgroup reproduces: partial[t] = the pixels
 * t, t + 1024, ... in increasing order; then per group of 64 partials a balanced pairwise tree (neighbours first); then the 16
 * group sums left to right. (OpenCV's own order -- SIMD blocks -- is neither documented nor stable across builds.)
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ECC_THREADS 1024
#define AB_BITS 10
#define AB_SCALE (1 << AB_BITS)
#define INTER_BITS 5
#define INTER_TAB (1 << INTER_BITS)

static inline int sat_int(double v) { return (int)lrint(v); }                 /* cv::saturate_cast<int>(double) = cvRound */
static inline int reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }

typedef struct { double v[13]; } sums_t;

/* the fixed-order reduction described in the header: vals[k][p] for pixel p, k < nk */
static void reduce_fixed(const double *vals, int nk, int npx, double *out)
{
    double *part = malloc(sizeof(double) * ECC_THREADS);
    for (int k = 0; k < nk; ++k) {
        for (int t = 0; t < ECC_THREADS; ++t) {
            double s = 0.0;
            for (int p = t; p < npx; p += ECC_THREADS) s += vals[(size_t)k * npx + p];
            part[t] = s;
        }
        for (int step = 1; step < 64; step <<= 1)
            for (int t = 0; t < ECC_THREADS; t += 2 * step) part[t] = part[t] + part[t + step];
        double s = 0.0;
        for (int g = 0; g < ECC_THREADS / 64; ++g) s += part[g * 64];
        out[k] = s;
    }
    free(part);
}

/* one bilinear tap set of cv::warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, BORDER_CONSTANT 0) at destination (x, y); src(yy, xx) via `at` */
typedef float (*pix_fn)(const float *img, int h, int w, int y, int x);
static float pix_plain(const float *img, int h, int w, int y, int x) { (void)h; return img[(size_t)y * w + x]; }
/* filter2D(image, [-0.5, 0, 0.5]) / its transpose with BORDER_REFLECT_101, evaluated where it is sampled */
static float pix_gx(const float *img, int h, int w, int y, int x) { (void)h; return 0.5f * img[(size_t)y * w + reflect101(x + 1, w)] - 0.5f * img[(size_t)y * w + reflect101(x - 1, w)]; }
static float pix_gy(const float *img, int h, int w, int y, int x) { return 0.5f * img[(size_t)reflect101(y + 1, h) * w + x] - 0.5f * img[(size_t)reflect101(y - 1, h) * w + x]; }

static float warp_linear(const float *img, int h, int w, pix_fn at, int X, int Y)
{
    const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;                      /* (saturation to short: images far below 32768 px) */
    const float fx = (float)(X & (INTER_TAB - 1)) * (1.f / INTER_TAB), fy = (float)(Y & (INTER_TAB - 1)) * (1.f / INTER_TAB);
    const float wx0 = 1.f - fx, wx1 = fx, wy0 = 1.f - fy, wy1 = fy;
    const float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
    if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) return 0.f;
    const int x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
    const float v0 = (x0 && y0) ? at(img, h, w, sy, sx) : 0.f, v1 = (x1 && y0) ? at(img, h, w, sy, sx + 1) : 0.f;
    const float v2 = (x0 && y1) ? at(img, h, w, sy + 1, sx) : 0.f, v3 = (x1 && y1) ? at(img, h, w, sy + 1, sx + 1) : 0.f;
    float r = v0 * w0;
    r = r + v1 * w1; r = r + v2 * w2; r = r + v3 * w3;
    return r;
}

/* cv2.findTransformECC on two (h, w) uint8 images, MOTION_EUCLIDEAN from the identity. map: (2, 3) float32 out.
 * Returns the number of iterations run (>= 1), or -1 where OpenCV throws (the caller then skips the camera update). rho_out optional. */
int orc_ecc_find_transform(const uint8_t *templ, const uint8_t *image, int h, int w, int max_iter, double eps, float *map, double *rho_out)
{
    const int npx = h * w;
    float *tf = malloc(sizeof(float) * npx), *imf = malloc(sizeof(float) * npx);
    float *iw = malloc(sizeof(float) * npx), *gxw = malloc(sizeof(float) * npx), *gyw = malloc(sizeof(float) * npx), *tz = malloc(sizeof(float) * npx);
    float *j0 = malloc(sizeof(float) * npx);
    uint8_t *mask = malloc((size_t)npx);
    double *vals = malloc(sizeof(double) * 13 * (size_t)npx);
    for (int p = 0; p < npx; ++p) { tf[p] = (float)templ[p]; imf[p] = (float)image[p]; }
    map[0] = 1.f; map[1] = 0.f; map[2] = 0.f; map[3] = 0.f; map[4] = 1.f; map[5] = 0.f;
    double rho = -1.0, last_rho = -eps;
    int it = 0, status = 0;
    for (int i = 1; i <= max_iter && fabs(rho - last_rho) >= eps; ++i) {
        it = i;
        double M[6];
        for (int k = 0; k < 6; ++k) M[k] = (double)map[k];
        /* pass A: warps + masked first and second moments */
        for (int y = 0; y < h; ++y) {
            const int X0l = sat_int((M[1] * y + M[2]) * AB_SCALE) + AB_SCALE / INTER_TAB / 2, Y0l = sat_int((M[4] * y + M[5]) * AB_SCALE) + AB_SCALE / INTER_TAB / 2;
            const int X0n = sat_int((M[1] * y + M[2]) * AB_SCALE) + AB_SCALE / 2, Y0n = sat_int((M[4] * y + M[5]) * AB_SCALE) + AB_SCALE / 2;
            for (int x = 0; x < w; ++x) {
                const int p = y * w + x;
                const int ad = sat_int(M[0] * x * AB_SCALE), bd = sat_int(M[3] * x * AB_SCALE);
                const int X = (X0l + ad) >> (AB_BITS - INTER_BITS), Y = (Y0l + bd) >> (AB_BITS - INTER_BITS);
                iw[p] = warp_linear(imf, h, w, pix_plain, X, Y);
                gxw[p] = warp_linear(imf, h, w, pix_gx, X, Y);
                gyw[p] = warp_linear(imf, h, w, pix_gy, X, Y);
                const int nx = (X0n + ad) >> AB_BITS, ny = (Y0n + bd) >> AB_BITS;
                mask[p] = (nx >= 0 && nx < w && ny >= 0 && ny < h) ? 1 : 0;
                const double m = mask[p] ? 1.0 : 0.0;
                vals[0 * (size_t)npx + p] = m;
                vals[1 * (size_t)npx + p] = m * (double)iw[p];
                vals[2 * (size_t)npx + p] = m * ((double)iw[p] * (double)iw[p]);
                vals[3 * (size_t)npx + p] = m * (double)tf[p];
                vals[4 * (size_t)npx + p] = m * ((double)tf[p] * (double)tf[p]);
            }
        }
        double a[13];
        reduce_fixed(vals, 5, npx, a);
        const double nz = a[0], scale = nz != 0.0 ? 1.0 / nz : 0.0;
        const double img_mean = a[1] * scale, tmp_mean = a[3] * scale;
        double img_var = a[2] * scale - img_mean * img_mean, tmp_var = a[4] * scale - tmp_mean * tmp_mean;
        if (img_var < 0.0) img_var = 0.0;
        if (tmp_var < 0.0) tmp_var = 0.0;
        const double img_std = sqrt(img_var), tmp_std = sqrt(tmp_var);
        const double tmp_norm = sqrt(nz * tmp_std * tmp_std), img_norm = sqrt(nz * img_std * img_std);
        /* pass B: zero-mean, Jacobian, Hessian and projections */
        const float h0 = map[0], h1 = map[3], im_f = (float)img_mean, tm_f = (float)tmp_mean;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int p = y * w + x;
                if (mask[p]) { iw[p] = iw[p] - im_f; tz[p] = tf[p] - tm_f; } else tz[p] = 0.f;
                const float X = (float)x, Y = (float)y;
                const float hx_a = X * (-h1), hx_b = Y * (-h0), hy_a = X * h0, hy_b = Y * (-h1);
                const float hatX = hx_a + hx_b, hatY = hy_a + hy_b;
                const float ja = gxw[p] * hatX, jb = gyw[p] * hatY;
                j0[p] = ja + jb;
                const double J0 = j0[p], J1 = gxw[p], J2 = gyw[p], I = iw[p], T = tz[p];
                double *v = vals + p;
                v[0 * (size_t)npx] = J0 * J0; v[1 * (size_t)npx] = J0 * J1; v[2 * (size_t)npx] = J0 * J2;
                v[3 * (size_t)npx] = J1 * J1; v[4 * (size_t)npx] = J1 * J2; v[5 * (size_t)npx] = J2 * J2;
                v[6 * (size_t)npx] = J0 * I; v[7 * (size_t)npx] = J1 * I; v[8 * (size_t)npx] = J2 * I;
                v[9 * (size_t)npx] = J0 * T; v[10 * (size_t)npx] = J1 * T; v[11 * (size_t)npx] = J2 * T;
                v[12 * (size_t)npx] = T * I;
            }
        reduce_fixed(vals, 13, npx, a);
        /* Hessian (float, symmetric), inverse by the adjugate in double (cv::invert, 3 x 3 CV_32F), results back to float */
        const float H[3][3] = {{(float)a[0], (float)a[1], (float)a[2]}, {(float)a[1], (float)a[3], (float)a[4]}, {(float)a[2], (float)a[4], (float)a[5]}};
        float Hi[3][3] = {{0}};
        double d = (double)H[0][0] * ((double)H[1][1] * H[2][2] - (double)H[1][2] * H[2][1]) - (double)H[0][1] * ((double)H[1][0] * H[2][2] - (double)H[1][2] * H[2][0])
                 + (double)H[0][2] * ((double)H[1][0] * H[2][1] - (double)H[1][1] * H[2][0]);
        if (d != 0.0) {
            d = 1.0 / d;
            Hi[0][0] = (float)(((double)H[1][1] * H[2][2] - (double)H[1][2] * H[2][1]) * d);
            Hi[0][1] = (float)(((double)H[0][2] * H[2][1] - (double)H[0][1] * H[2][2]) * d);
            Hi[0][2] = (float)(((double)H[0][1] * H[1][2] - (double)H[0][2] * H[1][1]) * d);
            Hi[1][0] = (float)(((double)H[1][2] * H[2][0] - (double)H[1][0] * H[2][2]) * d);
            Hi[1][1] = (float)(((double)H[0][0] * H[2][2] - (double)H[0][2] * H[2][0]) * d);
            Hi[1][2] = (float)(((double)H[0][2] * H[1][0] - (double)H[0][0] * H[1][2]) * d);
            Hi[2][0] = (float)(((double)H[1][0] * H[2][1] - (double)H[1][1] * H[2][0]) * d);
            Hi[2][1] = (float)(((double)H[0][1] * H[2][0] - (double)H[0][0] * H[2][1]) * d);
            Hi[2][2] = (float)(((double)H[0][0] * H[1][1] - (double)H[0][1] * H[1][0]) * d);
        }
        const double correlation = a[12];
        last_rho = rho;
        rho = correlation / (img_norm * tmp_norm);
        if (isnan(rho)) { status = -1; break; }
        const float ip[3] = {(float)a[6], (float)a[7], (float)a[8]}, tp[3] = {(float)a[9], (float)a[10], (float)a[11]};
        float iph[3];
        for (int r = 0; r < 3; ++r) iph[r] = (float)((double)Hi[r][0] * ip[0] + (double)Hi[r][1] * ip[1] + (double)Hi[r][2] * ip[2]);
        const double lambda_n = img_norm * img_norm - ((double)ip[0] * iph[0] + (double)ip[1] * iph[1] + (double)ip[2] * iph[2]);
        const double lambda_d = correlation - ((double)tp[0] * iph[0] + (double)tp[1] * iph[1] + (double)tp[2] * iph[2]);
        if (lambda_d <= 0.0) { rho = -1; status = -1; break; }
        const double lambda = lambda_n / lambda_d;
        /* pass C: error image and its projection */
        const float lam_f = (float)lambda;
        for (int p = 0; p < npx; ++p) {
            const float lt = lam_f * tz[p];
            const double e = (double)(lt - iw[p]);
            vals[0 * (size_t)npx + p] = (double)j0[p] * e; vals[1 * (size_t)npx + p] = (double)gxw[p] * e; vals[2 * (size_t)npx + p] = (double)gyw[p] * e;
        }
        reduce_fixed(vals, 3, npx, a);
        const float ep[3] = {(float)a[0], (float)a[1], (float)a[2]};
        float dp[3];
        for (int r = 0; r < 3; ++r) dp[r] = (float)((double)Hi[r][0] * ep[0] + (double)Hi[r][1] * ep[1] + (double)Hi[r][2] * ep[2]);
        /* update_warping_matrix_ECC, MOTION_EUCLIDEAN */
        double new_theta = (double)dp[0];
        new_theta += asin((double)map[3]);
        map[2] += dp[1]; map[5] += dp[2];
        map[0] = map[4] = (float)cos(new_theta);
        map[3] = (float)sin(new_theta);
        map[1] = -map[3];
    }
    if (rho_out) *rho_out = rho;
    free(tf); free(imf); free(iw); free(gxw); free(gyw); free(tz); free(j0); free(mask); free(vals);
    return status < 0 ? -1 : it;
}

/* ECC(src, dst) of sort/track.py:129-211 on two (h, w, 3) uint8 frames: warp (2, 3) float32 with the translation rescaled to frame
 * pixels. Returns the iteration count, or -1 when the reference returns (None, None). scale = 0.1: dsize = round(size * 0.1). */
int orc_ecc_frames(const uint8_t *prev_hwc3, const uint8_t *cur_hwc3, int h, int w, float *warp6, double *rho_out)
{
    const int dh = (int)lrint(h * 0.1), dw = (int)lrint(w * 0.1);
    uint8_t *g = malloc((size_t)h * w), *a = malloc((size_t)dh * dw), *b = malloc((size_t)dh * dw);
    orc_cmc_gray(prev_hwc3, h, w, g); orc_cmc_resize_gray(g, h, w, a, dh, dw);
    orc_cmc_gray(cur_hwc3, h, w, g); orc_cmc_resize_gray(g, h, w, b, dh, dw);
    const int it = orc_ecc_find_transform(a, b, dh, dw, 100, 1e-5, warp6, rho_out);
    if (it >= 0) { warp6[2] = warp6[2] / (float)0.1; warp6[5] = warp6[5] / (float)0.1; }
    free(g); free(a); free(b);
    return it;
}
