/*
 * oracle/src/cmc.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Camera-motion estimation of BoT-SORT, `GMC.applySparseOptFlow` (plugins/track/bot_sort/gmc.py:239-303; the same chain is
 * deep_oc_sort/cmc.py:136-166): cvtColor(BGR2GRAY) -> resize to (w // 2, h // 2) -> goodFeaturesToTrack(maxCorners 1000, quality
 * 0.01, minDistance 1, blockSize 3) -> calcOpticalFlowPyrLK(prev, cur, prev corners) with its defaults (21 x 21 window, 3 pyramid
 * levels above the image, 30 iterations / eps 0.01, minEigThreshold 1e-4) -> estimateAffinePartial2D(RANSAC: threshold 3, 2000
 * iterations, confidence 0.99) -> translation times the downscale.
 *
 * PARITY UNPINNED: every step here is third-party OpenCV (4.x; not installed, not vendored, no fixture in the reference). The
 * functions below restate the published algorithms (imgproc/color_rgb.simd.hpp RGB2Gray, resize.cpp, corner.cpp + featureselect.cpp,
 * pyramids.cpp, video/lkpyramid.cpp, calib3d/ptsetreg.cpp, core/rand.cpp) as closely as they can be restated from their
 * documentation; tests/golden/make_cmc_golden.py produces fixtures wherever OpenCV is installed. What IS pinned: the HIP kernels
 * (tracklab_amd/csrc/tlk_cmc.hip) against this file, and the recovered warp against the known transform of synthetic frame pairs.
 * r04: the refinement on the inliers is OpenCV's own -- cv::LMSolver (Nash's Levenberg-Marquardt: lambda on diag(J^T J), gain-ratio
 * thresholds 0.25 / 0.75, at most 10 iterations) from the RANSAC winner's 2-point model -- instead of the closed-form least-squares similarity
 * of r02-r03.  Measured (tests/test_oracle_cmc.py): the model being linear, the first step's gain ratio exceeds 0.75, lambda falls to 0 and
 * the second step lands on the least-squares solution, so the two agree to ~1e-12 -- the shortcut was numerically harmless, but the code
 * path is now the library's.  Still a restatement: the 4 x 4 solve is a Cholesky where OpenCV runs an SVD.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* cv::cvtColor(COLOR_BGR2GRAY) on an (h, w, 3) uint8 image: channel 0 gets the "B" weight whatever the image holds (the reference
 * hands its RGB frame to it). 15-bit fixed point: (c0 * 3735 + c1 * 19235 + c2 * 9798 + 16384) >> 15. */
void orc_cmc_gray(const uint8_t *img, int h, int w, uint8_t *gray)
{
    for (size_t i = 0; i < (size_t)h * w; ++i)
        gray[i] = (uint8_t)((img[i * 3] * 3735 + img[i * 3 + 1] * 19235 + img[i * 3 + 2] * 9798 + (1 << 14)) >> 15);
}

/* cv::resize(INTER_LINEAR) of a single-channel uint8 image (resize.cpp fixed point: 11-bit coefficients) */
static void lin_coef(int d, int ssize, int dsize, int is_col, int *s0, int *w0, int *w1)
{
    float f = (float)(((double)d + 0.5) * ((double)ssize / (double)dsize) - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (is_col) { if (s < 0) { f = 0.f; s = 0; } if (s >= ssize - 1) { f = 0.f; s = ssize - 1; } }
    *s0 = s; *w0 = (int)(short)(int)rintf((1.f - f) * 2048.f); *w1 = (int)(short)(int)rintf(f * 2048.f);
}
void orc_cmc_resize_gray(const uint8_t *src, int sh, int sw, uint8_t *dst, int dh, int dw)
{
    for (int dy = 0; dy < dh; ++dy) {
        int sy, b0, b1;
        lin_coef(dy, sh, dh, 0, &sy, &b0, &b1);
        const uint8_t *r0 = src + (size_t)(sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy)) * sw;
        const uint8_t *r1 = src + (size_t)(sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1)) * sw;
        for (int dx = 0; dx < dw; ++dx) {
            int sx, a0, a1;
            lin_coef(dx, sw, dw, 1, &sx, &a0, &a1);
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            const int S0 = r0[sx] * a0 + r0[sx1] * a1, S1 = r1[sx] * a0 + r1[sx1] * a1;
            const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dw + dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

/* cv::cornerMinEigenVal(blockSize 3, ksize 3, BORDER_REFLECT_101): Sobel derivatives scaled by 1 / (4 * 3 * 255), products summed
 * over the 3 x 3 block (not normalised), smaller eigenvalue of [[a, b], [b, c]] in float32 */
void orc_cmc_min_eigen(const uint8_t *img, int h, int w, float *eig)
{
    const float scale = (float)(1.0 / (4.0 * 3.0 * 255.0));
    float *cxx = malloc(sizeof(float) * (size_t)h * w), *cxy = malloc(sizeof(float) * (size_t)h * w), *cyy = malloc(sizeof(float) * (size_t)h * w);
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = img + (size_t)reflect101(y - 1, h) * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; ++x) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            const int gx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
            const int gy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
            const float dx = (float)gx * scale, dy = (float)gy * scale;
            cxx[(size_t)y * w + x] = dx * dx; cxy[(size_t)y * w + x] = dx * dy; cyy[(size_t)y * w + x] = dy * dy;
        }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float a = 0.f, b = 0.f, c = 0.f;
            for (int j = -1; j <= 1; ++j) {                      /* rows top to bottom, inside a row left to right */
                const size_t rb = (size_t)reflect101(y + j, h) * w;
                for (int i = -1; i <= 1; ++i) { const size_t k = rb + reflect101(x + i, w); a += cxx[k]; b += cxy[k]; c += cyy[k]; }
            }
            a *= 0.5f; c *= 0.5f;
            eig[(size_t)y * w + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        }
    free(cxx); free(cxy); free(cyy);
}

typedef struct { float v; int idx; } corner_t;
static int cmp_corner(const void *pa, const void *pb)
{
    const corner_t *a = pa, *b = pb;                             /* featureselect.cpp greaterThanPtr: value desc, ties: larger address first */
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (b->idx > a->idx) - (b->idx < a->idx);
}
/* cv::goodFeaturesToTrack(img, maxCorners, quality, minDistance = 1, mask, blockSize 3): pts (x, y) float32; returns the count.
 * With minDistance 1 distinct pixels are never closer than the limit: the strongest local maxima are taken as they come. */
int orc_cmc_good_features(const uint8_t *img, int h, int w, const uint8_t *mask, int max_corners, double quality, float *pts)
{
    float *eig = malloc(sizeof(float) * (size_t)h * w);
    orc_cmc_min_eigen(img, h, w, eig);
    float maxv = 0.f; int any = 0;
    for (size_t i = 0; i < (size_t)h * w; ++i) if (!mask || mask[i]) { if (!any || eig[i] > maxv) maxv = eig[i]; any = 1; }
    const float thr = (float)((double)maxv * quality);
    for (size_t i = 0; i < (size_t)h * w; ++i) if (!(eig[i] > thr)) eig[i] = 0.f;          /* THRESH_TOZERO */
    corner_t *c = malloc(sizeof(corner_t) * (size_t)h * w);
    int n = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float v = eig[(size_t)y * w + x];
            if (v == 0.f || (mask && !mask[(size_t)y * w + x])) continue;
            float m = v;                                          /* 3 x 3 dilation */
            for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i) { const float t = eig[(size_t)(y + j) * w + x + i]; if (t > m) m = t; }
            if (v == m) { c[n].v = v; c[n].idx = y * w + x; ++n; }
        }
    qsort(c, n, sizeof(corner_t), cmp_corner);
    if (n > max_corners) n = max_corners;
    for (int k = 0; k < n; ++k) { pts[2 * k] = (float)(c[k].idx % w); pts[2 * k + 1] = (float)(c[k].idx / w); }
    free(c); free(eig);
    return n;
}

/* cv::pyrDown (uint8): separable [1 4 6 4 1] / 16, BORDER_REFLECT_101, result (sum + 128) >> 8; dst is ((h+1)/2, (w+1)/2) */
void orc_cmc_pyr_down(const uint8_t *src, int h, int w, uint8_t *dst)
{
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    int *row = malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            const uint8_t *s = src + (size_t)reflect101(2 * y - 2 + k, h) * w;
            for (int x = 0; x < dw; ++x) {
                const int x0 = 2 * x;
                row[k * dw + x] = s[reflect101(x0 - 2, w)] + 4 * s[reflect101(x0 - 1, w)] + 6 * s[x0] + 4 * s[reflect101(x0 + 1, w)] + s[reflect101(x0 + 2, w)];
            }
        }
        for (int x = 0; x < dw; ++x)
            dst[(size_t)y * dw + x] = (uint8_t)((row[x] + 4 * row[dw + x] + 6 * row[2 * dw + x] + 4 * row[3 * dw + x] + row[4 * dw + x] + 128) >> 8);
    }
    free(row);
}

/* lkpyramid.cpp calcSharrDeriv: int16 (dx, dy) interleaved, BORDER_REFLECT_101 */
void orc_cmc_scharr(const uint8_t *src, int h, int w, int16_t *d)
{
    int *t0 = malloc(sizeof(int) * (size_t)(w + 2)), *t1 = malloc(sizeof(int) * (size_t)(w + 2));
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = src + (size_t)reflect101(y - 1, h) * w, *r1 = src + (size_t)y * w, *r2 = src + (size_t)reflect101(y + 1, h) * w;
        for (int x = -1; x <= w; ++x) { const int xx = reflect101(x, w); t0[x + 1] = (r0[xx] + r2[xx]) * 3 + r1[xx] * 10; t1[x + 1] = r2[xx] - r0[xx]; }
        for (int x = 0; x < w; ++x) {
            d[((size_t)y * w + x) * 2] = (int16_t)(t0[x + 2] - t0[x]);
            d[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
}

#define LK_WIN 21
#define LK_LEVELS 4                                               /* maxLevel 3 */
typedef struct { int h, w; uint8_t *img; int16_t *der; } lk_level;
struct orc_cmc_pyr { int nlev; lk_level L[LK_LEVELS]; };

orc_cmc_pyr *orc_cmc_pyr_build(const uint8_t *gray, int h, int w)
{
    orc_cmc_pyr *P = calloc(1, sizeof(*P));
    P->L[0].h = h; P->L[0].w = w;
    P->L[0].img = malloc((size_t)h * w); memcpy(P->L[0].img, gray, (size_t)h * w);
    P->nlev = 1;
    for (int l = 1; l < LK_LEVELS; ++l) {                         /* buildOpticalFlowPyramid stops before a level no larger than the window */
        const int ph = P->L[l - 1].h, pw = P->L[l - 1].w, nh = (ph + 1) / 2, nw = (pw + 1) / 2;
        if (nw <= LK_WIN || nh <= LK_WIN) break;
        P->L[l].h = nh; P->L[l].w = nw; P->L[l].img = malloc((size_t)nh * nw);
        orc_cmc_pyr_down(P->L[l - 1].img, ph, pw, P->L[l].img);
        P->nlev = l + 1;
    }
    for (int l = 0; l < P->nlev; ++l) { P->L[l].der = malloc(sizeof(int16_t) * 2 * (size_t)P->L[l].h * P->L[l].w); orc_cmc_scharr(P->L[l].img, P->L[l].h, P->L[l].w, P->L[l].der); }
    return P;
}
void orc_cmc_pyr_free(orc_cmc_pyr *P) { if (!P) return; for (int l = 0; l < P->nlev; ++l) { free(P->L[l].img); free(P->L[l].der); } free(P); }
int orc_cmc_pyr_levels(const orc_cmc_pyr *P) { return P->nlev; }
void orc_cmc_pyr_level(const orc_cmc_pyr *P, int l, int *h, int *w, const uint8_t **img, const int16_t **der)
{ *h = P->L[l].h; *w = P->L[l].w; if (img) *img = P->L[l].img; if (der) *der = P->L[l].der; }

/* pixel of the padded pyramid image: BORDER_REFLECT_101 for the image, constant 0 for the derivatives */
static inline int img_at(const lk_level *L, int y, int x) { return L->img[(size_t)reflect101(y, L->h) * L->w + reflect101(x, L->w)]; }
static inline int der_at(const lk_level *L, int y, int x, int c) { return (y < 0 || y >= L->h || x < 0 || x >= L->w) ? 0 : L->der[((size_t)y * L->w + x) * 2 + c]; }

/* cv::calcOpticalFlowPyrLK(prev, next, pts) with its defaults; next_pts (n, 2) float32, status (n) */
void orc_cmc_lk(const orc_cmc_pyr *A, const orc_cmc_pyr *Bp, const float *pts, int n, float *next_pts, uint8_t *status)
{
    const int nlev = A->nlev < Bp->nlev ? A->nlev : Bp->nlev;
    const float half = (LK_WIN - 1) * 0.5f, FLT_SCALE = 1.f / (1 << 20);
    for (int i = 0; i < n; ++i) {
        status[i] = 1;
        float nx = 0.f, ny = 0.f;
        for (int level = nlev - 1; level >= 0; --level) {
            const lk_level *I = &A->L[level], *J = &Bp->L[level];
            const float sc = 1.f / (float)(1 << level);
            float px = pts[2 * i] * sc, py = pts[2 * i + 1] * sc;
            if (level == nlev - 1) { nx = px; ny = py; } else { nx *= 2.f; ny *= 2.f; }
            next_pts[2 * i] = nx; next_pts[2 * i + 1] = ny;
            px -= half; py -= half;
            const int ipx = cv_floor(px), ipy = cv_floor(py);
            if (ipx < -LK_WIN || ipx >= I->w || ipy < -LK_WIN || ipy >= I->h) { if (level == 0) status[i] = 0; continue; }
            float a = px - ipx, b = py - ipy;
            int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << 14)), iw01 = cv_round(a * (1.f - b) * (1 << 14)), iw10 = cv_round((1.f - a) * b * (1 << 14));
            int iw11 = (1 << 14) - iw00 - iw01 - iw10;
            short Iw[LK_WIN * LK_WIN], dIx[LK_WIN * LK_WIN], dIy[LK_WIN * LK_WIN];
            double sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < LK_WIN; ++y)
                for (int x = 0; x < LK_WIN; ++x) {
                    const int yy = ipy + y, xx = ipx + x;
                    const int iv = DESCALE(img_at(I, yy, xx) * iw00 + img_at(I, yy, xx + 1) * iw01 + img_at(I, yy + 1, xx) * iw10 + img_at(I, yy + 1, xx + 1) * iw11, 14 - 5);
                    const int ix = DESCALE(der_at(I, yy, xx, 0) * iw00 + der_at(I, yy, xx + 1, 0) * iw01 + der_at(I, yy + 1, xx, 0) * iw10 + der_at(I, yy + 1, xx + 1, 0) * iw11, 14);
                    const int iy = DESCALE(der_at(I, yy, xx, 1) * iw00 + der_at(I, yy, xx + 1, 1) * iw01 + der_at(I, yy + 1, xx, 1) * iw10 + der_at(I, yy + 1, xx + 1, 1) * iw11, 14);
                    Iw[y * LK_WIN + x] = (short)iv; dIx[y * LK_WIN + x] = (short)ix; dIy[y * LK_WIN + x] = (short)iy;
                    sA11 += (double)(ix * ix); sA12 += (double)(ix * iy); sA22 += (double)(iy * iy);
                }
            const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LK_WIN * LK_WIN);
            if (minEig < 1e-4f || D < 1.1920929e-07f) { if (level == 0) status[i] = 0; continue; }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < 30; ++j) {
                const int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -LK_WIN || inx >= J->w || iny < -LK_WIN || iny >= J->h) { if (level == 0) status[i] = 0; break; }
                a = nx - inx; b = ny - iny;
                iw00 = cv_round((1.f - a) * (1.f - b) * (1 << 14)); iw01 = cv_round(a * (1.f - b) * (1 << 14)); iw10 = cv_round((1.f - a) * b * (1 << 14));
                iw11 = (1 << 14) - iw00 - iw01 - iw10;
                double sb1 = 0, sb2 = 0;
                for (int y = 0; y < LK_WIN; ++y)
                    for (int x = 0; x < LK_WIN; ++x) {
                        const int yy = iny + y, xx = inx + x;
                        const int diff = DESCALE(img_at(J, yy, xx) * iw00 + img_at(J, yy, xx + 1) * iw01 + img_at(J, yy + 1, xx) * iw10 + img_at(J, yy + 1, xx + 1) * iw11, 14 - 5)
                                         - Iw[y * LK_WIN + x];
                        sb1 += (double)(diff * dIx[y * LK_WIN + x]); sb2 += (double)(diff * dIy[y * LK_WIN + x]);
                    }
                const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                next_pts[2 * i] = nx + half; next_pts[2 * i + 1] = ny + half;
                if (dx * dx + dy * dy <= 0.01f * 0.01f) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) { next_pts[2 * i] -= dx * 0.5f; next_pts[2 * i + 1] -= dy * 0.5f; break; }
                pdx = dx; pdy = dy;
            }
            nx = next_pts[2 * i]; ny = next_pts[2 * i + 1];
            if (status[i] && level == 0) {                        /* the error pass drops points whose final window left the image */
                const int fx = cv_round(nx - half), fy = cv_round(ny - half);
                if (fx < -LK_WIN || fx >= J->w || fy < -LK_WIN || fy >= J->h) status[i] = 0;
            }
        }
    }
}

/* core/rand.cpp: multiply-with-carry generator; RANSAC seeds it with (uint64)-1 */
typedef struct { uint64_t s; } cv_rng;
static unsigned rng_next(cv_rng *r) { r->s = (uint64_t)(unsigned)r->s * 4164903690U + (unsigned)(r->s >> 32); return (unsigned)r->s; }
static int rng_uniform(cv_rng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

static int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p); ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1 - p; if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    double denom = 1 - pow(1 - ep, model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

/* the subsets the RANSAC loop would draw: ids (max_iters, 2); returns how many were drawn before the sampler gave up (normally max_iters) */
int orc_cmc_ransac_subsets(int count, int max_iters, int32_t *ids)
{
    cv_rng r = {0xffffffffffffffffULL};
    for (int it = 0; it < max_iters; ++it) {
        int idx[2], i = 0, iters = 0;
        for (; iters < 10000; ++iters) {
            for (i = 0; i < 2 && iters < 10000;) {
                const int v = idx[i] = rng_uniform(&r, 0, count);
                int j = 0;
                for (; j < i; ++j) if (v == idx[j]) break;
                if (j < i) continue;
                ++i;
            }
            break;                                                /* checkSubset: always true for two points */
        }
        if (!(i == 2 && iters < 10000)) return it;
        ids[2 * it] = idx[0]; ids[2 * it + 1] = idx[1];
    }
    return max_iters;
}

static void partial_from_two(const float *f, const float *t, int i0, int i1, double *M)
{
    const double x1 = f[2 * i0], y1 = f[2 * i0 + 1], x2 = f[2 * i1], y2 = f[2 * i1 + 1];
    const double X1 = t[2 * i0], Y1 = t[2 * i0 + 1], X2 = t[2 * i1], Y2 = t[2 * i1 + 1];
    const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
    const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[4] = S0; M[5] = S3;
}
static int count_inliers(const float *f, const float *t, int n, const double *M, float thr2, uint8_t *mask)
{
    const float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2], F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
    int good = 0;
    for (int i = 0; i < n; ++i) {
        const float a = F0 * f[2 * i] + F1 * f[2 * i + 1] + F2 - t[2 * i], b = F3 * f[2 * i] + F4 * f[2 * i + 1] + F5 - t[2 * i + 1];
        const int in = a * a + b * b <= thr2;
        if (mask) mask[i] = (uint8_t)in;
        good += in;
    }
    return good;
}

/* x = A^-1 b for the symmetric positive definite 4 x 4 system of the LM step (Cholesky).  OpenCV calls cv::solve(Ap, v, d, DECOMP_SVD) here
 * (DECOMP_EIG before 4.5.x): the same solution to ~1e-15 relative; which of the two it is cannot be pinned without cv2. */
static int solve4_spd(const double *A, const double *b, double *x)
{
    double L[16] = {0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 4 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 4 + k] * L[j * 4 + k];
            if (i == j) { if (!(s > 0)) return 0; L[i * 4 + i] = sqrt(s); }
            else L[i * 4 + j] = s / L[j * 4 + j];
        }
    double y[4];
    for (int i = 0; i < 4; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * 4 + k] * y[k]; y[i] = s / L[i * 4 + i]; }
    for (int i = 3; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 4; ++k) s -= L[k * 4 + i] * x[k]; x[i] = s / L[i * 4 + i]; }
    return 1;
}

/* AffinePartial2DRefineCallback::compute (calib3d/src/ptsetreg.cpp): residuals r (2 per inlier) of h = (a, b, tx, ty) and, with want_v,
 * v = J^T r for J rows (x, -y, 1, 0), (y, x, 0, 1).  Returns S = |r|^2.  Sums run over the inliers in index order. */
static double partial_residuals(const float *f, const float *t, const uint8_t *in, int n, const double *h, double *v)
{
    double S = 0;
    if (v) v[0] = v[1] = v[2] = v[3] = 0;
    for (int i = 0; i < n; ++i) if (in[i]) {
        const double Mx = f[2 * i], My = f[2 * i + 1];
        const double rx = h[0] * Mx - h[1] * My + h[2] - t[2 * i], ry = h[1] * Mx + h[0] * My + h[3] - t[2 * i + 1];
        S += rx * rx; S += ry * ry;
        if (v) { v[0] += Mx * rx; v[1] += -My * rx; v[2] += rx; v[0] += My * ry; v[1] += Mx * ry; v[3] += ry; }
    }
    return S;
}

/* cv::LMSolver::run (calib3d/src/levmarq.cpp, Nash's variant: lambda scales diag(A) of the FIRST linearisation, gain-ratio thresholds 0.25 /
 * 0.75, lambda halved or multiplied by nu in [2, 10], eps = FLT_EPSILON on |d|_inf and |r|_inf) on the four parameters of the partial affine
 * model, as cv::estimateAffinePartial2D calls it with refineIters = 10.  The model is linear in its parameters, so J -- and A = J^T J -- are
 * the same in every iteration; OpenCV recomputes them, with the same values.  M (2 x 3) in / out. */
static void lm_refine_partial(const float *f, const float *t, const uint8_t *in, int n, double *M, int max_iters)
{
    double x[4] = {M[0], M[3], M[2], M[5]}, xd[4], v[4], vd[4], d[4], A[16] = {0}, D[4];
    double sxx = 0, sx = 0, sy = 0, cnt = 0;
    for (int i = 0; i < n; ++i) if (in[i]) {
        const double Mx = f[2 * i], My = f[2 * i + 1];
        sxx += Mx * Mx; sxx += My * My; sx += Mx; sy += My; cnt += 1.0;
    }
    A[0] = sxx; A[5] = sxx; A[2] = A[8] = sx; A[3] = A[12] = sy; A[6] = A[9] = -sy; A[7] = A[13] = sx; A[10] = cnt; A[15] = cnt;
    double S = partial_residuals(f, t, in, n, x, v);
    for (int i = 0; i < 4; ++i) D[i] = A[i * 5];
    const double Rlo = 0.25, Rhi = 0.75, eps = 1.1920928955078125e-07;
    double lambda = 1, lc = 0.75;
    double rinf = 0;
    for (int iter = 0;;) {
        double Ap[16];
        memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < 4; ++i) Ap[i * 5] += lambda * D[i];
        if (!solve4_spd(Ap, v, d)) break;
        for (int i = 0; i < 4; ++i) xd[i] = x[i] - d[i];
        const double Sd = partial_residuals(f, t, in, n, xd, vd);
        double dS = 0, tdot = 0;
        for (int i = 0; i < 4; ++i) { double td = 2 * v[i]; for (int k = 0; k < 4; ++k) td -= A[i * 4 + k] * d[k]; dS += d[i] * td; tdot += d[i] * v[i]; }
        const double R = (S - Sd) / (fabs(dS) > 2.220446049250313e-16 ? dS : 1);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (R < Rlo) {
            double nu = (Sd - S) / (fabs(tdot) > 2.220446049250313e-16 ? tdot : 1) + 2;
            nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
            if (lambda == 0) {
                double maxval = 2.220446049250313e-16;
                for (int c = 0; c < 4; ++c) {                 /* diag of A^-1 (invert(A, Ap, DECOMP_EIG)) */
                    double e[4] = {0, 0, 0, 0}, col[4];
                    e[c] = 1;
                    if (solve4_spd(A, e, col) && fabs(col[c]) > maxval) maxval = fabs(col[c]);
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) { S = Sd; memcpy(x, xd, sizeof(x)); memcpy(v, vd, sizeof(v)); }
        ++iter;
        /* |r|_inf of the CURRENT x (OpenCV keeps r of the last accepted point) */
        rinf = 0;
        for (int i = 0; i < n; ++i) if (in[i]) {
            const double Mx = f[2 * i], My = f[2 * i + 1];
            const double rx = fabs(x[0] * Mx - x[1] * My + x[2] - t[2 * i]), ry = fabs(x[1] * Mx + x[0] * My + x[3] - t[2 * i + 1]);
            if (rx > rinf) rinf = rx;
            if (ry > rinf) rinf = ry;
        }
        double dinf = 0;
        for (int i = 0; i < 4; ++i) if (fabs(d[i]) > dinf) dinf = fabs(d[i]);
        if (!(iter < max_iters && dinf >= eps && rinf >= eps)) break;
    }
    M[0] = x[0]; M[4] = x[0]; M[1] = -x[1]; M[3] = x[1]; M[2] = x[2]; M[5] = x[3];
}

/* cv::estimateAffinePartial2D(from, to, RANSAC, 3.0, 2000, 0.99, refineIters): M (2, 3) float64, inliers (n). Returns 0 when no
 * model was found (OpenCV returns an empty matrix then). */
int orc_cmc_estimate_affine_partial(const float *from, const float *to, int n, double *M, uint8_t *inliers)
{
    if (n < 2) return 0;
    const int max_iters = 2000;
    const float thr2 = 3.0f * 3.0f;
    int32_t *ids = malloc(sizeof(int32_t) * 2 * max_iters);
    const int drawn = n > 2 ? orc_cmc_ransac_subsets(n, max_iters, ids) : 1;
    if (n == 2) { ids[0] = 0; ids[1] = 1; }
    uint8_t *mask = malloc((size_t)n), *best = malloc((size_t)n);
    int niters = max_iters, max_good = 0;
    double bestM[6] = {1, 0, 0, 0, 1, 0};
    for (int it = 0; it < niters; ++it) {
        if (it >= drawn) break;
        double Mi[6];
        partial_from_two(from, to, ids[2 * it], ids[2 * it + 1], Mi);
        const int good = count_inliers(from, to, n, Mi, thr2, mask);
        if (good > (max_good > 1 ? max_good : 1)) {
            memcpy(best, mask, (size_t)n); memcpy(bestM, Mi, sizeof(Mi)); max_good = good;
            niters = ransac_update_iters(0.99, (double)(n - good) / n, 2, niters);
        }
    }
    int ok = max_good > 0;
    if (ok) {
        /* refinement on the inliers, as OpenCV does it (r04; r02-r03 used the closed-form least-squares similarity instead): 10 iterations of
         * cv::LMSolver from the RANSAC winner's 2-point model */
        if (n > 2) lm_refine_partial(from, to, best, n, bestM, 10);            /* ptsetreg.cpp: `if (result && count > 2 && refineIters)`, count = all points */
        memcpy(M, bestM, sizeof(bestM));
        if (inliers) memcpy(inliers, best, (size_t)n);
    }
    free(ids); free(mask); free(best);
    return ok;
}

/* ------------------------------------------------------------------------------------------------ GMC.applySparseOptFlow */
struct orc_cmc { int h, w, downscale, have_prev, nprev; orc_cmc_pyr *prev; float *prev_pts; };

orc_cmc *orc_cmc_create(int h, int w, int downscale)
{
    orc_cmc *C = calloc(1, sizeof(*C));
    C->h = h; C->w = w; C->downscale = downscale < 1 ? 1 : downscale;
    C->prev_pts = malloc(sizeof(float) * 2 * 1000);
    return C;
}
void orc_cmc_destroy(orc_cmc *C) { if (!C) return; orc_cmc_pyr_free(C->prev); free(C->prev_pts); free(C); }

/* frame (h, w, 3) uint8 as the tracker receives it; H (2, 3) float64 (identity on the first frame and whenever fewer than five
 * points survive -- where the reference would print a warning or, without a RANSAC model, fail on `None`). Returns the number of
 * RANSAC inliers (0 when H is the identity by default). */
int orc_cmc_apply(orc_cmc *C, const uint8_t *frame, double *H)
{
    H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0;
    const int h = C->h, w = C->w, dh = C->downscale > 1 ? h / C->downscale : h, dw = C->downscale > 1 ? w / C->downscale : w;
    uint8_t *gray = malloc((size_t)h * w), *small = gray;
    orc_cmc_gray(frame, h, w, gray);
    if (C->downscale > 1) { small = malloc((size_t)dh * dw); orc_cmc_resize_gray(gray, h, w, small, dh, dw); }
    float *pts = malloc(sizeof(float) * 2 * 1000);
    const int npts = orc_cmc_good_features(small, dh, dw, NULL, 1000, 0.01, pts);
    orc_cmc_pyr *cur = orc_cmc_pyr_build(small, dh, dw);
    int inl = 0;
    if (C->have_prev) {
        float *np_ = malloc(sizeof(float) * 2 * 1000), *pf = malloc(sizeof(float) * 2 * 1000), *pt = malloc(sizeof(float) * 2 * 1000);
        uint8_t *st = malloc(1000);
        orc_cmc_lk(C->prev, cur, C->prev_pts, C->nprev, np_, st);
        int m = 0;
        for (int i = 0; i < C->nprev; ++i) if (st[i]) { pf[2 * m] = C->prev_pts[2 * i]; pf[2 * m + 1] = C->prev_pts[2 * i + 1]; pt[2 * m] = np_[2 * i]; pt[2 * m + 1] = np_[2 * i + 1]; ++m; }
        if (m > 4) {
            double M[6]; uint8_t *in = malloc((size_t)m);
            if (orc_cmc_estimate_affine_partial(pf, pt, m, M, in)) {
                memcpy(H, M, sizeof(M));
                if (C->downscale > 1) { H[2] *= C->downscale; H[5] *= C->downscale; }
                for (int i = 0; i < m; ++i) inl += in[i];
            }
            free(in);
        }
        free(np_); free(pf); free(pt); free(st);
    }
    orc_cmc_pyr_free(C->prev);
    C->prev = cur; C->have_prev = 1; C->nprev = npts;
    memcpy(C->prev_pts, pts, sizeof(float) * 2 * (size_t)npts);
    if (small != gray) free(small);
    free(gray); free(pts);
    return inl;
}
