// tlk_cmc.hip -- camera-motion estimation on the device: BoT-SORT's GMC.applySparseOptFlow
// (plugins/track/bot_sort/gmc.py:239-303; the same chain is deep_oc_sort/cmc.py:136-166):
//   cvtColor(BGR2GRAY) -> resize to (w // 2, h // 2) -> goodFeaturesToTrack(1000, 0.01, minDistance 1, blockSize 3) ->
//   calcOpticalFlowPyrLK(prev, cur, prev corners) -> estimateAffinePartial2D(RANSAC) -> translation x downscale.
// Every step is third-party OpenCV in the reference: PARITY UNPINNED (no cv2 here, no fixture in the reference). The kernels follow
// oracle/src/cmc.c operation for operation (integer pyramids / derivatives / window sums are exact, so the tracked points are
// bit-identical to the oracle's; the least-squares refinement sums in another order: warp equal to ~1e-12). The frame never leaves
// the device and nothing synchronises with the host: the (2,3) warp lands in device memory for tlk_botsort_update_dev_gmc.
//
// Stages (all on the caller's stream):
//   gray_resize_kernel      frame (h,w,3) u8 -> downscaled grey image (grey values recomputed per tap: 4 taps per output pixel)
//   cov_kernel, eig_kernel  Sobel products, 3x3 box sums, smaller eigenvalue (float32, the oracle's summation order)
//   max_kernel, key_kernel  global maximum -> threshold; every interior local maximum becomes a 64-bit key (value, pixel index)
//   rocprim radix sort      keys descending: the first min(count, 1000) are the corners, strongest first, ties by larger address
//   pyr_down / scharr       3 pyramid levels above the image + int16 Scharr derivatives per level
//   lk_kernel               one wavefront per corner: 21 x 21 window = 7 pixels per lane, integer window sums reduced across the wave
//   compact_kernel          tracked pairs in order; ransac_subsets_kernel: OpenCV's RNG + subset sampler (one lane, data independent)
//   ransac_eval_kernel      one wavefront per hypothesis: similarity from two pairs, inlier count over all pairs
//   ransac_pick_kernel      the sequential RANSAC loop replayed over the counts (adaptive iteration bound), inlier mask of the winner,
//                           least-squares similarity on the inliers, translation x downscale -> warp
#include "tlk_common.hpp"
#include "tlk_cv.hpp"

#include <rocprim/rocprim.hpp>

using namespace tlk;
using namespace tlk::cv;

namespace {

constexpr int LK_WIN = 21, LK_LEVELS = 4, MAX_CORNERS_CAP = 1024, RANSAC_ITERS = 2000;

__global__ void __launch_bounds__(BLOCK) cov_kernel(const unsigned char *__restrict__ img, int h, int w, float *__restrict__ cov)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const unsigned char *r0 = img + (size_t)reflect101(y - 1, h) * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)reflect101(y + 1, h) * w;
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    const int gx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
    const int gy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
    const float scale = (float)(1.0 / (4.0 * 3.0 * 255.0));
    const float dx = (float)gx * scale, dy = (float)gy * scale;
    cov[(size_t)i * 3] = dx * dx; cov[(size_t)i * 3 + 1] = dx * dy; cov[(size_t)i * 3 + 2] = dy * dy;
}

__global__ void __launch_bounds__(BLOCK) eig_kernel(const float *__restrict__ cov, int h, int w, float *__restrict__ eig)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int j = -1; j <= 1; ++j) {
        const size_t rb = (size_t)reflect101(y + j, h) * w;
        for (int k = -1; k <= 1; ++k) { const size_t q = (rb + reflect101(x + k, w)) * 3; a += cov[q]; b += cov[q + 1]; c += cov[q + 2]; }
    }
    a *= 0.5f; c *= 0.5f;
    eig[i] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
}

__device__ __forceinline__ unsigned int f32_key(float v) { const unsigned int u = __float_as_uint(v); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float f32_unkey(unsigned int k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

__global__ void __launch_bounds__(BLOCK) max_kernel(const float *__restrict__ eig, int n, unsigned int *__restrict__ maxkey)
{
    __shared__ unsigned int s_m[NWAVES];
    unsigned int m = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) { const unsigned int k = f32_key(eig[i]); m = k > m ? k : m; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_xor(m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { for (int k = 1; k < NWAVES; ++k) m = s_m[k] > m ? s_m[k] : m; atomicMax(maxkey, m); }
}

// interior local maxima above quality * max -> key = (order-preserving value bits << 32) | pixel index; everything else key 0
__global__ void __launch_bounds__(BLOCK) key_kernel(const float *__restrict__ eig, int h, int w, const unsigned int *__restrict__ maxkey, double quality,
                                                    unsigned long long *__restrict__ keys, int *__restrict__ n_cand)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    unsigned long long key = 0;
    if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
        const float thr = (float)((double)f32_unkey(*maxkey) * quality);
        const float v = eig[i];
        if (v > thr) {
            float m = v;
            for (int j = -1; j <= 1; ++j) for (int k = -1; k <= 1; ++k) { float t = eig[(size_t)(y + j) * w + x + k]; t = t > thr ? t : 0.f; m = t > m ? t : m; }
            if (v == m) key = ((unsigned long long)f32_key(v) << 32) | (unsigned int)i;
        }
    }
    keys[i] = key;
    const unsigned long long bal = __ballot(key != 0);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_cand, __popcll(bal));
}

__global__ void __launch_bounds__(BLOCK) corners_kernel(const unsigned long long *__restrict__ sorted, const int *__restrict__ n_cand, int w, int max_corners,
                                                        float *__restrict__ pts, int *__restrict__ n_pts)
{
    const int n = min(*n_cand, max_corners);
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_pts = n;
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) {
        const int idx = (int)(sorted[k] & 0xffffffffu);
        pts[2 * k] = (float)(idx % w); pts[2 * k + 1] = (float)(idx / w);
    }
}

__global__ void __launch_bounds__(BLOCK) pyr_down_kernel(const unsigned char *__restrict__ src, int h, int w, unsigned char *__restrict__ dst, int dh, int dw)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= dh * dw) return;
    const int y = i / dw, x = i - y * dw, x0 = 2 * x;
    const int xs[5] = {reflect101(x0 - 2, w), reflect101(x0 - 1, w), x0, reflect101(x0 + 1, w), reflect101(x0 + 2, w)};
    const int wt[5] = {1, 4, 6, 4, 1};
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const unsigned char *s = src + (size_t)reflect101(2 * y - 2 + k, h) * w;
        acc += wt[k] * (s[xs[0]] + 4 * s[xs[1]] + 6 * s[xs[2]] + 4 * s[xs[3]] + s[xs[4]]);
    }
    dst[i] = (unsigned char)((acc + 128) >> 8);
}

__global__ void __launch_bounds__(BLOCK) scharr_kernel(const unsigned char *__restrict__ src, int h, int w, short *__restrict__ d)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const unsigned char *r0 = src + (size_t)reflect101(y - 1, h) * w, *r1 = src + (size_t)y * w, *r2 = src + (size_t)reflect101(y + 1, h) * w;
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
    const int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
    d[(size_t)i * 2] = (short)(t0p - t0m);
    d[(size_t)i * 2 + 1] = (short)((t1p + t1m) * 3 + t1c * 10);
}

struct LkLevel { const unsigned char *img; const short *der; int h, w; };
struct LkPyr { LkLevel L[LK_LEVELS]; int nlev; };

// The window reaches at most LK_WIN pixels outside a level: with every level at least LK_WIN + 1 pixels wide and high one reflection
// is enough and the taps are branch-free (SMALL = false); images smaller than that keep the general loop.
__device__ __forceinline__ int reflect101_once(int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * n - 2 - p : p; }
template <bool SMALL>
__device__ __forceinline__ int lk_img(const LkLevel &L, int y, int x)
{
    if constexpr (SMALL) return L.img[(size_t)reflect101(y, L.h) * L.w + reflect101(x, L.w)];
    else return L.img[(size_t)reflect101_once(y, L.h) * L.w + reflect101_once(x, L.w)];
}
__device__ __forceinline__ int lk_der(const LkLevel &L, int y, int x, int c) { return (y < 0 || y >= L.h || x < 0 || x >= L.w) ? 0 : L.der[((size_t)y * L.w + x) * 2 + c]; }
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// cv::calcOpticalFlowPyrLK with its defaults, one wavefront per point; lane l owns window pixels l, l + 64, ... (7 of 441)
template <bool SMALL>
__global__ void __launch_bounds__(BLOCK) lk_kernel(LkPyr A, LkPyr Bp, const float *__restrict__ pts, const int *__restrict__ n_pts,
                                                   float *__restrict__ next_pts, unsigned char *__restrict__ status)
{
    const int i = blockIdx.x * NWAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= *n_pts) return;
    constexpr int PER = (LK_WIN * LK_WIN + WAVE - 1) / WAVE;      // 7
    const int nlev = A.nlev < Bp.nlev ? A.nlev : Bp.nlev;
    const float half = (LK_WIN - 1) * 0.5f, FLT_SCALE = 1.f / (1 << 20);
    int st = 1;
    float nx = 0.f, ny = 0.f, ox = 0.f, oy = 0.f;
    for (int level = nlev - 1; level >= 0; --level) {
        const LkLevel &I = A.L[level], &J = Bp.L[level];
        const float sc = 1.f / (float)(1 << level);
        float px = pts[2 * i] * sc, py = pts[2 * i + 1] * sc;
        if (level == nlev - 1) { nx = px; ny = py; } else { nx *= 2.f; ny *= 2.f; }
        ox = nx; oy = ny;
        px -= half; py -= half;
        const int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
        if (ipx < -LK_WIN || ipx >= I.w || ipy < -LK_WIN || ipy >= I.h) { if (level == 0) st = 0; continue; }
        float a = px - ipx, b = py - ipy;
        int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << 14)), iw01 = cv_round_f(a * (1.f - b) * (1 << 14)), iw10 = cv_round_f((1.f - a) * b * (1 << 14));
        int iw11 = (1 << 14) - iw00 - iw01 - iw10;
        short Iw[PER], dIx[PER], dIy[PER];
        long long s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int k = q * WAVE + lane;
            Iw[q] = 0; dIx[q] = 0; dIy[q] = 0;
            if (k < LK_WIN * LK_WIN) {
                const int y = k / LK_WIN, x = k - y * LK_WIN, yy = ipy + y, xx = ipx + x;
                const int iv = DESCALE(lk_img<SMALL>(I, yy, xx) * iw00 + lk_img<SMALL>(I, yy, xx + 1) * iw01 + lk_img<SMALL>(I, yy + 1, xx) * iw10 + lk_img<SMALL>(I, yy + 1, xx + 1) * iw11, 14 - 5);
                const int ix = DESCALE(lk_der(I, yy, xx, 0) * iw00 + lk_der(I, yy, xx + 1, 0) * iw01 + lk_der(I, yy + 1, xx, 0) * iw10 + lk_der(I, yy + 1, xx + 1, 0) * iw11, 14);
                const int iy = DESCALE(lk_der(I, yy, xx, 1) * iw00 + lk_der(I, yy, xx + 1, 1) * iw01 + lk_der(I, yy + 1, xx, 1) * iw10 + lk_der(I, yy + 1, xx + 1, 1) * iw11, 14);
                Iw[q] = (short)iv; dIx[q] = (short)ix; dIy[q] = (short)iy;
                s11 += (long long)(ix * ix); s12 += (long long)(ix * iy); s22 += (long long)(iy * iy);
            }
        }
        s11 = wave_sum_i64(s11); s12 = wave_sum_i64(s12); s22 = wave_sum_i64(s22);      // integer sums: exact, order-free
        const float A11 = (float)(double)s11 * FLT_SCALE, A12 = (float)(double)s12 * FLT_SCALE, A22 = (float)(double)s22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LK_WIN * LK_WIN);
        if (minEig < 1e-4f || D < 1.1920929e-07f) { if (level == 0) st = 0; continue; }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < 30; ++j) {
            const int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
            if (inx < -LK_WIN || inx >= J.w || iny < -LK_WIN || iny >= J.h) { if (level == 0) st = 0; break; }
            a = nx - inx; b = ny - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << 14)); iw01 = cv_round_f(a * (1.f - b) * (1 << 14)); iw10 = cv_round_f((1.f - a) * b * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            long long sb1 = 0, sb2 = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int k = q * WAVE + lane;
                if (k < LK_WIN * LK_WIN) {
                    const int y = k / LK_WIN, x = k - y * LK_WIN, yy = iny + y, xx = inx + x;
                    const int diff = DESCALE(lk_img<SMALL>(J, yy, xx) * iw00 + lk_img<SMALL>(J, yy, xx + 1) * iw01 + lk_img<SMALL>(J, yy + 1, xx) * iw10 + lk_img<SMALL>(J, yy + 1, xx + 1) * iw11, 14 - 5) - Iw[q];
                    sb1 += (long long)(diff * dIx[q]); sb2 += (long long)(diff * dIy[q]);
                }
            }
            sb1 = wave_sum_i64(sb1); sb2 = wave_sum_i64(sb2);
            const float b1 = (float)(double)sb1 * FLT_SCALE, b2 = (float)(double)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            ox = nx + half; oy = ny + half;
            if (dx * dx + dy * dy <= 0.01f * 0.01f) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) { ox -= dx * 0.5f; oy -= dy * 0.5f; break; }
            pdx = dx; pdy = dy;
        }
        nx = ox; ny = oy;
        if (st && level == 0) {
            const int fx = cv_round_f(nx - half), fy = cv_round_f(ny - half);
            if (fx < -LK_WIN || fx >= J.w || fy < -LK_WIN || fy >= J.h) st = 0;
        }
    }
    if (lane == 0) { next_pts[2 * i] = ox; next_pts[2 * i + 1] = oy; status[i] = (unsigned char)st; }
}

// tracked pairs in their original order (one workgroup)
__global__ void __launch_bounds__(BLOCK) compact_kernel(const float *__restrict__ pts, const float *__restrict__ next_pts, const unsigned char *__restrict__ status,
                                                        const int *__restrict__ n_pts, float *__restrict__ from, float *__restrict__ to, int *__restrict__ m_out)
{
    __shared__ int s_scan[NWAVES];
    const int n = *n_pts;
    const int m = block_compact(n, [&](int i) { return status[i] != 0; },
                                [&](int i, int pos) { from[2 * pos] = pts[2 * i]; from[2 * pos + 1] = pts[2 * i + 1]; to[2 * pos] = next_pts[2 * i]; to[2 * pos + 1] = next_pts[2 * i + 1]; }, s_scan);
    if (threadIdx.x == 0) *m_out = m;
}

// the index pairs OpenCV's RANSAC would draw for `count` points (core/rand.cpp multiply-with-carry generator seeded with (uint64)-1,
// ptsetreg.cpp getSubset): sequential by nature, one lane
__global__ void ransac_subsets_kernel(const int *__restrict__ count_p, int *__restrict__ ids, int *__restrict__ n_drawn)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int count = *count_p;
    if (count <= 2) { ids[0] = 0; ids[1] = 1; *n_drawn = count == 2 ? 1 : 0; return; }
    unsigned long long s = 0xffffffffffffffffULL;
    int drawn = RANSAC_ITERS;
    for (int it = 0; it < RANSAC_ITERS; ++it) {
        int idx[2], i = 0, iters = 0;
        for (; iters < 10000; ++iters) {
            for (i = 0; i < 2 && iters < 10000;) {
                s = (unsigned long long)(unsigned int)s * 4164903690U + (unsigned int)(s >> 32);
                const int v = idx[i] = (int)((unsigned int)s % (unsigned int)count);
                int j = 0;
                for (; j < i; ++j) if (v == idx[j]) break;
                if (j < i) continue;
                ++i;
            }
            break;
        }
        if (!(i == 2 && iters < 10000)) { drawn = it; break; }
        ids[2 * it] = idx[0]; ids[2 * it + 1] = idx[1];
    }
    *n_drawn = drawn;
}

__device__ __forceinline__ void partial_from_two(const float *f, const float *t, int i0, int i1, double *M)
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
__device__ __forceinline__ bool is_inlier(const float *f, const float *t, int i, const float (&F)[6])
{
    const float a = F[0] * f[2 * i] + F[1] * f[2 * i + 1] + F[2] - t[2 * i], b = F[3] * f[2 * i] + F[4] * f[2 * i + 1] + F[5] - t[2 * i + 1];
    return a * a + b * b <= 9.0f;
}

// one wavefront per hypothesis
__global__ void __launch_bounds__(BLOCK) ransac_eval_kernel(const float *__restrict__ from, const float *__restrict__ to, const int *__restrict__ m_p,
                                                            const int *__restrict__ ids, const int *__restrict__ n_drawn, int *__restrict__ good)
{
    const int it = blockIdx.x * NWAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (it >= RANSAC_ITERS) return;
    const int m = *m_p;
    if (it >= *n_drawn || m < 2) { if (lane == 0) good[it] = 0; return; }
    double M[6];
    partial_from_two(from, to, ids[2 * it], ids[2 * it + 1], M);
    const float F[6] = {(float)M[0], (float)M[1], (float)M[2], (float)M[3], (float)M[4], (float)M[5]};
    int g = 0;
    for (int i = lane; i < m; i += WAVE) g += is_inlier(from, to, i, F) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g += __shfl_xor(g, off);
    if (lane == 0) good[it] = g;
}

__device__ __forceinline__ int ransac_update_iters(double p, double ep, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p); ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1 - p; if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    double denom = 1 - (1 - ep) * (1 - ep);                       // pow(1 - ep, modelPoints = 2)
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// RANSAC's sequential loop over the precomputed counts, then the refinement on the winner's inliers (one workgroup)
__global__ void __launch_bounds__(BLOCK) ransac_pick_kernel(const float *__restrict__ from, const float *__restrict__ to, const int *__restrict__ m_p,
                                                            const int *__restrict__ ids, const int *__restrict__ n_drawn, const int *__restrict__ good,
                                                            int have_prev, double downscale, double *__restrict__ warp, int *__restrict__ n_inliers)
{
    __shared__ int s_best, s_ok;
    __shared__ double s_red[NWAVES][8];
    const int m = *m_p, tid = threadIdx.x;
    if (tid == 0) {
        int best = -1, max_good = 0, niters = RANSAC_ITERS;
        if (have_prev && m > 4) {
            const int drawn = *n_drawn;
            for (int it = 0; it < niters && it < drawn; ++it) {
                const int g = good[it];
                if (g > (max_good > 1 ? max_good : 1)) { best = it; max_good = g; niters = ransac_update_iters(0.99, (double)(m - g) / m, niters); }
            }
        }
        s_best = best; s_ok = best >= 0;
    }
    __syncthreads();
    if (!s_ok) { if (tid == 0) { warp[0] = 1; warp[1] = 0; warp[2] = 0; warp[3] = 0; warp[4] = 1; warp[5] = 0; *n_inliers = 0; } return; }
    double M[6];
    partial_from_two(from, to, ids[2 * s_best], ids[2 * s_best + 1], M);
    const float F[6] = {(float)M[0], (float)M[1], (float)M[2], (float)M[3], (float)M[4], (float)M[5]};
    // Refinement on the inliers = OpenCV's: <= 10 iterations of cv::LMSolver (Nash's Levenberg-Marquardt) on h = (a, b, tx, ty) from the 2-point
    // model (oracle/src/cmc.c::lm_refine_partial is the sequential statement).  The model is linear in h, so J^T J is built once; every
    // iteration is ONE pass over the points (residuals of the trial point, their J^T r and |r|^2, |r|_inf) reduced across the workgroup in a
    // fixed order, then a 4 x 4 Cholesky solve that every thread repeats for itself.  Sums are block-parallel (the oracle's are sequential):
    // warp equal to ~1e-12, as with the closed form this replaces.
    auto block_sum8 = [&](double (&x)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x[k] += __shfl_xor(x[k], off);
        }
        __syncthreads();
        if ((tid & 63) == 0) for (int k = 0; k < 8; ++k) s_red[tid >> 6][k] = x[k];
        __syncthreads();
        for (int k = 0; k < 8; ++k) { double t = 0; for (int q = 0; q < NWAVES; ++q) t += s_red[q][k]; x[k] = t; }
    };
    auto block_max = [&](double x) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(x, off); x = o > x ? o : x; }
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6][0] = x;
        __syncthreads();
        double t = 0;
        for (int q = 0; q < NWAVES; ++q) t = s_red[q][0] > t ? s_red[q][0] : t;
        return t;
    };
    // this thread's inliers: points tid, tid + BLOCK, ... (m <= 1000: at most 4 per thread, kept in registers)
    double px[4], py[4], qx_[4], qy_[4];
    int nmine = 0;
    double v0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < m && nmine < 4; i += BLOCK)
        if (is_inlier(from, to, i, F)) {
            px[nmine] = from[2 * i]; py[nmine] = from[2 * i + 1]; qx_[nmine] = to[2 * i]; qy_[nmine] = to[2 * i + 1];
            v0[0] += px[nmine] * px[nmine]; v0[0] += py[nmine] * py[nmine]; v0[1] += px[nmine]; v0[2] += py[nmine]; v0[3] += 1.0;
            ++nmine;
        }
    block_sum8(v0);
    const double sxx = v0[0], sx = v0[1], sy = v0[2], cnt = v0[3];
    double A[16] = {sxx, 0, sx, sy,  0, sxx, -sy, sx,  sx, -sy, cnt, 0,  sy, sx, 0, cnt};
    auto solve4 = [&](const double *Am, const double *bv, double *xo) -> bool {
        double L[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) L[i] = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j <= i; ++j) {
                double sacc = Am[i * 4 + j];
                for (int k = 0; k < j; ++k) sacc -= L[i * 4 + k] * L[j * 4 + k];
                if (i == j) { if (!(sacc > 0)) return false; L[i * 4 + i] = sqrt(sacc); }
                else L[i * 4 + j] = sacc / L[j * 4 + j];
            }
        double y[4];
        for (int i = 0; i < 4; ++i) { double sacc = bv[i]; for (int k = 0; k < i; ++k) sacc -= L[i * 4 + k] * y[k]; y[i] = sacc / L[i * 4 + i]; }
        for (int i = 3; i >= 0; --i) { double sacc = y[i]; for (int k = i + 1; k < 4; ++k) sacc -= L[k * 4 + i] * xo[k]; xo[i] = sacc / L[i * 4 + i]; }
        return true;
    };
    // one pass: S = |r|^2, v = J^T r, |r|_inf of parameters h
    auto residual_pass = [&](const double *h, double &S, double *v, double &rinf) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double mx = 0;
        for (int q = 0; q < nmine; ++q) {
            const double rx = h[0] * px[q] - h[1] * py[q] + h[2] - qx_[q], ry = h[1] * px[q] + h[0] * py[q] + h[3] - qy_[q];
            acc[4] += rx * rx; acc[4] += ry * ry;
            acc[0] += px[q] * rx; acc[1] += -py[q] * rx; acc[2] += rx; acc[0] += py[q] * ry; acc[1] += px[q] * ry; acc[3] += ry;
            mx = fabs(rx) > mx ? fabs(rx) : mx; mx = fabs(ry) > mx ? fabs(ry) : mx;
        }
        block_sum8(acc);
        rinf = block_max(mx);
        S = acc[4]; v[0] = acc[0]; v[1] = acc[1]; v[2] = acc[2]; v[3] = acc[3];
    };
    if (m > 2) {
        double x[4] = {M[0], M[3], M[2], M[5]}, xd[4], v[4], vd[4], d[4], S, Sd, rinf, rinf_d;
        residual_pass(x, S, v, rinf);
        const double Dg[4] = {A[0], A[5], A[10], A[15]};
        const double Rlo = 0.25, Rhi = 0.75, eps = 1.1920928955078125e-07, DEPS = 2.220446049250313e-16;
        double lambda = 1, lc = 0.75;
        for (int iter = 0;;) {
            double Ap[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) Ap[i] = A[i];
            for (int i = 0; i < 4; ++i) Ap[i * 5] += lambda * Dg[i];
            if (!solve4(Ap, v, d)) break;
            for (int i = 0; i < 4; ++i) xd[i] = x[i] - d[i];
            residual_pass(xd, Sd, vd, rinf_d);
            double dS = 0, tdot = 0;
            for (int i = 0; i < 4; ++i) { double td = 2 * v[i]; for (int k = 0; k < 4; ++k) td -= A[i * 4 + k] * d[k]; dS += d[i] * td; tdot += d[i] * v[i]; }
            const double R = (S - Sd) / (fabs(dS) > DEPS ? dS : 1);
            if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
            else if (R < Rlo) {
                double nu = (Sd - S) / (fabs(tdot) > DEPS ? tdot : 1) + 2;
                nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
                if (lambda == 0) {
                    double maxval = DEPS;
                    for (int c = 0; c < 4; ++c) {
                        double e[4] = {0, 0, 0, 0}, col[4];
                        e[c] = 1;
                        if (solve4(A, e, col) && fabs(col[c]) > maxval) maxval = fabs(col[c]);
                    }
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            if (Sd < S) { S = Sd; rinf = rinf_d; for (int i = 0; i < 4; ++i) { x[i] = xd[i]; v[i] = vd[i]; } }
            ++iter;
            double dinf = 0;
            for (int i = 0; i < 4; ++i) dinf = fabs(d[i]) > dinf ? fabs(d[i]) : dinf;
            if (!(iter < 10 && dinf >= eps && rinf >= eps)) break;
        }
        M[0] = x[0]; M[4] = x[0]; M[1] = -x[1]; M[3] = x[1]; M[2] = x[2]; M[5] = x[3];
    }
    if (tid == 0) {
        if (downscale > 1.0) { M[2] *= downscale; M[5] *= downscale; }
        for (int k = 0; k < 6; ++k) warp[k] = M[k];
        *n_inliers = (int)cnt;
    }
}

}  // namespace

struct tlk_cmc {
    int device, h, w, dh, dw, downscale, max_corners, have_prev, cur;
    unsigned char *gray[2][LK_LEVELS]; short *der[2][LK_LEVELS]; int lh[LK_LEVELS], lw[LK_LEVELS], nlev;
    float *cov, *eig; unsigned long long *keys, *keys_sorted; void *sort_tmp; size_t sort_tmp_bytes;
    unsigned int *maxkey; int *n_cand, *n_pts[2], *m, *n_drawn, *good, *ids, *n_inl;
    float *pts[2], *next_pts, *from, *to; unsigned char *status;
    double *warp; unsigned char *frame_stage;
};

static void cmc_free(tlk_cmc *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    for (int b = 0; b < 2; ++b) for (int l = 0; l < LK_LEVELS; ++l) { if (c->gray[b][l]) hipFree(c->gray[b][l]); if (c->der[b][l]) hipFree(c->der[b][l]); }
    void *ptrs[] = {c->cov, c->eig, c->keys, c->keys_sorted, c->sort_tmp, c->maxkey, c->n_cand, c->n_pts[0], c->n_pts[1], c->m, c->n_drawn, c->good, c->ids, c->n_inl,
                    c->pts[0], c->pts[1], c->next_pts, c->from, c->to, c->status, c->warp, c->frame_stage};
    for (void *p : ptrs) if (p) hipFree(p);
    delete c;
}

extern "C" int tlk_cmc_create(int h, int w, int downscale, int max_corners, int device, tlk_cmc **out)
{
    if (!out || h < 32 || w < 32 || downscale < 1 || max_corners < 8 || max_corners > MAX_CORNERS_CAP) return fail(TLK_EINVAL, "tlk_cmc_create: bad argument (max_corners in [8, 1024])");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_cmc_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_cmc_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_cmc *c = new tlk_cmc();
    memset(c, 0, sizeof(*c));
    c->device = device; c->h = h; c->w = w; c->downscale = downscale; c->max_corners = max_corners;
    c->dh = downscale > 1 ? h / downscale : h; c->dw = downscale > 1 ? w / downscale : w;
    c->lh[0] = c->dh; c->lw[0] = c->dw; c->nlev = 1;
    for (int l = 1; l < LK_LEVELS; ++l) {
        const int nh = (c->lh[l - 1] + 1) / 2, nw = (c->lw[l - 1] + 1) / 2;
        if (nw <= LK_WIN || nh <= LK_WIN) break;
        c->lh[l] = nh; c->lw[l] = nw; c->nlev = l + 1;
    }
    const size_t npx = (size_t)c->dh * c->dw;
#define CMC_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { cmc_free(c); return fail(TLK_EHIP, std::string("tlk_cmc_create: hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    for (int b = 0; b < 2; ++b) for (int l = 0; l < c->nlev; ++l) { CMC_ALLOC(c->gray[b][l], (size_t)c->lh[l] * c->lw[l]); CMC_ALLOC(c->der[b][l], sizeof(short) * 2 * (size_t)c->lh[l] * c->lw[l]); }
    CMC_ALLOC(c->cov, sizeof(float) * 3 * npx); CMC_ALLOC(c->eig, sizeof(float) * npx);
    CMC_ALLOC(c->keys, sizeof(unsigned long long) * npx); CMC_ALLOC(c->keys_sorted, sizeof(unsigned long long) * npx);
    if (rocprim::radix_sort_keys_desc(nullptr, c->sort_tmp_bytes, c->keys, c->keys_sorted, npx, 0, 64, (hipStream_t)0) != hipSuccess) { cmc_free(c); return fail(TLK_EHIP, "tlk_cmc_create: sort workspace query failed"); }
    CMC_ALLOC(c->sort_tmp, c->sort_tmp_bytes ? c->sort_tmp_bytes : 16);
    CMC_ALLOC(c->maxkey, sizeof(unsigned int)); CMC_ALLOC(c->n_cand, sizeof(int)); CMC_ALLOC(c->n_pts[0], sizeof(int)); CMC_ALLOC(c->n_pts[1], sizeof(int));
    CMC_ALLOC(c->m, sizeof(int)); CMC_ALLOC(c->n_drawn, sizeof(int)); CMC_ALLOC(c->good, sizeof(int) * RANSAC_ITERS); CMC_ALLOC(c->ids, sizeof(int) * 2 * RANSAC_ITERS); CMC_ALLOC(c->n_inl, sizeof(int));
    for (int b = 0; b < 2; ++b) CMC_ALLOC(c->pts[b], sizeof(float) * 2 * MAX_CORNERS_CAP);
    CMC_ALLOC(c->next_pts, sizeof(float) * 2 * MAX_CORNERS_CAP); CMC_ALLOC(c->from, sizeof(float) * 2 * MAX_CORNERS_CAP); CMC_ALLOC(c->to, sizeof(float) * 2 * MAX_CORNERS_CAP);
    CMC_ALLOC(c->status, MAX_CORNERS_CAP); CMC_ALLOC(c->warp, sizeof(double) * 6); CMC_ALLOC(c->frame_stage, (size_t)h * w * 3);
#undef CMC_ALLOC
    TLK_HIP(hipMemset(c->n_pts[0], 0, sizeof(int))); TLK_HIP(hipMemset(c->n_pts[1], 0, sizeof(int)));
    TLK_HIP(hipDeviceSynchronize());
    *out = c;
    return TLK_OK;
}

extern "C" int tlk_cmc_destroy(tlk_cmc *c) { cmc_free(c); return TLK_OK; }
extern "C" int tlk_cmc_reset(tlk_cmc *c)
{
    if (!c) return fail(TLK_EINVAL, "tlk_cmc_reset: null handle");
    TLK_HIP(hipSetDevice(c->device));
    c->have_prev = 0; c->cur = 0;
    // (no corners of an earlier video: a gated first frame without detections rolls back onto THIS state, and the frame after it must find nothing to track)
    TLK_HIP(hipMemset(c->n_pts[0], 0, sizeof(int))); TLK_HIP(hipMemset(c->n_pts[1], 0, sizeof(int)));
    return TLK_OK;
}

extern "C" int tlk_cmc_apply_dev(tlk_cmc *c, const uint8_t *frame_dev, double *warp6_dev, void *hip_stream)
{
    if (!c || !frame_dev || !warp6_dev) return fail(TLK_EINVAL, "tlk_cmc_apply_dev: null pointer");
    TLK_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int b = c->cur, pb = 1 - b, npx = c->dh * c->dw;
    const unsigned g = (unsigned)((npx + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(gray_resize_kernel, dim3(g), dim3(BLOCK), 0, st, frame_dev, c->h, c->w, c->gray[b][0], c->dh, c->dw);
    hipLaunchKernelGGL(cov_kernel, dim3(g), dim3(BLOCK), 0, st, (const unsigned char *)c->gray[b][0], c->dh, c->dw, c->cov);
    hipLaunchKernelGGL(eig_kernel, dim3(g), dim3(BLOCK), 0, st, (const float *)c->cov, c->dh, c->dw, c->eig);
    TLK_HIP(hipMemsetAsync(c->maxkey, 0, sizeof(unsigned int), st));
    TLK_HIP(hipMemsetAsync(c->n_cand, 0, sizeof(int), st));
    hipLaunchKernelGGL(max_kernel, dim3(g < 1024 ? g : 1024), dim3(BLOCK), 0, st, (const float *)c->eig, npx, c->maxkey);
    hipLaunchKernelGGL(key_kernel, dim3(g), dim3(BLOCK), 0, st, (const float *)c->eig, c->dh, c->dw, (const unsigned int *)c->maxkey, 0.01, c->keys, c->n_cand);
    if (rocprim::radix_sort_keys_desc(c->sort_tmp, c->sort_tmp_bytes, c->keys, c->keys_sorted, (size_t)npx, 0, 64, st) != hipSuccess) return fail(TLK_EHIP, "tlk_cmc_apply_dev: sort failed");
    hipLaunchKernelGGL(corners_kernel, dim3(4), dim3(BLOCK), 0, st, (const unsigned long long *)c->keys_sorted, (const int *)c->n_cand, c->dw, c->max_corners, c->pts[b], c->n_pts[b]);
    for (int l = 1; l < c->nlev; ++l) {
        const int n = c->lh[l] * c->lw[l];
        hipLaunchKernelGGL(pyr_down_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, (const unsigned char *)c->gray[b][l - 1], c->lh[l - 1], c->lw[l - 1], c->gray[b][l], c->lh[l], c->lw[l]);
    }
    for (int l = 0; l < c->nlev; ++l) {
        const int n = c->lh[l] * c->lw[l];
        hipLaunchKernelGGL(scharr_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, (const unsigned char *)c->gray[b][l], c->lh[l], c->lw[l], c->der[b][l]);
    }
    if (c->have_prev) {
        LkPyr A, Bp;
        A.nlev = Bp.nlev = c->nlev;
        for (int l = 0; l < c->nlev; ++l) { A.L[l] = LkLevel{c->gray[pb][l], c->der[pb][l], c->lh[l], c->lw[l]}; Bp.L[l] = LkLevel{c->gray[b][l], c->der[b][l], c->lh[l], c->lw[l]}; }
        const bool small = c->dh <= LK_WIN || c->dw <= LK_WIN;      // (levels above 0 are only built when larger than the window)
        const dim3 lg((c->max_corners + NWAVES - 1) / NWAVES);
        if (small) hipLaunchKernelGGL(lk_kernel<true>, lg, dim3(BLOCK), 0, st, A, Bp, (const float *)c->pts[pb], (const int *)c->n_pts[pb], c->next_pts, c->status);
        else hipLaunchKernelGGL(lk_kernel<false>, lg, dim3(BLOCK), 0, st, A, Bp, (const float *)c->pts[pb], (const int *)c->n_pts[pb], c->next_pts, c->status);
        hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(BLOCK), 0, st, (const float *)c->pts[pb], (const float *)c->next_pts, (const unsigned char *)c->status, (const int *)c->n_pts[pb], c->from, c->to, c->m);
        hipLaunchKernelGGL(ransac_subsets_kernel, dim3(1), dim3(64), 0, st, (const int *)c->m, c->ids, c->n_drawn);
        hipLaunchKernelGGL(ransac_eval_kernel, dim3((RANSAC_ITERS + NWAVES - 1) / NWAVES), dim3(BLOCK), 0, st, (const float *)c->from, (const float *)c->to, (const int *)c->m, (const int *)c->ids,
                           (const int *)c->n_drawn, c->good);
    }
    hipLaunchKernelGGL(ransac_pick_kernel, dim3(1), dim3(BLOCK), 0, st, (const float *)c->from, (const float *)c->to, (const int *)c->m, (const int *)c->ids, (const int *)c->n_drawn,
                       (const int *)c->good, c->have_prev, (double)c->downscale, warp6_dev, c->n_inl);
    TLK_HIP(hipGetLastError());
    c->have_prev = 1; c->cur = pb;
    return TLK_OK;
}

// state of one ping-pong side copied onto the other when the gate is closed: <= 2 * LK_LEVELS + 2 buffers
struct CmcCopyTab { const unsigned char *src[2 * LK_LEVELS + 2]; unsigned char *dst[2 * LK_LEVELS + 2]; unsigned int bytes[2 * LK_LEVELS + 2]; int n; };
__global__ void __launch_bounds__(BLOCK) cmc_rollback_kernel(const int *__restrict__ count, CmcCopyTab tab)
{
    if (*count != 0) return;
    const unsigned int tid = blockIdx.x * BLOCK + threadIdx.x, nthr = gridDim.x * BLOCK;
    for (int i = 0; i < tab.n; ++i) {
        const unsigned int n16 = tab.bytes[i] >> 4;                      // (hipMalloc'ed buffers: 256-byte aligned)
        const uint4 *s4 = reinterpret_cast<const uint4 *>(tab.src[i]);
        uint4 *d4 = reinterpret_cast<uint4 *>(tab.dst[i]);
        for (unsigned int k = tid; k < n16; k += nthr) d4[k] = s4[k];
        for (unsigned int k = (n16 << 4) + tid; k < tab.bytes[i]; k += nthr) tab.dst[i][k] = tab.src[i][k];
    }
}

extern "C" int tlk_cmc_apply_dev_gated(tlk_cmc *c, const uint8_t *frame_dev, double *warp6_dev, const int32_t *count_dev, void *hip_stream)
{
    if (!count_dev) return fail(TLK_EINVAL, "tlk_cmc_apply_dev_gated: null pointer");
    const int rc = tlk_cmc_apply_dev(c, frame_dev, warp6_dev, hip_stream);
    if (rc != TLK_OK) return rc;
    // the frame just processed sits in side `now` (= the next call's "previous"); the state before the call is side `old`
    const int now = 1 - c->cur, old = c->cur;
    CmcCopyTab tab;
    tab.n = 0;
    auto add = [&](const void *s, void *d, size_t bytes) { tab.src[tab.n] = (const unsigned char *)s; tab.dst[tab.n] = (unsigned char *)d; tab.bytes[tab.n] = (unsigned int)bytes; ++tab.n; };
    for (int l = 0; l < c->nlev; ++l) {
        add(c->gray[old][l], c->gray[now][l], (size_t)c->lh[l] * c->lw[l]);
        add(c->der[old][l], c->der[now][l], sizeof(short) * 2 * (size_t)c->lh[l] * c->lw[l]);
    }
    add(c->pts[old], c->pts[now], sizeof(float) * 2 * MAX_CORNERS_CAP);
    add(c->n_pts[old], c->n_pts[now], sizeof(int));
    hipLaunchKernelGGL(cmc_rollback_kernel, dim3(256), dim3(BLOCK), 0, (hipStream_t)hip_stream, (const int *)count_dev, tab);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_cmc_apply(tlk_cmc *c, const uint8_t *frame_host, double *warp6_host, int *n_inliers)
{
    if (!c || !frame_host || !warp6_host) return fail(TLK_EINVAL, "tlk_cmc_apply: null pointer");
    TLK_HIP(hipSetDevice(c->device));
    TLK_HIP(hipMemcpy(c->frame_stage, frame_host, (size_t)c->h * c->w * 3, hipMemcpyHostToDevice));
    const int rc = tlk_cmc_apply_dev(c, c->frame_stage, c->warp, nullptr);
    if (rc != TLK_OK) return rc;
    TLK_HIP(hipMemcpy(warp6_host, c->warp, sizeof(double) * 6, hipMemcpyDeviceToHost));
    if (n_inliers) TLK_HIP(hipMemcpy(n_inliers, c->n_inl, sizeof(int), hipMemcpyDeviceToHost));
    return TLK_OK;
}

// debug / test: what = 0 downscaled grey image (dh*dw u8), 1 eigenvalue image (dh*dw f32), 2 corners of the LAST frame (n, 2) f32 [n -> *n_out],
// 3 tracked points of the last LK run (n_prev, 2) f32, 4 their status (n_prev u8), 10 + l pyramid level l image, 20 + l its derivatives (int16 x 2)
extern "C" int tlk_cmc_debug_get(tlk_cmc *c, int what, void *host_buf, size_t cap_bytes, int *n_out)
{
    if (!c || !host_buf) return fail(TLK_EINVAL, "tlk_cmc_debug_get: null pointer");
    TLK_HIP(hipSetDevice(c->device));
    TLK_HIP(hipDeviceSynchronize());
    const int last = 1 - c->cur, prev = c->cur;                   // buffers flip after every apply
    const void *src = nullptr; size_t bytes = 0; int n = 0;
    if (what == 0) { src = c->gray[last][0]; bytes = (size_t)c->dh * c->dw; n = c->dh * c->dw; }
    else if (what == 1) { src = c->eig; bytes = sizeof(float) * (size_t)c->dh * c->dw; n = c->dh * c->dw; }
    else if (what == 2) { TLK_HIP(hipMemcpy(&n, c->n_pts[last], sizeof(int), hipMemcpyDeviceToHost)); src = c->pts[last]; bytes = sizeof(float) * 2 * (size_t)n; }
    else if (what == 3 || what == 4) { TLK_HIP(hipMemcpy(&n, c->n_pts[prev], sizeof(int), hipMemcpyDeviceToHost)); src = what == 3 ? (const void *)c->next_pts : (const void *)c->status; bytes = what == 3 ? sizeof(float) * 2 * (size_t)n : (size_t)n; }
    else if (what >= 10 && what < 10 + c->nlev) { const int l = what - 10; src = c->gray[last][l]; bytes = (size_t)c->lh[l] * c->lw[l]; n = c->lw[l]; }
    else if (what >= 20 && what < 20 + c->nlev) { const int l = what - 20; src = c->der[last][l]; bytes = sizeof(short) * 2 * (size_t)c->lh[l] * c->lw[l]; n = c->lw[l]; }
    else return fail(TLK_EINVAL, "tlk_cmc_debug_get: unknown item");
    if (bytes > cap_bytes) return fail(TLK_ECAPACITY, "tlk_cmc_debug_get: buffer too small");
    if (bytes) TLK_HIP(hipMemcpy(host_buf, src, bytes, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    return TLK_OK;
}
