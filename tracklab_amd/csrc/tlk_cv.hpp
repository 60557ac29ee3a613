// tlk_cv.hpp -- the OpenCV image primitives the camera-motion estimators share (tlk_cmc.hip: sparse optical flow; tlk_ecc.hip: ECC):
// cvtColor(BGR2GRAY) in 15-bit fixed point and resize(INTER_LINEAR) in 11-bit fixed point, fused (grey values recomputed per tap).
// OpenCV's arithmetic restated: PARITY UNPINNED (oracle/src/cmc.c holds the CPU restatement these are checked against).
#pragma once
#include "tlk_common.hpp"

namespace tlk {
namespace cv {

__device__ __forceinline__ int reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }
__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int cv_floor_f(float v) { return (int)floorf(v); }
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

__device__ __forceinline__ int gray_at(const unsigned char *img, int w, int y, int x)
{
    const unsigned char *p = img + ((size_t)y * w + x) * 3;
    return (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
}
__device__ __forceinline__ void lin_coef(int d, int ssize, int dsize, bool is_col, int &s0, int &w0, int &w1)
{
    float f = (float)(((double)d + 0.5) * ((double)ssize / (double)dsize) - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (is_col) { if (s < 0) { f = 0.f; s = 0; } if (s >= ssize - 1) { f = 0.f; s = ssize - 1; } }
    s0 = s; w0 = (int)(short)(int)rintf((1.f - f) * 2048.f); w1 = (int)(short)(int)rintf(f * 2048.f);
}

static __global__ void __launch_bounds__(BLOCK) gray_resize_kernel(const unsigned char *__restrict__ frame, int h, int w, unsigned char *__restrict__ out, int dh, int dw)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= dh * dw) return;
    const int dy = i / dw, dx = i - dy * dw;
    if (dh == h && dw == w) { out[i] = (unsigned char)gray_at(frame, w, dy, dx); return; }
    int sy, b0, b1, sx, a0, a1;
    lin_coef(dy, h, dh, false, sy, b0, b1);
    lin_coef(dx, w, dw, true, sx, a0, a1);
    const int y0 = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy), y1 = sy + 1 < 0 ? 0 : (sy + 1 > h - 1 ? h - 1 : sy + 1);
    const int sx1 = sx + 1 < w ? sx + 1 : w - 1;
    const int S0 = gray_at(frame, w, y0, sx) * a0 + gray_at(frame, w, y0, sx1) * a1, S1 = gray_at(frame, w, y1, sx) * a0 + gray_at(frame, w, y1, sx1) * a1;
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    out[i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

}  // namespace cv
}  // namespace tlk
