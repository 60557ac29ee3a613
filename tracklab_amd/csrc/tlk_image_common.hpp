// tlk_image_common.hpp -- helpers shared by the image kernels of libtlk (tlk_image.hip: letterbox + cv2-semantics crops; tlk_pil.hip:
// Pillow-semantics crops): cv2 fixed-point coefficients, element conversion, 16-byte packs, streaming stores.  File-local (anonymous
// namespace): every translation unit gets its own copy.
#pragma once
#include <hip/hip_fp16.h>

#include "tlk_common.hpp"

using namespace tlk;

namespace {


// ---- cv2.resize(INTER_LINEAR, uint8) coefficients, OpenCV imgproc/resize.cpp (INTER_RESIZE_COEF_BITS = 11)
struct Coef { int s; int w0, w1; };
__host__ __device__ __forceinline__ Coef cv_coef_s(int d, int ssize, double scale, bool is_col)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (is_col) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    Coef c;
    c.s = s;
    c.w0 = (int)(short)(int)rintf((1.f - f) * 2048.f);      // round-half-even, as cvRound
    c.w1 = (int)(short)(int)rintf(f * 2048.f);
    return c;
}
__host__ __device__ __forceinline__ Coef cv_coef(int d, int ssize, int dsize, bool is_col)
{
    return cv_coef_s(d, ssize, (double)ssize / (double)dsize, is_col);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one bilinear sample of 3 interleaved channels; region = (rh x rw) pixels at `base`, row stride in bytes
__device__ __forceinline__ void sample3(const unsigned char *__restrict__ base, int stride, int rh, int rw, Coef cy, Coef cx,
                                        int (&v)[3])
{
    const unsigned char *r0 = base + (size_t)clampi(cy.s, 0, rh - 1) * stride;
    const unsigned char *r1 = base + (size_t)clampi(cy.s + 1, 0, rh - 1) * stride;
    const int x0 = cx.s * 3, x1 = (cx.s + 1 < rw ? cx.s + 1 : rw - 1) * 3;
    const bool need_x1 = cx.w1 != 0, need_r1 = cy.w1 != 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int S0 = (int)r0[x0 + c] * cx.w0;
        if (need_x1) S0 += (int)r0[x1 + c] * cx.w1;
        int S1 = 0;
        if (need_r1) { S1 = (int)r1[x0 + c] * cx.w0; if (need_x1) S1 += (int)r1[x1 + c] * cx.w1; }
        const int r = (((cy.w0 * (S0 >> 4)) >> 16) + ((cy.w1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v[c] = clampi(r, 0, 255);
    }
}

template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ __half cvt<__half>(float v) { return __float2half_rn(v); }
struct bf16_t { unsigned short x; };
template <> __device__ __forceinline__ bf16_t cvt<bf16_t>(float v)
{
    unsigned int u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);          // round to nearest even (inputs here are finite)
    bf16_t r; r.x = (unsigned short)(u >> 16); return r;
}

template <typename T, int N> struct alignas(sizeof(T) * N) Pack { T v[N]; };
// Streaming store of a 16- or 32-byte pack: the crop / letterbox outputs are written once and read by the next kernel only after hundreds of MB
// more have gone by; written with the nontemporal hint they do not push the source rows and tables of the running workgroups out of L2
// (crop kernel: 244 -> 218 us, profiles/r02_crop_fat_phases.txt). ONLY for fully coalesced stores -- whole cache lines per instruction: strided
// 16-byte pieces written with the hint are not merged in L2 and each becomes a partial write to memory (crop_fat_kernel 266 -> 514 us, fp32 2.4 ms)
typedef unsigned int tlk_u32x4 __attribute__((ext_vector_type(4)));
template <typename P>
__device__ __forceinline__ void stream_store(P *dst, const P &v)
{
    static_assert(sizeof(P) % 16 == 0, "packs of 16 bytes");
    const tlk_u32x4 *src = reinterpret_cast<const tlk_u32x4 *>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(P) / 16; ++i) __builtin_nontemporal_store(src[i], reinterpret_cast<tlk_u32x4 *>(dst) + i);
}

enum { LAYOUT_NCHW = 0, LAYOUT_NHWC = 1, LAYOUT_FOCUS_NHWC = 2 };


// ---- tile geometry shared by the crop kernels of both files
constexpr int STAGE_PAD = 32;           // bytes of slack per staged row (alignment shift + clamped x+1 tap)
constexpr int CS_BAND = 16;                            // output rows per workgroup
constexpr int CS_ROWS = 18;                            // staged source rows: CS_BAND * scale + 2 <= 18 for scale <= 1 (up-scaling / same size)
constexpr int CS_ROW_BYTES = 512;                      // 32 chunks of 16 bytes: crops up to 160 px wide
constexpr int CS_LUT_N = 1024;                         // entries per channel (t <= 1020)
constexpr int CF_BANDS = 8;
constexpr int WV_ROWS = 4;                             // output rows of a mini-band (lane = (row, group of 8 px) in the vertical pass)
constexpr int WV_SRC = 6;                              // staged source rows per mini-band: WV_ROWS * scale + 2 for scale <= 1; 2-row mini-bands up to scale 2
constexpr int WV_WAVE_LDS = WV_SRC * CS_ROW_BYTES + WV_SRC * 128 * 6;

}  // namespace
