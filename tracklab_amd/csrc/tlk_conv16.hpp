// tlk_conv16.hpp -- what the 16-bit MFMA convolution kernels share (tlk_conv16.hip: the register-staged and 128 x 128 direct-to-LDS
// kernels of r04; tlk_conv16x.hip: the large-tile direct-to-LDS kernels of r05): argument block, modes, epilogue arithmetic.
#pragma once
#include <hip/hip_fp16.h>

#include "tlk_common.hpp"

namespace tlk {
namespace c16 {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
enum { MODE_F16 = 0, MODE_SPLIT = 1, MODE_F32 = 2 };      // MODE_F32 (tlk_conv16x.hip only): fp32 tensors on the exact fp32-input MFMA, same loader / ring
constexpr int ROW_BYTES = 128;                            // one K slice of a tile row in LDS: 64 f16 of one plane, or 32 hi | 32 lo
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

struct Conv16Args {
    const _Float16 *x, *x_lo, *w, *w_lo, *res, *res_lo;
    const float *bias;
    _Float16 *y, *y_lo;
    float *y32;
    long long M;
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, K;
    int x_pix, y_pix, r_pix;      // elements between two pixels of x / y / residual
    int tiles_n;
    long long tiles;
    int res_post;                 // 1: the residual is added AFTER the activation (y = act(conv + bias) + r)
    const int *n_dyn;             // not NULL: live image count in device memory (tlk_conv_set_dynamic_batch); the call's n is the capacity
    // r06, split mode only -- SCALED planes: a tensor's value is scale * (hi + lo * 2^-11) with scale a power of two kept in device memory beside
    // the tensor, so that activations beyond float16's range (65504) no longer saturate.  s_in / s_res: the scale of the input / residual planes
    // (NULL = 1); s_out: this layer's output state {scale in use, bits of the largest |output| seen since the last tlk_split_scale_update}
    // (NULL = unscaled output, nothing recorded).  Powers of two: every multiplication by a scale is exact.
    const float *s_in = nullptr, *s_res = nullptr;
    float *s_out = nullptr;
};

#if defined(__HIPCC__)
template <int ACT> __device__ __forceinline__ float act16(float v)
{
    if (ACT == ACT_RELU) return v < 0.f ? 0.f : v;          // (this form lets NaN through, like torch.relu: an overflow upstream must stay visible, r05)
    if (ACT == ACT_SILU) return v / (1.f + __expf(-v));
    return v;
}

__device__ __forceinline__ void split_f32(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * LO_SCALE);
}

struct SplitScales { float in, res, out_inv; };
__device__ __forceinline__ SplitScales load_scales(const Conv16Args &p)
{
    SplitScales s;
    s.in = p.s_in ? p.s_in[0] : 1.f;
    s.res = p.s_res ? p.s_res[0] : 1.f;
    s.out_inv = p.s_out ? 1.f / p.s_out[0] : 1.f;          // (a power of two: the reciprocal and the products with it are exact)
    return s;
}
// largest |value| of this wavefront -> a state word (non-negative floats order like their bit patterns).  The atomic is GUARDED by a relaxed
// device-scope load of the word: a launch has ~10^5-10^6 wavefronts and same-address atomics retire one at a time in the L2 (~12 ns each: measured,
// an unguarded atomic per wavefront took a 2.5 ms layer to 12 ms); once the first few wavefronts have raised the word almost nobody exceeds it, and a
// stale read only costs a redundant atomic, never a missed maximum.
__device__ __forceinline__ void record_amax(float *word, float am)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((threadIdx.x & 63) == 0 && am > 0.f) {
        unsigned int *w = reinterpret_cast<unsigned int *>(word);
        const unsigned int mine = __float_as_uint(am);
        if (mine > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, mine);
    }
}
__device__ __forceinline__ void note_amax(const Conv16Args &p, float am)
{
    if (p.s_out) record_amax(p.s_out + 1, am);
}

// rows of this launch that exist: the static M, or n_dyn[0] images' worth when a dynamic batch is set
__device__ __forceinline__ long long live_rows(const Conv16Args &p)
{
    long long M = p.M;
    if (p.n_dyn) { const long long md = (long long)p.n_dyn[0] * p.Ho * p.Wo; M = md < M ? (md < 0 ? 0 : md) : M; }
    return M;
}

// XCD-aware tile order: hardware deals consecutive workgroups round-robin to the 8 XCDs; give each XCD a contiguous run of tiles
__device__ __forceinline__ long long xcd_tile(long long tiles)
{
    const long long b = blockIdx.x, q = tiles >> 3;
    const int r = (int)(tiles & 7), xcd = (int)(b & 7);
    return (long long)xcd * q + (xcd < r ? xcd : r) + (b >> 3);
}

#endif  // __HIPCC__

const unsigned char *zero_page();      // 256 zero bytes on the current device (tlk_conv16.hip)

// tlk_conv16x.hip: the large-tile kernels.  cfg: 0 = choose by shape (may decline: returns 1 = "not mine", the caller keeps its own kernel),
// > 0 = force that tile configuration (probes).  Returns TLK_OK when launched.
int launch16x(Conv16Args &a, bool split, bool out32, int act, int cfg, hipStream_t st);
extern int g_last_cfg16x;     // what tlk_conv16_last_config reports
// the same kernels on fp32 tensors (a.x / a.w / a.res / a.y32 hold float pointers): the memory-bound layers of the fp32 networks
int launch32x(Conv16Args &a, int act, int cfg, hipStream_t st);

}  // namespace c16
}  // namespace tlk
