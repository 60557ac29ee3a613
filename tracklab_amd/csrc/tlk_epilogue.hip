// tlk_epilogue.hip -- fused convolution epilogue for the PyTorch-ROCm backbones: y = act(x + bias[c] (+ residual)),
// in place on a channels-last (N*H*W, C) fp16/bf16 activation. MIOpen runs conv, bias (two op-tensor launches),
// activation and the residual add as separate full passes over activations that are GBs large at ReID batch sizes;
// this kernel makes it one read (two with a residual) + one write: HBM-bound, 16 B per lane, grid-stride.
#include <hip/hip_fp16.h>

#include "tlk_common.hpp"

using namespace tlk;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

struct alignas(16) H8 { __half2 v[4]; };

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
struct alignas(16) B8 { unsigned short v[8]; };

template <int ACT> __device__ __forceinline__ float act(float v)
{
    if (ACT == ACT_RELU) return v < 0.f ? 0.f : v;          // (this form lets NaN through, like torch.relu: an overflow upstream must stay visible, r05)
    if (ACT == ACT_SILU) return v / (1.f + __expf(-v));
    return v;
}

template <int ACT, bool RES>
__global__ void __launch_bounds__(BLOCK) bias_act_f16_kernel(__half *__restrict__ x, const __half *__restrict__ bias,
                                                             const __half *__restrict__ res, long long n_vec, int c_vec)
{
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n_vec; i += stride) {
        const int cv = (int)(i % c_vec);
        H8 a = reinterpret_cast<const H8 *>(x)[i];
        const H8 b = reinterpret_cast<const H8 *>(bias)[cv];
        H8 r;
        if (RES) r = reinterpret_cast<const H8 *>(res)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float2 fa = __half22float2(a.v[k]);
            const float2 fb = __half22float2(b.v[k]);
            fa.x += fb.x; fa.y += fb.y;
            if (RES) { const float2 fr = __half22float2(r.v[k]); fa.x += fr.x; fa.y += fr.y; }
            fa.x = act<ACT>(fa.x); fa.y = act<ACT>(fa.y);
            a.v[k] = __float22half2_rn(fa);
        }
        reinterpret_cast<H8 *>(x)[i] = a;
    }
}

template <int ACT, bool RES>
__global__ void __launch_bounds__(BLOCK) bias_act_bf16_kernel(unsigned short *__restrict__ x, const unsigned short *__restrict__ bias,
                                                              const unsigned short *__restrict__ res, long long n_vec, int c_vec)
{
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n_vec; i += stride) {
        const int cv = (int)(i % c_vec);
        B8 a = reinterpret_cast<const B8 *>(x)[i];
        const B8 b = reinterpret_cast<const B8 *>(bias)[cv];
        B8 r;
        if (RES) r = reinterpret_cast<const B8 *>(res)[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = bf16_to_f32(a.v[k]) + bf16_to_f32(b.v[k]);
            if (RES) v += bf16_to_f32(r.v[k]);
            a.v[k] = f32_to_bf16(act<ACT>(v));
        }
        reinterpret_cast<B8 *>(x)[i] = a;
    }
}

}  // namespace

extern "C" int tlk_bias_act_nhwc(void *x_dev, const void *bias_dev, const void *residual_dev, long long rows, int channels,
                                 int act_kind, int dtype, void *hip_stream)
{
    if (rows < 0 || channels <= 0 || channels % 8 != 0) return fail(TLK_EINVAL, "tlk_bias_act_nhwc: channels must be a positive multiple of 8");
    if (act_kind < 0 || act_kind > 2 || (dtype != TLK_F16 && dtype != TLK_BF16)) return fail(TLK_EINVAL, "tlk_bias_act_nhwc: bad act/dtype");
    if (rows == 0) return TLK_OK;
    if (!x_dev || !bias_dev) return fail(TLK_EINVAL, "tlk_bias_act_nhwc: null pointer");
    const int c_vec = channels / 8;
    const long long n_vec = rows * c_vec;
    long long blocks = (n_vec + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 16) blocks = 256 * 16;            // grid-stride beyond 16 blocks per CU
    hipStream_t st = (hipStream_t)hip_stream;
    const bool res = residual_dev != nullptr;
#define LAUNCH(KERN, T, A)                                                                                                   \
    do {                                                                                                                     \
        if (res) hipLaunchKernelGGL((KERN<A, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (T *)x_dev, (const T *)bias_dev, \
                                    (const T *)residual_dev, n_vec, c_vec);                                                  \
        else hipLaunchKernelGGL((KERN<A, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (T *)x_dev, (const T *)bias_dev,     \
                                (const T *)nullptr, n_vec, c_vec);                                                           \
    } while (0)
    if (dtype == TLK_F16) {
        if (act_kind == 0) LAUNCH(bias_act_f16_kernel, __half, ACT_NONE);
        else if (act_kind == 1) LAUNCH(bias_act_f16_kernel, __half, ACT_RELU);
        else LAUNCH(bias_act_f16_kernel, __half, ACT_SILU);
    } else {
        if (act_kind == 0) LAUNCH(bias_act_bf16_kernel, unsigned short, ACT_NONE);
        else if (act_kind == 1) LAUNCH(bias_act_bf16_kernel, unsigned short, ACT_RELU);
        else LAUNCH(bias_act_bf16_kernel, unsigned short, ACT_SILU);
    }
#undef LAUNCH
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
