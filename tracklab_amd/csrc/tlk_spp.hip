// tlk_spp.hip -- the spatial-pyramid-pooling block of YOLOX's CSPDarknet and RTMPose's CSPNeXt (SPPBottleneck, kernel sizes 5 / 9 / 13, stride
// 1, "same" padding with -inf): ONE pass that reads the activation once and writes the concatenation [x | max5 | max9 | max13] the following
// 1 x 1 convolution consumes.  The reference runs both networks inside ONNXRuntime (tracklab/wrappers/bbox_detector/rtmlib_api.py:21,
// wrappers/pose_estimator/rtmlib_api.py:21); the library route here was three max-pool launches + a concatenation copy (3.9 of the 45 ms of
// the config-4 pose forward on 8 x 6 maps).
//
// Algorithmic bytes = 1 read + 4 writes of the map; measured 0.5-1.2 TB/s on them (profiles/r04_dwconv_spp.txt: the 169-tap window walk with
// three running maxima is VALU work, and the maps are small) -- 8.7x (pose, 8 x 6 maps) / 8.4x (YOLOX, 20 x 20) faster than three max_pool2d
// launches + a concatenation.  One lane owns 16 bytes of channels of one pixel; lanes are laid out
// (pixel, channel group) with the channel group fastest, so loads and stores are contiguous NHWC runs; the 13 x 13 window is walked once
// (neighbours re-read the same lines from L1), rows / columns that no lane of the wavefront needs are skipped.  max is exact in every
// precision, so the result is bit-identical to any other evaluation order (inputs must be NaN-free: the hardware max returns the number).
#include "tlk_common.hpp"

using namespace tlk;

namespace {

struct SppArgs {
    const void *x;
    void *y;
    int N, H, W, C, CG;
    int x_pix, y_pix;            // elements between two pixels of x / y (y_pix >= 4 * C)
    long long items;             // N * H * W * CG
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NaN-propagating max, like torch.max_pool2d (__builtin_elementwise_max has fmax semantics and DROPS a NaN -- a NaN produced
// upstream must stay visible to the pipeline's non-finite check, the reason the kernels' ReLU was rewritten in r05).  gfx950 has the
// IEEE-754-2019 maximum as ONE instruction (v_maximum3_f32 / v_pk_maximum3_f16): same cost as the NaN-dropping v_max it replaces (a first
// r06 form with compares and selects took the SPP kernel from 459 to 1274 us on the RTMPose maps).
template <typename V> __device__ __forceinline__ V vmax(V a, V b)
{
    return __builtin_elementwise_maximum(a, b);
}

template <typename T, typename V, int VN>
__global__ void __launch_bounds__(256) spp_kernel(const SppArgs p)
{
    // XCD-aware block order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2); give every XCD a
    // contiguous run of blocks, so that the workgroups sharing a map / neighbouring columns read it through ONE L2 (measured before: the
    // 8 x 6 pose maps were fetched from memory 8 times, once per XCD)
    long long blk;
    {
        const long long b = blockIdx.x, q = (long long)gridDim.x >> 3;
        const int r = (int)(gridDim.x & 7), xcd = (int)(b & 7);
        blk = (long long)xcd * q + (xcd < r ? xcd : r) + (b >> 3);
    }
    const long long item = blk * 256 + threadIdx.x;
    const bool live = item < p.items;
    const long long it = live ? item : p.items - 1;
    const int cg = (int)(it % p.CG);
    long long t = it / p.CG;
    const int x = (int)(t % p.W);
    t /= p.W;
    const int y = (int)(t % p.H);
    const long long n = t / p.H;
    const T *xb = (const T *)p.x + (n * p.H * p.W) * p.x_pix + cg * VN;
    V ninf;
#pragma unroll
    for (int c = 0; c < VN; ++c) ninf[c] = (T)(-__builtin_inff());
    V m5 = ninf, m9 = ninf, m13 = ninf, ctr = ninf;
#pragma unroll
    for (int dy = -6; dy <= 6; ++dy) {
        const int iy = y + dy;
        const bool rowok = live && (unsigned)iy < (unsigned)p.H;
        if (__builtin_amdgcn_ballot_w64(rowok) == 0) continue;
        V h5 = ninf, h9 = ninf, h13 = ninf;
#pragma unroll
        for (int dx = -6; dx <= 6; ++dx) {
            const int ix = x + dx;
            const bool ok = rowok && (unsigned)ix < (unsigned)p.W;
            if (__builtin_amdgcn_ballot_w64(ok) == 0) continue;
            V v = ninf;
            if (ok) v = *(const V *)(xb + ((long long)iy * p.W + ix) * p.x_pix);
            h13 = vmax(h13, v);
            if (dx >= -4 && dx <= 4) h9 = vmax(h9, v);
            if (dx >= -2 && dx <= 2) h5 = vmax(h5, v);
            if (dx == 0 && dy == 0) ctr = v;
        }
        m13 = vmax(m13, h13);
        if (dy >= -4 && dy <= 4) m9 = vmax(m9, h9);
        if (dy >= -2 && dy <= 2) m5 = vmax(m5, h5);
    }
    if (!live) return;
    T *yo = (T *)p.y + ((n * p.H + y) * p.W + x) * p.y_pix + cg * VN;
    *(V *)(yo) = ctr;
    *(V *)(yo + p.C) = m5;
    *(V *)(yo + 2 * p.C) = m9;
    *(V *)(yo + 3 * p.C) = m13;
}

// Plain k x k / stride s max pooling (-inf padding) of a channels-last map: ResNet-50's stem pool (3 x 3, stride 2, pad 1) -- torch's
// max_pool_forward_nhwc took 2.6 ms of the fp32 step and 2.4 ms of the f16 one (r04 / r05 profiles).  One lane = 16 bytes of channels of one
// output pixel, channel group fastest (contiguous NHWC runs both ways); the k * k taps of neighbouring outputs overlap and come from L1 / L2.
// max is exact.  Honours tlk_conv_set_dynamic_batch like the convolutions around it.
struct PoolArgs {
    const void *x;
    void *y;
    int N, H, W, Ho, Wo, C, CG, k, stride, pad, x_pix, y_pix;
    long long items;
    const int *n_dyn;
};

template <typename T, typename V, int VN>
__global__ void __launch_bounds__(256) maxpool_kernel(const PoolArgs p)
{
    long long items = p.items;
    if (p.n_dyn) { const long long live = (long long)p.n_dyn[0] * p.Ho * p.Wo * p.CG; items = live < items ? (live < 0 ? 0 : live) : items; }
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    const int cg = (int)(item % p.CG);
    long long t = item / p.CG;
    const int ox = (int)(t % p.Wo);
    t /= p.Wo;
    const int oy = (int)(t % p.Ho);
    const long long n = t / p.Ho;
    const T *xb = (const T *)p.x + (n * p.H * p.W) * p.x_pix + cg * VN;
    V m;
#pragma unroll
    for (int c = 0; c < VN; ++c) m[c] = (T)(-__builtin_inff());
    for (int dy = 0; dy < p.k; ++dy) {
        const int iy = oy * p.stride - p.pad + dy;
        if ((unsigned)iy >= (unsigned)p.H) continue;
        for (int dx = 0; dx < p.k; ++dx) {
            const int ix = ox * p.stride - p.pad + dx;
            if ((unsigned)ix >= (unsigned)p.W) continue;
            m = vmax(m, *(const V *)(xb + ((long long)iy * p.W + ix) * p.x_pix));
        }
    }
    *(V *)((T *)p.y + ((n * p.Ho + oy) * p.Wo + ox) * p.y_pix + cg * VN) = m;
}

}  // namespace

extern "C" int tlk_maxpool2d_nhwc(const void *x_dev, void *y_dev, int n, int h, int w, int c, int k, int stride, int pad, int dtype, int x_pix_stride,
                                  int y_pix_stride, void *hip_stream)
{
    if (!x_dev || !y_dev) return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: null pointer");
    if (dtype != TLK_F32 && dtype != TLK_F16) return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: dtype must be TLK_F32 or TLK_F16");
    const int v = dtype == TLK_F16 ? 8 : 4;
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || c % v != 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k)
        return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: bad shape (channels a multiple of 16 bytes, pad <= k / 2)");
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: empty output");
    const int xp = x_pix_stride ? x_pix_stride : c, yp = y_pix_stride ? y_pix_stride : c;
    if (xp < c || yp < c || xp % v != 0 || yp % v != 0) return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: pixel strides must be >= channels and multiples of 16 bytes");
    if (((uintptr_t)x_dev | (uintptr_t)y_dev) & 15) return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: x, y must be 16-byte aligned");
    if (n == 0) return TLK_OK;
    PoolArgs a;
    a.x = x_dev; a.y = y_dev; a.N = n; a.H = h; a.W = w; a.Ho = ho; a.Wo = wo; a.C = c; a.CG = c / v; a.k = k; a.stride = stride; a.pad = pad;
    a.x_pix = xp; a.y_pix = yp; a.items = (long long)n * ho * wo * a.CG; a.n_dyn = conv_dynamic_batch();
    const long long blocks = (a.items + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_maxpool2d_nhwc: more than 2^31 - 1 workgroups");
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == TLK_F16) hipLaunchKernelGGL((maxpool_kernel<_Float16, f16x8, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((maxpool_kernel<float, f32x4, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_spp_maxpool_nhwc(const void *x_dev, void *y_dev, int n, int h, int w, int c, int dtype, int x_pix_stride, int y_pix_stride,
                                    void *hip_stream)
{
    if (!x_dev || !y_dev) return fail(TLK_EINVAL, "tlk_spp_maxpool_nhwc: null pointer");
    if (dtype != TLK_F32 && dtype != TLK_F16) return fail(TLK_EINVAL, "tlk_spp_maxpool_nhwc: dtype must be TLK_F32 or TLK_F16");
    const int v = dtype == TLK_F16 ? 8 : 4;
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || c % v != 0) return fail(TLK_EINVAL, "tlk_spp_maxpool_nhwc: channels must be a positive multiple of 16 bytes");
    const int xp = x_pix_stride ? x_pix_stride : c, yp = y_pix_stride ? y_pix_stride : 4 * c;
    if (xp < c || yp < 4 * c || xp % v != 0 || yp % v != 0)
        return fail(TLK_EINVAL, "tlk_spp_maxpool_nhwc: pixel strides must be >= channels (x) / 4 * channels (y) and multiples of 16 bytes");
    if (((uintptr_t)x_dev | (uintptr_t)y_dev) & 15) return fail(TLK_EINVAL, "tlk_spp_maxpool_nhwc: x, y must be 16-byte aligned");
    if (n == 0) return TLK_OK;
    SppArgs a;
    a.x = x_dev; a.y = y_dev; a.N = n; a.H = h; a.W = w; a.C = c; a.CG = c / v; a.x_pix = xp; a.y_pix = yp;
    a.items = (long long)n * h * w * a.CG;
    const long long blocks = (a.items + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_spp_maxpool_nhwc: more than 2^31 - 1 workgroups");
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == TLK_F16) hipLaunchKernelGGL((spp_kernel<_Float16, f16x8, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((spp_kernel<float, f32x4, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
