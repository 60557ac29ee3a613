// tlk_dwconv.hip -- depthwise k x k convolution (stride 1, pad k/2) of a channels-last activation with bias + activation inside:
// the 5 x 5 depthwise halves of RTMPose's CSPNeXt blocks (mmdet DepthwiseSeparableConvModule; the reference runs the network behind
// tracklab/wrappers/pose_estimator/rtmlib_api.py:21-36, configs/modules/pose_estimator/rtmpose_rtmlib.yaml).  The library route (MIOpen's
// grouped-convolution kernels + a separate bias / SiLU pass) spent 33 of the 76 ms of the config-4 pose forward (2400 crops, f16) here,
// 18x above the HBM time of these layers.
//
// Every input element is read from HBM once and every output element written once:
//   * one lane owns 16 bytes of channels (8 x f16 / 4 x f32) of ONE image column and marches down the rows of its strip; lanes are laid
//     out (column, channel group) with the channel group fastest, so every wave-wide load / store is one contiguous run of the NHWC row;
//   * the k taps of a row are k loads per lane (neighbouring lanes re-read the same lines: L1 hits, not HBM traffic), issued one row ahead
//     of the arithmetic;
//   * the k x k weights of the lane's channels stay in registers (k*k*4 VGPRs) for the whole march;
//   * a ring of k partial output rows (fp32) collects the contributions: input row r adds its k taps into output rows r-k+1 .. r, the
//     oldest of which is then complete -> + bias, activation, 16-byte store.  No halo is re-read inside a strip; strips (blockIdx.y)
//     exist only to give small batches enough wavefronts and re-read k-1 rows each.
// What bounds it (measured, profiles/r04_dwconv_spp.txt): 5 x 5 in f16 is 200 fp32-accumulating FMAs per 32 bytes moved -- 6.25 FMA / byte,
// i.e. 25 TFMA/s at 4 TB/s against the 39 TFMA/s the VALUs issue -- so the kernel sits on VALU issue, not on HBM: 2.0 TB/s in f16 (time is
// linear in the tap count: 14.8 us per tap + 184 us on 2400 x 64 x 48 x 48), 2.8 TB/s in fp32 (half the FMAs per byte), SiLU's exact
// division another 14 %.  Still 9x faster than the library route (6.4 ms -> 0.72 ms per stage-1 layer).  Next: v_dot2_f32_f16 on tap pairs.
// Arithmetic: fp32 accumulation for both element types (f16: v_fma_mix_f32 -- the f16 operands are read straight into an fp32 fma).
// Each output element is ONE fmaf chain over (ky ascending, kx ascending), rows outside the image skipped, columns outside the image
// entering as zero terms, then + bias, activation (oracle/src/conv.c: orc_dwconv2d_nhwc_f32 walks the same chain; fp32 results are
// bit-identical, SiLU within the device exp's error).
#include "tlk_common.hpp"

using namespace tlk;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

struct DwArgs {
    const void *x, *w;
    const float *bias;
    void *y;
    int N, H, W, C, CG;           // CG = channel groups of 16 bytes
    int x_pix, y_pix;             // elements between two pixels of x / y (>= C: a call may read / write a channel slice of a wider tensor)
    int rows_per_strip;
    long long items;              // N * W * CG
};

template <int ACT> __device__ __forceinline__ float act_f32(float v)
{
    if (ACT == ACT_RELU) return v < 0.f ? 0.f : v;          // (this form lets NaN through, like torch.relu: an overflow upstream must stay visible, r05)
    if (ACT == ACT_SILU) return v / (1.f + __expf(-v));
    return v;
}

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef float type __attribute__((ext_vector_type(4)));
};
template <> struct Vec<_Float16> {
    static constexpr int N = 8;
    typedef unsigned type __attribute__((ext_vector_type(4)));      // eight f16 as four packed registers
};

// f16 x f16 + f32 -> f32 in one VALU instruction, the f16 operands read in place from the low / high half of a packed register.  Spelled in
// assembly because the compiler otherwise hoists 200 weight conversions out of the row loop (v_cvt_f32_f16 + v_fma_f32: 316 registers, one
// wavefront per SIMD).
__device__ __forceinline__ float fma_mix_lo(unsigned a, unsigned b, float c)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned a, unsigned b, float c)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <typename T, int K, int ACT>
__global__ void __launch_bounds__(256) dwconv_kernel(const DwArgs p)
{
    constexpr int V = Vec<T>::N;
    constexpr int PAD = K / 2;
    typedef typename Vec<T>::type vec_t;
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
    if (item >= p.items) return;
    const int cg = (int)(item % p.CG);
    const long long t = item / p.CG;
    const int x = (int)(t % p.W);
    const long long n = t / p.W;
    const int ys = blockIdx.y * p.rows_per_strip;
    const int rs = min(p.rows_per_strip, p.H - ys);           // output rows of this strip
    const T *xb = (const T *)p.x + (n * p.H * p.W) * p.x_pix + cg * V;
    T *yb = (T *)p.y + (n * p.H * p.W) * p.y_pix + cg * V;

    vec_t wr[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wr[ky][kx] = *(const vec_t *)((const T *)p.w + (ky * K + kx) * p.C + cg * V);
    float bs[V];
#pragma unroll
    for (int c = 0; c < V; ++c) bs[c] = p.bias ? p.bias[cg * V + c] : 0.f;

    bool colok[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) colok[kx] = (unsigned)(x + kx - PAD) < (unsigned)p.W;

    float acc[K][V];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int c = 0; c < V; ++c) acc[s][c] = 0.f;

    const vec_t zero = {};
    auto load_row = [&](vec_t (&dst)[K], int r) {             // input row r of the strip = image row ys - PAD + r
        const int iy = ys - PAD + r;
        const bool rowok = (unsigned)iy < (unsigned)p.H && r < rs + 2 * PAD;
        const T *rowp = xb + ((long long)iy * p.W + (x - PAD)) * p.x_pix;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) dst[kx] = (rowok && colok[kx]) ? *(const vec_t *)(rowp + (long long)kx * p.x_pix) : zero;
    };

    vec_t cur[K], nxt[K];
    load_row(nxt, 0);
    const int rtot = rs + 2 * PAD;                            // input rows this strip walks
    for (int r0 = 0; r0 < rtot; r0 += K) {
#pragma unroll
        for (int ph = 0; ph < K; ++ph) {
            const int r = r0 + ph;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) cur[kx] = nxt[kx];
            load_row(nxt, r + 1);
            const int iy = ys - PAD + r;
            if ((unsigned)iy < (unsigned)p.H && r < rtot) {
#pragma unroll
                for (int ky = K - 1; ky >= 0; --ky) {         // output row o = r - ky lives in ring slot (ph - ky) mod K
                    constexpr int KK = K;
                    const int slot = (ph - ky + KK) % KK;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        if constexpr (sizeof(T) == 4) {
#pragma unroll
                            for (int c = 0; c < V; ++c) acc[slot][c] = fmaf(cur[kx][c], wr[ky][kx][c], acc[slot][c]);
                        } else {
#pragma unroll
                            for (int c = 0; c < V; c += 2) {
                                acc[slot][c] = fma_mix_lo(cur[kx][c >> 1], wr[ky][kx][c >> 1], acc[slot][c]);
                                acc[slot][c + 1] = fma_mix_hi(cur[kx][c >> 1], wr[ky][kx][c >> 1], acc[slot][c + 1]);
                            }
                        }
                    }
                }
            }
            // output row o = r - (K - 1) has now seen its last input row
            const int o = r - (K - 1);
            const int slot = (ph + 1) % K;
            if (o >= 0 && o < rs) {
                vec_t out;
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int c = 0; c < V; ++c) out[c] = act_f32<ACT>(acc[slot][c] + bs[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < V; c += 2) {
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        h2 pr = {(_Float16)act_f32<ACT>(acc[slot][c] + bs[c]), (_Float16)act_f32<ACT>(acc[slot][c + 1] + bs[c + 1])};
                        out[c >> 1] = __builtin_bit_cast(unsigned, pr);
                    }
                }
                *(vec_t *)(yb + ((long long)(ys + o) * p.W + x) * p.y_pix) = out;
            }
#pragma unroll
            for (int c = 0; c < V; ++c) acc[slot][c] = 0.f;
        }
    }
}

template <typename T, int K> int launch_k(const DwArgs &a, int act, dim3 grid, hipStream_t st)
{
    if (act == 0) hipLaunchKernelGGL((dwconv_kernel<T, K, ACT_NONE>), grid, dim3(256), 0, st, a);
    else if (act == 1) hipLaunchKernelGGL((dwconv_kernel<T, K, ACT_RELU>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwconv_kernel<T, K, ACT_SILU>), grid, dim3(256), 0, st, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

}  // namespace

extern "C" int tlk_dwconv2d_nhwc(const void *x_dev, const void *w_dev, const float *bias_dev, void *y_dev, int n, int h, int w, int c, int k,
                                 int act_kind, int dtype, int x_pix_stride, int y_pix_stride, void *hip_stream)
{
    if (!x_dev || !w_dev || !y_dev) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: null pointer");
    if (dtype != TLK_F32 && dtype != TLK_F16) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: dtype must be TLK_F32 or TLK_F16");
    if (k != 3 && k != 5) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: k must be 3 or 5 (stride 1, pad k/2)");
    if (act_kind < 0 || act_kind > 2) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: act_kind must be 0 (none), 1 (ReLU) or 2 (SiLU)");
    const int v = dtype == TLK_F16 ? 8 : 4;
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || c % v != 0) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: channels must be a positive multiple of 16 bytes");
    const int xp = x_pix_stride ? x_pix_stride : c, yp = y_pix_stride ? y_pix_stride : c;
    if (xp < c || yp < c || xp % v != 0 || yp % v != 0) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: pixel strides must be >= channels and multiples of 16 bytes");
    if (((uintptr_t)x_dev | (uintptr_t)w_dev | (uintptr_t)y_dev) & 15) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: x, w, y must be 16-byte aligned");
    if (n == 0) return TLK_OK;
    DwArgs a;
    a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.y = y_dev;
    a.N = n; a.H = h; a.W = w; a.C = c; a.CG = c / v; a.x_pix = xp; a.y_pix = yp;
    a.items = (long long)n * w * a.CG;
    const long long blocks = (a.items + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_dwconv2d_nhwc: more than 2^31 - 1 workgroups");
    // strips only where one march per column would leave the chip short of wavefronts (target: >= 8 waves on each of 256 CUs), never
    // shorter than 8 rows (each strip re-reads k - 1 rows)
    int strips = 1;
    const long long want_blocks = 256 * 2;
    if (blocks < want_blocks) strips = (int)std::min<long long>((want_blocks + blocks - 1) / blocks, std::max(1, h / 8));
    a.rows_per_strip = (h + strips - 1) / strips;
    strips = (h + a.rows_per_strip - 1) / a.rows_per_strip;
    const dim3 grid((unsigned)blocks, (unsigned)strips);
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == TLK_F16) return k == 5 ? launch_k<_Float16, 5>(a, act_kind, grid, st) : launch_k<_Float16, 3>(a, act_kind, grid, st);
    return k == 5 ? launch_k<float, 5>(a, act_kind, grid, st) : launch_k<float, 3>(a, act_kind, grid, st);
}
