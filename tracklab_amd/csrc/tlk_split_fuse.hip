// tlk_split_fuse.hip -- the element-wise joints of a split-precision network (r06): the sum of up to four tensors brought to one resolution,
// an optional ReLU, and the result written as scaled (hi, lo) float16 planes -- one HBM pass where the composition of library passes
// (merge every term to fp32, nearest up-sampling, n - 1 additions, ReLU, split) takes ten.  What HRNet's exchange units
// (tracklab/configs/modules/reid/bpbreid.yaml:53 backbone "hrnet32") do between their branches, and -- with one term and a channel offset --
// the concatenation of the branches in front of the part-based head, which has to bring four tensors with four scales onto one.
//
// HBM-bound: algorithmic bytes per output element = 4 (planes written) + per term 4 (planes) or 4 (fp32) / 4^shift.
#include <hip/hip_runtime.h>

#include "tlk.h"
#include "tlk_conv16.hpp"

using namespace tlk;
using namespace tlk::c16;

namespace {

constexpr int FUSE_BLOCK = 256;
constexpr int MAX_TERMS = 4;

struct FuseTerm {
    const _Float16 *hi, *lo;      // planes (both set), or
    const float *f32;             // a plain fp32 tensor, or
    const _Float16 *h16;          // a plain f16 tensor (OUT = 2 only)
    const float *scale;           // the planes' scale (NULL = 1)
    int shift;                    // the term lives at (h >> shift, w >> shift): nearest up-sampling by 2^shift
    int pix;                      // elements between two of its pixels
};

struct FuseArgs {
    FuseTerm t[MAX_TERMS];
    int nt, h, w, c, relu, y_pix;
    long long n;
    _Float16 *yhi, *ylo;
    float *y32;                   // OUT = 1 (tlk_fuse_sum_f32): the result as plain fp32, no planes, no state
    _Float16 *y16;                // OUT = 2 (tlk_fuse_sum_f16): f16 terms, f16 result, every partial sum rounded to f16 (torch's half `y = y + t`)
    float *state;                 // {scale of the planes written, recorded maximum} or NULL (scale 1, nothing recorded)
    const int *n_dyn;             // live image count (tlk_conv_set_dynamic_batch) or NULL
};

// One item = 8 channels of one output pixel (16 bytes of each plane); consecutive lanes, consecutive 16 bytes.  (A two-items-per-lane form with all
// loads issued ahead was measured SLOWER, 416 vs 348 us per launch over HRNet-W32's 30 joints: the kernel is not short of loads in flight.)
template <int OUT> __global__ void __launch_bounds__(FUSE_BLOCK) split_fuse_sum_kernel(const FuseArgs p)
{
    const int cg = p.c >> 3;
    long long n = p.n;
    if (p.n_dyn) { const long long nd = p.n_dyn[0]; n = nd < n ? (nd < 0 ? 0 : nd) : n; }
    const long long hw = (long long)p.h * p.w;
    const long long items = n * hw * cg;
    const float inv = p.state ? 1.f / p.state[0] : 1.f;
    float sc[MAX_TERMS];
#pragma unroll
    for (int t = 0; t < MAX_TERMS; ++t) sc[t] = (t < p.nt && p.t[t].hi && p.t[t].scale) ? p.t[t].scale[0] : 1.f;
    float am = 0.f;
    // XCD-aware order: hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2); every XCD takes a CONTIGUOUS eighth of
    // a pass, so the output pixels that share an up-sampled source pixel meet in one L2 (PMC: the reduce's residual sum fetched 2.0 x its terms before)
    const long long lb = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);       // (the launcher makes gridDim.x a multiple of 8)
    for (long long i = lb * FUSE_BLOCK + threadIdx.x; i < items; i += (long long)gridDim.x * FUSE_BLOCK) {
        const long long px = i / cg;
        const int cv = (int)(i - px * cg) * 8;
        const long long img = px / hw;
        const int rem = (int)(px - img * hw);
        const int y = rem / p.w, x = rem - y * p.w;
        // acc = term 0, then += term 1, 2, 3 in order: ((t0 + t1) + t2) + t3, the association of torch's `y = y + t` chain -- with fp32 terms and the
        // fp32 output the result is that composition's, bit for bit
        float acc[8];
#pragma unroll
        for (int t = 0; t < MAX_TERMS; ++t) {
            if (t >= p.nt) break;
            const FuseTerm &T = p.t[t];
            const int s = T.shift, ws = p.w >> s;
            const long long sp = (img * (p.h >> s) + (y >> s)) * ws + (x >> s);
            const long long off = sp * T.pix + cv;
            float v[8];
            if (OUT == 2) {
                const h16x8 hv = *reinterpret_cast<const h16x8 *>(T.h16 + off);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)hv[k];
            } else if (T.hi) {
                const h16x8 hh = *reinterpret_cast<const h16x8 *>(T.hi + off), ll = *reinterpret_cast<const h16x8 *>(T.lo + off);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = ((float)hh[k] + (float)ll[k] * LO_INV) * sc[t];
            } else {
                const float4 a = *reinterpret_cast<const float4 *>(T.f32 + off), b = *reinterpret_cast<const float4 *>(T.f32 + off + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = t == 0 ? v[k] : (OUT == 2 ? (float)(_Float16)(acc[k] + v[k]) : acc[k] + v[k]);
        }
        const long long yo = px * p.y_pix + cv;
        if (OUT != 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (p.relu) acc[k] = acc[k] < 0.f ? 0.f : acc[k];      // (lets NaN through, like torch.relu)
            if (OUT == 1) {
                *reinterpret_cast<float4 *>(p.y32 + yo) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4 *>(p.y32 + yo + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            } else {
                h16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (_Float16)acc[k];       // (exact: every partial sum already is an f16 value)
                *reinterpret_cast<h16x8 *>(p.y16 + yo) = o;
            }
        } else {
            h16x8 oh, ol;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = acc[k];
                if (p.relu) v = v < 0.f ? 0.f : v;
                am = fmaxf(am, fabsf(v));
                _Float16 a, b;
                split_f32(v * inv, a, b);
                oh[k] = a; ol[k] = b;
            }
            *reinterpret_cast<h16x8 *>(p.yhi + yo) = oh;
            *reinterpret_cast<h16x8 *>(p.ylo + yo) = ol;
        }
    }
    if (OUT == 0 && p.state) record_amax(p.state + 1, am);
}

}  // namespace

static int fuse_entry(int n_terms, const void *const *hi_dev, const void *const *lo_dev, const float *const *f32_dev,
                      const float *const *scale_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                      void *y_hi_dev, void *y_lo_dev, float *y_f32_dev, int y_pix_stride, float *out_state_dev, int dynamic_batch, void *hip_stream,
                      const void *const *h16_dev = nullptr, void *y_f16_dev = nullptr)
{
    static const void *const no_ptrs[MAX_TERMS] = {nullptr, nullptr, nullptr, nullptr};
    static const float *const no_f32[MAX_TERMS] = {nullptr, nullptr, nullptr, nullptr};
    if (y_f32_dev || y_f16_dev) {                      // the fp32 / f16 forms: no plane tables
        if (!hi_dev) hi_dev = no_ptrs;
        if (!lo_dev) lo_dev = no_ptrs;
        if (!f32_dev) f32_dev = no_f32;
    }
    if (n_terms < 1 || n_terms > MAX_TERMS) return fail(TLK_EINVAL, "tlk_split_fuse_sum: 1..4 terms");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 != 0) return fail(TLK_EINVAL, "tlk_split_fuse_sum: bad shape (channels a multiple of 8)");
    if (!hi_dev || !lo_dev || !f32_dev || !shift) return fail(TLK_EINVAL, "tlk_split_fuse_sum: null term table");
    if (n == 0) return TLK_OK;
    if (!y_f32_dev && !y_f16_dev && (!y_hi_dev || !y_lo_dev)) return fail(TLK_EINVAL, "tlk_split_fuse_sum: no output");
    if (y_f16_dev && !h16_dev) return fail(TLK_EINVAL, "tlk_fuse_sum_f16: null term table");
    FuseArgs a{};
    a.nt = n_terms; a.n = n; a.h = h; a.w = w; a.c = c; a.relu = relu ? 1 : 0;
    a.y_pix = y_pix_stride > 0 ? y_pix_stride : c;
    a.yhi = (_Float16 *)y_hi_dev; a.ylo = (_Float16 *)y_lo_dev; a.y32 = y_f32_dev; a.y16 = (_Float16 *)y_f16_dev;
    a.state = (y_f32_dev || y_f16_dev) ? nullptr : out_state_dev;
    a.n_dyn = dynamic_batch ? conv_dynamic_batch() : nullptr;
    uintptr_t align = (uintptr_t)y_hi_dev | (uintptr_t)y_lo_dev | (uintptr_t)y_f32_dev | (uintptr_t)y_f16_dev;
    if (a.y_pix < c || a.y_pix % 8 != 0) return fail(TLK_EINVAL, "tlk_split_fuse_sum: the output's pixel stride must cover the channels and be a multiple of 8");
    for (int t = 0; t < n_terms; ++t) {
        FuseTerm &T = a.t[t];
        T.hi = (const _Float16 *)hi_dev[t]; T.lo = (const _Float16 *)lo_dev[t]; T.f32 = f32_dev[t];
        T.h16 = y_f16_dev ? (const _Float16 *)h16_dev[t] : nullptr;
        T.scale = scale_dev ? scale_dev[t] : nullptr;
        T.shift = shift[t];
        T.pix = pix_stride && pix_stride[t] > 0 ? pix_stride[t] : c;
        if (y_f16_dev) {
            if (!T.h16 || T.hi || T.lo || T.f32) return fail(TLK_EINVAL, "tlk_fuse_sum_f16: every term is one f16 tensor");
        } else if ((T.hi != nullptr) != (T.lo != nullptr) || (T.hi != nullptr) == (T.f32 != nullptr))
            return fail(TLK_EINVAL, "tlk_split_fuse_sum: a term is a (hi, lo) plane pair or one fp32 tensor");
        if (T.shift < 0 || T.shift > 8 || (h >> T.shift) << T.shift != h || (w >> T.shift) << T.shift != w)
            return fail(TLK_EINVAL, "tlk_split_fuse_sum: a term's resolution must divide the output's by a power of two");
        if (T.pix < c || T.pix % 8 != 0) return fail(TLK_EINVAL, "tlk_split_fuse_sum: a term's pixel stride must cover the channels and be a multiple of 8");
        align |= (uintptr_t)T.hi | (uintptr_t)T.lo | (uintptr_t)T.f32 | (uintptr_t)T.h16;
    }
    if (align & 15) return fail(TLK_EINVAL, "tlk_split_fuse_sum: every pointer must be 16-byte aligned");
    const long long items = (long long)n * h * w * (c / 8);
    long long blocks = (items + FUSE_BLOCK - 1) / FUSE_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    blocks = (blocks + 7) / 8 * 8;
    if (y_f16_dev) hipLaunchKernelGGL(split_fuse_sum_kernel<2>, dim3((unsigned)blocks), dim3(FUSE_BLOCK), 0, (hipStream_t)hip_stream, a);
    else if (y_f32_dev) hipLaunchKernelGGL(split_fuse_sum_kernel<1>, dim3((unsigned)blocks), dim3(FUSE_BLOCK), 0, (hipStream_t)hip_stream, a);
    else hipLaunchKernelGGL(split_fuse_sum_kernel<0>, dim3((unsigned)blocks), dim3(FUSE_BLOCK), 0, (hipStream_t)hip_stream, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_split_fuse_sum(int n_terms, const void *const *hi_dev, const void *const *lo_dev, const float *const *f32_dev,
                                  const float *const *scale_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                                  void *y_hi_dev, void *y_lo_dev, int y_pix_stride, float *out_state_dev, int dynamic_batch, void *hip_stream)
{
    return fuse_entry(n_terms, hi_dev, lo_dev, f32_dev, scale_dev, shift, pix_stride, n, h, w, c, relu, y_hi_dev, y_lo_dev, nullptr, y_pix_stride,
                      out_state_dev, dynamic_batch, hip_stream);
}

extern "C" int tlk_fuse_sum_f32(int n_terms, const float *const *x_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                                float *y_dev, int y_pix_stride, int dynamic_batch, void *hip_stream)
{
    if (!y_dev) return fail(TLK_EINVAL, "tlk_fuse_sum_f32: no output");
    return fuse_entry(n_terms, nullptr, nullptr, x_dev, nullptr, shift, pix_stride, n, h, w, c, relu, nullptr, nullptr, y_dev, y_pix_stride, nullptr,
                      dynamic_batch, hip_stream);
}

extern "C" int tlk_fuse_sum_f16(int n_terms, const void *const *x_dev, const int *shift, const int *pix_stride, int n, int h, int w, int c, int relu,
                                void *y_dev, int y_pix_stride, int dynamic_batch, void *hip_stream)
{
    if (!y_dev) return fail(TLK_EINVAL, "tlk_fuse_sum_f16: no output");
    return fuse_entry(n_terms, nullptr, nullptr, nullptr, nullptr, shift, pix_stride, n, h, w, c, relu, nullptr, nullptr, nullptr, y_pix_stride, nullptr,
                      dynamic_batch, hip_stream, x_dev, y_dev);
}
