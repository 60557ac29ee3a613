// tlk_heads.hip -- r06: the prediction heads of the backbones as ONE launch each, so that a config-3 step has no library convolution and no
// element-wise torch glue left behind its two networks (the detector head was 9 MIOpen / CK convolutions with 1 / 4 / 1
// output channels + sigmoid / cat / flatten / permute / cast passes, ~40 launches; the ReID head a 6-channel convolution + softmax + bmm +
// division + amax + two index_select gathers + the non-finite check, ~14 launches).
//
//   tlk_yolox_head_nhwc     rtmlib YOLOX's decoupled head outputs (third-party ONNX graph behind tracklab/wrappers/bbox_detector/rtmlib_api.py:21,30):
//                           per level reg (4) / obj (1) 1 x 1 predictions of the regression branch and cls (C) of the classification branch,
//                           sigmoid on obj / cls, written straight into the (B, A, 5 + C) fp32 tensor tlk_yolox_decode_nms reads.
//   tlk_reid_part_head      the part-based ReID head (torchreid BPBReID / KPR behind tracklab/wrappers/reid/kpreid_api.py:147-182): pixel-wise part
//                           classifier (K x D 1 x 1), softmax over the parts, attention-weighted average of the feature map per part, visibility
//                           = strongest attention of the part; rows are written in the tracker's (frame, detection slot) layout from the DENSE
//                           batch (crop_slot_bases), padding slots zero-filled, "an embedding is not finite" ORed into a device flag.
//
// Both stage 64 pixels x C channels of the NHWC map into LDS with coalesced 16-byte loads (converted to fp32), then lane = pixel walks its row.
// Arithmetic contract (oracle/src/heads.c restates it loop for loop): every dot product is ONE fmaf chain over the channels ASCENDING starting
// from 0, then + bias; sigmoid(v) = 1 / (1 + exp(-v)); softmax_k = exp(l_k - max) / sum over k ascending; the pooled sums are fmaf chains over
// the pixels ascending.  Bit-exact against the oracle up to the device's exp (2 ulp; the tests allow 2e-6 relative where an exp is involved).
#include <hip/hip_fp16.h>

#include "tlk_common.hpp"

using namespace tlk;

namespace {

constexpr int HP = 64;                   // pixels per chunk: one per lane
constexpr int MAXLV = 4;
constexpr int MAXK = 8;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// rows [p0, p0 + np) x C channels of an NHWC map (pixel stride `pix` elements) -> xs[row][C + 4] fp32; C % VN == 0, all threads of the block call
template <typename T> __device__ __forceinline__ void stage_rows(const T *__restrict__ src, long long p0, int np, int C, int pix, float *xs)
{
    constexpr int VN = sizeof(T) == 2 ? 8 : 4;
    const int cv = C / VN, ld = C + 4;
    for (int idx = threadIdx.x; idx < np * cv; idx += BLOCK) {
        const int row = idx / cv, c = (idx - row * cv) * VN;
        const T *g = src + (p0 + row) * (long long)pix + c;
        float *d = xs + row * ld + c;
        if constexpr (sizeof(T) == 2) {
            const h16x8 v = *reinterpret_cast<const h16x8 *>(g);
            f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
            *reinterpret_cast<f32x4 *>(d) = a;
            *reinterpret_cast<f32x4 *>(d + 4) = b;
        } else {
            *reinterpret_cast<f32x4 *>(d) = *reinterpret_cast<const f32x4 *>(g);
        }
    }
}

// one fmaf chain over c = 0 .. C-1 of xs_row[c] * w[c] (w wave-uniform: scalar loads)
__device__ __forceinline__ float dot_chain(const float *__restrict__ xr, const float *__restrict__ w, int C)
{
    float acc = 0.f;
    for (int c = 0; c < C; c += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4 *>(xr + c);
        acc = __builtin_fmaf(x[0], w[c], acc);
        acc = __builtin_fmaf(x[1], w[c + 1], acc);
        acc = __builtin_fmaf(x[2], w[c + 2], acc);
        acc = __builtin_fmaf(x[3], w[c + 3], acc);
    }
    return acc;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

struct YHLevel {
    const void *cls, *reg;               // (B, hw, C) NHWC maps of the classification / regression branch
    const float *w, *b;                  // [(5 + ncls)][C] rows: reg 0..3, obj 4, cls 5..; [(5 + ncls)] biases
    int hw, a_off, cls_pix, reg_pix, chunk0;
};
struct YHArgs {
    YHLevel lv[MAXLV];
    float *out;                          // (B, A, 5 + ncls)
    int levels, C, ncls, A, chunks;      // chunks = per image, all levels
};

template <typename T> __global__ void __launch_bounds__(BLOCK) yolox_head_kernel(const YHArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NO = 5 + p.ncls, ld = p.C + 4;
    float *xs = lds, *so = lds + HP * ld;                           // so[HP][NO]
    const int b = blockIdx.x / p.chunks, ch = blockIdx.x - b * p.chunks;
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAXLV; ++i) l += (i < p.levels && ch >= p.lv[i].chunk0) ? 1 : 0;
    const YHLevel &L = p.lv[l];
    const int p0 = (ch - L.chunk0) * HP, np = min(HP, L.hw - p0);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *xr = xs + lane * ld;
    // regression branch: outputs 0..4
    stage_rows(reinterpret_cast<const T *>(L.reg), (long long)b * L.hw + p0, np, p.C, L.reg_pix, xs);
    __syncthreads();
    if (lane < np)
        for (int o = wv; o < 5; o += NWAVES) {
            const float v = dot_chain(xr, L.w + (long long)o * p.C, p.C) + L.b[o];
            so[lane * NO + o] = o == 4 ? sigmoidf_(v) : v;
        }
    __syncthreads();
    // classification branch: outputs 5..
    stage_rows(reinterpret_cast<const T *>(L.cls), (long long)b * L.hw + p0, np, p.C, L.cls_pix, xs);
    __syncthreads();
    if (lane < np)
        for (int o = 5 + wv; o < NO; o += NWAVES) so[lane * NO + o] = sigmoidf_(dot_chain(xr, L.w + (long long)o * p.C, p.C) + L.b[o]);
    __syncthreads();
    float *g = p.out + ((long long)b * p.A + L.a_off + p0) * NO;    // the chunk's rows are contiguous in the output
    for (int i = threadIdx.x; i < np * NO; i += BLOCK) g[i] = so[i];
}

struct PHArgs {
    const void *f;                       // (rows_dense, hw, D) NHWC feature map
    const float *wc, *bc;                // [K][D], [K]
    const int *counts, *slot_base;       // per frame (nullable): live detections, first dense row
    float *emb;                          // (rows, K, D)
    unsigned char *vis;                  // (rows, K)
    unsigned char *flag;                 // nullable: set to 1 when a live embedding is not finite
    int hw, D, K, f_pix, maxd;
    float vis_thr;
};

template <typename T> __global__ void __launch_bounds__(BLOCK) reid_part_head_kernel(const PHArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ld = p.D + 4, K = p.K, D = p.D;
    float *xs = lds, *lg = lds + HP * ld, *att = lg + HP * MAXK, *sden = att + HP * MAXK;      // lg[HP][MAXK], att[HP][MAXK], sden[MAXK]
    const int r = blockIdx.x, tid = threadIdx.x;
    long long src = r;
    if (p.counts) {
        const int b = r / p.maxd, j = r - b * p.maxd;
        if (j >= p.counts[b]) {                                     // padding slot: defined content (zeros), never garbage
            for (int i = tid; i < K * D; i += BLOCK) p.emb[(long long)r * K * D + i] = 0.f;
            if (tid < K) p.vis[(long long)r * K + tid] = 0;
            return;
        }
        if (p.slot_base) src = p.slot_base[b] + j;
    }
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const T *fm = reinterpret_cast<const T *>(p.f) + src * (long long)p.hw * p.f_pix;
    float e[MAXK][2];                                               // pooled sums of channels tid and tid + 256
#pragma unroll
    for (int k = 0; k < MAXK; ++k) e[k][0] = e[k][1] = 0.f;
    float den = 0.f, mx = 0.f;                                      // part tid < K: sum and maximum of its attention over the pixels
    for (int p0 = 0; p0 < p.hw; p0 += HP) {
        const int np = min(HP, p.hw - p0);
        if (p0) __syncthreads();                                    // everybody is done with the previous chunk
        stage_rows(fm, p0, np, D, p.f_pix, xs);
        __syncthreads();
        if (lane < np)
            for (int k = wv; k < K; k += NWAVES) lg[lane * MAXK + k] = dot_chain(xs + lane * ld, p.wc + (long long)k * D, D) + p.bc[k];
        __syncthreads();
        if (tid < np) {                                             // softmax over the parts of pixel tid
            float m = lg[tid * MAXK];
            for (int k = 1; k < K; ++k) m = fmaxf(m, lg[tid * MAXK + k]);
            float ex[MAXK], s = 0.f;
#pragma unroll
            for (int k = 0; k < MAXK; ++k) {
                ex[k] = k < K ? expf(lg[tid * MAXK + k] - m) : 0.f;
                if (k < K) s += ex[k];
            }
#pragma unroll
            for (int k = 0; k < MAXK; ++k) att[tid * MAXK + k] = k < K ? ex[k] / s : 0.f;
        }
        __syncthreads();
        if (tid < K)
            for (int q = 0; q < np; ++q) {
                const float a = att[q * MAXK + tid];
                den += a;
                mx = (a != a || a > mx) ? a : mx;                   // (NaN-propagating, like torch.amax)
            }
#pragma unroll
        for (int di = 0; di < 2; ++di) {
            const int d = tid + di * BLOCK;
            if (d < D)
                for (int q = 0; q < np; ++q) {
                    const float x = xs[q * ld + d];
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(att + q * MAXK), a1 = *reinterpret_cast<const f32x4 *>(att + q * MAXK + 4);
                    e[0][di] = __builtin_fmaf(a0[0], x, e[0][di]); e[1][di] = __builtin_fmaf(a0[1], x, e[1][di]);
                    e[2][di] = __builtin_fmaf(a0[2], x, e[2][di]); e[3][di] = __builtin_fmaf(a0[3], x, e[3][di]);
                    e[4][di] = __builtin_fmaf(a1[0], x, e[4][di]); e[5][di] = __builtin_fmaf(a1[1], x, e[5][di]);
                    e[6][di] = __builtin_fmaf(a1[2], x, e[6][di]); e[7][di] = __builtin_fmaf(a1[3], x, e[7][di]);
                }
        }
    }
    if (tid < K) {
        sden[tid] = den < 1e-6f ? 1e-6f : den;                     // clamp_min(1e-6) (a NaN sum stays NaN: the comparison is false)
        p.vis[(long long)r * K + tid] = (tid == 0 || mx > p.vis_thr) ? 1 : 0;
    }
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (int di = 0; di < 2; ++di) {
        const int d = tid + di * BLOCK;
        if (d < D)
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < K) {
                    const float v = e[k][di] / sden[k];
                    p.emb[((long long)r * K + k) * D + d] = v;
                    bad |= !(fabsf(v) <= 3.4028234664e38f);
                }
    }
    if (p.flag && __any(bad) && lane == 0) *p.flag = 1;
}

}  // namespace

extern "C" int tlk_yolox_head_nhwc(const void *const *cls_feat_dev, const void *const *reg_feat_dev, const int *hw, const int *cls_pix_stride,
                                   const int *reg_pix_stride, const float *const *w_dev, const float *const *b_dev, int levels, int batch, int channels,
                                   int num_classes, int dtype, float *out_dev, void *hip_stream)
{
    if (levels < 1 || levels > MAXLV) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: 1..4 levels");
    if (dtype != TLK_F32 && dtype != TLK_F16) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: dtype must be TLK_F32 or TLK_F16");
    const int vn = dtype == TLK_F16 ? 8 : 4;
    if (batch < 0 || channels <= 0 || channels % vn != 0 || channels > 1024 || num_classes < 1 || num_classes > 256)
        return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: channels a positive multiple of 16 bytes (<= 1024), 1..256 classes");
    if (!cls_feat_dev || !reg_feat_dev || !hw || !w_dev || !b_dev || !out_dev) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: null pointer");
    YHArgs a;
    memset(&a, 0, sizeof(a));
    int A = 0, chunks = 0;
    for (int l = 0; l < levels; ++l) {
        if (hw[l] <= 0 || !cls_feat_dev[l] || !reg_feat_dev[l] || !w_dev[l] || !b_dev[l]) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: empty level / null pointer");
        const int cp = cls_pix_stride && cls_pix_stride[l] ? cls_pix_stride[l] : channels, rp = reg_pix_stride && reg_pix_stride[l] ? reg_pix_stride[l] : channels;
        if (cp < channels || rp < channels || cp % vn != 0 || rp % vn != 0) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: pixel strides must be >= channels and multiples of 16 bytes");
        if (((uintptr_t)cls_feat_dev[l] | (uintptr_t)reg_feat_dev[l]) & 15) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: feature maps must be 16-byte aligned");
        a.lv[l].cls = cls_feat_dev[l]; a.lv[l].reg = reg_feat_dev[l]; a.lv[l].w = w_dev[l]; a.lv[l].b = b_dev[l];
        a.lv[l].hw = hw[l]; a.lv[l].a_off = A; a.lv[l].cls_pix = cp; a.lv[l].reg_pix = rp; a.lv[l].chunk0 = chunks;
        A += hw[l]; chunks += (hw[l] + HP - 1) / HP;
    }
    if (batch == 0) return TLK_OK;
    a.out = out_dev; a.levels = levels; a.C = channels; a.ncls = num_classes; a.A = A; a.chunks = chunks;
    const long long blocks = (long long)batch * chunks;
    if (blocks > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: more than 2^31 - 1 workgroups");
    const size_t smem = ((size_t)HP * (channels + 4) + (size_t)HP * (5 + num_classes)) * sizeof(float);
    if (smem > 160 * 1024) return fail(TLK_EINVAL, "tlk_yolox_head_nhwc: channels / classes beyond the LDS tile");
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == TLK_F16) {
        static bool set16 = false;
        if (!set16) { TLK_HIP(hipFuncSetAttribute((const void *)yolox_head_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set16 = true; }
        hipLaunchKernelGGL(yolox_head_kernel<_Float16>, dim3((unsigned)blocks), dim3(BLOCK), smem, st, a);
    } else {
        static bool set32 = false;
        if (!set32) { TLK_HIP(hipFuncSetAttribute((const void *)yolox_head_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set32 = true; }
        hipLaunchKernelGGL(yolox_head_kernel<float>, dim3((unsigned)blocks), dim3(BLOCK), smem, st, a);
    }
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_reid_part_head(const void *feat_dev, int feat_pix_stride, int dtype, int hw, int dim, int parts, const float *w_dev, const float *b_dev,
                                  const int32_t *counts_dev, const int32_t *slot_base_dev, int rows, int max_dets, float vis_threshold,
                                  float *emb_dev, unsigned char *vis_dev, unsigned char *nonfinite_flag_dev, void *hip_stream)
{
    if (dtype != TLK_F32 && dtype != TLK_F16) return fail(TLK_EINVAL, "tlk_reid_part_head: dtype must be TLK_F32 or TLK_F16");
    const int vn = dtype == TLK_F16 ? 8 : 4;
    if (rows < 0 || hw <= 0 || dim <= 0 || dim % vn != 0 || dim > 2 * BLOCK || parts < 1 || parts > MAXK)
        return fail(TLK_EINVAL, "tlk_reid_part_head: dim a positive multiple of 16 bytes (<= 512), 1..8 parts");
    const int fp = feat_pix_stride ? feat_pix_stride : dim;
    if (fp < dim || fp % vn != 0) return fail(TLK_EINVAL, "tlk_reid_part_head: pixel stride must be >= dim and a multiple of 16 bytes");
    if (!feat_dev || !w_dev || !b_dev || !emb_dev || !vis_dev) return fail(TLK_EINVAL, "tlk_reid_part_head: null pointer");
    if ((uintptr_t)feat_dev & 15) return fail(TLK_EINVAL, "tlk_reid_part_head: the feature map must be 16-byte aligned");
    if (counts_dev && (max_dets <= 0 || rows % max_dets != 0)) return fail(TLK_EINVAL, "tlk_reid_part_head: rows must be frames x max_dets when counts are given");
    if (slot_base_dev && !counts_dev) return fail(TLK_EINVAL, "tlk_reid_part_head: slot bases need the per-frame counts");
    if (rows == 0) return TLK_OK;
    PHArgs a;
    a.f = feat_dev; a.wc = w_dev; a.bc = b_dev; a.counts = counts_dev; a.slot_base = slot_base_dev; a.emb = emb_dev; a.vis = vis_dev; a.flag = nonfinite_flag_dev;
    a.hw = hw; a.D = dim; a.K = parts; a.f_pix = fp; a.maxd = max_dets > 0 ? max_dets : 1; a.vis_thr = vis_threshold;
    const size_t smem = ((size_t)HP * (dim + 4) + (size_t)2 * HP * MAXK + MAXK) * sizeof(float);
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == TLK_F16) {
        static bool set16 = false;
        if (!set16) { TLK_HIP(hipFuncSetAttribute((const void *)reid_part_head_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set16 = true; }
        hipLaunchKernelGGL(reid_part_head_kernel<_Float16>, dim3((unsigned)rows), dim3(BLOCK), smem, st, a);
    } else {
        static bool set32 = false;
        if (!set32) { TLK_HIP(hipFuncSetAttribute((const void *)reid_part_head_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set32 = true; }
        hipLaunchKernelGGL(reid_part_head_kernel<float>, dim3((unsigned)rows), dim3(BLOCK), smem, st, a);
    }
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
