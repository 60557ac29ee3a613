// tlk_conv_stem.hip -- the RGB stem convolutions (Cin = 3: ResNet-50's 7 x 7 stride 2, RTMPose's 3 x 3 stride 2) at fp32, direct form.
//
// Through the implicit-GEMM kernel (tlk_conv.hip) the stem was the slowest layer of the fp32 step by far: 14.2 ms for the ReID network's
// 2400 crops = 39 TFLOP/s (profiles/r04_conv16_shapes.txt).  That kernel gathers 16 bytes per tap, so the image had to be padded to four
// channels first (one more pass over it) and K = 7 * 7 * 4 = 196 ran as seven 32-wide steps, a quarter of whose MFMAs multiplied the zero
// channel or the tail beyond K; seven steps do not amortise a tile's prologue and its 32 KB epilogue through LDS either.
//
// Here: no padding of the image, no K tail, no LDS round trip for the output.
//   * a workgroup (4 wavefronts) owns a strip of TR = 4 output rows x 64 output columns of one image; the input patch the strip reads --
//     ((TR - 1) * S + KH) rows x (63 * S + KW) pixels x 3 floats, zeros outside the image -- is staged in LDS once (coalesced dword loads);
//     the weights sit in LDS as [tap][channel][cout] with one all-zero tap appended;
//   * a wavefront computes one output row: 64 pixels x Cout = 2 x NCO tiles of v_mfma_f32_32x32x2_f32.  Operands are single floats read
//     straight from LDS (ds_read_b32): lane l supplies pixel (l & 31) of a tile and tap (t0 + (l >> 5)) of the tap pair being multiplied, so
//     ONE MFMA folds (tap t0, channel c) and (tap t0 + 1, channel c) into the accumulators.  At 64 cycles per MFMA two 4-byte LDS reads per
//     MFMA are noise: the kernel is bound by the matrix pipe, 75 MFMAs per tile (25 tap pairs x 3 channels) against 112 before;
//   * summation order = the implicit-GEMM kernel's on the 4-channel-padded problem, minus its zero terms: k = (tap, channel) pairs
//     (2j, c), (2j + 1, c) for c = 0, 1, 2, j ascending -- an fmaf chain (the f32 MFMA is exactly that), so the result is BIT-IDENTICAL to
//     oracle/src/conv.c on the padded input (a zero product leaves an fmaf chain unchanged): tests/test_gpu_conv.py compares them;
//   * epilogue: bias + ReLU / SiLU in registers, then every 32 x 32 tile passes through a private LDS tile of its wavefront (the patch area,
//     dead by then) so that a lane stores 16 contiguous bytes and a pixel's 32 channels leave as one 128-byte run; a workgroup stages the
//     weights once for two row strips.
// Honours tlk_conv_set_dynamic_batch (images beyond the live count are neither read nor written).
#include "tlk_common.hpp"

using namespace tlk;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StemArgs {
    const float *x, *w, *bias;
    float *y;
    int N, H, W, Ho, Wo, Cout, pad, x_pix, y_pix, act;
    int col_tiles, row_tiles;
    const int *n_dyn;
};

constexpr int TR = 4, TC = 64;          // output rows (one per wavefront) x output columns of a workgroup

constexpr int RSTRIPS = 2;              // row strips per workgroup (the weights are staged once for both)
constexpr int OST = 36;                 // floats per pixel row of a wavefront's 32 x 32 output staging tile (32 + 4: conflict-free float4 reads)

template <int KH, int KW, int S, int NCO>
__global__ void __launch_bounds__(256) conv_stem3_kernel(const StemArgs p)
{
    constexpr int TAPS = KH * KW, PAIRS = (TAPS + 1) / 2;
    constexpr int PR = (TR - 1) * S + KH, PW = (TC - 1) * S + KW;        // patch rows x pixels
    constexpr int PWF = PW * 3 + 1;                                      // floats per patch row (+ 1: odd stride, rows start on different banks)
    constexpr int CO = NCO * 32;
    constexpr int PATCH_FLOATS = ((PR * PWF > 4 * 32 * OST ? PR * PWF : 4 * 32 * OST) + 3) & ~3;      // the patch area doubles as the four wavefronts' output staging tiles
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *patch = lds;                                                  // [PR][PWF]; after the MFMAs: 4 x [32][OST] output staging
    float *wl = lds + PATCH_FLOATS;                                      // [2 * PAIRS][3][CO], taps >= TAPS all zero
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (image, pair of row strips, column strip)
    int b = blockIdx.x;
    const int ct = b % p.col_tiles; b /= p.col_tiles;
    const int rt2 = b % p.row_tiles; const int n = b / p.row_tiles;
    if (p.n_dyn && n >= p.n_dyn[0]) return;
    const int wo0 = ct * TC, wi0 = wo0 * S - p.pad;
    // ---- weights ([cout][tap][3] in global memory = torch's channels_last (Cout, 3, KH, KW)), once per workgroup
    // (every load is issued before the first store: ONE memory round trip, not one per element -- the first version of this kernel spent more
    //  time in its two staging loops, a dependent load -> store chain of ~60 links per workgroup, than in its 600 MFMAs)
    {
        constexpr int NW_ = 2 * PAIRS * 3 * CO, WI = (NW_ + 255) / 256;
        float wv[WI];
#pragma unroll
        for (int k = 0; k < WI; ++k) {
            const int e = k * 256 + tid;
            const int co = e % CO, c = (e / CO) % 3, t = e / (3 * CO);
            wv[k] = (e < NW_ && t < TAPS && co < p.Cout) ? p.w[((size_t)co * TAPS + t) * 3 + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < WI; ++k) { const int e = k * 256 + tid; if (e < NW_) wl[e] = wv[k]; }
    }
    const float *img = p.x + (size_t)n * p.H * p.W * p.x_pix;
    const int half = lane >> 5, pl = lane & 31;
    float bias[NCO];
#pragma unroll
    for (int jj = 0; jj < NCO; ++jj) bias[jj] = (p.bias && jj * 32 + pl < p.Cout) ? p.bias[jj * 32 + pl] : 0.f;
    for (int strip = 0; strip < RSTRIPS; ++strip) {
        const int ho0 = (rt2 * RSTRIPS + strip) * TR;
        if (ho0 >= p.Ho) break;                                          // uniform
        const int hi0 = ho0 * S - p.pad;
        if (strip) __syncthreads();                                      // the staging tiles of the previous strip have been read
        // ---- input patch of the strip: zeros outside the image
        {
            constexpr int QI = (PW * 3 + 255) / 256;
            float pv[PR][QI];
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int hi = hi0 + r;
                const bool row_ok = (unsigned)hi < (unsigned)p.H;
                const float *src = img + (size_t)(row_ok ? hi : 0) * p.W * p.x_pix;
#pragma unroll
                for (int k = 0; k < QI; ++k) {
                    const int q = k * 256 + tid;
                    const int px = q / 3, c = q - px * 3, wi = wi0 + px;
                    pv[r][k] = (q < PW * 3 && row_ok && (unsigned)wi < (unsigned)p.W) ? src[(size_t)wi * p.x_pix + c] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < PR; ++r)
#pragma unroll
                for (int k = 0; k < QI; ++k) { const int q = k * 256 + tid; if (q < PW * 3) patch[r * PWF + q] = pv[r][k]; }
        }
        __syncthreads();
        const int ho = ho0 + wave;
        const bool live = ho < p.Ho;                                     // wave-uniform
        f32x16 acc[2][NCO];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCO; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (live) {
            // this lane's pixel of tile i is output column wo0 + 32 i + pl: patch pixel (32 i + pl) * S + kw, patch row wave * S + kh
            const float *arow = patch + (wave * S) * PWF + (pl * S) * 3;
            const float *brow = wl + pl;
#pragma unroll
            for (int j = 0; j < PAIRS; ++j) {
                // tap of this half: t = 2 j + half
                const int t_lo = 2 * j, t_hi = 2 * j + 1;
                const int off_lo = (t_lo / KW) * PWF + (t_lo % KW) * 3;
                const int off_hi = t_hi < TAPS ? (t_hi / KW) * PWF + (t_hi % KW) * 3 : off_lo;     // (the appended tap: any finite value times a zero weight)
                const int aoff = half ? off_hi : off_lo;
                const int boff = (2 * j + half) * 3 * CO;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float av[2], bv[NCO];
#pragma unroll
                    for (int i = 0; i < 2; ++i) av[i] = arow[aoff + i * 32 * S * 3 + c];
#pragma unroll
                    for (int jj = 0; jj < NCO; ++jj) bv[jj] = brow[boff + c * CO + jj * 32];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < NCO; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jj], acc[i][jj], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                 // every wavefront is done with the patch: it becomes the staging area
        if (!live) continue;
        // ---- epilogue.  C/D map: column (= cout) = lane & 31, row (= pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Each 32 x 32 tile goes
        // through this wavefront's staging tile so that a lane then stores 16 contiguous bytes: 8 lanes cover the 32 channels of a pixel (128-byte
        // runs), four store instructions per tile instead of sixteen 4-byte ones.
        float *stg = patch + wave * (32 * OST);
        float *yrow = p.y + ((size_t)n * p.Ho + ho) * p.Wo * (size_t)p.y_pix;
        const bool vec_ok = ((p.y_pix | p.Cout) & 3) == 0 && ((uintptr_t)p.y & 15) == 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NCO; ++jj) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][jj][r] + bias[jj];
                    if (p.act == ACT_RELU) v = v < 0.f ? 0.f : v;
                    else if (p.act == ACT_SILU) v = v / (1.f + __expf(-v));
                    stg[((r & 3) + 8 * (r >> 2) + 4 * half) * OST + pl] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                    const int wo = wo0 + i * 32 + row, co = jj * 32 + c4;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + row * OST + c4);
                    if (wo >= p.Wo || co >= p.Cout) continue;
                    float *o = yrow + (size_t)wo * p.y_pix + co;
                    if (vec_ok) *reinterpret_cast<float4 *>(o) = v;
                    else { o[0] = v.x; if (co + 1 < p.Cout) o[1] = v.y; if (co + 2 < p.Cout) o[2] = v.z; if (co + 3 < p.Cout) o[3] = v.w; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
    }
}

template <int KH, int KW, int S, int NCO> int launch_stem(StemArgs &a, hipStream_t st)
{
    constexpr int TAPS = KH * KW, PAIRS = (TAPS + 1) / 2;
    constexpr int PR = (TR - 1) * S + KH, PW = (TC - 1) * S + KW, PWF = PW * 3 + 1;
    constexpr int PATCH_FLOATS = ((PR * PWF > 4 * 32 * OST ? PR * PWF : 4 * 32 * OST) + 3) & ~3;
    constexpr size_t LDS_BYTES = ((size_t)PATCH_FLOATS + (size_t)2 * PAIRS * 3 * NCO * 32) * sizeof(float);
    static_assert(LDS_BYTES <= 160 * 1024, "patch + weights must fit the CU's LDS");
    a.col_tiles = (a.Wo + TC - 1) / TC;
    a.row_tiles = (a.Ho + TR * RSTRIPS - 1) / (TR * RSTRIPS);
    const long long wgs = (long long)a.N * a.row_tiles * a.col_tiles;
    if (wgs > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: too many output strips for one launch");
    auto kern = conv_stem3_kernel<KH, KW, S, NCO>;
    static bool attr_set = false;
    if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), LDS_BYTES, st, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

}  // namespace

namespace tlk {

// 1 = "not a stem this file has a kernel for" (the caller answers as before), else the launch status
int conv_stem3_f32(const float *x, const float *w, const float *bias, float *y, int n, int h, int wd, int cout, int kh, int kw, int stride, int pad, int act,
                   int x_pix, int y_pix, hipStream_t st)
{
    if (cout > 64) return 1;
    StemArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.N = n; a.H = h; a.W = wd; a.Cout = cout; a.pad = pad; a.act = act;
    a.Ho = (h + 2 * pad - kh) / stride + 1; a.Wo = (wd + 2 * pad - kw) / stride + 1;
    a.x_pix = x_pix; a.y_pix = y_pix;
    a.n_dyn = conv_dynamic_batch();
    const int nco = cout > 32 ? 2 : 1;
    if (kh == 7 && kw == 7 && stride == 2) return nco == 2 ? launch_stem<7, 7, 2, 2>(a, st) : launch_stem<7, 7, 2, 1>(a, st);
    if (kh == 3 && kw == 3 && stride == 2) return nco == 2 ? launch_stem<3, 3, 2, 2>(a, st) : launch_stem<3, 3, 2, 1>(a, st);
    return 1;
}

}  // namespace tlk
