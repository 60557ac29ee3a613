// tlk_dwconv.hip -- depthwise k x k convolution (stride 1, pad k/2) of a channels-last activation with bias + activation inside:
// the 5 x 5 depthwise halves of RTMPose's CSPNeXt blocks (mmdet DepthwiseSeparableConvModule; the reference runs the network behind
// tracklab/wrappers/pose_estimator/rtmlib_api.py:21-36, configs/modules/pose_estimator/rtmpose_rtmlib.yaml).  The library route (MIOpen's
// grouped-convolution kernels + a separate bias / SiLU pass) spent 33 of the 76 ms of the config-4 pose forward (2400 crops, f16) here,
// 18x above the HBM time of these layers.
//
// Every input element is read from HBM once and every output element written once:
//   * one lane owns FOUR channels of COLS neighbouring image columns and marches down the rows of its strip; lanes are laid out (column
//     group, channel group) with the channel group fastest, so every wave-wide load / store covers whole pixels of the NHWC row;
//   * the COLS + k - 1 input pixels a row contributes are that many loads per lane (neighbouring lanes re-read the same lines: L1 hits, not
//     HBM traffic), issued one row ahead of the arithmetic;
//   * the k x k weights of the lane's channels stay in registers as fp32 (k*k*4 VGPRs) for the whole march;
//   * a ring of k partial output rows (fp32) collects the contributions: input row r adds its k taps into output rows r-k+1 .. r, the
//     oldest of which is then complete -> + bias, activation, store.  No halo is re-read inside a strip; strips (blockIdx.y) exist only to
//     give small batches enough wavefronts and re-read k-1 rows each.
// What bounds it: VALU issue, not HBM -- r04 measured the time linear in the tap count (5 x 5 in f16 with one v_fma_mix_f32 per tap and
// channel: 200 instructions per 32 bytes moved, 2.0 TB/s; fp32 2.8 TB/s; profiles/r04_dwconv_spp.txt).  r06 halves the instruction count:
//   * the arithmetic is v_pk_fma_f32 -- two channels per instruction, each half one ordinary fused multiply-add, so the results are the
//     r04 kernel's bit for bit.  Spelled in assembly and in the PLAIN form only (no op_sel): the form r03 found unsafe beside MFMA kernels
//     of another stream is the one with op_sel on src1 (tools/audit_pk_f32.py checks the built library);
//   * f16 inputs are converted to fp32 once per loaded pixel (v_cvt_f32_f16) and a lane owns TWO columns, which share four of the six
//     pixels they read: 24 conversions + 100 packed FMAs per 8 outputs, against 200 v_fma_mix_f32 before;
//   * input rows are loaded two rows ahead with unconditional loads from clamped addresses (conditional loads made the compiler wait for
//     ALL outstanding loads at every row).
// Measured (2400 crops, 5 x 5 + SiLU, three boxes; profiles/r06_dwconv_spp.txt): f16 48 ch @ 64 x 48 643 -> 388-408 us = 3.5-3.65 TB/s
// (0.43-0.46 of 8 TB/s; ReLU 371 us = 0.48), fp32 990 -> 528-553 us = 5.1-5.4 TB/s (0.64-0.67).
// Each output element is ONE fmaf chain over (ky ascending, kx ascending), rows outside the image skipped, columns outside the image
// entering as zero terms, then + bias, activation (oracle/src/conv.c: orc_dwconv2d_nhwc_f32 walks the same chain; fp32 results are
// bit-identical, f16 results are that chain on the f16 operands rounded once, SiLU within the device exp's and reciprocal's error: ~3e-7
// relative).
#include "tlk_common.hpp"

#include <atomic>

using namespace tlk;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
constexpr int CH = 4;             // channels per lane

struct DwArgs {
    const void *x, *w;
    const float *bias;
    void *y;
    int N, H, W, C, CG, WG;       // CG = channel groups of CH channels, WG = column groups of COLS columns
    int x_pix, y_pix;             // elements between two pixels of x / y (>= C: a call may read / write a channel slice of a wider tensor)
    int rows_per_strip;
    long long items;              // N * WG * CG
};

std::atomic<int> g_cfg{0};        // tlk_dwconv_set_config (probes)

template <int ACT> __device__ __forceinline__ float act_f32(float v)
{
    if (ACT == ACT_RELU) return v < 0.f ? 0.f : v;          // (this form lets NaN through, like torch.relu: an overflow upstream must stay visible, r05)
    // SiLU = v * rcp(1 + exp(-v)): the device reciprocal (<= 1 ulp) instead of the IEEE division of the pointwise kernels -- a division is ten
    // VALU instructions per output element here, as many as the 25 taps of the convolution itself after r06; limits as for the division
    // (v -> -inf: -0, NaN stays NaN)
    if (ACT == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    return v;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// four channels of one pixel / weight tap as two fp32 pairs
struct Px { f2 lo, hi; };
template <typename T> struct Raw;
template <> struct Raw<float> {
    typedef f4 type;
    static __device__ __forceinline__ Px widen(f4 v) { return Px{f2{v[0], v[1]}, f2{v[2], v[3]}}; }
    static __device__ __forceinline__ f4 narrow(float a, float b, float c, float d) { return f4{a, b, c, d}; }
};
template <> struct Raw<_Float16> {
    typedef u2 type;              // four f16 as two packed registers
    static __device__ __forceinline__ Px widen(u2 v)
    {
        // (element copies first: hipcc 7.2 compiles __builtin_bit_cast(h2, v[1]) of a vector ELEMENT as a cast of element 0)
        const unsigned v0 = v[0], v1 = v[1];
        const h2 a = __builtin_bit_cast(h2, v0), b = __builtin_bit_cast(h2, v1);
        return Px{f2{(float)a[0], (float)a[1]}, f2{(float)b[0], (float)b[1]}};
    }
    static __device__ __forceinline__ u2 narrow(float a, float b, float c, float d)
    {
        const h2 lo = {(_Float16)a, (_Float16)b}, hi = {(_Float16)c, (_Float16)d};
        return u2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    }
};

// c += a * b on two channels: two independent fused multiply-adds (plain form, see the header)
__device__ __forceinline__ void pk_fma(f2 &c, f2 a, f2 b)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <typename T, int K, int ACT, int COLS, int PF>
__global__ void __launch_bounds__(256) dwconv_kernel(const DwArgs p)
{
    constexpr int PAD = K / 2;
    constexpr int NPX = COLS + K - 1;                         // input pixels of a row that the lane's columns read
    typedef typename Raw<T>::type raw_t;
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
    const int x0 = (int)(t % p.WG) * COLS;
    const long long n = t / p.WG;
    const int ys = blockIdx.y * p.rows_per_strip;
    const int rs = min(p.rows_per_strip, p.H - ys);           // output rows of this strip
    const T *xb = (const T *)p.x + (n * p.H * p.W) * p.x_pix + cg * CH;
    T *yb = (T *)p.y + (n * p.H * p.W) * p.y_pix + cg * CH;

    Px wr[K][K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wr[ky][kx] = Raw<T>::widen(*(const raw_t *)((const T *)p.w + (ky * K + kx) * p.C + cg * CH));
    float bs[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bs[c] = p.bias ? p.bias[cg * CH + c] : 0.f;

    // columns outside the image enter as zero terms: the loads go to a CLAMPED column (unconditional, straight-line -- the compiler counts them
    // in its s_waitcnt, which conditional loads defeat), and wavefronts that hold a lane at the left / right image border zero those
    // pixels after the load (a wave-uniform branch per row; interior wavefronts skip it)
    bool colok[NPX];
    int coff[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int xc = x0 + j - PAD;
        colok[j] = (unsigned)xc < (unsigned)p.W;
        coff[j] = min(max(xc, 0), p.W - 1) * p.x_pix;
    }
    const bool edge_l = __builtin_amdgcn_ballot_w64(!colok[0]) != 0;
    const bool edge_r = __builtin_amdgcn_ballot_w64(!colok[NPX - 1]) != 0;

    Px acc[K][COLS];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int j = 0; j < COLS; ++j) acc[s][j] = Px{f2{0.f, 0.f}, f2{0.f, 0.f}};

    const raw_t zero = {};
    auto load_row = [&](raw_t (&dst)[NPX], int r) {           // input row r of the strip = image row ys - PAD + r (clamped: rows outside are never used)
        const int iy = min(max(ys - PAD + r, 0), p.H - 1);
        const T *rowp = xb + (long long)iy * p.W * p.x_pix;
#pragma unroll
        for (int j = 0; j < NPX; ++j) dst[j] = *(const raw_t *)(rowp + coff[j]);
    };

    // input rows are loaded PF rows ahead of the arithmetic into a ring of PF + 1 register buffers; the row loop is unrolled over
    // lcm(PF + 1, K) phases so that both the buffer of a phase and its ring slots are compile-time constants (no register copies)
    constexpr int NB = PF + 1;
    constexpr int U = (NB % K == 0) ? NB : (K % NB == 0 ? K : NB * K);
    raw_t buf[NB][NPX];
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(buf[i], i);
    const int rtot = rs + 2 * PAD;                            // input rows this strip walks
    for (int r0 = 0; r0 < rtot; r0 += U) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            constexpr int KK = K;
            const int ph = q % KK;
            const int r = r0 + q;
            if (r >= rtot) break;
            load_row(buf[(q + PF) % NB], r + PF);
            const int iy = ys - PAD + r;
            if ((unsigned)iy < (unsigned)p.H) {
                if (edge_l) {
#pragma unroll
                    for (int j = 0; j < PAD; ++j) buf[q % NB][j] = colok[j] ? buf[q % NB][j] : zero;
                }
                if (edge_r) {
#pragma unroll
                    for (int j = PAD + 1; j < NPX; ++j) buf[q % NB][j] = colok[j] ? buf[q % NB][j] : zero;
                }
                Px cur[NPX];
#pragma unroll
                for (int j = 0; j < NPX; ++j) cur[j] = Raw<T>::widen(buf[q % NB][j]);
#pragma unroll
                for (int ky = K - 1; ky >= 0; --ky) {         // output row o = r - ky lives in ring slot (ph - ky) mod K
                    const int slot = (ph - ky + KK) % KK;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int j = 0; j < COLS; ++j) {
                            pk_fma(acc[slot][j].lo, cur[j + kx].lo, wr[ky][kx].lo);
                            pk_fma(acc[slot][j].hi, cur[j + kx].hi, wr[ky][kx].hi);
                        }
                }
            }
            // output row o = r - (K - 1) has now seen its last input row
            const int o = r - (K - 1);
            const int slot = (ph + 1) % KK;
            if (o >= 0 && o < rs) {
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    if (x0 + j < p.W) {
                        const Px &a = acc[slot][j];
                        *(raw_t *)(yb + ((long long)(ys + o) * p.W + x0 + j) * p.y_pix) =
                            Raw<T>::narrow(act_f32<ACT>(a.lo[0] + bs[0]), act_f32<ACT>(a.lo[1] + bs[1]), act_f32<ACT>(a.hi[0] + bs[2]), act_f32<ACT>(a.hi[1] + bs[3]));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < COLS; ++j) acc[slot][j] = Px{f2{0.f, 0.f}, f2{0.f, 0.f}};
        }
    }
}

template <typename T, int K, int COLS, int PF> int launch_p(const DwArgs &a, int act, dim3 grid, hipStream_t st)
{
    if (act == 0) hipLaunchKernelGGL((dwconv_kernel<T, K, ACT_NONE, COLS, PF>), grid, dim3(256), 0, st, a);
    else if (act == 1) hipLaunchKernelGGL((dwconv_kernel<T, K, ACT_RELU, COLS, PF>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwconv_kernel<T, K, ACT_SILU, COLS, PF>), grid, dim3(256), 0, st, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
template <typename T, int K, int COLS> int launch_c(const DwArgs &a, int act, int pf, dim3 grid, hipStream_t st)
{
    return pf == 1 ? launch_p<T, K, COLS, 1>(a, act, grid, st) : launch_p<T, K, COLS, 2>(a, act, grid, st);
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
    const int cfg = g_cfg.load(std::memory_order_relaxed);
    const int cols = cfg ? 1 + ((cfg - 1) & 1) : 2;
    const int pf = cfg ? 1 + ((cfg - 1) >> 1) : 2;
    DwArgs a;
    a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.y = y_dev;
    a.N = n; a.H = h; a.W = w; a.C = c; a.CG = c / CH; a.WG = (w + cols - 1) / cols; a.x_pix = xp; a.y_pix = yp;
    a.items = (long long)n * a.WG * a.CG;
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
    if (dtype == TLK_F16) {
        if (cols == 2) return k == 5 ? launch_c<_Float16, 5, 2>(a, act_kind, pf, grid, st) : launch_c<_Float16, 3, 2>(a, act_kind, pf, grid, st);
        return k == 5 ? launch_c<_Float16, 5, 1>(a, act_kind, pf, grid, st) : launch_c<_Float16, 3, 1>(a, act_kind, pf, grid, st);
    }
    if (cols == 2) return k == 5 ? launch_c<float, 5, 2>(a, act_kind, pf, grid, st) : launch_c<float, 3, 2>(a, act_kind, pf, grid, st);
    return k == 5 ? launch_c<float, 5, 1>(a, act_kind, pf, grid, st) : launch_c<float, 3, 1>(a, act_kind, pf, grid, st);
}

// probes: 0 = the default (2 columns per lane, rows loaded 2 ahead: measured best or equal on every RTMPose shape in both element types,
// profiles/r06_dwconv_spp.txt), 1..4 = (columns per lane, rows ahead) = (1, 1), (2, 1), (1, 2), (2, 2)
extern "C" int tlk_dwconv_set_config(int cfg)
{
    if (cfg < 0 || cfg > 4) return fail(TLK_EINVAL, "tlk_dwconv_set_config: 0 (default) or 1..4");
    g_cfg.store(cfg, std::memory_order_relaxed);
    return TLK_OK;
}
