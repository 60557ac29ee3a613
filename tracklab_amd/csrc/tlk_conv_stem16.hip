// tlk_conv_stem16.hip -- the RGB stem convolutions in f16 (Cin = 3: ResNet-50's 7 x 7 stride 2, the 3 x 3 stride 2 stems of HRNet / RTMPose), direct
// form on v_mfma_f32_32x32x16_f16, with the ResNet stem's 3 x 3 / stride 2 / pad 1 max-pool FUSED behind it.
//
// Why.  r05's steady-state trace of the f16 step (profiles/r05_config3_f16_rocprof.md): with every other convolution on libtlk's own kernels, the
// ReID stem was what remained on the libraries -- MIOpen's implicit GEMM 3.34 ms + its bias pass 0.88 ms + tlk_bias_act 1.61 ms + the max-pool
// 0.87 ms = 6.7 ms of a 56 ms step, for a layer whose tensors are 0.7 GB in and (after the pool) 0.94 GB out (this kernel: 1.67 ms).  The 3.8 GB convolution output was
// written, read and re-written three times.  Here the 192 x 64 x 64 map never exists in memory.
//
//   * a workgroup (4 wavefronts) owns strips of PT = 4 pooled rows x the full width of one image (the convolution map is at most 64 wide: one
//     column strip); the input patch a strip needs -- 8 S + KH rows x (63 S + 4 G) pixels, zeros outside the image -- is staged in LDS once with
//     every pixel padded to 4 halfs (8 bytes);
//   * k order = (kh | kw, c4): one image row of the window is 4 G taps x 4 channels = G slices of 16 -- 7 x 7: kw 0..7 (the 8th tap and the
//     4th channel meet zero weights), two v_mfma_f32_32x32x16_f16 per kernel row; lane (pixel p, half h) reads its 8 halfs = the two pixels
//     (32 i + p) S + 4 g + 2 h, + 1 of the patch row as ONE aligned ds_read_b128, conflict-free (consecutive lanes, consecutive 16-byte slots);
//   * the weights arrive PRE-PACKED in fragment order (tlk_conv_stem16_pack: [kh][g][cout tile][lane][8 halfs]) and sit in LDS behind the patch,
//     copied once per workgroup: a fragment is one conflict-free ds_read_b128 (held in registers -- 112 of them -- they left no room for the
//     next strip's pixels in flight, and that prefetch is worth more: 2.17 -> 1.67 ms);
//   * the pixels of strip s + 1 are loaded (two aligned dwords per pixel, unconditionally, from clamped addresses) before strip s is multiplied
//     and wait in registers: the memory round trip hides behind the MFMAs;
//   * a wavefront computes pooled row p: convolution rows 2 p - 1, 2 p, 2 p + 1, each + bias, activation, rounding to f16 (max commutes with
//     the rounding), 3 -> 1 max along the row INSIDE the accumulator layout -- a lane holds 4 consecutive columns x 4 groups of one channel;
//     the one neighbour column it lacks sits in lane ^ 32 (two packed shuffles per tile) -- then a running v_pk_max_f16 over the three rows.
//     Every convolution row with an odd index is computed by two wavefronts (1.5 x the MFMAs: 12 of the 56 per row that the pipe has room for);
//   * the pooled row leaves through a private LDS tile of the wavefront as 16-byte stores (8 channels of a pixel per lane).
// POOL = false (the 3 x 3 stems, or a 7 x 7 one without the pool): a wavefront computes one convolution row of a 4-row x 64-column strip.
// Arithmetic = the f16 route's everywhere else: f16 operands, fp32 accumulation, fp32 bias, f16 result.  Honours tlk_conv_set_dynamic_batch.
#include "tlk_common.hpp"

using namespace tlk;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Stem16Args {
    const _Float16 *x, *wp;               // image (N, H, W, x_pix halfs per pixel, 3 used); weights packed by tlk_conv_stem16_pack
    const float *bias;
    _Float16 *y;                          // POOL: (N, Hp, Wp, y_pix); else (N, Ho, Wo, y_pix)
    int N, H, W, Ho, Wo, Hp, Wp, Cout, pad, x_pix, y_pix, act;
    int col_tiles, row_tiles;
    const int *n_dyn;
};

constexpr int TC = 64;                    // convolution columns of a strip
constexpr int PT = 4;                     // rows of a strip (pooled rows with POOL, convolution rows without): one per wavefront
constexpr int RSTRIPS = 3;                // strips per workgroup (the weight fragments are fetched once for all of them)
constexpr int OST = 72;                   // halfs per pixel row of a wavefront's output staging tile (64 + 8: 16-byte aligned rows on different banks)

__device__ __forceinline__ h16x2 pkmax(h16x2 a, h16x2 b) { return __builtin_elementwise_max(a, b); }

template <int KH, int KW, int S, int NCO, bool POOL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) conv_stem16_kernel(const Stem16Args p)
{
    constexpr int G = (KW * 4 + 15) / 16;                               // 16-wide k slices per kernel row
    constexpr int CROWS = POOL ? 2 * PT + 1 : PT;                       // convolution rows a strip touches
    constexpr int PR = (CROWS - 1) * S + KH, PWP = (TC - 1) * S + 4 * G;        // patch rows x pixels (8 bytes each)
    static_assert((PWP * 8) % 16 == 0, "patch rows must stay 16-byte aligned");
    constexpr int SROWS = POOL ? 32 : 64;                               // pixels of a staged output row
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint2 *patch = reinterpret_cast<uint2 *>(lds);                      // [PR][PWP]
    _Float16 *stg_all = reinterpret_cast<_Float16 *>(lds + ((PR * PWP * 8 + 15) & ~15));       // 4 x [SROWS][OST]
    i32x4 *wl = reinterpret_cast<i32x4 *>(lds + ((PR * PWP * 8 + 15) & ~15) + 4 * SROWS * OST * 2);      // [KH][G][NCO][64 lanes] weight fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, pl = lane & 31;
    int b = blockIdx.x;
    const int ct = b % p.col_tiles; b /= p.col_tiles;
    const int rt = b % p.row_tiles; const int n = b / p.row_tiles;
    if (p.n_dyn && n >= p.n_dyn[0]) return;
    const int wo0 = ct * TC, wi0 = wo0 * S - p.pad;
    // ---- weight fragments: [kh][g][jj][lane][8 halfs], zero where kw >= KW, c == 3 or cout >= Cout (the packer wrote them); into LDS once per
    // workgroup -- held in registers (112 of them for 7 x 7 x 64) they left no room for the next strip's pixels in flight
    {
        constexpr int NF = KH * G * NCO * 64, FI = (NF + 255) / 256;
        i32x4 wv[FI];
#pragma unroll
        for (int k = 0; k < FI; ++k) { const int e = k * 256 + tid; wv[k] = reinterpret_cast<const i32x4 *>(p.wp)[e < NF ? e : 0]; }
#pragma unroll
        for (int k = 0; k < FI; ++k) { const int e = k * 256 + tid; if (e < NF) wl[e] = wv[k]; }
    }
    float bias[NCO];
#pragma unroll
    for (int jj = 0; jj < NCO; ++jj) bias[jj] = (p.bias && jj * 32 + pl < p.Cout) ? p.bias[jj * 32 + pl] : 0.f;
    const _Float16 *img = p.x + (size_t)n * p.H * p.W * p.x_pix;
    _Float16 *stg = stg_all + wave * (SROWS * OST);
    const int out_rows = POOL ? p.Hp : p.Ho;
    // ---- input patch of a strip.  A pixel's three halfs straddle two aligned dwords: both are fetched UNCONDITIONALLY from a clamped address and
    // the halfs picked by the parity of the pixel's offset (the first version guarded three 2-byte loads per pixel with branches: the compiler
    // then waited for each one in turn -- 21 dependent round trips per strip, 16 us; the kernel ran at a third of the library's speed instead of
    // five times it).  The loads of strip s + 1 are issued BEFORE strip s is multiplied and wait in registers: the memory round trip hides behind
    // the MFMAs.
    constexpr int NPX = PR * PWP, QI = (NPX + 255) / 256;
    const unsigned par0 = (unsigned)(((size_t)n * p.H * p.W * p.x_pix) & 1);                      // an image with an odd number of halfs starts mid-dword
    const unsigned *img32 = reinterpret_cast<const unsigned *>(img - par0);
    const unsigned last_dw = (unsigned)((par0 + (long long)p.H * p.W * p.x_pix - 1) >> 1);          // (one image stays below 2^31 halfs: host check)
    unsigned d0[QI], d1[QI];
    auto strip_hi0 = [&](int strip) { const int r0 = (rt * RSTRIPS + strip) * PT; return (POOL ? 2 * r0 - 1 : r0) * S - p.pad; };
    auto issue_loads = [&](int hi0) {
#pragma unroll
        for (int k = 0; k < QI; ++k) {
            const int e = k * 256 + tid;
            const int r = e / PWP, px = e - r * PWP;
            const int hi = hi0 + r, wi = wi0 + px;
            const bool ok = e < NPX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned dw = ok ? (par0 + (unsigned)(hi * p.W + wi) * (unsigned)p.x_pix) >> 1 : 0u;
            d0[k] = img32[dw];
            d1[k] = img32[dw < last_dw ? dw + 1 : last_dw];
        }
    };
    auto store_patch = [&](int hi0) {
#pragma unroll
        for (int k = 0; k < QI; ++k) {
            const int e = k * 256 + tid;
            const int r = e / PWP, px = e - r * PWP;
            const int hi = hi0 + r, wi = wi0 + px;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const bool odd = ((par0 + (unsigned)(hi * p.W + wi) * (unsigned)p.x_pix) & 1u) != 0;
            // even offset: halfs (d0.lo, d0.hi, d1.lo); odd: (d0.hi, d1.lo, d1.hi)
            const unsigned lo = odd ? ((d0[k] >> 16) | (d1[k] << 16)) : d0[k];
            const unsigned hi3 = odd ? (d1[k] >> 16) : (d1[k] & 0xffffu);
            if (e < NPX) patch[e] = ok ? make_uint2(lo, hi3) : make_uint2(0u, 0u);
        }
    };
    issue_loads(strip_hi0(0));
    for (int strip = 0; strip < RSTRIPS; ++strip) {
        const int r0 = (rt * RSTRIPS + strip) * PT;                      // first (pooled / convolution) row of the strip
        if (r0 >= out_rows) break;                                       // uniform
        const int c0 = POOL ? 2 * r0 - 1 : r0;                           // first convolution row the strip touches
        if (strip) __syncthreads();                                      // everybody is done with the previous patch
        store_patch(strip_hi0(strip));
        __syncthreads();
        if (strip + 1 < RSTRIPS && r0 + PT < out_rows) issue_loads(strip_hi0(strip + 1));
        const int orow = r0 + wave;                                      // this wavefront's output row
        if (orow >= out_rows) continue;                                  // wave-uniform (the barrier above is the strip's last)
        constexpr int NV = POOL ? 3 : 1;
        h16x2 rmax[2][NCO][4];                                           // POOL: running max of the pooled pairs (columns 16 i + 4 k + 2 half, + 1)
        // v_pk_max_f16 / fmaxf16 have fmax semantics: a NaN among the pooled values would be DROPPED (torch.max_pool2d propagates it -- ADVICE r05;
        // the pipeline's non-finite check must see what the stem produced).  Every convolution output of the row triple is folded into pz as
        // 0 * v: NaN iff one of them is NaN or infinite; the pooled outputs of that channel and row then leave as NaN (a superset of the windows
        // torch would poison: never fewer).  16 FMAs per MFMA tile.
        float pz[NCO];
#pragma unroll
        for (int jj = 0; jj < NCO; ++jj) pz[jj] = 0.f;
        if (POOL) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                    for (int k = 0; k < 4; ++k) rmax[i][jj][k] = h16x2{(_Float16)-65504.f, (_Float16)-65504.f};
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int crow = POOL ? 2 * orow - 1 + v : orow;             // convolution row
            if (POOL && (crow < 0 || crow >= p.Ho)) continue;            // the pool's padding rows: -inf, nothing to fold in (uniform)
            const int cr = crow - c0;                                    // row within the strip's patch
            const unsigned char *arow = reinterpret_cast<const unsigned char *>(patch) + ((cr * S) * PWP + pl * S + 2 * half) * 8;
            f32x16 acc[2][NCO];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    h16x8 a[2], bfr[NCO];
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const h16x8 *>(arow + (kh * PWP + i * 32 * S + 4 * g) * 8);
#pragma unroll
                    for (int jj = 0; jj < NCO; ++jj) bfr[jj] = __builtin_bit_cast(h16x8, wl[((kh * G + g) * NCO + jj) * 64 + lane]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < NCO; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bfr[jj], acc[i][jj], 0, 0, 0);
                }
            _Float16 carry[NCO];                                         // last column of tile 0 as the other half holds it: tile 1's left neighbour
#pragma unroll
            for (int jj = 0; jj < NCO; ++jj) carry[jj] = (_Float16)-65504.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // bias + activation on the accumulators, one uniform switch per tile (a per-element switch compiled to a thousand branches)
#pragma unroll
                for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jj][r] += bias[jj];
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][jj][r] = acc[i][jj][r] < 0.f ? 0.f : acc[i][jj][r];
                } else if (p.act == ACT_SILU) {
#pragma unroll
                    for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][jj][r] = acc[i][jj][r] / (1.f + __expf(-acc[i][jj][r]));
                }
                // C/D map of the 32x32 tile: column (= cout) = lane & 31, row (= pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
                if (POOL) {
#pragma unroll
                    for (int jj = 0; jj < NCO; ++jj) {
                        _Float16 hv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = wo0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            hv[r] = col < p.Wo ? (_Float16)acc[i][jj][r] : (_Float16)-65504.f;
                            pz[jj] = __builtin_fmaf(col < p.Wo ? acc[i][jj][r] : 0.f, 0.f, pz[jj]);
                        }
                        // the column left of each 4-group lives in lane ^ 32: its registers 3, 7, 11, 15 (packed two to a dword)
                        _Float16 p3[4];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            const h16x2 mine = {hv[8 * k2 + 3], hv[8 * k2 + 7]};
                            const h16x2 theirs = __builtin_bit_cast(h16x2, __shfl_xor(__builtin_bit_cast(int, mine), 32, 64));
                            p3[2 * k2] = theirs.x; p3[2 * k2 + 1] = theirs.y;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            // half 1: columns 8 k + 4..7, left neighbour = the other half's 8 k + 3; half 0: columns 8 k..8 k + 3, left neighbour =
                            // the other half's 8 (k - 1) + 7 (the previous tile's last column for k = 0; the pool's padding at the row's start)
                            const _Float16 lo = k > 0 ? p3[k - 1] : carry[jj];
                            const _Float16 prev = half ? p3[k] : lo;
                            const _Float16 q0 = __builtin_fmaxf16(__builtin_fmaxf16(prev, hv[4 * k]), hv[4 * k + 1]);
                            const _Float16 q1 = __builtin_fmaxf16(__builtin_fmaxf16(hv[4 * k + 1], hv[4 * k + 2]), hv[4 * k + 3]);
                            rmax[i][jj][k] = pkmax(rmax[i][jj][k], h16x2{q0, q1});
                        }
                        carry[jj] = p3[3];
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) stg[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * OST + jj * 32 + pl] = (_Float16)acc[i][jj][r];
                }
            }
        }
        if (POOL) {
#pragma unroll
            for (int jj = 0; jj < NCO; ++jj) pz[jj] += __shfl_xor(pz[jj], 32, 64);      // the other half's columns are this half's window neighbours
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NCO; ++jj)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = 16 * i + 4 * k + 2 * half;
                        const bool bad = pz[jj] != pz[jj];
                        stg[q * OST + jj * 32 + pl] = bad ? (_Float16)__builtin_nanf("") : rmax[i][jj][k].x;
                        stg[(q + 1) * OST + jj * 32 + pl] = bad ? (_Float16)__builtin_nanf("") : rmax[i][jj][k].y;
                    }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- the staged row leaves as 16-byte stores: 8 lanes cover the 64 channels of a pixel
        const int wout = POOL ? p.Wp : p.Wo, q00 = POOL ? 0 : wo0;
        _Float16 *yrow = p.y + ((size_t)n * out_rows + orow) * wout * (size_t)p.y_pix;
#pragma unroll
        for (int it = 0; it < SROWS / 8; ++it) {
            const int q = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + q * OST + c8);
            if (q00 + q < wout && c8 < p.Cout) *reinterpret_cast<i32x4 *>(yrow + (size_t)(q00 + q) * p.y_pix + c8) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// weights (Cout, KH, KW, 3) f16 -> fragment order [kh][g][jj][lane = half * 32 + cout % 32][8 halfs]: k = 8 half + e <-> tap kw = 4 g + 2 half + e / 4,
// channel e % 4
__global__ void stem16_pack_kernel(const _Float16 *w, _Float16 *wp, int cout, int kh_, int kw_, int G, int NCO)
{
    const int total = kh_ * G * NCO * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int el = e & 7, lane = (e >> 3) & 63;
        int f = e >> 9;
        const int jj = f % NCO; f /= NCO;
        const int g = f % G, kh = f / G;
        const int half = lane >> 5, co = jj * 32 + (lane & 31);
        const int kw = 4 * g + 2 * half + (el >> 2), c = el & 3;
        wp[e] = (co < cout && kw < kw_ && c < 3) ? w[((size_t)(co * kh_ + kh) * kw_ + kw) * 3 + c] : (_Float16)0.f;
    }
}

template <int KH, int KW, int S, int NCO, bool POOL> int launch_stem16(Stem16Args &a, hipStream_t st)
{
    constexpr int G = (KW * 4 + 15) / 16;
    constexpr int CROWS = POOL ? 2 * PT + 1 : PT;
    constexpr int PR = (CROWS - 1) * S + KH, PWP = (TC - 1) * S + 4 * G, SROWS = POOL ? 32 : 64;
    constexpr size_t LDS_BYTES = (size_t)((PR * PWP * 8 + 15) & ~15) + (size_t)4 * SROWS * OST * 2 + (size_t)KH * G * NCO * 64 * 16;
    static_assert(LDS_BYTES <= 160 * 1024, "patch + staging must fit the CU's LDS");
    const int out_rows = POOL ? a.Hp : a.Ho;
    a.col_tiles = POOL ? 1 : (a.Wo + TC - 1) / TC;
    a.row_tiles = (out_rows + PT * RSTRIPS - 1) / (PT * RSTRIPS);
    const long long wgs = (long long)a.N * a.row_tiles * a.col_tiles;
    if (wgs > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_conv_stem16: too many output strips for one launch");
    if (wgs == 0) return TLK_OK;
    auto kern = conv_stem16_kernel<KH, KW, S, NCO, POOL>;
    static bool attr_set = false;
    if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), LDS_BYTES, st, a);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

int stem_geometry(int cout, int kh, int kw, int stride, int *g, int *nco)
{
    if (!((kh == 7 && kw == 7) || (kh == 3 && kw == 3)) || stride != 2 || cout < 1 || cout > 64) return 0;
    *g = (kw * 4 + 15) / 16;
    *nco = cout > 32 ? 2 : 1;
    return 1;
}

}  // namespace

extern "C" long long tlk_conv_stem16_packed_halfs(int cout, int kh, int kw, int stride)
{
    int g, nco;
    if (!stem_geometry(cout, kh, kw, stride, &g, &nco)) return fail(TLK_EINVAL, "tlk_conv_stem16: the f16 stem kernel takes 7 x 7 / 3 x 3, stride 2, Cout <= 64");
    return (long long)kh * g * nco * 64 * 8;
}

extern "C" int tlk_conv_stem16_pack(const void *w_dev, void *packed_dev, int cout, int kh, int kw, int stride, void *hip_stream)
{
    int g, nco;
    if (!stem_geometry(cout, kh, kw, stride, &g, &nco)) return fail(TLK_EINVAL, "tlk_conv_stem16_pack: the f16 stem kernel takes 7 x 7 / 3 x 3, stride 2, Cout <= 64");
    if (!w_dev || !packed_dev || ((uintptr_t)packed_dev & 15)) return fail(TLK_EINVAL, "tlk_conv_stem16_pack: NULL or unaligned buffer");
    hipLaunchKernelGGL(stem16_pack_kernel, dim3(32), dim3(256), 0, (hipStream_t)hip_stream, (const _Float16 *)w_dev, (_Float16 *)packed_dev, cout, kh, kw, g, nco);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_conv_stem16_nhwc(const void *x_dev, const void *packed_w_dev, const float *bias_dev, void *y_dev, int n, int h, int w, int cout, int kh, int kw,
                                    int stride, int pad, int act, int pool, int x_pix_stride, int y_pix_stride, void *hip_stream)
{
    int g, nco;
    if (!stem_geometry(cout, kh, kw, stride, &g, &nco)) return fail(TLK_EINVAL, "tlk_conv_stem16_nhwc: the f16 stem kernel takes 7 x 7 / 3 x 3, stride 2, Cout <= 64");
    if (n < 0 || h < 1 || w < 1 || pad < 0 || pad > kh / 2 || act < 0 || act > 2 || (long long)h * w * (x_pix_stride > 0 ? x_pix_stride : 3) >= 0x7fffffffLL)
        return fail(TLK_EINVAL, "tlk_conv_stem16_nhwc: bad shape");
    if (n == 0) return TLK_OK;
    Stem16Args a;
    a.x = (const _Float16 *)x_dev; a.wp = (const _Float16 *)packed_w_dev; a.bias = bias_dev; a.y = (_Float16 *)y_dev;
    a.N = n; a.H = h; a.W = w; a.Cout = cout; a.pad = pad; a.act = act;
    a.Ho = (h + 2 * pad - kh) / stride + 1; a.Wo = (w + 2 * pad - kw) / stride + 1;
    a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;                // 3 x 3 / stride 2 / pad 1
    a.x_pix = x_pix_stride > 0 ? x_pix_stride : 3;
    a.y_pix = y_pix_stride > 0 ? y_pix_stride : cout;
    a.n_dyn = conv_dynamic_batch();
    if (!x_dev || !packed_w_dev || !y_dev) return fail(TLK_EINVAL, "tlk_conv_stem16_nhwc: NULL buffer");
    if (a.x_pix < 3 || a.y_pix < cout || cout % 8 != 0 || a.y_pix % 8 != 0 || (((uintptr_t)y_dev | (uintptr_t)packed_w_dev) & 15) || ((uintptr_t)x_dev & 3))
        return fail(TLK_EINVAL, "tlk_conv_stem16_nhwc: Cout and the output pixel stride must be multiples of 8, y and the packed weights 16-byte aligned, x 4-byte aligned");
    if (pool && a.Wo > TC) return fail(TLK_EINVAL, "tlk_conv_stem16_nhwc: the fused max-pool takes maps up to 64 columns wide (one column strip)");
    hipStream_t st = (hipStream_t)hip_stream;
    if (kh == 7) {
        if (pool) return nco == 2 ? launch_stem16<7, 7, 2, 2, true>(a, st) : launch_stem16<7, 7, 2, 1, true>(a, st);
        return nco == 2 ? launch_stem16<7, 7, 2, 2, false>(a, st) : launch_stem16<7, 7, 2, 1, false>(a, st);
    }
    if (pool) return nco == 2 ? launch_stem16<3, 3, 2, 2, true>(a, st) : launch_stem16<3, 3, 2, 1, true>(a, st);
    return nco == 2 ? launch_stem16<3, 3, 2, 2, false>(a, st) : launch_stem16<3, 3, 2, 1, false>(a, st);
}
