// tlk_image.hip -- detector / ReID pre- and post-processing on gfx950:
//   * batched letterbox (rtmlib YOLOX.preprocess semantics, cv2 INTER_LINEAR fixed-point bilinear)
//   * ROI crop -> resize -> normalize for ReID patches (KPReId.preprocess semantics)
//   * YOLOX grid decode + per-class greedy NMS with wavefront ballot bitmasks
// HBM-bound byte movers: each thread produces 8 consecutive output pixels so every store is a
// 16-byte-per-lane coalesced global_store_dwordx4; source rows are re-used through L1/L2.
#include <hip/hip_fp16.h>

#include "tlk_common.hpp"

#include <map>
#include <mutex>
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------
// Letterbox: frames (B, H, W, 3) u8 -> (B, 3, S, S) [NCHW] | (B, S, S, 3) [NHWC] | (B, S/2, S/2, 12) [FOCUS]
// rtmlib YOLOX.preprocess: ratio = min(S/H, S/W); resized (int(H*ratio), int(W*ratio)) pasted top-left on 114.
// Thread = 8 consecutive x of one output row (FOCUS: of two rows).
// ---------------------------------------------------------------------------------------------
template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) letterbox_kernel(const unsigned char *__restrict__ frames, int B, int H, int W, int S,
                                                          int rh, int rw, T *__restrict__ out, int swap_rb)
{
    constexpr int ROWS = (LAYOUT == LAYOUT_FOCUS_NHWC) ? 2 : 1;
    const int groups_per_row = S / 8;
    const int rows_units = S / ROWS;
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long per_frame = (long long)groups_per_row * rows_units;
    if (gid >= per_frame * B) return;
    const int b = (int)(gid / per_frame);
    const int rem = (int)(gid - (long long)b * per_frame);
    const int yu = rem / groups_per_row, xg = rem - yu * groups_per_row;
    const int x_base = xg * 8;
    const unsigned char *img = frames + (size_t)b * H * W * 3;
    T px[ROWS][8][3];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = yu * ROWS + r;
        const bool yin = y < rh;
        Coef cy = yin ? cv_coef(y, H, rh, false) : Coef{0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = x_base + i;
            int v[3] = {114, 114, 114};
            if (yin && x < rw) sample3(img, W * 3, H, W, cy, cv_coef(x, W, rw, true), v);
#pragma unroll
            for (int c = 0; c < 3; ++c) px[r][i][c] = cvt<T>((float)v[c]);
            if (swap_rb) { const T t0 = px[r][i][0]; px[r][i][0] = px[r][i][2]; px[r][i][2] = t0; }
        }
    }
    if (LAYOUT == LAYOUT_NCHW) {
        const int y = yu;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Pack<T, 8> p;
#pragma unroll
            for (int i = 0; i < 8; ++i) p.v[i] = px[0][i][c];
            *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)b * 3 + c) * S + y) * S + x_base) = p;
        }
    } else if (LAYOUT == LAYOUT_NHWC) {
        const int y = yu;
        T *o = out + (((size_t)b * S + y) * S + x_base) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Pack<T, 8> p;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[0][idx / 3][idx % 3]; }
            *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
        }
    } else {   // FOCUS: channel = c + 3*((x&1)*2 + (y&1)); YOLOX Focus order (tl, bl, tr, br)
        const int S2 = S / 2;
        T *o = out + (((size_t)b * S2 + yu) * S2 + x_base / 2) * 12;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            Pack<T, 8> p;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = k * 8 + e;            // 0..47 = 4 focus pixels x 12 channels
                const int fp = idx / 12, ch = idx % 12;
                const int grp = ch / 3, c = ch % 3;
                const int xo = grp >> 1, yo = grp & 1;
                p.v[e] = px[yo][fp * 2 + xo][c];
            }
            *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ROI crop -> resize(OHxOW, cv2 INTER_LINEAR) -> (x - 255*mean) * (1/(255*std)).
// boxes: (B, max_n, 4) float32 ltwh as emitted by the detector; clip+round per coordinates.py:216-267.
// out: (B*max_n, 3, OH, OW) NCHW or (B*max_n, OH, OW, 3) NHWC; slots >= counts[b] are zero-filled.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void crop_ltrb(const float *ltwh, int W, int H, int &l, int &t, int &r, int &b)
{
    double b0 = ltwh[0], b1 = ltwh[1], b2 = ltwh[2], b3 = ltwh[3];
    b0 = fmax(0.0, fmin(b0, (double)(W - 2)));
    b1 = fmax(0.0, fmin(b1, (double)(H - 2)));
    b2 = fmax(1.0, fmin(b2, (double)(W - 1) - b0));
    b3 = fmax(1.0, fmin(b3, (double)(H - 1) - b1));
    l = (int)rint(b0); t = (int)rint(b1); r = (int)rint(b0 + b2); b = (int)rint(b1 + b3);
}

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) crop_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                     const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                     int OH, int OW, float m0, float m1, float m2, float d0, float d1, float d2,
                                                     T *__restrict__ out, int swap_rb)
{
    const int groups_per_row = OW / 8;
    const int per_crop = groups_per_row * OH;
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= (long long)per_crop * B * max_n) return;
    const int slot = (int)(gid / per_crop);
    const int rem = (int)(gid - (long long)slot * per_crop);
    const int y = rem / groups_per_row, x_base = (rem - y * groups_per_row) * 8;
    const int b = slot / max_n, i = slot - b * max_n;
    const float mean[3] = {m0, m1, m2}, den[3] = {d0, d1, d2};
    T px[8][3];
    bool valid = i < counts[b];
    int l = 0, t = 0, r = 0, bt = 0;
    if (valid) { crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt); valid = (r > l) && (bt > t); }
    if (valid) {
        const int cw = r - l, ch = bt - t;
        const unsigned char *base = frames + ((size_t)b * H * W + (size_t)t * W + l) * 3;
        const Coef cy = cv_coef(y, ch, OH, false);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int v[3];
            sample3(base, W * 3, ch, cw, cy, cv_coef(x_base + k, cw, OW, true), v);
#pragma unroll
            for (int c = 0; c < 3; ++c) { float f = (float)v[c]; f -= mean[c]; f *= den[c]; px[k][c] = cvt<T>(f); }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) px[k][c] = cvt<T>(0.f);
    }
    if (swap_rb) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
    }
    if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Pack<T, 8> p;
#pragma unroll
            for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
            *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
        }
    } else {
        T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Pack<T, 8> p;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
            *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged variants (the fast paths). A workgroup owns a band of output rows: it (1) stages the source-row
// segments the band samples into LDS with coalesced 16-byte global loads (each source byte leaves HBM/L2 once per band
// instead of once per tap through the texture-address path), (2) builds the cv2 coefficient tables of the band once,
// (3) lets every thread produce 8 output pixels from LDS and emit 16-byte stores.
// ---------------------------------------------------------------------------------------------
constexpr int STAGE_PAD = 32;           // bytes of slack per staged row (alignment shift + clamped x+1 tap)

// Copy bytes [g0, g0+len) of global memory to LDS so that byte g lands at lds[(g - (g0 & ~15))]; 16-byte loads on the
// aligned body, byte loads on a tail that would cross `gend` (end of the allocation).
__device__ __forceinline__ void stage_row(const unsigned char *__restrict__ g0, int len, unsigned char *lds, const unsigned char *gend,
                                          int tid, int nthreads)
{
    const int shift = (int)((uintptr_t)g0 & 15);
    const int chunks = (shift + len + 15) >> 4;
    const unsigned char *src = g0 - shift;            // (pointer arithmetic: the loads stay global_load)
    for (int c = tid; c < chunks; c += nthreads) {
        const unsigned char *p = src + (size_t)c * 16;
        if (p + 16 <= gend) *reinterpret_cast<uint4 *>(lds + c * 16) = *reinterpret_cast<const uint4 *>(p);
        else for (int k = 0; k < 16 && p + k < gend; ++k) lds[c * 16 + k] = p[k];
    }
}

struct XCoef { short off; short w0, w1, pad; };       // byte offset of tap 0 inside the staged row, weights; tap 1 = off + step
static_assert(sizeof(XCoef) == 8, "XCoef");

// ---- crop: workgroup = (slot, band of CROP_BAND output rows); OW <= 256
constexpr int CROP_BAND = 32;
constexpr int CROP_LDS_ROW_BYTES = 544;                // staged segment per source row: crops up to 170 px wide; 136 words -> rows
                                                       // start 8 banks apart (a 128-byte-multiple stride made every row hit the same banks)
constexpr int CROP_LDS_ROWS = 34;                      // 32 output rows of an upscaled band touch <= 34 source rows (18 KB -> 8 workgroups/CU)

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) crop_lds_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                         const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                         int OH, int OW, float m0, float m1, float m2, float d0, float d1, float d2,
                                                         T *__restrict__ out, int swap_rb)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[CROP_LDS_ROWS * CROP_LDS_ROW_BYTES];
    __shared__ int4 s_xc[256];          // (byte offset of tap 0, w0, w1, tap-1 step) stored k-major: [k * groups + xg] -> lanes read consecutive 16 B
    __shared__ int s_y0[CROP_BAND], s_y1[CROP_BAND], s_yw0[CROP_BAND], s_yw1[CROP_BAND];
    __shared__ int s_hdr[8];
    const int tid = threadIdx.x;
    const int bands = (OH + CROP_BAND - 1) / CROP_BAND;
    const int slot = blockIdx.x / bands, band = blockIdx.x - slot * bands;
    const int b = slot / max_n, i = slot - b * max_n;
    const int y_base = band * CROP_BAND;
    const int groups_per_row = OW / 8;
    const float mean[3] = {m0, m1, m2}, den[3] = {d0, d1, d2};
    bool valid = i < counts[b];
    int l = 0, t = 0, r = 0, bt = 0;
    if (valid) { crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt); valid = (r > l) && (bt > t); }
    const int cw = r - l, ch = bt - t;
    bool staged = false;
    if (valid) {
        // band row range (uniform): first/last source rows touched by output rows [y_base, y_base+CROP_BAND)
        const int ylast = min(y_base + CROP_BAND, OH) - 1;
        const double scale_y = (double)ch / (double)OH, scale_x = (double)cw / (double)OW;     // one fp64 division per axis
        const Coef c_first = cv_coef_s(y_base, ch, scale_y, false), c_last = cv_coef_s(ylast, ch, scale_y, false);
        const int r_lo = clampi(c_first.s, 0, ch - 1), r_hi = clampi(c_last.s + 1, 0, ch - 1);
        const int nrows = r_hi - r_lo + 1;
        staged = nrows <= CROP_LDS_ROWS && cw * 3 + STAGE_PAD <= CROP_LDS_ROW_BYTES;
        if (staged) {
            const unsigned char *gend = frames + (size_t)B * H * W * 3;
            // all (row, 16-byte chunk) pairs of the band are fetched in ONE sweep (a row is only ~10 chunks: staging row after
            // row serialised nrows global round trips per workgroup and left 60 % of the wave cycles waiting)
            const int cmax = (cw * 3 + 30) >> 4;
            for (int idx = tid; idx < nrows * cmax; idx += BLOCK) {
                const int rr = idx / cmax, c = idx - rr * cmax;
                const unsigned char *g0 = frames + ((size_t)b * H * W + (size_t)(t + r_lo + rr) * W + l) * 3;
                const int mis = (int)((uintptr_t)g0 & 15);          // pointer ARITHMETIC keeps the global address space (an integer round trip makes the loads flat_load, which also count against lgkmcnt)
                const int chunks = (mis + cw * 3 + 15) >> 4;
                if (c < chunks) {
                    const unsigned char *p = g0 - mis + (size_t)c * 16;
                    unsigned char *lds = s_rows + rr * CROP_LDS_ROW_BYTES + c * 16;
                    if (p + 16 <= gend) *reinterpret_cast<uint4 *>(lds) = *reinterpret_cast<const uint4 *>(p);
                    else for (int k = 0; k < 16 && p + k < gend; ++k) lds[k] = p[k];
                }
            }
            if (tid < OW) {
                const Coef cx = cv_coef_s(tid, cw, scale_x, true);
                s_xc[(tid & 7) * groups_per_row + (tid >> 3)] = make_int4(cx.s * 3, cx.w0, cx.w1, (cx.s + 1 < cw ? 3 : 0));
            }
            if (tid < CROP_BAND && y_base + tid < OH) {
                const Coef cy = cv_coef_s(y_base + tid, ch, scale_y, false);
                s_y0[tid] = clampi(cy.s, 0, ch - 1) - r_lo; s_y1[tid] = clampi(cy.s + 1, 0, ch - 1) - r_lo;
                s_yw0[tid] = cy.w0; s_yw1[tid] = cy.w1;
            }
            if (tid == 0) s_hdr[0] = r_lo;
        }
    }
    __syncthreads();
    const int row_in_band = tid / groups_per_row, xg = tid - row_in_band * groups_per_row;
    // thread -> (row, 8-px group); bands of CROP_BAND rows x groups_per_row groups may exceed BLOCK: loop
    for (int unit = tid; unit < CROP_BAND * groups_per_row; unit += BLOCK) {
        const int ry = unit / groups_per_row, x_base = (unit - ry * groups_per_row) * 8;
        const int y = y_base + ry;
        if (y >= OH) continue;
        T px[8][3];
        if (valid && staged) {
            // alignment shift of each staged row: rows are W*3 apart in memory -> shift differs per row
            const uintptr_t g0 = (uintptr_t)(frames + ((size_t)b * H * W + (size_t)(t + s_hdr[0] + s_y0[ry]) * W + l) * 3);
            const uintptr_t g1 = (uintptr_t)(frames + ((size_t)b * H * W + (size_t)(t + s_hdr[0] + s_y1[ry]) * W + l) * 3);
            const unsigned char *p0 = s_rows + s_y0[ry] * CROP_LDS_ROW_BYTES + (int)(g0 & 15);
            const unsigned char *p1 = s_rows + s_y1[ry] * CROP_LDS_ROW_BYTES + (int)(g1 & 15);
            const int b0 = s_yw0[ry], b1 = s_yw1[ry];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int4 cx = s_xc[k * groups_per_row + (x_base >> 3)];
                const int o0 = cx.x, o1 = o0 + cx.w, a0 = cx.y, a1 = cx.z;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int S0 = (int)p0[o0 + c] * a0 + (int)p0[o1 + c] * a1;
                    const int S1 = (int)p1[o0 + c] * a0 + (int)p1[o1 + c] * a1;
                    const int v = clampi((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, 0, 255);
                    float f = (float)v; f -= mean[c]; f *= den[c];
                    px[k][c] = cvt<T>(f);
                }
            }
        } else if (valid) {
            const unsigned char *base = frames + ((size_t)b * H * W + (size_t)t * W + l) * 3;
            const Coef cy = cv_coef(y, ch, OH, false);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int v[3];
                sample3(base, W * 3, ch, cw, cy, cv_coef(x_base + k, cw, OW, true), v);
#pragma unroll
                for (int c = 0; c < 3; ++c) { float f = (float)v[c]; f -= mean[c]; f *= den[c]; px[k][c] = cvt<T>(f); }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[k][c] = cvt<T>(0.f);
        }
        if (swap_rb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
        }
        if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Pack<T, 8> p;
#pragma unroll
                for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
            }
        } else {
            T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        }
    }
    (void)row_in_band; (void)xg;
}

// ---------------------------------------------------------------------------------------------
// Separable form of the same crop -> cv2 INTER_LINEAR resize -> normalise (the fast path since round 2).
// cv2's fixed-point bilinear is  S_r = p[r][x0]*a0 + p[r][x1]*a1  (11-bit weights) for the two source rows r = y0, y1, then
// v = (((b0 * (S_y0 >> 4)) >> 16) + ((b1 * (S_y1 >> 4)) >> 16) + 2) >> 2.  H[r][x][c] = S_r >> 4 depends on the SOURCE row only, and
// an upscaled crop (the normal case: boxes ~80 x 176 px -> 384 x 128) samples every source row from ~4.4 output rows. PMC counters
// put round 1's kernel (and a first separable version) at ~620 VALU instructions per thread, 4 cycles each: issue-bound at 0.31 of
// the HBM roof whatever the memory side did. This version is built around the instruction count. A workgroup = (slot, band of
// CS_BAND output rows):
//   0. wavefront 0 alone does the crop geometry (float64 clip + round, band row range) and the x / y coefficient tables and hands
//      them over through LDS; wavefronts 1-3 meanwhile copy the normalisation table (below) into LDS;
//   1. all: the band's source-row segments -> LDS (16-byte loads, lane = chunk, 8 rows per sweep: no index divisions);
//   2. horizontal pass ONCE per (source row, x, channel): thread = one x (its tap offsets and packed weights stay in registers), loop
//      over the staged rows: two byte taps packed in one register, one v_dot2_u32_u16 against (a0, a1), >> 4 -> 16-bit LDS plane;
//   3. vertical pass per output value from 16-byte reads of that plane: two 24-bit multiplies and ONE add that picks the high halves
//      (SDWA), then a table look-up: the table is indexed by t = (b0*H0 >> 16) + (b1*H1 >> 16) directly (t <= 1020) and holds
//      T((float)((t + 2) >> 2) - 255*mean) * (1 / (255*std))) -- exactly the reference's float arithmetic, built once per
//      (device, statistics, dtype) by a tiny kernel and cached;
//   4. the band's output (ONE contiguous block for NHWC) is assembled in LDS and leaves as fully coalesced 16-byte stores.
// Padding slots (i >= counts[b]) are not touched at all. Blocks are remapped so that the bands of one crop run on ONE XCD.
// ---------------------------------------------------------------------------------------------
constexpr int CS_BAND = 16;                            // output rows per workgroup
constexpr int CS_ROWS = 18;                            // staged source rows: CS_BAND * scale + 2 <= 18 for scale <= 1 (up-scaling / same size)
constexpr int CS_ROW_BYTES = 512;                      // 32 chunks of 16 bytes: crops up to 160 px wide
constexpr int CS_LUT_N = 1024;                         // entries per channel (t <= 1020)

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
template <typename T> struct LutBits;
template <> struct LutBits<float> { using type = unsigned int; };
template <> struct LutBits<__half> { using type = unsigned short; };
template <> struct LutBits<bf16_t> { using type = unsigned short; };

// region 0 holds the staged source rows and, once the horizontal pass is done with them, the band's output rows on their way to
// fully coalesced stores (CS_BAND rows x OW x 3 elements)
__host__ __device__ inline size_t crop_sep_region0(int OW, size_t elem)
{
    const size_t a = (size_t)CS_ROWS * CS_ROW_BYTES, b = (size_t)CS_BAND * OW * 3 * elem;
    return ((a > b ? a : b) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t crop_sep_lds_bytes(int OW, size_t elem)
{
    return crop_sep_region0(OW, elem) + (size_t)CS_ROWS * OW * 3 * 2 + (size_t)OW * 8 + (size_t)3 * CS_LUT_N * elem;
}

// lut[c][t] = T(((float)min((t + 2) >> 2, 255) - m[c]) * d[c]): float32 subtract then multiply, then the conversion -- value for
// value what crop_kernel / the oracle compute per element
template <typename T>
__global__ void crop_lut_kernel(T *__restrict__ lut, float m0, float m1, float m2, float d0, float d1, float d2)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= CS_LUT_N) return;
    const float mean[3] = {m0, m1, m2}, den[3] = {d0, d1, d2};
    int v = (t + 2) >> 2;
    v = v > 255 ? 255 : v;
    for (int c = 0; c < 3; ++c) { float f = (float)v; f -= mean[c]; f *= den[c]; lut[c * CS_LUT_N + t] = cvt<T>(f); }
}

struct CropPar { int l, t, cw, ch, r_lo, nrows, staged, valid; };

template <typename T, int LAYOUT, int OWC>
__global__ void __launch_bounds__(BLOCK) crop_sep_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                         const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                         int OH, int OW_rt, const T *__restrict__ lut_g, float m0, float m1, float m2,
                                                         float d0, float d1, float d2, T *__restrict__ out, int swap_rb, int nwg)
{
    static_assert(BLOCK == 256, "thread <-> (x, row parity) mapping of the horizontal pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_y[CS_BAND * 2];                   // per band row: (index of staged row y0) | (y1 << 16), (b0 | b1 << 16)
    __shared__ int s_rsh[CS_ROWS];                     // alignment shift of every staged row
    __shared__ CropPar s_par;
    const int OW = OWC ? OWC : OW_rt;
    const int tid = threadIdx.x;
    // bijective XCD-aware remap (cdna_hip_programming.md: block b runs on XCD b % 8)
    int wg;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int bands = (OH + CS_BAND - 1) / CS_BAND;
    const int slot = wg / bands, band = wg - slot * bands;
    const int b = slot / max_n, i = slot - b * max_n;
    if (i >= counts[b]) return;                         // padding slot: left untouched
    const int y_base = band * CS_BAND;
    const int nb = min(CS_BAND, OH - y_base);
    const int groups_per_row = OW >> 3;
    const size_t R0 = crop_sep_region0(OW, sizeof(T));
    unsigned char *s_rows = s_dyn;
    unsigned short *s_h = reinterpret_cast<unsigned short *>(s_dyn + R0);
    int2 *s_xc = reinterpret_cast<int2 *>(s_dyn + R0 + (size_t)CS_ROWS * OW * 6);
    T *s_lut = reinterpret_cast<T *>(s_dyn + R0 + (size_t)CS_ROWS * OW * 6 + (size_t)OW * 8);
    const int HS = OW * 3;                              // 16-bit elements per plane row
    const size_t frame_off = (size_t)b * H * W * 3;
    if (tid < WAVE) {
        // ---- 0. geometry + coefficient tables: ONE wavefront (the float64 clip / round / scale arithmetic is ~150 instructions)
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const bool valid = (r > l) && (bt > t);
        const int cw = r - l, ch = bt - t;
        int r_lo = 0, nrows = 0;
        bool staged = false;
        if (valid) {
            const double scale_y = (double)ch / (double)OH, scale_x = (double)cw / (double)OW;     // one fp64 division per axis
            const Coef c_first = cv_coef_s(y_base, ch, scale_y, false), c_last = cv_coef_s(y_base + nb - 1, ch, scale_y, false);
            r_lo = clampi(c_first.s, 0, ch - 1);
            nrows = clampi(c_last.s + 1, 0, ch - 1) - r_lo + 1;
            staged = nrows <= CS_ROWS && cw * 3 + STAGE_PAD <= CS_ROW_BYTES;
            if (staged) {
                for (int x = tid; x < OW; x += WAVE) {
                    const Coef cx = cv_coef_s(x, cw, scale_x, true);
                    s_xc[x] = make_int2((cx.s * 3) | ((cx.s + 1 < cw ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
                }
                if (tid < nb) {
                    const Coef cy = cv_coef_s(y_base + tid, ch, scale_y, false);
                    s_y[tid * 2] = (clampi(cy.s, 0, ch - 1) - r_lo) | ((clampi(cy.s + 1, 0, ch - 1) - r_lo) << 16);
                    s_y[tid * 2 + 1] = (cy.w0 & 0xffff) | (cy.w1 << 16);
                }
                if (tid < nrows) s_rsh[tid] = (int)((uintptr_t)(frames + frame_off + ((size_t)(t + r_lo + tid) * W + l) * 3) & 15);
            }
        }
        if (tid == 0) { CropPar p; p.l = l; p.t = t; p.cw = cw; p.ch = ch; p.r_lo = r_lo; p.nrows = nrows; p.staged = staged; p.valid = valid; s_par = p; }
    } else {
        // wavefronts 1-3: the normalisation table -> LDS (16-byte chunks)
        const int n16 = (int)(3 * CS_LUT_N * sizeof(T) / 16);
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *d = reinterpret_cast<uint4 *>(s_lut);
        for (int c = tid - WAVE; c < n16; c += BLOCK - WAVE) d[c] = g[c];
    }
    __syncthreads();
    const CropPar par = s_par;
    const bool staged = par.staged != 0, valid = par.valid != 0;
    if (staged) {
        // ---- 1. source rows -> LDS: lane = 16-byte chunk of a row (32 per row), 8 rows per sweep
        const unsigned char *gend = frames + (size_t)B * H * W * 3;
        const int c = tid & 31;
        for (int rr = tid >> 5; rr < par.nrows; rr += 8) {
            const unsigned char *g0 = frames + frame_off + ((size_t)(par.t + par.r_lo + rr) * W + par.l) * 3;
            const int mis = (int)((uintptr_t)g0 & 15);          // pointer ARITHMETIC keeps the global address space (an integer round trip makes the loads flat_load, which also count against lgkmcnt)
            const int chunks = (mis + par.cw * 3 + 15) >> 4;
            if (c < chunks) {
                const unsigned char *p = g0 - mis + (size_t)c * 16;
                unsigned char *lds = s_rows + rr * CS_ROW_BYTES + c * 16;
                if (p + 16 <= gend) *reinterpret_cast<uint4 *>(lds) = *reinterpret_cast<const uint4 *>(p);
                else for (int k = 0; k < 16 && p + k < gend; ++k) lds[k] = p[k];
            }
        }
        __syncthreads();
        // ---- 2. horizontal pass: thread = one x, every (BLOCK / OW)-th staged row
        if (OWC == 128) {
            const int x = tid & 127;
            const int2 xc = s_xc[x];
            const int o0 = xc.x & 0xffff, o1 = o0 + (xc.x >> 16);
            const us2_t A = __builtin_bit_cast(us2_t, xc.y);
            for (int rr = tid >> 7; rr < par.nrows; rr += 2) {
                const unsigned char *p = s_rows + rr * CS_ROW_BYTES + s_rsh[rr];
                unsigned short *o = s_h + rr * HS + x * 3;
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const unsigned int P = (unsigned int)p[o0 + c3] | ((unsigned int)p[o1 + c3] << 16);
                    o[c3] = (unsigned short)(__builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, P), A, 0u, false) >> 4);
                }
            }
        } else {
            for (int idx = tid; idx < par.nrows * OW; idx += BLOCK) {
                const int rr = idx / OW, x = idx - rr * OW;
                const int2 xc = s_xc[x];
                const int o0 = xc.x & 0xffff, o1 = o0 + (xc.x >> 16);
                const unsigned char *p = s_rows + rr * CS_ROW_BYTES + s_rsh[rr];
                const us2_t A = __builtin_bit_cast(us2_t, xc.y);
                unsigned short *o = s_h + rr * HS + x * 3;
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const unsigned int P = (unsigned int)p[o0 + c3] | ((unsigned int)p[o1 + c3] << 16);
                    o[c3] = (unsigned short)(__builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, P), A, 0u, false) >> 4);
                }
            }
        }
        __syncthreads();
    }
    // ---- 3. vertical pass + normalisation: thread -> (row of the band, 8 consecutive x)
    const bool use_lds_store = LAYOUT == LAYOUT_NHWC && ((size_t)OW * 3 * sizeof(T)) % 16 == 0;       // staged or not, uniform
    for (int unit = tid; unit < nb * groups_per_row; unit += BLOCK) {
        const int ry = OWC == 128 ? (unit >> 4) : unit / groups_per_row;
        const int x_base = (unit - ry * groups_per_row) * 8;
        const int y = y_base + ry;
        T px[8][3];
        if (staged) {
            const unsigned int yi = (unsigned int)s_y[ry * 2], yw = (unsigned int)s_y[ry * 2 + 1];
            const uint4 *h0 = reinterpret_cast<const uint4 *>(s_h + (yi & 0xffffu) * HS + x_base * 3);
            const uint4 *h1 = reinterpret_cast<const uint4 *>(s_h + (yi >> 16) * HS + x_base * 3);
            const unsigned int b0 = yw & 0xffffu, b1 = yw >> 16;
            unsigned int w0[12], w1[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint4 u = h0[k], v = h1[k];
                w0[k * 4] = u.x; w0[k * 4 + 1] = u.y; w0[k * 4 + 2] = u.z; w0[k * 4 + 3] = u.w;
                w1[k * 4] = v.x; w1[k * 4 + 1] = v.y; w1[k * 4 + 2] = v.z; w1[k * 4 + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const unsigned int a = (q & 1) ? (w0[q >> 1] >> 16) : (w0[q >> 1] & 0xffffu);
                const unsigned int c1 = (q & 1) ? (w1[q >> 1] >> 16) : (w1[q >> 1] & 0xffffu);
                // t <= 1020 always: H <= (255 * 2049) >> 4 and b0 + b1 <= 2049
                // (the compiler distributes the table's element size over the two shifted terms: 5 VALU per value; adding the two
                //  HIGH HALVES in one SDWA add keeps it at 4)
                const unsigned int xa = __umul24(b0, a), xb = __umul24(b1, c1);
                unsigned int t;
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(xa), "v"(xb));
                px[q / 3][q % 3] = s_lut[(q % 3) * CS_LUT_N + t];
            }
        } else if (valid) {
            const float mean[3] = {m0, m1, m2}, den[3] = {d0, d1, d2};
            const unsigned char *base = frames + frame_off + ((size_t)par.t * W + par.l) * 3;
            const Coef cy = cv_coef(y, par.ch, OH, false);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int v[3];
                sample3(base, W * 3, par.ch, par.cw, cy, cv_coef(x_base + k, par.cw, OW, true), v);
#pragma unroll
                for (int c = 0; c < 3; ++c) { float f = (float)v[c]; f -= mean[c]; f *= den[c]; px[k][c] = cvt<T>(f); }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[k][c] = cvt<T>(0.f);
        }
        if (swap_rb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
        }
        if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Pack<T, 8> p;
#pragma unroll
                for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
            }
        } else {
            // NHWC: a thread's 24 elements are contiguous but 24 elements apart from its neighbour's, so a direct 16-byte store
            // instruction touches every 128-byte line of the band partially. The band's output (nb rows = ONE contiguous block of
            // memory) is put together in region 0 instead and leaves below as fully coalesced 16-byte stores.
            T *o = use_lds_store ? reinterpret_cast<T *>(s_dyn) + ((size_t)ry * OW + x_base) * 3 : out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        }
    }
    if (use_lds_store) {
        __syncthreads();
        const int n16 = (int)((size_t)nb * OW * 3 * sizeof(T) / 16);
        uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + y_base) * OW * 3);
        const uint4 *l4 = reinterpret_cast<const uint4 *>(s_dyn);
        for (int c = tid; c < n16; c += BLOCK) stream_store(g + c, l4[c]);
    }
}

// ---------------------------------------------------------------------------------------------
// crop_fat_kernel: the same arithmetic as crop_sep_kernel for the ReID shape (OW = 128), but a workgroup owns CF_BANDS consecutive
// bands of ONE crop. With a workgroup per band the kernel was bound by neither VALU issue nor HBM (the time did not move when the
// VALU instruction count fell by 44 %): every workgroup ran geometry -> barrier -> global loads -> barrier -> horizontal ->
// barrier -> vertical -> store as a ~6 us dependent chain with 4 workgroups per CU in flight, 55 rounds of that per launch.
// Here geometry, the x table, the y tables of all CF_BANDS bands and the normalisation table are set up ONCE per workgroup, and
// the source rows of band k+1 are already in flight (in registers) while band k goes through its vertical pass: two barriers per
// band, no global-load latency on the critical path after the first band.
// ---------------------------------------------------------------------------------------------
constexpr int CF_BANDS = 8;

// rare paths kept out of line so that their registers do not count against the main loop's occupancy
__device__ __noinline__ uint4 load16_clipped(const unsigned char *p, const unsigned char *gend)     // 16 bytes at p, zero beyond gend
{
    unsigned int w[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16 && p + k < gend; ++k) w[k >> 2] |= (unsigned int)p[k] << ((k & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <typename T, int LAYOUT>
__device__ __noinline__ void crop_direct_unit(const unsigned char *__restrict__ base, int W, int ch, int cw, int OH, int OW, int y, int x_base,
                                              float m0, float m1, float m2, float d0, float d1, float d2, int swap_rb, T *__restrict__ out, size_t slot)
{
    const float mean[3] = {m0, m1, m2}, den[3] = {d0, d1, d2};
    const Coef cy = cv_coef(y, ch, OH, false);
    T px[8][3];
    for (int k = 0; k < 8; ++k) {
        int v[3];
        sample3(base, W * 3, ch, cw, cy, cv_coef(x_base + k, cw, OW, true), v);
        for (int c = 0; c < 3; ++c) { float f = (float)v[c]; f -= mean[c]; f *= den[c]; px[k][swap_rb ? 2 - c : c] = cvt<T>(f); }
    }
    for (int k = 0; k < 8; ++k)
        for (int c = 0; c < 3; ++c) {
            if (LAYOUT == LAYOUT_NCHW) out[((slot * 3 + c) * OH + y) * OW + x_base + k] = px[k][c];
            else out[((slot * OH + y) * OW + x_base + k) * 3 + c] = px[k][c];
        }
}

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) crop_fat_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                         const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                         int OH, const T *__restrict__ lut_g, float m0, float m1, float m2,
                                                         float d0, float d1, float d2, T *__restrict__ out, int swap_rb, int nwg)
{
    static_assert(BLOCK == 256, "thread <-> (x, row parity) mapping of the horizontal pass");
    constexpr int OW = 128, HS = OW * 3, GROUPS = OW / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_y[CF_BANDS * CS_BAND * 2];         // per output row of the chunk: (source row y0 | y1 << 16) relative to the crop, (b0 | b1 << 16)
    __shared__ int s_rsh[CS_ROWS];
    __shared__ CropPar s_par;
    const int tid = threadIdx.x;
    int wg;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int bands = (OH + CS_BAND - 1) / CS_BAND, chunks = (bands + CF_BANDS - 1) / CF_BANDS;
    const int slot = wg / chunks, chunk = wg - slot * chunks;
    const int b = slot / max_n, i = slot - b * max_n;
    if (i >= counts[b]) return;                         // padding slot: left untouched
    const int band0 = chunk * CF_BANDS, nbands = min(CF_BANDS, bands - band0);
    unsigned char *s_rows = s_dyn;
    unsigned short *s_h = reinterpret_cast<unsigned short *>(s_dyn + CS_ROWS * CS_ROW_BYTES);
    int2 *s_xc = reinterpret_cast<int2 *>(s_dyn + CS_ROWS * CS_ROW_BYTES + CS_ROWS * HS * 2);
    T *s_lut = reinterpret_cast<T *>(s_dyn + CS_ROWS * CS_ROW_BYTES + CS_ROWS * HS * 2 + OW * 8);
    const size_t frame_off = (size_t)b * H * W * 3;
    // ---- set-up, once per workgroup: wave 0 geometry + x table, waves 1-2 the y tables of the chunk, wave 3 the normalisation table
    if (tid < WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const bool valid = (r > l) && (bt > t);
        const int cw = r - l, ch = bt - t;
        const bool wide_ok = cw * 3 + STAGE_PAD <= CS_ROW_BYTES;
        if (valid && wide_ok) {
            const double scale_x = (double)cw / (double)OW;
            for (int x = tid; x < OW; x += WAVE) {
                const Coef cx = cv_coef_s(x, cw, scale_x, true);
                s_xc[x] = make_int2((cx.s * 3) | ((cx.s + 1 < cw ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
            }
        }
        if (tid == 0) { CropPar p; p.l = l; p.t = t; p.cw = cw; p.ch = ch; p.r_lo = 0; p.nrows = 0; p.staged = wide_ok; p.valid = valid; s_par = p; }
    } else if (tid < 3 * WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const int ch = bt - t, row = tid - WAVE, y = band0 * CS_BAND + row;
        if (bt > t && r > l && y < OH && row < nbands * CS_BAND) {
            const Coef cy = cv_coef_s(y, ch, (double)ch / (double)OH, false);
            s_y[row * 2] = clampi(cy.s, 0, ch - 1) | (clampi(cy.s + 1, 0, ch - 1) << 16);
            s_y[row * 2 + 1] = (cy.w0 & 0xffff) | (cy.w1 << 16);
        }
    } else {
        const int n16 = (int)(3 * CS_LUT_N * sizeof(T) / 16);
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *d = reinterpret_cast<uint4 *>(s_lut);
        for (int c = tid - 3 * WAVE; c < n16; c += WAVE) d[c] = g[c];
    }
    __syncthreads();
    const CropPar par = s_par;
    const bool valid = par.valid != 0;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const int c16 = tid & 31, rsub = tid >> 5;
    uint4 stage[3];
    int r_lo_next = 0, nrows_next = 0;
    bool staged_next = false;
    // source rows of band kb -> registers (lane = 16-byte chunk of a row, rows rsub, rsub + 8, rsub + 16)
    auto fetch = [&](int kb) {
        const int row_first = kb * CS_BAND, row_last = min(kb * CS_BAND + CS_BAND, OH - band0 * CS_BAND) - 1;
        r_lo_next = s_y[row_first * 2] & 0xffff;
        nrows_next = (int)((unsigned int)s_y[row_last * 2] >> 16) - r_lo_next + 1;
        staged_next = par.staged && nrows_next <= CS_ROWS;
        if (!staged_next) return;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int rr = rsub + 8 * j;
            stage[j] = make_uint4(0, 0, 0, 0);
            if (rr < nrows_next) {
                const unsigned char *g0 = frames + frame_off + ((size_t)(par.t + r_lo_next + rr) * W + par.l) * 3;
                const int mis = (int)((uintptr_t)g0 & 15);          // pointer ARITHMETIC keeps the global address space (an integer round trip makes the loads flat_load, which also count against lgkmcnt)
                const int nchunks = (mis + par.cw * 3 + 15) >> 4;
                const unsigned char *p = g0 - mis + (size_t)c16 * 16;
                if (c16 < nchunks) {
                    if (p + 16 <= gend) stage[j] = *reinterpret_cast<const uint4 *>(p);
                    else stage[j] = load16_clipped(p, gend);
                }
            }
        }
    };
    if (valid) fetch(0);
    for (int kb = 0; kb < nbands; ++kb) {
        const int y_base = (band0 + kb) * CS_BAND, nb = min(CS_BAND, OH - y_base);
        const int r_lo = r_lo_next, nrows = nrows_next;
        const bool staged = valid && staged_next;
        if (staged) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int rr = rsub + 8 * j;
                if (rr < nrows) *reinterpret_cast<uint4 *>(s_rows + rr * CS_ROW_BYTES + c16 * 16) = stage[j];
            }
            if (tid < nrows) s_rsh[tid] = (int)((uintptr_t)(frames + frame_off + ((size_t)(par.t + r_lo + tid) * W + par.l) * 3) & 15);
        }
        __syncthreads();
        if (staged) {                                   // horizontal pass: thread = one x, every second staged row
            const int x = tid & 127;
            const int2 xc = s_xc[x];
            const int o0 = xc.x & 0xffff, o1 = o0 + (xc.x >> 16);
            const us2_t A = __builtin_bit_cast(us2_t, xc.y);
            for (int rr = tid >> 7; rr < nrows; rr += 2) {
                const unsigned char *p = s_rows + rr * CS_ROW_BYTES + s_rsh[rr];
                unsigned short *o = s_h + rr * HS + x * 3;
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const unsigned int P = (unsigned int)p[o0 + c3] | ((unsigned int)p[o1 + c3] << 16);
                    o[c3] = (unsigned short)(__builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, P), A, 0u, false) >> 4);
                }
            }
        }
        __syncthreads();
        if (valid && kb + 1 < nbands) fetch(kb + 1);    // in flight during the vertical pass
        const int ry = tid >> 4, x_base = (tid & (GROUPS - 1)) * 8;
        const int y = y_base + ry;
        if (ry < nb && !staged) {
            // crops too tall / wide for the staged path: direct sampling (mean / std per SOURCE channel were swapped on the host for swap_rb)
            if (valid) crop_direct_unit<T, LAYOUT>(frames + frame_off + ((size_t)par.t * W + par.l) * 3, W, par.ch, par.cw, OH, OW, y, x_base, m0, m1, m2, d0, d1, d2,
                                                   swap_rb, out, (size_t)slot);
            else {
                for (int k = 0; k < 8; ++k)
                    for (int c = 0; c < 3; ++c) {
                        if (LAYOUT == LAYOUT_NCHW) out[(((size_t)slot * 3 + c) * OH + y) * OW + x_base + k] = cvt<T>(0.f);
                        else out[(((size_t)slot * OH + y) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
                    }
            }
        } else if (ry < nb) {
            T px[8][3];
            {
                const int row = kb * CS_BAND + ry;
                const unsigned int yi = (unsigned int)s_y[row * 2], yw = (unsigned int)s_y[row * 2 + 1];
                const uint4 *h0 = reinterpret_cast<const uint4 *>(s_h + ((yi & 0xffffu) - r_lo) * HS + x_base * 3);
                const uint4 *h1 = reinterpret_cast<const uint4 *>(s_h + ((yi >> 16) - r_lo) * HS + x_base * 3);
                const unsigned int b0 = yw & 0xffffu, b1 = yw >> 16;
                unsigned int w0[12], w1[12];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint4 u = h0[k], v = h1[k];
                    w0[k * 4] = u.x; w0[k * 4 + 1] = u.y; w0[k * 4 + 2] = u.z; w0[k * 4 + 3] = u.w;
                    w1[k * 4] = v.x; w1[k * 4 + 1] = v.y; w1[k * 4 + 2] = v.z; w1[k * 4 + 3] = v.w;
                }
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    const unsigned int a = (q & 1) ? (w0[q >> 1] >> 16) : (w0[q >> 1] & 0xffffu);
                    const unsigned int c1 = (q & 1) ? (w1[q >> 1] >> 16) : (w1[q >> 1] & 0xffffu);
                    const unsigned int xa = __umul24(b0, a), xb = __umul24(b1, c1);
                    unsigned int t;                     // t <= 1020 always (see crop_sep_kernel)
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(xa), "v"(xb));
                    px[q / 3][q % 3] = s_lut[(q % 3) * CS_LUT_N + t];
                }
            }
            if (swap_rb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
            }
            if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                    *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
                }
            } else {
                T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                    *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// crop_wave_kernel: the arithmetic of crop_fat_kernel with NO workgroup barrier in the band loop. The ablation of crop_fat_kernel
// (profiles/r02_crop_fat_phases.txt) showed its two passes' arithmetic fully hidden and its time spent in a skeleton of 16 barriers per
// workgroup plus source loads and output stores that do not overlap: the four wavefronts of a workgroup move in lockstep, and within a
// wavefront the wait for prefetched rows is also a wait for output stores issued after them (reads and writes share vmcnt and complete out of
// order with respect to each other). Here the set-up (geometry, x / y tables, normalisation table) is still shared by the workgroup -- ONE
// barrier -- but then every wavefront owns WV_MB consecutive mini-bands of WV_ROWS output rows, with its own LDS staging area and 16-bit
// plane: load (prefetched one mini-band ahead) -> LDS -> horizontal pass -> vertical pass -> 3 KB of contiguous output. The wavefronts drift
// apart, so while one waits for its stores the other fifteen of the CU compute, load or store.
// ---------------------------------------------------------------------------------------------
constexpr int WV_ROWS = 4;                             // output rows of a mini-band (lane = (row, group of 8 px) in the vertical pass)
constexpr int WV_SRC = 6;                              // staged source rows per mini-band: WV_ROWS * scale + 2 for scale <= 1; 2-row mini-bands up to scale 2
constexpr int WV_WAVE_LDS = WV_SRC * CS_ROW_BYTES + WV_SRC * 128 * 6;

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) crop_wave_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                          const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                          int OH, const T *__restrict__ lut_g, float m0, float m1, float m2,
                                                          float d0, float d1, float d2, T *__restrict__ out, int swap_rb, int nwg)
{
    constexpr int OW = 128, HS = OW * 3, GROUPS = OW / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_y[CF_BANDS * CS_BAND * 2];         // per output row of the chunk: (source row y0 | y1 << 16) relative to the crop, (b0 | b1 << 16)
    __shared__ CropPar s_par;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int wg;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int bands = (OH + CS_BAND - 1) / CS_BAND, chunks = (bands + CF_BANDS - 1) / CF_BANDS;
    const int slot = wg / chunks, chunk = wg - slot * chunks;
    const int b = slot / max_n, i = slot - b * max_n;
    if (i >= counts[b]) return;                         // padding slot: left untouched
    const int band0 = chunk * CF_BANDS, nbands = min(CF_BANDS, bands - band0);
    int2 *s_xc = reinterpret_cast<int2 *>(s_dyn);
    T *s_lut = reinterpret_cast<T *>(s_dyn + OW * 8);
    unsigned char *s_rows = s_dyn + OW * 8 + ((3 * CS_LUT_N * sizeof(T) + 15) & ~(size_t)15) + (size_t)wv * WV_WAVE_LDS;      // this wavefront's staging area ...
    unsigned short *s_h = reinterpret_cast<unsigned short *>(s_rows + WV_SRC * CS_ROW_BYTES);                                  // ... and 16-bit plane
    const size_t frame_off = (size_t)b * H * W * 3;
    // ---- set-up, once per workgroup: wave 0 geometry + x table, waves 1-2 the y tables of the chunk, wave 3 the normalisation table
    if (tid < WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const bool valid = (r > l) && (bt > t);
        const int cw = r - l, ch = bt - t;
        const bool wide_ok = cw * 3 + STAGE_PAD <= CS_ROW_BYTES;
        if (valid && wide_ok) {
            const double scale_x = (double)cw / (double)OW;
            for (int x = tid; x < OW; x += WAVE) {
                const Coef cx = cv_coef_s(x, cw, scale_x, true);
                s_xc[x] = make_int2((cx.s * 3) | ((cx.s + 1 < cw ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
            }
        }
        if (tid == 0) { CropPar p; p.l = l; p.t = t; p.cw = cw; p.ch = ch; p.r_lo = 0; p.nrows = 0; p.staged = wide_ok; p.valid = valid; s_par = p; }
    } else if (tid < 3 * WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const int ch = bt - t, row = tid - WAVE, y = band0 * CS_BAND + row;
        if (bt > t && r > l && y < OH && row < nbands * CS_BAND) {
            const Coef cy = cv_coef_s(y, ch, (double)ch / (double)OH, false);
            s_y[row * 2] = clampi(cy.s, 0, ch - 1) | (clampi(cy.s + 1, 0, ch - 1) << 16);
            s_y[row * 2 + 1] = (cy.w0 & 0xffff) | (cy.w1 << 16);
        }
    } else {
        const int n16 = (int)(3 * CS_LUT_N * sizeof(T) / 16);
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *d = reinterpret_cast<uint4 *>(s_lut);
        for (int c = tid - 3 * WAVE; c < n16; c += WAVE) d[c] = g[c];
    }
    __syncthreads();
    const CropPar par = s_par;
    const bool valid = par.valid != 0;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const int row0 = band0 * CS_BAND, rows_chunk = min(nbands * CS_BAND, OH - row0);       // output rows of this workgroup
    // mini-bands of 4 rows while the crop is not taller than the output (<= 6 source rows each), of 2 rows up to twice as tall
    const int mbh = par.ch <= OH ? WV_ROWS : 2;
    const int n_mb = (rows_chunk + mbh - 1) / mbh, mb_per_wave = (n_mb + NWAVES - 1) / NWAVES;
    const int mb_lo = wv * mb_per_wave, mb_hi = min(n_mb, mb_lo + mb_per_wave);
    const int cmax = (par.cw * 3 + 30) >> 4;             // 16-byte chunks per staged row
    // ---- is every mini-band of this wavefront on the fast path? (staged crop, <= WV_SRC source rows, <= 3 x 64 chunks, no load near the end of the frames)
    bool fast = valid && par.staged && par.ch <= 2 * OH;
    if (fast) {
        fast = frames + frame_off + ((size_t)(par.t + par.ch - 1) * W + par.l) * 3 + 34 * 16 <= gend;
        for (int mb = mb_lo + lane; mb < mb_hi; mb += WAVE) {
            const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
            const int n = (int)((unsigned int)s_y[rb * 2] >> 16) - (s_y[ra * 2] & 0xffff) + 1;
            if (n > WV_SRC || n * cmax > 3 * WAVE) fast = false;
        }
        fast = __all(fast);
    }
    if (!fast) {
        // rare: invalid / very tall / very wide crops, the last rows of the last frame -- direct sampling, a unit (row, 8 px) per lane
        for (int mb = mb_lo; mb < mb_hi; ++mb)
            for (int u = lane; u < mbh * GROUPS; u += WAVE) {
                const int ry = u >> 4, x_base = (u & (GROUPS - 1)) * 8, y = row0 + mb * mbh + ry;
                if (y >= OH || mb * mbh + ry >= rows_chunk) continue;
                if (valid) crop_direct_unit<T, LAYOUT>(frames + frame_off + ((size_t)par.t * W + par.l) * 3, W, par.ch, par.cw, OH, OW, y, x_base, m0, m1, m2, d0, d1, d2,
                                                       swap_rb, out, (size_t)slot);
                else
                    for (int k = 0; k < 8; ++k)
                        for (int c = 0; c < 3; ++c) {
                            if (LAYOUT == LAYOUT_NCHW) out[(((size_t)slot * 3 + c) * OH + y) * OW + x_base + k] = cvt<T>(0.f);
                            else out[(((size_t)slot * OH + y) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
                        }
            }
        return;
    }
    // ---- fast path: no call, no workgroup barrier below this line
    const unsigned char *crop0 = frames + frame_off + ((size_t)par.t * W + par.l) * 3;
    const unsigned int a_step = ((unsigned int)W * 3u) & 15u;
    // horizontal pass: this lane's two adjacent x (2 lane, 2 lane + 1). The two taps of the three channels of an x are 6 consecutive source bytes
    // (3 when the right tap is clamped onto the left one): three ALIGNED dword reads + v_alignbyte bring them, v_perm_b32 with per-lane selectors
    // builds the (left, right) pairs for v_dot2 -- byte-granular ds_read_u8 taps cost 9 x the LDS time (profiles/r02_lds_microbench.txt), and in this
    // kernel the LDS pipe is the shared resource the sixteen free-running wavefronts of a CU compete for
    const int4 xc2 = *reinterpret_cast<const int4 *>(&s_xc[2 * lane]);
    const int oA = xc2.x & 0xffff, oB = xc2.z & 0xffff;
    const unsigned int selA = 0x0c000c00u | ((unsigned int)(xc2.x >> 16) << 16), selB = 0x0c000c00u | ((unsigned int)(xc2.z >> 16) << 16);
    const us2_t wA = __builtin_bit_cast(us2_t, xc2.y), wB = __builtin_bit_cast(us2_t, xc2.w);
    // source rows of a mini-band -> three registers per lane (flat sweep over (row, chunk)); unconditional loads: lanes past the band re-read its first
    // chunk. TWO mini-bands are kept in flight (register sets X and Y): the wait for the rows of mini-band m then has the loads of m + 1 behind it, and
    // the rule "reads and writes complete out of order with respect to each other" no longer forces it to drain the stores of m - 1 just issued
    struct RowRegs { uint4 a, b, c; int r_lo, nrows; };
    RowRegs X{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), 0, 0}, Y = X;
    auto fetch = [&](int mb, RowRegs &R) {
        const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
        const int r_lo_n = s_y[ra * 2] & 0xffff;
        const int nrows_n = (int)((unsigned int)s_y[rb * 2] >> 16) - r_lo_n + 1;
        const unsigned char *rowp = crop0 + (size_t)r_lo_n * W * 3;
        auto one = [&](int idx) {
            const int rr = idx / cmax, c = idx - rr * cmax;
            const bool in = rr < nrows_n;
            const unsigned char *g0 = rowp + (size_t)(in ? rr : 0) * W * 3;
            const int mis = (int)((uintptr_t)g0 & 15);
            return *reinterpret_cast<const uint4 *>(g0 - mis + (size_t)((in && c < ((mis + par.cw * 3 + 15) >> 4)) ? c : 0) * 16);
        };
        R.a = one(lane); R.b = one(lane + WAVE); R.c = one(lane + 2 * WAVE);
        R.r_lo = r_lo_n; R.nrows = nrows_n;
    };
    auto mini_band = [&](int mb, RowRegs &R) {
        const int r_lo = R.r_lo, nrows = R.nrows;
        // registers -> this wavefront's staging rows
        {
            const int i0 = lane, i1 = lane + WAVE, i2 = lane + 2 * WAVE;
            const int r0 = i0 / cmax, r1 = i1 / cmax, r2 = i2 / cmax;
            if (r0 < nrows) *reinterpret_cast<uint4 *>(s_rows + r0 * CS_ROW_BYTES + (i0 - r0 * cmax) * 16) = R.a;
            if (r1 < nrows) *reinterpret_cast<uint4 *>(s_rows + r1 * CS_ROW_BYTES + (i1 - r1 * cmax) * 16) = R.b;
            if (r2 < nrows) *reinterpret_cast<uint4 *>(s_rows + r2 * CS_ROW_BYTES + (i2 - r2 * cmax) * 16) = R.c;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (mb + 2 < mb_hi) fetch(mb + 2, R);            // this register set is free again: the mini-band after the next one
        // horizontal pass into the PLANAR 16-bit plane s_h[row][channel][x]: one aligned ds_write_b32 per channel and row for the lane's x pair
        {
            const unsigned int a_lo = (unsigned int)(uintptr_t)(crop0 + (size_t)r_lo * W * 3) & 15u;
            auto taps = [&](int addr, unsigned int &lo, unsigned int &hi) {
                const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                const unsigned int d0 = q[0], d1 = q[1], d2 = q[2];
                lo = __builtin_amdgcn_alignbyte(d1, d0, (unsigned int)addr & 3u);
                hi = __builtin_amdgcn_alignbyte(d2, d1, (unsigned int)addr & 3u);
            };
            for (int rr = 0; rr < nrows; ++rr) {
                const int base = rr * CS_ROW_BYTES + (int)((a_lo + (unsigned int)rr * a_step) & 15u);
                unsigned int loA, hiA, loB, hiB;
                taps(base + oA, loA, hiA);
                taps(base + oB, loB, hiB);
                unsigned int *o = reinterpret_cast<unsigned int *>(s_h + rr * HS) + lane;
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const unsigned int va = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiA, loA, selA + 0x00010001u * c3)), wA, 0u, false) >> 4;
                    const unsigned int vb = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiB, loB, selB + 0x00010001u * c3)), wB, 0u, false) >> 4;
                    o[c3 * (OW / 2)] = va | (vb << 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // vertical pass + normalisation + store: lane = (row of the mini-band, group of 8 px)
        const int ry = lane >> 4, x_base = (lane & (GROUPS - 1)) * 8;
        const int row = mb * mbh + ry, y = row0 + row;
        if (ry < mbh && row < rows_chunk) {
            T px[8][3];
            {
                const unsigned int yi = (unsigned int)s_y[row * 2], yw = (unsigned int)s_y[row * 2 + 1];
                const unsigned short *h0 = s_h + ((yi & 0xffffu) - r_lo) * HS + x_base;       // planar: [channel][x]
                const unsigned short *h1 = s_h + ((yi >> 16) - r_lo) * HS + x_base;
                const unsigned int b0 = yw & 0xffffu, b1 = yw >> 16;
                unsigned int w0[12], w1[12];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(h0 + c * OW), v = *reinterpret_cast<const uint4 *>(h1 + c * OW);
                    w0[c * 4] = u.x; w0[c * 4 + 1] = u.y; w0[c * 4 + 2] = u.z; w0[c * 4 + 3] = u.w;
                    w1[c * 4] = v.x; w1[c * 4 + 1] = v.y; w1[c * 4 + 2] = v.z; w1[c * 4 + 3] = v.w;
                }
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    const int c = q >> 3, kk = q & 7;                           // channel plane, pixel of the group
                    const unsigned int a = (kk & 1) ? (w0[c * 4 + (kk >> 1)] >> 16) : (w0[c * 4 + (kk >> 1)] & 0xffffu);
                    const unsigned int c1 = (kk & 1) ? (w1[c * 4 + (kk >> 1)] >> 16) : (w1[c * 4 + (kk >> 1)] & 0xffffu);
                    const unsigned int xa = __umul24(b0, a), xb = __umul24(b1, c1);
                    unsigned int t;                     // t <= 1020 always (see crop_sep_kernel)
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(xa), "v"(xb));
                    px[kk][c] = s_lut[c * CS_LUT_N + t];
                }
            }
            if (swap_rb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
            }
            if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                    *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
                }
            } else if (mbh == WV_ROWS && (mb + 1) * mbh <= rows_chunk) {
                // a full mini-band is ONE contiguous 3 KB (fp32: 6 KB) block of the output: assemble it in the (now dead) staging rows + plane of this
                // wavefront and write it with store instructions of 64 x 16 consecutive bytes -- stored straight from the registers every instruction scatters 16-byte pieces
                // 48 bytes apart and the memory system sees three partial writes per cache line
                T *o = reinterpret_cast<T *>(s_rows) + ((size_t)ry * OW + x_base) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                    *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + row0 + mb * mbh) * OW * 3);
                const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
#pragma unroll
                for (int k = 0; k < 3 * (int)sizeof(T) / 2; ++k) stream_store(g + k * WAVE + lane, l4[k * WAVE + lane]);
            } else {
                T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                    *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the next mini-band overwrites staging rows and plane
        __builtin_amdgcn_wave_barrier();
    };
    if (mb_lo < mb_hi) fetch(mb_lo, X);
    if (mb_lo + 1 < mb_hi) fetch(mb_lo + 1, Y);
    for (int mb = mb_lo; mb < mb_hi; mb += 2) {
        mini_band(mb, X);
        if (mb + 1 < mb_hi) mini_band(mb + 1, Y);
    }
}

// crop_wave2_kernel (r03): crop_wave_kernel's arithmetic and launch shape with the memory pipeline re-ordered around ONE fact read from its
// ISA: gfx950 counts loads and stores on the same vmcnt, they complete out of order with respect to each other, so the compiler waits for
// a load with `s_waitcnt vmcnt(0)` whenever a store is pending -- crop_wave_kernel therefore drained the 3 KB of stores it had JUST issued (and
// the prefetch behind them) at the top of every mini-band. Here the source loads are inline asm with a hand-placed `s_waitcnt vmcnt(3)`
// (three younger loads in flight => this set has landed, whatever the stores do), and the next mini-band's rows are staged BEFORE the
// current mini-band's stores are issued, so a store has a full mini-band of arithmetic to complete before anything can wait on it.
template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) crop_wave2_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                          const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                          int OH, const T *__restrict__ lut_g, float m0, float m1, float m2,
                                                          float d0, float d1, float d2, T *__restrict__ out, int swap_rb, int nwg)
{
    constexpr int OW = 128, HS = OW * 3, GROUPS = OW / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_y[CF_BANDS * CS_BAND * 2];         // per output row of the chunk: (source row y0 | y1 << 16) relative to the crop, (b0 | b1 << 16)
    __shared__ CropPar s_par;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the mini-band loop is scalar control flow
    int wg;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int bands = (OH + CS_BAND - 1) / CS_BAND, chunks = (bands + CF_BANDS - 1) / CF_BANDS;
    const int slot = wg / chunks, chunk = wg - slot * chunks;
    const int b = slot / max_n, i = slot - b * max_n;
    if (i >= counts[b]) return;                         // padding slot: left untouched
    const int band0 = chunk * CF_BANDS, nbands = min(CF_BANDS, bands - band0);
    int2 *s_xc = reinterpret_cast<int2 *>(s_dyn);
    T *s_lut = reinterpret_cast<T *>(s_dyn + OW * 8);
    unsigned char *s_rows = s_dyn + OW * 8 + ((3 * CS_LUT_N * sizeof(T) + 15) & ~(size_t)15) + (size_t)wv * WV_WAVE_LDS;      // this wavefront's staging area ...
    unsigned short *s_h = reinterpret_cast<unsigned short *>(s_rows + WV_SRC * CS_ROW_BYTES);                                  // ... and 16-bit plane
    const size_t frame_off = (size_t)b * H * W * 3;
    // ---- set-up, once per workgroup: wave 0 geometry + x table, waves 1-2 the y tables of the chunk, wave 3 the normalisation table
    if (tid < WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const bool valid = (r > l) && (bt > t);
        const int cw = r - l, ch = bt - t;
        const bool wide_ok = cw * 3 + STAGE_PAD <= CS_ROW_BYTES;
        if (valid && wide_ok) {
            const double scale_x = (double)cw / (double)OW;
            for (int x = tid; x < OW; x += WAVE) {
                const Coef cx = cv_coef_s(x, cw, scale_x, true);
                s_xc[x] = make_int2((cx.s * 3) | ((cx.s + 1 < cw ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
            }
        }
        if (tid == 0) { CropPar p; p.l = l; p.t = t; p.cw = cw; p.ch = ch; p.r_lo = 0; p.nrows = 0; p.staged = wide_ok; p.valid = valid; s_par = p; }
    } else if (tid < 3 * WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const int ch = bt - t, row = tid - WAVE, y = band0 * CS_BAND + row;
        if (bt > t && r > l && y < OH && row < nbands * CS_BAND) {
            const Coef cy = cv_coef_s(y, ch, (double)ch / (double)OH, false);
            s_y[row * 2] = clampi(cy.s, 0, ch - 1) | (clampi(cy.s + 1, 0, ch - 1) << 16);
            s_y[row * 2 + 1] = (cy.w0 & 0xffff) | (cy.w1 << 16);
        }
    } else {
        const int n16 = (int)(3 * CS_LUT_N * sizeof(T) / 16);
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *d = reinterpret_cast<uint4 *>(s_lut);
        for (int c = tid - 3 * WAVE; c < n16; c += WAVE) d[c] = g[c];
    }
    __syncthreads();
    const CropPar par = s_par;
    const bool valid = par.valid != 0;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const int row0 = band0 * CS_BAND, rows_chunk = min(nbands * CS_BAND, OH - row0);       // output rows of this workgroup
    // mini-bands of 4 rows while the crop is not taller than the output (<= 6 source rows each), of 2 rows up to twice as tall
    const int mbh = par.ch <= OH ? WV_ROWS : 2;
    const int n_mb = (rows_chunk + mbh - 1) / mbh, mb_per_wave = (n_mb + NWAVES - 1) / NWAVES;
    const int mb_lo = wv * mb_per_wave, mb_hi = min(n_mb, mb_lo + mb_per_wave);
    const int cmax = (par.cw * 3 + 30) >> 4;             // 16-byte chunks per staged row
    // ---- is every mini-band of this wavefront on the fast path? (staged crop, <= WV_SRC source rows, <= 3 x 64 chunks, no load near the end of the frames)
    bool fast = valid && par.staged && par.ch <= 2 * OH;
    if (fast) {
        fast = frames + frame_off + ((size_t)(par.t + par.ch - 1) * W + par.l) * 3 + 34 * 16 <= gend;
        for (int mb = mb_lo + lane; mb < mb_hi; mb += WAVE) {
            const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
            const int n = (int)((unsigned int)s_y[rb * 2] >> 16) - (s_y[ra * 2] & 0xffff) + 1;
            if (n > WV_SRC || n * cmax > 3 * WAVE) fast = false;
        }
        fast = __all(fast);
    }
    if (!fast) {
        // rare: invalid / very tall / very wide crops, the last rows of the last frame -- direct sampling, a unit (row, 8 px) per lane
        for (int mb = mb_lo; mb < mb_hi; ++mb)
            for (int u = lane; u < mbh * GROUPS; u += WAVE) {
                const int ry = u >> 4, x_base = (u & (GROUPS - 1)) * 8, y = row0 + mb * mbh + ry;
                if (y >= OH || mb * mbh + ry >= rows_chunk) continue;
                if (valid) crop_direct_unit<T, LAYOUT>(frames + frame_off + ((size_t)par.t * W + par.l) * 3, W, par.ch, par.cw, OH, OW, y, x_base, m0, m1, m2, d0, d1, d2,
                                                       swap_rb, out, (size_t)slot);
                else
                    for (int k = 0; k < 8; ++k)
                        for (int c = 0; c < 3; ++c) {
                            if (LAYOUT == LAYOUT_NCHW) out[(((size_t)slot * 3 + c) * OH + y) * OW + x_base + k] = cvt<T>(0.f);
                            else out[(((size_t)slot * OH + y) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
                        }
            }
        return;
    }
    // ---- fast path: no call, no workgroup barrier below this line
    const unsigned char *crop0 = frames + frame_off + ((size_t)par.t * W + par.l) * 3;
    const unsigned int a_step = ((unsigned int)W * 3u) & 15u;
    // horizontal pass: this lane's two adjacent x (2 lane, 2 lane + 1). The two taps of the three channels of an x are 6 consecutive source bytes
    // (3 when the right tap is clamped onto the left one): three ALIGNED dword reads + v_alignbyte bring them, v_perm_b32 with per-lane selectors
    // builds the (left, right) pairs for v_dot2 -- byte-granular ds_read_u8 taps cost 9 x the LDS time (profiles/r02_lds_microbench.txt), and in this
    // kernel the LDS pipe is the shared resource the sixteen free-running wavefronts of a CU compete for
    // source rows of a mini-band -> three registers per lane (flat sweep over (row, chunk)); unconditional loads: lanes past the band re-read its first
    // chunk. TWO mini-bands are kept in flight (register sets X and Y): the wait for the rows of mini-band m then has the loads of m + 1 behind it, and
    // the rule "reads and writes complete out of order with respect to each other" no longer forces it to drain the stores of m - 1 just issued
    struct RowRegs { tlk_u32x4 a, b, c; int r_lo, nrows; };
    RowRegs X{tlk_u32x4{0, 0, 0, 0}, tlk_u32x4{0, 0, 0, 0}, tlk_u32x4{0, 0, 0, 0}, 0, 0}, Y = X;
    // The three loads of a mini-band are inline asm: the compiler does not track them, so it cannot turn the wait for them into the
    // `s_waitcnt vmcnt(0)` it must use whenever loads AND stores are pending on the one counter gfx950 has for both (that full drain at the top of
    // every mini-band -- the stores of the previous one included -- was the un-overlapped "skeleton" of crop_wave_kernel, r03 ISA reading).
    // wait_rows() is the hand-placed wait: loads return in order among themselves, so with three younger loads always issued (the next
    // fetch; past the end a repeat of the last one) "at most 3 operations outstanding" implies this set has landed, whatever the stores do.
    // lane -> (source row of the mini-band, 16-byte chunk of that row) for its three load slots: fixed for the whole chunk of the crop, so the
    // two integer divisions per slot are done ONCE here, not in every fetch and every stage (r03: they were a fifth of the kernel's instructions)
    int sl_rr[3], sl_c[3], sl_lds[3];
    unsigned int sl_goff[3];
    const unsigned int W3 = (unsigned int)W * 3u;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int idx = lane + q * WAVE;
        sl_rr[q] = idx / cmax; sl_c[q] = idx - sl_rr[q] * cmax;
        sl_lds[q] = sl_rr[q] * CS_ROW_BYTES + sl_c[q] * 16;
        sl_goff[q] = (unsigned int)sl_rr[q] * W3;           // <= 5 rows: fits 32 bits
    }
    const int cw3 = par.cw * 3;
    auto fetch = [&](int mb_req, RowRegs &R) {
        const int mb = min(mb_req, mb_hi - 1);           // past the wavefront's last mini-band: fetch that one again (L2 hits) -- every wait then has its three younger loads
        const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
        const int r_lo_n = __builtin_amdgcn_readfirstlane(s_y[ra * 2] & 0xffff);
        const int nrows_n = __builtin_amdgcn_readfirstlane((int)((unsigned int)s_y[rb * 2] >> 16)) - r_lo_n + 1;
        const unsigned char *rowp = crop0 + (size_t)r_lo_n * W3;      // wave-uniform
        auto addr = [&](int q) {
            const bool in = sl_rr[q] < nrows_n;
            const unsigned char *g0 = rowp + (in ? sl_goff[q] : 0u);
            const int mis = (int)((uintptr_t)g0 & 15);
            return g0 - mis + (size_t)((in && sl_c[q] < ((mis + cw3 + 15) >> 4)) ? sl_c[q] : 0) * 16;
        };
        const unsigned char *p0 = addr(0), *p1 = addr(1), *p2 = addr(2);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.a) : "v"(p0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.b) : "v"(p1));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.c) : "v"(p2));
        R.r_lo = r_lo_n; R.nrows = nrows_n;
    };
    auto wait_rows = [&](RowRegs &R) {                   // ONE form of the wait (two forms would meet in a phi: register copies of loads still in flight)
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(R.a), "+v"(R.b), "+v"(R.c));
    };
    int st_r_lo = 0, st_nrows = 0;                       // the mini-band whose rows are in this wavefront's staging area
    auto stage = [&](const RowRegs &R) {
        const int nrows = R.nrows;
        if (sl_rr[0] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[0]) = R.a;
        if (sl_rr[1] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[1]) = R.b;
        if (sl_rr[2] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[2]) = R.c;
        st_r_lo = R.r_lo; st_nrows = nrows;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int4 xc2 = *reinterpret_cast<const int4 *>(&s_xc[2 * lane]);
    const int oA = xc2.x & 0xffff, oB = xc2.z & 0xffff;
    const unsigned int selA = 0x0c000c00u | ((unsigned int)(xc2.x >> 16) << 16), selB = 0x0c000c00u | ((unsigned int)(xc2.z >> 16) << 16);
    const us2_t wA = __builtin_bit_cast(us2_t, xc2.y), wB = __builtin_bit_cast(us2_t, xc2.w);
    // one mini-band: rows of `mb` are staged; N holds (or will hold) the rows of mb + 1. Order: horizontal pass, vertical pass, output block
    // assembled in LDS and read back into registers, THEN wait for N + stage it + issue the fetch of mb + 3, and only then the stores of mb:
    // a store has a whole mini-band of arithmetic to complete before the next wait can stall on it
    auto mini_band = [&](int mb, RowRegs &N) {
        const int r_lo = st_r_lo, nrows = st_nrows;
        {
            const unsigned int a_lo = (unsigned int)(uintptr_t)(crop0 + (size_t)r_lo * W * 3) & 15u;
            auto taps = [&](int addr, unsigned int &lo, unsigned int &hi) {
                const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                const unsigned int d0 = q[0], d1 = q[1], d2 = q[2];
                lo = __builtin_amdgcn_alignbyte(d1, d0, (unsigned int)addr & 3u);
                hi = __builtin_amdgcn_alignbyte(d2, d1, (unsigned int)addr & 3u);
            };
            for (int rr = 0; rr < nrows; ++rr) {
                const int base = rr * CS_ROW_BYTES + (int)((a_lo + (unsigned int)rr * a_step) & 15u);
                unsigned int loA, hiA, loB, hiB;
                taps(base + oA, loA, hiA);
                taps(base + oB, loB, hiB);
                unsigned int *o = reinterpret_cast<unsigned int *>(s_h + rr * HS) + lane;
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const unsigned int va = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiA, loA, selA + 0x00010001u * c3)), wA, 0u, false) >> 4;
                    const unsigned int vb = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiB, loB, selB + 0x00010001u * c3)), wB, 0u, false) >> 4;
                    o[c3 * (OW / 2)] = va | (vb << 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ry = lane >> 4, x_base = (lane & (GROUPS - 1)) * 8;
        const int row = mb * mbh + ry, y = row0 + row;
        const bool act = ry < mbh && row < rows_chunk;
        const bool block = LAYOUT == LAYOUT_NHWC && mbh == WV_ROWS && (mb + 1) * mbh <= rows_chunk;      // wave-uniform: one contiguous output block
        T px[8][3];
        if (act) {
            const unsigned int yi = (unsigned int)s_y[row * 2], yw = (unsigned int)s_y[row * 2 + 1];
            const unsigned short *h0 = s_h + ((yi & 0xffffu) - r_lo) * HS + x_base;       // planar: [channel][x]
            const unsigned short *h1 = s_h + ((yi >> 16) - r_lo) * HS + x_base;
            const unsigned int b0 = yw & 0xffffu, b1 = yw >> 16;
            unsigned int w0[12], w1[12];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint4 u = *reinterpret_cast<const uint4 *>(h0 + c * OW), v = *reinterpret_cast<const uint4 *>(h1 + c * OW);
                w0[c * 4] = u.x; w0[c * 4 + 1] = u.y; w0[c * 4 + 2] = u.z; w0[c * 4 + 3] = u.w;
                w1[c * 4] = v.x; w1[c * 4 + 1] = v.y; w1[c * 4 + 2] = v.z; w1[c * 4 + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const int c = q >> 3, kk = q & 7;
                const unsigned int a = (kk & 1) ? (w0[c * 4 + (kk >> 1)] >> 16) : (w0[c * 4 + (kk >> 1)] & 0xffffu);
                const unsigned int c1 = (kk & 1) ? (w1[c * 4 + (kk >> 1)] >> 16) : (w1[c * 4 + (kk >> 1)] & 0xffffu);
                const unsigned int xa = __umul24(b0, a), xb = __umul24(b1, c1);
                unsigned int t;                     // t <= 1020 always (see crop_sep_kernel)
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(xa), "v"(xb));
                px[kk][c] = s_lut[c * CS_LUT_N + t];
            }
            if (swap_rb) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
            }
        }
        constexpr int NST = 3 * (int)sizeof(T) / 2;      // 16-byte stores per lane of a full mini-band block
        uint4 blk[NST];
        if (block) {
            // assemble the mini-band's contiguous block in the (now dead) staging rows + plane, read it back as whole cache lines per instruction
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            T *o = reinterpret_cast<T *>(s_rows) + ((size_t)ry * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
#pragma unroll
            for (int k = 0; k < NST; ++k) blk[k] = l4[k * WAVE + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // staging rows and plane are dead from here
        __builtin_amdgcn_wave_barrier();
        if (mb + 1 < mb_hi) {
            wait_rows(N);
            stage(N);
            fetch(mb + 3, N);
        }
        if (block) {
            uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + row0 + mb * mbh) * OW * 3);
#pragma unroll
            for (int k = 0; k < NST; ++k) stream_store(g + k * WAVE + lane, blk[k]);
        } else if (act) {
            if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                    *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
                }
            } else {
                T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                    *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                }
            }
        }
    };
    if (mb_lo < mb_hi) {
        fetch(mb_lo, X);
        fetch(mb_lo + 1, Y);
        wait_rows(X);
        stage(X);
        fetch(mb_lo + 2, X);
    }
    for (int mb = mb_lo; mb < mb_hi; mb += 2) {
        mini_band(mb, Y);
        if (mb + 1 < mb_hi) mini_band(mb + 1, X);
    }
}


// crop_wave3_kernel (r03): crop_wave2_kernel with a SLIDING WINDOW over the source rows. PMC of crop_wave2_kernel (profiles/r03_crop_pmc.txt):
// VALU 63 % and LDS 68 % busy, no memory stall left to remove -- the kernel is bound by its own instruction count. A wavefront walks its eight
// mini-bands top to bottom and consecutive mini-bands share source rows; here the 16-bit plane is a ring of WV_SRC rows addressed by
// (source row mod WV_SRC), so a source row is fetched, staged and taken through the horizontal pass ONCE per wavefront range.
// P16 (r03 late): the frames' row pitch W * 3 is a multiple of 16 bytes (1920-, 1280-, 640-wide frames): every source row of a crop then has the SAME
// misalignment, so (1) a lane's global offset for each of its load slots is a constant of the crop and the loads take the SGPR-base + 32-bit-VGPR-offset
// form (three 64-bit pointer computations per fetch gone), (2) the tap windows' aligned LDS offsets and byte shifts are constants of the crop (ten address
// instructions per staged row gone); and the R/B swap is done by the order in which the output block is assembled, not by register copies. The kernel is
// VALU-issue bound (profiles/r03_crop_pmc.txt): these are ~60 of its ~375 vector instructions per mini-band.
template <typename T, int LAYOUT, bool P16, bool PREFETCH3 = false>
__global__ void __launch_bounds__(BLOCK) crop_wave3_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                          const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                          int OH, const T *__restrict__ lut_g, float m0, float m1, float m2,
                                                          float d0, float d1, float d2, T *__restrict__ out, int swap_flags, int nwg)
{
    const int swap_rb = swap_flags & 1;                  // bit 1 of swap_flags (experiment, TLK_CROP_NT=0): plain instead of streaming stores
    const bool plain_stores = (swap_flags & 2) != 0;
    constexpr int OW = 128, HS = OW * 3, GROUPS = OW / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_y[CF_BANDS * CS_BAND * 2];         // per output row of the chunk: (source row y0 | y1 << 16) relative to the crop, (b0 | b1 << 16)
    __shared__ CropPar s_par;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the mini-band loop is scalar control flow
    int wg;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int bands = (OH + CS_BAND - 1) / CS_BAND, chunks = (bands + CF_BANDS - 1) / CF_BANDS;
    const int slot = wg / chunks, chunk = wg - slot * chunks;
    const int b = slot / max_n, i = slot - b * max_n;
    if (i >= counts[b]) return;                         // padding slot: left untouched
    const int band0 = chunk * CF_BANDS, nbands = min(CF_BANDS, bands - band0);
    int2 *s_xc = reinterpret_cast<int2 *>(s_dyn);
    T *s_lut = reinterpret_cast<T *>(s_dyn + OW * 8);
    unsigned char *s_rows = s_dyn + OW * 8 + ((3 * CS_LUT_N * sizeof(T) + 15) & ~(size_t)15) + (size_t)wv * WV_WAVE_LDS;      // this wavefront's staging area ...
    unsigned short *s_h = reinterpret_cast<unsigned short *>(s_rows + WV_SRC * CS_ROW_BYTES);                                  // ... and 16-bit plane
    const size_t frame_off = (size_t)b * H * W * 3;
    // ---- set-up, once per workgroup: wave 0 geometry + x table, waves 1-2 the y tables of the chunk, wave 3 the normalisation table
    if (tid < WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const bool valid = (r > l) && (bt > t);
        const int cw = r - l, ch = bt - t;
        const bool wide_ok = cw * 3 + STAGE_PAD <= CS_ROW_BYTES;
        if (valid && wide_ok) {
            const double scale_x = (double)cw / (double)OW;
            for (int x = tid; x < OW; x += WAVE) {
                const Coef cx = cv_coef_s(x, cw, scale_x, true);
                s_xc[x] = make_int2((cx.s * 3) | ((cx.s + 1 < cw ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
            }
        }
        if (tid == 0) { CropPar p; p.l = l; p.t = t; p.cw = cw; p.ch = ch; p.r_lo = 0; p.nrows = 0; p.staged = wide_ok; p.valid = valid; s_par = p; }
    } else if (tid < 3 * WAVE) {
        int l, t, r, bt;
        crop_ltrb(boxes + ((size_t)b * max_n + i) * 4, W, H, l, t, r, bt);
        const int ch = bt - t, row = tid - WAVE, y = band0 * CS_BAND + row;
        if (bt > t && r > l && y < OH && row < nbands * CS_BAND) {
            const Coef cy = cv_coef_s(y, ch, (double)ch / (double)OH, false);
            const int y0 = clampi(cy.s, 0, ch - 1), y1 = clampi(cy.s + 1, 0, ch - 1);      // < 2048: bits 12-14 / 28-30 carry the rows' slots in the ring plane
            s_y[row * 2] = y0 | ((y0 % WV_SRC) << 12) | (y1 << 16) | ((y1 % WV_SRC) << 28);
            s_y[row * 2 + 1] = (cy.w0 & 0xffff) | (cy.w1 << 16);
        }
    } else {
        const int n16 = (int)(3 * CS_LUT_N * sizeof(T) / 16);
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *d = reinterpret_cast<uint4 *>(s_lut);
        for (int c = tid - 3 * WAVE; c < n16; c += WAVE) d[c] = g[c];
    }
    __syncthreads();
    const CropPar par = s_par;
    const bool valid = par.valid != 0;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const int row0 = band0 * CS_BAND, rows_chunk = min(nbands * CS_BAND, OH - row0);       // output rows of this workgroup
    // mini-bands of 4 rows while the crop is not taller than the output (<= 6 source rows each), of 2 rows up to twice as tall
    const int mbh = par.ch <= OH ? WV_ROWS : 2;
    const int n_mb = (rows_chunk + mbh - 1) / mbh, mb_per_wave = (n_mb + NWAVES - 1) / NWAVES;
    const int mb_lo = wv * mb_per_wave, mb_hi = min(n_mb, mb_lo + mb_per_wave);
    const int cmax = (par.cw * 3 + 30) >> 4;             // 16-byte chunks per staged row
    // ---- is every mini-band of this wavefront on the fast path? (staged crop, <= WV_SRC source rows, <= 3 x 64 chunks, no load near the end of the frames)
    bool fast = valid && par.staged && par.ch <= 2 * OH;
    if (fast) {
        fast = frames + frame_off + ((size_t)(par.t + par.ch - 1) * W + par.l) * 3 + 34 * 16 <= gend;
        for (int mb = mb_lo + lane; mb < mb_hi; mb += WAVE) {
            const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
            const int n = (int)(((unsigned int)s_y[rb * 2] >> 16) & 0x7ffu) - (s_y[ra * 2] & 0x7ff) + 1;
            if (n > WV_SRC || n * cmax > 3 * WAVE) fast = false;
        }
        fast = __all(fast);
    }
    if (!fast) {
        // rare: invalid / very tall / very wide crops, the last rows of the last frame -- direct sampling, a unit (row, 8 px) per lane
        for (int mb = mb_lo; mb < mb_hi; ++mb)
            for (int u = lane; u < mbh * GROUPS; u += WAVE) {
                const int ry = u >> 4, x_base = (u & (GROUPS - 1)) * 8, y = row0 + mb * mbh + ry;
                if (y >= OH || mb * mbh + ry >= rows_chunk) continue;
                if (valid) crop_direct_unit<T, LAYOUT>(frames + frame_off + ((size_t)par.t * W + par.l) * 3, W, par.ch, par.cw, OH, OW, y, x_base, m0, m1, m2, d0, d1, d2,
                                                       swap_rb, out, (size_t)slot);
                else
                    for (int k = 0; k < 8; ++k)
                        for (int c = 0; c < 3; ++c) {
                            if (LAYOUT == LAYOUT_NCHW) out[(((size_t)slot * 3 + c) * OH + y) * OW + x_base + k] = cvt<T>(0.f);
                            else out[(((size_t)slot * OH + y) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
                        }
            }
        return;
    }
    // ---- fast path: no call, no workgroup barrier below this line
    const unsigned char *crop0 = frames + frame_off + ((size_t)par.t * W + par.l) * 3;
    const unsigned int a_step = ((unsigned int)W * 3u) & 15u;
    // horizontal pass: this lane's two adjacent x (2 lane, 2 lane + 1). The two taps of the three channels of an x are 6 consecutive source bytes
    // (3 when the right tap is clamped onto the left one): three ALIGNED dword reads + v_alignbyte bring them, v_perm_b32 with per-lane selectors
    // builds the (left, right) pairs for v_dot2 -- byte-granular ds_read_u8 taps cost 9 x the LDS time (profiles/r02_lds_microbench.txt), and in this
    // kernel the LDS pipe is the shared resource the sixteen free-running wavefronts of a CU compete for
    // source rows of a mini-band -> three registers per lane (flat sweep over (row, chunk)); unconditional loads: lanes past the band re-read its first
    // chunk. TWO mini-bands are kept in flight (register sets X and Y): the wait for the rows of mini-band m then has the loads of m + 1 behind it, and
    // the rule "reads and writes complete out of order with respect to each other" no longer forces it to drain the stores of m - 1 just issued
    struct RowRegs { tlk_u32x4 a, b, c; int r_lo, nrows; };
    RowRegs X{tlk_u32x4{0, 0, 0, 0}, tlk_u32x4{0, 0, 0, 0}, tlk_u32x4{0, 0, 0, 0}, 0, 0}, Y = X, Z = X;
    // DEPTH mini-bands of source rows in flight (register sets X, Y[, Z]). Counters with COLD sources (profiles/r03_bytekernels_pmc.txt: waves wait 54 % of
    // their cycles, 33 % warm) say the kernel waits for memory inside the pipeline: the 16-byte-pitch variant, whose address arithmetic left room in the
    // register file, keeps three sets in flight
    constexpr int DEPTH = PREFETCH3 ? 3 : 2;
    // The three loads of a mini-band are inline asm: the compiler does not track them, so it cannot turn the wait for them into the
    // `s_waitcnt vmcnt(0)` it must use whenever loads AND stores are pending on the one counter gfx950 has for both (that full drain at the top of
    // every mini-band -- the stores of the previous one included -- was the un-overlapped "skeleton" of crop_wave_kernel, r03 ISA reading).
    // wait_rows() is the hand-placed wait: loads return in order among themselves, so with three younger loads always issued (the next
    // fetch; past the end a repeat of the last one) "at most 3 operations outstanding" implies this set has landed, whatever the stores do.
    // lane -> (source row of the mini-band, 16-byte chunk of that row) for its three load slots: fixed for the whole chunk of the crop, so the
    // two integer divisions per slot are done ONCE here, not in every fetch and every stage (r03: they were a fifth of the kernel's instructions)
    int sl_rr[3], sl_c[3], sl_lds[3];
    unsigned int sl_goff[3];
    const unsigned int W3 = (unsigned int)W * 3u;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int idx = lane + q * WAVE;
        sl_rr[q] = idx / cmax; sl_c[q] = idx - sl_rr[q] * cmax;
        sl_lds[q] = sl_rr[q] * CS_ROW_BYTES + sl_c[q] * 16;
        sl_goff[q] = (unsigned int)sl_rr[q] * W3;           // <= 5 rows: fits 32 bits
    }
    const int cw3 = par.cw * 3;
    const int mis0 = (int)((uintptr_t)crop0 & 15);       // P16: the misalignment of EVERY source row of this crop
    unsigned int sl_off[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) sl_off[q] = sl_c[q] < ((mis0 + cw3 + 15) >> 4) ? sl_goff[q] + (unsigned int)sl_c[q] * 16u : 0u;
    auto fetch = [&](int mb_req, RowRegs &R) {
        const int mb = min(mb_req, mb_hi - 1);           // past the wavefront's last mini-band: fetch that one again (L2 hits) -- every wait then has its three younger loads
        const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
        // sliding window: the rows this wavefront's PREVIOUS mini-band already took through the horizontal pass are still in the ring plane
        // (consecutive mini-bands share one or two source rows; up-scaling crops 30-40 % of them)
        const int r_first = __builtin_amdgcn_readfirstlane(s_y[ra * 2] & 0x7ff);
        const int r_last = __builtin_amdgcn_readfirstlane((int)(((unsigned int)s_y[rb * 2] >> 16) & 0x7ffu));
        const int done = mb > mb_lo ? __builtin_amdgcn_readfirstlane((int)(((unsigned int)s_y[(ra - 1) * 2] >> 16) & 0x7ffu)) : -1;
        const int r_lo_n = max(r_first, done + 1);
        const int nrows_n = r_last - r_lo_n + 1;          // 0 .. WV_SRC new rows
        const unsigned char *rowp = crop0 + (size_t)min(r_lo_n, par.ch - 1) * W3;      // wave-uniform (no new row: r_lo_n may be one past the crop -- never address it)
        if constexpr (P16) {
            const unsigned char *base = rowp - mis0;     // SGPR pair, 16-byte aligned; lane offsets are constants of the crop
            const unsigned int o0 = sl_rr[0] < nrows_n ? sl_off[0] : 0u, o1 = sl_rr[1] < nrows_n ? sl_off[1] : 0u, o2 = sl_rr[2] < nrows_n ? sl_off[2] : 0u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(R.a) : "v"(o0), "s"(base));
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(R.b) : "v"(o1), "s"(base));
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(R.c) : "v"(o2), "s"(base));
            R.r_lo = r_lo_n; R.nrows = nrows_n;
            return;
        }
        auto addr = [&](int q) {
            const bool in = sl_rr[q] < nrows_n;
            const unsigned char *g0 = rowp + (in ? sl_goff[q] : 0u);
            const int mis = (int)((uintptr_t)g0 & 15);
            return g0 - mis + (size_t)((in && sl_c[q] < ((mis + cw3 + 15) >> 4)) ? sl_c[q] : 0) * 16;
        };
        const unsigned char *p0 = addr(0), *p1 = addr(1), *p2 = addr(2);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.a) : "v"(p0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.b) : "v"(p1));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.c) : "v"(p2));
        R.r_lo = r_lo_n; R.nrows = nrows_n;
    };
    auto wait_rows = [&](RowRegs &R) {                   // ONE form of the wait (two forms would meet in a phi: register copies of loads still in flight)
        if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(6)" : "+v"(R.a), "+v"(R.b), "+v"(R.c));
        else asm volatile("s_waitcnt vmcnt(3)" : "+v"(R.a), "+v"(R.b), "+v"(R.c));
    };
    int st_r_lo = 0, st_nrows = 0;                       // the mini-band whose rows are in this wavefront's staging area
    auto stage = [&](const RowRegs &R) {
        const int nrows = R.nrows;
        if (sl_rr[0] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[0]) = R.a;
        if (sl_rr[1] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[1]) = R.b;
        if (sl_rr[2] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[2]) = R.c;
        st_r_lo = R.r_lo; st_nrows = nrows;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int4 xc2 = *reinterpret_cast<const int4 *>(&s_xc[2 * lane]);
    const int oA = xc2.x & 0xffff, oB = xc2.z & 0xffff;
    const int oA4 = (oA + mis0) & ~3, oB4 = (oB + mis0) & ~3;                      // P16: aligned LDS offset and byte shift of the two tap windows
    const unsigned int shA = (unsigned int)(oA + mis0) & 3u, shB = (unsigned int)(oB + mis0) & 3u;
    const unsigned int selA = 0x0c000c00u | ((unsigned int)(xc2.x >> 16) << 16), selB = 0x0c000c00u | ((unsigned int)(xc2.z >> 16) << 16);
    const us2_t wA = __builtin_bit_cast(us2_t, xc2.y), wB = __builtin_bit_cast(us2_t, xc2.w);
    // one mini-band: rows of `mb` are staged; N holds (or will hold) the rows of mb + 1. Order: horizontal pass, vertical pass, output block
    // assembled in LDS and read back into registers, THEN wait for N + stage it + issue the fetch of mb + 3, and only then the stores of mb:
    // a store has a whole mini-band of arithmetic to complete before the next wait can stall on it
    auto mini_band = [&](int mb, RowRegs &N) {
        const int r_lo = st_r_lo, nrows = st_nrows;
        {
            const unsigned int a_lo = (unsigned int)(uintptr_t)(crop0 + (size_t)r_lo * W * 3) & 15u;
            auto taps = [&](int addr, unsigned int &lo, unsigned int &hi) {
                const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                const unsigned int d0 = q[0], d1 = q[1], d2 = q[2];
                lo = __builtin_amdgcn_alignbyte(d1, d0, (unsigned int)addr & 3u);
                hi = __builtin_amdgcn_alignbyte(d2, d1, (unsigned int)addr & 3u);
            };
            for (int rr = 0; rr < nrows; ++rr) {
                unsigned int loA, hiA, loB, hiB;
                if constexpr (P16) {
                    const unsigned int *qa = reinterpret_cast<const unsigned int *>(s_rows + rr * CS_ROW_BYTES + oA4);
                    const unsigned int *qb = reinterpret_cast<const unsigned int *>(s_rows + rr * CS_ROW_BYTES + oB4);
                    const unsigned int a0 = qa[0], a1 = qa[1], a2 = qa[2], b0_ = qb[0], b1_ = qb[1], b2_ = qb[2];
                    loA = __builtin_amdgcn_alignbyte(a1, a0, shA); hiA = __builtin_amdgcn_alignbyte(a2, a1, shA);
                    loB = __builtin_amdgcn_alignbyte(b1_, b0_, shB); hiB = __builtin_amdgcn_alignbyte(b2_, b1_, shB);
                } else {
                    const int base = rr * CS_ROW_BYTES + (int)((a_lo + (unsigned int)rr * a_step) & 15u);
                    taps(base + oA, loA, hiA);
                    taps(base + oB, loB, hiB);
                }
                unsigned int *o = reinterpret_cast<unsigned int *>(s_h + ((r_lo + rr) % WV_SRC) * HS) + lane;
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const unsigned int va = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiA, loA, selA + 0x00010001u * c3)), wA, 0u, false) >> 4;
                    const unsigned int vb = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiB, loB, selB + 0x00010001u * c3)), wB, 0u, false) >> 4;
                    o[c3 * (OW / 2)] = va | (vb << 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ry = lane >> 4, x_base = (lane & (GROUPS - 1)) * 8;
        const int row = mb * mbh + ry, y = row0 + row;
        const bool act = ry < mbh && row < rows_chunk;
        static_assert(sizeof(T) == 2, "crop_wave3_kernel assembles its 3 KB output block in the staging rows only: 16-bit element types");
        const bool block = LAYOUT == LAYOUT_NHWC && mbh == WV_ROWS && (mb + 1) * mbh <= rows_chunk;      // wave-uniform: one contiguous output block
        T px[8][3];
        if (act) {
            const unsigned int yi = (unsigned int)s_y[row * 2], yw = (unsigned int)s_y[row * 2 + 1];
            const unsigned short *h0 = s_h + ((yi >> 12) & 7u) * HS + x_base;             // planar: [channel][x]; row = its ring slot
            const unsigned short *h1 = s_h + ((yi >> 28) & 7u) * HS + x_base;
            const unsigned int b0 = yw & 0xffffu, b1 = yw >> 16;
            // output channel co <- source channel cs: the R/B swap is a wave-uniform choice of plane and table row, not a register shuffle afterwards
            // (the compiler had turned the swap into 16 selects + 8 copies per mini-band)
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                const int cs = swap_rb ? 2 - co : co;
                const uint4 u = *reinterpret_cast<const uint4 *>(h0 + cs * OW), v = *reinterpret_cast<const uint4 *>(h1 + cs * OW);
                const unsigned int w0[4] = {u.x, u.y, u.z, u.w}, w1[4] = {v.x, v.y, v.z, v.w};
                const T *lut_c = s_lut + cs * CS_LUT_N;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const unsigned int a = (kk & 1) ? (w0[kk >> 1] >> 16) : (w0[kk >> 1] & 0xffffu);
                    const unsigned int c1 = (kk & 1) ? (w1[kk >> 1] >> 16) : (w1[kk >> 1] & 0xffffu);
                    const unsigned int xa = __umul24(b0, a), xb = __umul24(b1, c1);
                    unsigned int t;                     // t <= 1020 always (see crop_sep_kernel)
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(xa), "v"(xb));
                    px[kk][co] = lut_c[t];
                }
            }
        }
        constexpr int NST = 3 * (int)sizeof(T) / 2;      // 16-byte stores per lane of a full mini-band block
        uint4 blk[NST];
        if (block) {
            // assemble the mini-band's contiguous block in the (now dead) staging rows + plane, read it back as whole cache lines per instruction
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            T *o = reinterpret_cast<T *>(s_rows) + ((size_t)ry * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
#pragma unroll
            for (int k = 0; k < NST; ++k) blk[k] = l4[k * WAVE + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // staging rows and plane are dead from here
        __builtin_amdgcn_wave_barrier();
        if (mb + 1 < mb_hi) {
            wait_rows(N);
            stage(N);
            fetch(mb + 1 + DEPTH, N);
        }
        if (block) {
            uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + row0 + mb * mbh) * OW * 3);
            if (plain_stores) {
#pragma unroll
                for (int k = 0; k < NST; ++k) g[k * WAVE + lane] = blk[k];
            } else {
#pragma unroll
                for (int k = 0; k < NST; ++k) stream_store(g + k * WAVE + lane, blk[k]);
            }
        } else if (act) {
            if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                    *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
                }
            } else {
                T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                    *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                }
            }
        }
    };
    if (mb_lo < mb_hi) {
        fetch(mb_lo, X);
        fetch(mb_lo + 1, Y);
        if constexpr (DEPTH == 3) fetch(mb_lo + 2, Z);
        wait_rows(X);
        stage(X);
        fetch(mb_lo + DEPTH, X);
    }
    if constexpr (DEPTH == 3) {
        for (int mb = mb_lo; mb < mb_hi; mb += 3) {
            mini_band(mb, Y);
            if (mb + 1 < mb_hi) mini_band(mb + 1, Z);
            if (mb + 2 < mb_hi) mini_band(mb + 2, X);
        }
    } else {
        for (int mb = mb_lo; mb < mb_hi; mb += 2) {
            mini_band(mb, Y);
            if (mb + 1 < mb_hi) mini_band(mb + 1, X);
        }
    }
}



// ---------------------------------------------------------------------------------------------
// crop_pw_kernel (r03, TLK_CROP_WAVE=4; measured EQUAL to crop_wave3_kernel -- 187-196 vs 187-191 us -- and therefore not the default): PERSISTENT WAVEFRONTS. Every earlier crop kernel launched a workgroup per (crop, 128 output rows): 7 000 workgroups
// each paid geometry + two tables + a 6 KB table copy behind a barrier before its first byte moved, and resident wavefronts averaged 62 % of
// the 16 per CU (PMC, profiles/r03_crop_pmc.txt). Here the grid is 2 workgroups of 8 wavefronts per CU for the whole launch; the
// normalisation table is copied ONCE per workgroup (the only barrier of the kernel); the VALID crops of the batch (counts[] prefix, padding
// slots are never visited) are cut into 4-row mini-bands and every wavefront owns ONE contiguous range of them -- about 55 mini-bands, i.e.
// half a crop, so geometry and the x table are rebuilt once or twice per wavefront, the y coefficients live in a 64-row ring refilled 32 rows
// ahead, and the sliding window of crop_wave3_kernel (a source row is fetched, staged and taken through the horizontal pass once) runs
// uninterrupted over the whole range. Mini-band arithmetic, hand-placed waits and store order are crop_wave3_kernel's. 16-bit outputs.
// ---------------------------------------------------------------------------------------------
constexpr int PW_WAVES = 8, PW_BLOCK = PW_WAVES * WAVE;
constexpr int PW_XTAB = 128 * 8, PW_YTAB = 64 * 8, PW_ROWS = WV_SRC * CS_ROW_BYTES, PW_PLANE = WV_SRC * 128 * 6;
constexpr int PW_WAVE_LDS = PW_XTAB + PW_YTAB + PW_ROWS + PW_PLANE;

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(PW_BLOCK, 4) crop_pw_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                           const float *__restrict__ boxes, const int *__restrict__ counts, int max_n,
                                                           int OH, const T *__restrict__ lut_g, float m0, float m1, float m2,
                                                           float d0, float d1, float d2, T *__restrict__ out, int swap_rb)
{
    static_assert(sizeof(T) == 2, "the 3 KB output block of a mini-band is assembled in the staging rows: 16-bit element types");
    constexpr int OW = 128, HS = OW * 3, GROUPS = OW / 8;
    constexpr int LUT_BYTES = (3 * CS_LUT_N * (int)sizeof(T) + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    T *s_lut = reinterpret_cast<T *>(s_dyn);
    unsigned char *wbase = s_dyn + LUT_BYTES + (size_t)wv * PW_WAVE_LDS;
    int2 *s_xc = reinterpret_cast<int2 *>(wbase);
    int *s_y = reinterpret_cast<int *>(wbase + PW_XTAB);                         // ring of 64 output rows: [(row & 63) * 2] = y0 | slot0 << 12 | y1 << 16 | slot1 << 28, [+ 1] = b0 | b1 << 16
    unsigned char *s_rows = wbase + PW_XTAB + PW_YTAB;
    unsigned short *s_h = reinterpret_cast<unsigned short *>(s_rows + PW_ROWS);
    {   // the only workgroup-wide step: the (u8 -> normalised T) table
        const int n16 = LUT_BYTES / 16;
        const uint4 *g = reinterpret_cast<const uint4 *>(lut_g);
        uint4 *d = reinterpret_cast<uint4 *>(s_lut);
        for (int c = tid; c < n16; c += PW_BLOCK) d[c] = g[c];
    }
    __syncthreads();
    int n_valid = 0;
    for (int b = 0; b < B; ++b) n_valid += counts[b];
    const int mbpc = OH / WV_ROWS;                                                 // 4-row units per crop (the host guarantees OH % 4 == 0)
    const long long total = (long long)n_valid * mbpc;
    const int nw = (int)gridDim.x * PW_WAVES, wg = (int)blockIdx.x * PW_WAVES + wv;
    long long u = total * wg / nw;
    const long long u_end = total * (wg + 1) / nw;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const unsigned int W3 = (unsigned int)W * 3u, a_step = W3 & 15u;
    while (u < u_end) {
        const int v = (int)(u / mbpc), ua = (int)(u - (long long)v * mbpc);
        const int ue = (int)min((long long)mbpc, ua + (u_end - u));                // units [ua, ue) of valid crop v
        u += ue - ua;
        int b = 0, rem = v;
        while (rem >= counts[b]) { rem -= counts[b]; ++b; }
        const int slot = b * max_n + rem;
        int l, t, r, bt;
        crop_ltrb(boxes + (size_t)slot * 4, W, H, l, t, r, bt);
        l = __builtin_amdgcn_readfirstlane(l); t = __builtin_amdgcn_readfirstlane(t);
        r = __builtin_amdgcn_readfirstlane(r); bt = __builtin_amdgcn_readfirstlane(bt);
        const int cw = r - l, ch = bt - t;
        const bool valid = cw > 0 && ch > 0;
        const size_t frame_off = (size_t)b * H * W * 3;
        const unsigned char *crop0 = frames + frame_off + ((size_t)t * W + l) * 3;
        const bool fast = valid && cw * 3 + STAGE_PAD <= CS_ROW_BYTES && ch <= 2 * OH &&
                          frames + frame_off + ((size_t)(t + ch - 1) * W + l) * 3 + 34 * 16 <= gend;
        const int row_a = ua * WV_ROWS, row_e = ue * WV_ROWS;
        if (!fast) {
            // rare: invalid / very tall / very wide crops, the last rows of the last frame -- direct sampling, a unit (row, 8 px) per lane
            for (int row = row_a + (lane >> 4); row < row_e; row += WAVE / GROUPS) {
                const int x_base = (lane & (GROUPS - 1)) * 8;
                if (valid) crop_direct_unit<T, LAYOUT>(crop0, W, ch, cw, OH, OW, row, x_base, m0, m1, m2, d0, d1, d2, swap_rb, out, (size_t)slot);
                else
                    for (int k = 0; k < 8; ++k)
                        for (int c = 0; c < 3; ++c) {
                            if (LAYOUT == LAYOUT_NCHW) out[(((size_t)slot * 3 + c) * OH + row) * OW + x_base + k] = cvt<T>(0.f);
                            else out[(((size_t)slot * OH + row) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
                        }
            }
            continue;
        }
        // ---- per crop range: x table, first 64 rows of the y ring, lane -> (row, chunk) map of the three load slots
        const double scale_x = (double)cw / (double)OW, scale_y = (double)ch / (double)OH;
        for (int x = lane; x < OW; x += WAVE) {
            const Coef cx = cv_coef_s(x, cw, scale_x, true);
            s_xc[x] = make_int2((cx.s * 3) | ((cx.s + 1 < cw ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
        }
        auto fill_y = [&](int first) {               // ring entries of output rows [first, first + 64) (first % 32 == 0: lanes 0..31 overwrite the older half when called for 32)
            const int y = first + lane;
            if (y < OH) {
                const Coef cy = cv_coef_s(y, ch, scale_y, false);
                const int y0 = clampi(cy.s, 0, ch - 1), y1 = clampi(cy.s + 1, 0, ch - 1);
                s_y[(y & 63) * 2] = y0 | ((y0 % WV_SRC) << 12) | (y1 << 16) | ((y1 % WV_SRC) << 28);
                s_y[(y & 63) * 2 + 1] = (cy.w0 & 0xffff) | (cy.w1 << 16);
            }
        };
        int y_tab = row_a & ~31;                      // the ring holds rows [y_tab, y_tab + 64)
        fill_y(y_tab);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int mbh = ch <= OH ? WV_ROWS : 2;       // mini-bands of 4 rows (<= 6 source rows each while the crop is not taller than the output), of 2 rows up to twice as tall
        const int mb_lo = row_a / mbh, mb_hi = row_e / mbh;
        const int cmax = (cw * 3 + 30) >> 4;          // 16-byte chunks per staged row (<= 32: three load slots cover 6 rows)
        const int cw3 = cw * 3;
        int sl_rr[3], sl_c[3], sl_lds[3];
        unsigned int sl_goff[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = lane + q * WAVE;
            sl_rr[q] = idx / cmax; sl_c[q] = idx - sl_rr[q] * cmax;
            sl_lds[q] = sl_rr[q] * CS_ROW_BYTES + sl_c[q] * 16;
            sl_goff[q] = (unsigned int)sl_rr[q] * W3;
        }
        auto ytab = [&](int row) -> const int * { return s_y + (row & 63) * 2; };
        struct RowRegs { tlk_u32x4 a, b, c; int r_lo, nrows; };
        RowRegs X{tlk_u32x4{0, 0, 0, 0}, tlk_u32x4{0, 0, 0, 0}, tlk_u32x4{0, 0, 0, 0}, 0, 0}, Y = X;
        auto fetch = [&](int mb_req, RowRegs &R) {    // inline-asm loads + hand-placed vmcnt(3): see crop_wave2_kernel; only the NEW source rows: see crop_wave3_kernel
            const int mb = min(mb_req, mb_hi - 1);
            const int ra = mb * mbh, rb = ra + mbh - 1;
            const int r_first = __builtin_amdgcn_readfirstlane(ytab(ra)[0] & 0x7ff);
            const int r_last = __builtin_amdgcn_readfirstlane((int)(((unsigned int)ytab(rb)[0] >> 16) & 0x7ffu));
            const int done = mb > mb_lo ? __builtin_amdgcn_readfirstlane((int)(((unsigned int)ytab(ra - 1)[0] >> 16) & 0x7ffu)) : -1;
            const int r_lo_n = max(r_first, done + 1);
            const int nrows_n = r_last - r_lo_n + 1;
            const unsigned char *rowp = crop0 + (size_t)min(r_lo_n, ch - 1) * W3;      // (no new row: r_lo_n may be one past the crop -- never address it)
            auto addr = [&](int q) {
                const bool in = sl_rr[q] < nrows_n;
                const unsigned char *g0 = rowp + (in ? sl_goff[q] : 0u);
                const int mis = (int)((uintptr_t)g0 & 15);
                return g0 - mis + (size_t)((in && sl_c[q] < ((mis + cw3 + 15) >> 4)) ? sl_c[q] : 0) * 16;
            };
            const unsigned char *p0 = addr(0), *p1 = addr(1), *p2 = addr(2);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.a) : "v"(p0));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.b) : "v"(p1));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.c) : "v"(p2));
            R.r_lo = r_lo_n; R.nrows = nrows_n;
        };
        auto wait_rows = [&](RowRegs &R) { asm volatile("s_waitcnt vmcnt(3)" : "+v"(R.a), "+v"(R.b), "+v"(R.c)); };
        int st_r_lo = 0, st_nrows = 0;
        auto stage = [&](const RowRegs &R) {
            const int nrows = R.nrows;
            if (sl_rr[0] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[0]) = R.a;
            if (sl_rr[1] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[1]) = R.b;
            if (sl_rr[2] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[2]) = R.c;
            st_r_lo = R.r_lo; st_nrows = nrows;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        const int4 xc2 = *reinterpret_cast<const int4 *>(&s_xc[2 * lane]);
        const int oA = xc2.x & 0xffff, oB = xc2.z & 0xffff;
        const unsigned int selA = 0x0c000c00u | ((unsigned int)(xc2.x >> 16) << 16), selB = 0x0c000c00u | ((unsigned int)(xc2.z >> 16) << 16);
        const us2_t wA = __builtin_bit_cast(us2_t, xc2.y), wB = __builtin_bit_cast(us2_t, xc2.w);
        auto mini_band = [&](int mb, RowRegs &N) {
            const int r_lo = st_r_lo, nrows = st_nrows;
            {
                const unsigned int a_lo = (unsigned int)(uintptr_t)(crop0 + (size_t)r_lo * W3) & 15u;
                auto taps = [&](int addr, unsigned int &lo, unsigned int &hi) {
                    const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                    const unsigned int e0 = q[0], e1 = q[1], e2 = q[2];
                    lo = __builtin_amdgcn_alignbyte(e1, e0, (unsigned int)addr & 3u);
                    hi = __builtin_amdgcn_alignbyte(e2, e1, (unsigned int)addr & 3u);
                };
                for (int rr = 0; rr < nrows; ++rr) {
                    const int base = rr * CS_ROW_BYTES + (int)((a_lo + (unsigned int)rr * a_step) & 15u);
                    unsigned int loA, hiA, loB, hiB;
                    taps(base + oA, loA, hiA);
                    taps(base + oB, loB, hiB);
                    unsigned int *o = reinterpret_cast<unsigned int *>(s_h + ((r_lo + rr) % WV_SRC) * HS) + lane;
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) {
                        const unsigned int va = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiA, loA, selA + 0x00010001u * c3)), wA, 0u, false) >> 4;
                        const unsigned int vb = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hiB, loB, selB + 0x00010001u * c3)), wB, 0u, false) >> 4;
                        o[c3 * (OW / 2)] = va | (vb << 16);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int ry = lane >> 4, x_base = (lane & (GROUPS - 1)) * 8;
            const int y = mb * mbh + ry;
            const bool act = ry < mbh;
            const bool block = LAYOUT == LAYOUT_NHWC && mbh == WV_ROWS;                   // wave-uniform: the mini-band is one contiguous 3 KB output block
            T px[8][3];
            if (act) {
                const unsigned int yi = (unsigned int)ytab(y)[0], yw = (unsigned int)ytab(y)[1];
                const unsigned short *h0 = s_h + ((yi >> 12) & 7u) * HS + x_base;             // planar: [channel][x]; row = its ring slot
                const unsigned short *h1 = s_h + ((yi >> 28) & 7u) * HS + x_base;
                const unsigned int b0 = yw & 0xffffu, b1 = yw >> 16;
                unsigned int w0[12], w1[12];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint4 p = *reinterpret_cast<const uint4 *>(h0 + c * OW), q = *reinterpret_cast<const uint4 *>(h1 + c * OW);
                    w0[c * 4] = p.x; w0[c * 4 + 1] = p.y; w0[c * 4 + 2] = p.z; w0[c * 4 + 3] = p.w;
                    w1[c * 4] = q.x; w1[c * 4 + 1] = q.y; w1[c * 4 + 2] = q.z; w1[c * 4 + 3] = q.w;
                }
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    const int c = q >> 3, kk = q & 7;
                    const unsigned int a = (kk & 1) ? (w0[c * 4 + (kk >> 1)] >> 16) : (w0[c * 4 + (kk >> 1)] & 0xffffu);
                    const unsigned int c1 = (kk & 1) ? (w1[c * 4 + (kk >> 1)] >> 16) : (w1[c * 4 + (kk >> 1)] & 0xffffu);
                    const unsigned int xa = __umul24(b0, a), xb = __umul24(b1, c1);
                    unsigned int tt;                    // <= 1020 always (see crop_sep_kernel)
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(tt) : "v"(xa), "v"(xb));
                    px[kk][c] = s_lut[c * CS_LUT_N + tt];
                }
                if (swap_rb) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
                }
            }
            constexpr int NST = 3 * (int)sizeof(T) / 2;
            uint4 blk[NST];
            if (block) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                T *o = reinterpret_cast<T *>(s_rows) + ((size_t)ry * OW + x_base) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                    *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
#pragma unroll
                for (int k = 0; k < NST; ++k) blk[k] = l4[k * WAVE + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the staging rows are dead from here (the ring plane lives on)
            __builtin_amdgcn_wave_barrier();
            if (mb + 1 < mb_hi) {
                const int nxt = (mb + 1) * mbh;                         // entering the second half of the y ring: refill the half behind (rows 32 ahead of it)
                if (nxt >= y_tab + 32) { y_tab += 32; if (lane < 32) fill_y(y_tab + 32); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
                wait_rows(N);
                stage(N);
                fetch(mb + 3, N);
            }
            if (block) {
                uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + (size_t)mb * mbh) * OW * 3);
#pragma unroll
                for (int k = 0; k < NST; ++k) stream_store(g + k * WAVE + lane, blk[k]);
            } else if (act) {
                if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        Pack<T, 8> p;
#pragma unroll
                        for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                        *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
                    }
                } else {
                    T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        Pack<T, 8> p;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                        *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
                    }
                }
            }
        };
        fetch(mb_lo, X);
        fetch(mb_lo + 1, Y);
        wait_rows(X);
        stage(X);
        fetch(mb_lo + 2, X);
        for (int mb = mb_lo; mb < mb_hi; mb += 2) {
            mini_band(mb, Y);
            if (mb + 1 < mb_hi) mini_band(mb + 1, X);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain before the next range re-uses the register sets (its waits count from a clean slate)
    }
}

// ---------------------------------------------------------------------------------------------
// Plain StrongSORT's ReID input: crop ori_img[y1:y2, x1:x2] of the int-truncated, clipped box
// (strong_sort.py:102-108, :135-141) -> Pillow Image.resize(BILINEAR) -> ToTensor -> Normalize
// (reid_multibackend.py:44-52, :184-195). Pillow's resample (src/libImaging/Resample.c) is separable with an 8-bit
// intermediate: horizontal pass (support = max(1, scale) source pixels either side, weights normalised, 22-bit fixed
// point, rounded and clipped to uint8), then the same vertically. Workgroup = (slot, band of PIL_BAND output rows):
// stage the source rows of the band in LDS, run the horizontal pass into a second LDS plane, then the vertical pass +
// normalisation straight to 16-byte stores. Crops too large for the LDS planes take the direct (recompute) branch.
// ---------------------------------------------------------------------------------------------
constexpr int PIL_BITS = 32 - 8 - 2;
constexpr int PIL_BAND = 32;                          // (16 rows measured slower: 428 vs 356 us -- twice the per-band set-up)
constexpr int PIL_KMAX = 5;                           // taps per axis handled from LDS tables: scale <= 2
constexpr int PIL_BPW = 4;                            // bands per workgroup
constexpr int PIL_KPAD = 8;                           // coefficient rows padded to 32 bytes: one ds_read_b128 + one b32 per row
constexpr int PIL_ROWS = 40;                          // staged source rows per band
constexpr int PIL_ROW_BYTES = 544;                    // as CROP_LDS_ROW_BYTES: crops up to 170 px wide
constexpr int PIL_OW_MAX = 128;

struct PilAxis { double scale, support, ss; int ksize; };
__host__ __device__ __forceinline__ PilAxis pil_axis(int inSize, int outSize)
{
    PilAxis a;
    a.scale = (double)inSize / (double)outSize;
    const double fs = a.scale < 1.0 ? 1.0 : a.scale;
    a.support = 1.0 * fs;
    a.ss = 1.0 / fs;
    a.ksize = (int)ceil(a.support) * 2 + 1;
    return a;
}
__host__ __device__ __forceinline__ void pil_bounds(const PilAxis &a, int inSize, int xx, int &xmin, int &xmax)
{
    const double center = 0.0 + (xx + 0.5) * a.scale;
    xmin = (int)(center - a.support + 0.5);
    if (xmin < 0) xmin = 0;
    xmax = (int)(center + a.support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
}
__host__ __device__ __forceinline__ double pil_tri(const PilAxis &a, int xx, int xmin, int x)
{
    const double center = 0.0 + (xx + 0.5) * a.scale;
    double v = (x + xmin - center + 0.5) * a.ss;
    if (v < 0.0) v = -v;
    return v < 1.0 ? 1.0 - v : 0.0;
}
__host__ __device__ __forceinline__ double pil_wsum(const PilAxis &a, int xx, int xmin, int xmax)
{
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pil_tri(a, xx, xmin, x);
    return ww;
}
__host__ __device__ __forceinline__ int pil_fixed(const PilAxis &a, int xx, int xmin, int x, double ww)
{
    double w = pil_tri(a, xx, xmin, x);
    if (ww != 0.0) w /= ww;
    return (int)(0.5 + w * (double)(1 << PIL_BITS));
}
__device__ __forceinline__ int pil_clip8(int v) { v >>= PIL_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ void ssort_crop_box(const double *xyxy, int W, int H, int &x1, int &y1, int &x2, int &y2)
{
    const double x = (xyxy[0] + xyxy[2]) / 2, y = (xyxy[1] + xyxy[3]) / 2, w = xyxy[2] - xyxy[0], h = xyxy[3] - xyxy[1];
    x1 = (int)(x - w / 2); x2 = (int)(x + w / 2); y1 = (int)(y - h / 2); y2 = (int)(y + h / 2);
    x1 = x1 > 0 ? x1 : 0; y1 = y1 > 0 ? y1 : 0;
    x2 = x2 < W - 1 ? x2 : W - 1; y2 = y2 < H - 1 ? y2 : H - 1;
}

// one horizontally resampled uint8 sample (3 channels) of source row `row` (global memory) at output column xx
__device__ __forceinline__ void pil_hsample(const unsigned char *__restrict__ row, const PilAxis &ax, int cw, int xx, int (&o)[3])
{
    int xmin, xmax;
    pil_bounds(ax, cw, xx, xmin, xmax);
    const double ww = pil_wsum(ax, xx, xmin, xmax);
    int s0 = 1 << (PIL_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int k = pil_fixed(ax, xx, xmin, x, ww);
        const unsigned char *p = row + (size_t)(x + xmin) * 3;
        s0 += (int)p[0] * k; s1 += (int)p[1] * k; s2 += (int)p[2] * k;
    }
    o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
}

// Image.resize's own rule (PIL/Image.py, Pillow 12.2.0: `if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]`): an image more than 100
// times taller than wide that shrinks vertically is resized VERTICALLY first and horizontally afterwards -- the uint8 rounding between the passes
// happens in the other order. A 3 x 301 px box; found by the r03 sweep fixture (tests/golden/pil_sweep.npz). Such crops take the direct path.
__host__ __device__ __forceinline__ bool pil_vertical_first(int cw, int ch, int OH) { return ch > cw * 100 && OH < ch; }

// one output pixel (3 channels, already clipped to 0..255) of the direct path, in Pillow's pass order for this crop; rolled loops on purpose
__device__ __forceinline__ void pil_direct_px(const unsigned char *__restrict__ base, int W, const PilAxis &ax, const PilAxis &ay, int cw, int ch, int OH,
                                              int y, int x, int (&s)[3])
{
    int ymin, ymax;
    pil_bounds(ay, ch, y, ymin, ymax);
    const double wwy = pil_wsum(ay, y, ymin, ymax);
    s[0] = s[1] = s[2] = 1 << (PIL_BITS - 1);
    if (!pil_vertical_first(cw, ch, OH)) {
        // every vertical tap recomputes its horizontally resampled (and uint8-rounded) sample from global memory
#pragma nounroll
        for (int t = 0; t < ymax; ++t) {
            const int kv = pil_fixed(ay, y, ymin, t, wwy);
            int hv[3];
            pil_hsample(base + (size_t)(ymin + t) * W * 3, ax, cw, x, hv);
            s[0] += hv[0] * kv; s[1] += hv[1] * kv; s[2] += hv[2] * kv;
        }
    } else {
        // every horizontal tap recomputes its vertically resampled (and uint8-rounded) sample
        int xmin, xmax;
        pil_bounds(ax, cw, x, xmin, xmax);
        const double wwx = pil_wsum(ax, x, xmin, xmax);
#pragma nounroll
        for (int xt = 0; xt < xmax; ++xt) {
            const int kh = pil_fixed(ax, x, xmin, xt, wwx);
            int v0 = 1 << (PIL_BITS - 1), v1 = v0, v2 = v0;
#pragma nounroll
            for (int t = 0; t < ymax; ++t) {
                const int kv = pil_fixed(ay, y, ymin, t, wwy);
                const unsigned char *p = base + ((size_t)(ymin + t) * W + (xmin + xt)) * 3;
                v0 += (int)p[0] * kv; v1 += (int)p[1] * kv; v2 += (int)p[2] * kv;
            }
            s[0] += pil_clip8(v0) * kh; s[1] += pil_clip8(v1) * kh; s[2] += pil_clip8(v2) * kh;
        }
    }
    s[0] = pil_clip8(s[0]); s[1] = pil_clip8(s[1]); s[2] = pil_clip8(s[2]);
}

// byte q of a little-endian word array (constant q: the shift folds into an SDWA byte select of the multiply)
template <int NW> __device__ __forceinline__ int byte_of(const unsigned (&w)[NW], int q) { return (int)((w[q >> 2] >> ((q & 3) * 8)) & 0xffu); }

// OWC: the output width as a compile-time constant (128: the ReID input of every tracker here), 0 = the run-time OW
template <typename T, int LAYOUT, int OWC>
__global__ void __launch_bounds__(BLOCK) pil_crop_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                         const double *__restrict__ boxes, int box_stride, const int *__restrict__ counts,
                                                         int max_n, int OH, int OW_rt, float m0, float m1, float m2, float d0, float d1, float d2,
                                                         T *__restrict__ out, int swap_rb)
{
    const int OW = OWC ? OWC : OW_rt;
    // the source rows; afterwards the band's output on its way to coalesced stores (PIL_BAND rows x 128 px x 3 two-byte elements)
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[PIL_ROWS * PIL_ROW_BYTES > PIL_BAND * PIL_OW_MAX * 6 ? PIL_ROWS * PIL_ROW_BYTES : PIL_BAND * PIL_OW_MAX * 6];
    // (+ PIL_KMAX - 1 rows: the vertical pass reads all of its taps unconditionally, the ones past a row's support with weight 0)
    __shared__ __attribute__((aligned(16))) unsigned char s_h[(PIL_ROWS + PIL_KMAX - 1) * (PIL_OW_MAX * 3 + 16)];
    __shared__ int s_hmin[PIL_OW_MAX];
    __shared__ __attribute__((aligned(16))) int s_hk[PIL_OW_MAX][PIL_KPAD];
    __shared__ int s_vmin[PIL_BAND];
    __shared__ __attribute__((aligned(16))) int s_vk[PIL_BAND][PIL_KPAD];
    // ToTensor + Normalize of an 8-bit value, per SOURCE channel, with exactly the reference's float32 arithmetic ((v / 255) - mean) / std:
    // one table entry per (channel, value) instead of two IEEE divisions per output element
    __shared__ T s_lut[3][256];
    static_assert(BLOCK == 256, "the look-up table is built one 8-bit value per thread");
    const int HS = OW * 3 + 16;
    const int tid = threadIdx.x;
    const int bands = (OH + PIL_BAND - 1) / PIL_BAND;
    // workgroup = (crop, PIL_BPW consecutive bands): the look-up table, the geometry and the horizontal coefficient rows (fp64, one IEEE
    // division per tap) are set up ONCE and shared by the bands (r02a: per band -- ~30 % of the kernel's VALU instructions)
    const int chunks = (bands + PIL_BPW - 1) / PIL_BPW;
    const int slot = blockIdx.x / chunks, chunk = blockIdx.x - slot * chunks;
    const int b = slot / max_n, i = slot - b * max_n;
    const int groups_per_row = OW / 8;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {d0, d1, d2};
    bool valid = i < counts[b];
    if (!valid) return;                                  // padding slot: left untouched
#pragma unroll
    for (int c = 0; c < 3; ++c) { float f = (float)tid / 255.0f; f = f - mean[c]; f = f / stdv[c]; s_lut[c][tid] = cvt<T>(f); }
    int x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (valid) { ssort_crop_box(boxes + ((size_t)b * max_n + i) * box_stride, W, H, x1, y1, x2, y2); valid = (x2 > x1) && (y2 > y1); }
    const int cw = x2 - x1, ch = y2 - y1;
    const PilAxis ax = pil_axis(valid ? cw : 1, OW), ay = pil_axis(valid ? ch : 1, OH);
    const bool h_ok = valid && OW <= PIL_OW_MAX && cw * 3 + STAGE_PAD <= PIL_ROW_BYTES && ax.ksize <= PIL_KMAX && ay.ksize <= PIL_KMAX &&
                      !pil_vertical_first(cw, ch, OH);        // (Pillow resizes such a crop vertically first: direct branch)
    if (h_ok && tid < OW) {                                                     // horizontal coefficient rows
        int xmin, xmax;
        pil_bounds(ax, cw, tid, xmin, xmax);
        const double ww = pil_wsum(ax, tid, xmin, xmax);
        s_hmin[tid] = xmin * 3;
        for (int k = 0; k < PIL_KMAX; ++k) s_hk[tid][k] = k < xmax ? pil_fixed(ax, tid, xmin, k, ww) : 0;
    }
    for (int band = chunk * PIL_BPW; band < min(bands, (chunk + 1) * PIL_BPW); ++band) {
    const int y_base = band * PIL_BAND;
    const int nb = min(PIL_BAND, OH - y_base);
    bool staged = false;
    int r_lo = 0, nrows = 0;
    if (valid) {
        int lo0, n0_, lo1, n1_;
        pil_bounds(ay, ch, y_base, lo0, n0_);
        pil_bounds(ay, ch, y_base + nb - 1, lo1, n1_);
        r_lo = lo0; nrows = lo1 + n1_ - lo0;
        staged = h_ok && nrows <= PIL_ROWS;
    }
    if (valid && staged) {
        const unsigned char *gend = frames + (size_t)B * H * W * 3;
        const int cmax = (cw * 3 + 30) >> 4;
        for (int idx = tid; idx < nrows * cmax; idx += BLOCK) {                 // source rows of the band, one flat sweep
            const int rr = idx / cmax, c = idx - rr * cmax;
            const unsigned char *g0 = frames + ((size_t)b * H * W + (size_t)(y1 + r_lo + rr) * W + x1) * 3;
            const int mis = (int)((uintptr_t)g0 & 15);          // pointer ARITHMETIC keeps the global address space (an integer round trip makes the loads flat_load, which also count against lgkmcnt)
            const int chunks = (mis + cw * 3 + 15) >> 4;
            if (c < chunks) {
                const unsigned char *p = g0 - mis + (size_t)c * 16;
                unsigned char *lds = s_rows + rr * PIL_ROW_BYTES + c * 16;
                if (p + 16 <= gend) *reinterpret_cast<uint4 *>(lds) = *reinterpret_cast<const uint4 *>(p);
                else for (int k = 0; k < 16 && p + k < gend; ++k) lds[k] = p[k];
            }
        }
        if (tid >= BLOCK - PIL_BAND && tid - (BLOCK - PIL_BAND) < nb) {         // vertical coefficient rows (last wavefront's lanes)
            const int ry = tid - (BLOCK - PIL_BAND);
            int ymin, ymax;
            pil_bounds(ay, ch, y_base + ry, ymin, ymax);
            const double ww = pil_wsum(ay, y_base + ry, ymin, ymax);
            s_vmin[ry] = ymin - r_lo;
            for (int k = 0; k < PIL_KMAX; ++k) s_vk[ry][k] = k < ymax ? pil_fixed(ay, y_base + ry, ymin, k, ww) : 0;
        }
        __syncthreads();
        // horizontal pass -> 8-bit plane. Every tap is read (taps past a column's support carry weight 0), so the loop has no divergent
        // branches and no serialised LDS round trips: five aligned dword reads bring the 3 x 5 source bytes of the pixel, one 16-byte +
        // one 4-byte read its weights (r01 form: a branch and 3 narrow reads per tap, ~115 instructions and 5 LDS latencies per pixel)
        const unsigned a_lo = (unsigned)(uintptr_t)(frames + ((size_t)b * H * W + (size_t)(y1 + r_lo) * W + x1) * 3) & 15u, row_step = ((unsigned)W * 3u) & 15u;
        const bool wide_h = ax.ksize > 3;                                       // (uniform: support > 1, i.e. the crop is wider than OW)
        for (int idx = tid; idx < nrows * OW; idx += BLOCK) {
            const int rr = idx / OW, x = idx - rr * OW;
            const int off = rr * PIL_ROW_BYTES + (int)((a_lo + (unsigned)rr * row_step) & 15u) + s_hmin[x];
            // 15 bytes from a byte-granular address: five ALIGNED dwords + v_alignbyte (an unaligned 16-byte LDS read is serialised per lane:
            // 65 vs 17 LDS cycles per wavefront, tools/micro/lds_unaligned.hip). The address stays an offset into s_rows: through a pointer ->
            // integer -> pointer round trip the compiler loses the LDS address space and emits flat loads.
            unsigned w[4];
            {
                const unsigned *q = reinterpret_cast<const unsigned *>(s_rows + (off & ~3));
                const unsigned sh = (unsigned)off & 3u;
                const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh); w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
                w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); w[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
            }
            const int4 c03 = *reinterpret_cast<const int4 *>(&s_hk[x][0]);
            int s0 = 1 << (PIL_BITS - 1), s1 = s0, s2 = s0;
            s0 += __mul24(byte_of(w, 0), c03.x); s1 += __mul24(byte_of(w, 1), c03.x); s2 += __mul24(byte_of(w, 2), c03.x);
            s0 += __mul24(byte_of(w, 3), c03.y); s1 += __mul24(byte_of(w, 4), c03.y); s2 += __mul24(byte_of(w, 5), c03.y);
            s0 += __mul24(byte_of(w, 6), c03.z); s1 += __mul24(byte_of(w, 7), c03.z); s2 += __mul24(byte_of(w, 8), c03.z);
            if (wide_h) {
                const int c4 = s_hk[x][4];
                s0 += __mul24(byte_of(w, 9), c03.w); s1 += __mul24(byte_of(w, 10), c03.w); s2 += __mul24(byte_of(w, 11), c03.w);
                s0 += __mul24(byte_of(w, 12), c4); s1 += __mul24(byte_of(w, 13), c4); s2 += __mul24(byte_of(w, 14), c4);
            }
            unsigned char *o = s_h + rr * HS + x * 3;
            o[0] = (unsigned char)pil_clip8(s0); o[1] = (unsigned char)pil_clip8(s1); o[2] = (unsigned char)pil_clip8(s2);
        }
    }
    __syncthreads();
    // (the direct branch reads global memory only, so the source-row area is free for the output in every case)
    const bool use_lds_store = LAYOUT == LAYOUT_NHWC && ((size_t)OW * 3 * sizeof(T)) % 16 == 0 && (size_t)nb * OW * 3 * sizeof(T) <= sizeof(s_rows);
    for (int unit = tid; unit < PIL_BAND * groups_per_row; unit += BLOCK) {
        const int ry = unit / groups_per_row, x_base = (unit - ry * groups_per_row) * 8;
        const int y = y_base + ry;
        if (y >= OH) continue;
        T px[8][3];
        if (valid && staged) {
            const unsigned char *p = s_h + s_vmin[ry] * HS + x_base * 3;      // 8-byte aligned: HS and 24 are multiples of 8
            const int4 c03 = *reinterpret_cast<const int4 *>(&s_vk[ry][0]);
            const int c4 = s_vk[ry][4];
            const int kv[PIL_KMAX] = {c03.x, c03.y, c03.z, c03.w, c4};
            const int ks = ay.ksize;                                            // (uniform per crop; taps past a row's support weigh 0)
            int acc[24];
#pragma unroll
            for (int q = 0; q < 24; ++q) acc[q] = 1 << (PIL_BITS - 1);
#pragma unroll
            for (int k = 0; k < PIL_KMAX; ++k)
                if (k < 3 || k < ks) {
                    unsigned w[6];
                    __builtin_memcpy(w, __builtin_assume_aligned(p + k * HS, 8), 24);
#pragma unroll
                    for (int q = 0; q < 24; ++q) acc[q] += __mul24(byte_of(w, q), kv[k]);
                }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[k][c] = s_lut[c][pil_clip8(acc[k * 3 + c])];
        } else if (valid) {
            // direct branch (pil_direct_px: recompute per output pixel, in Pillow's pass order for this crop)
            const unsigned char *base = frames + ((size_t)b * H * W + (size_t)y1 * W + x1) * 3;
            for (int k = 0; k < 8; ++k) {
                int s[3];
                pil_direct_px(base, W, ax, ay, cw, ch, OH, y, x_base + k, s);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float f = (float)s[c] / 255.0f; f = f - mean[c]; f = f / stdv[c];
                    px[k][c] = cvt<T>(f);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[k][c] = cvt<T>(0.f);
        }
        if (swap_rb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
        }
        if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Pack<T, 8> p;
#pragma unroll
                for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)slot * 3 + c) * OH + y) * OW + x_base) = p;
            }
        } else if (use_lds_store) {      // as crop_sep_kernel: the band's output is ONE contiguous block; assemble it in the (dead) source-row area
            T *o = reinterpret_cast<T *>(s_rows) + ((size_t)ry * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        } else {
            T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        }
    }
    if (use_lds_store) {
        __syncthreads();
        const int n16 = (int)((size_t)nb * OW * 3 * sizeof(T) / 16);
        uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + y_base) * OW * 3);
        const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
        for (int c = tid; c < n16; c += BLOCK) stream_store(g + c, l4[c]);
    }
    __syncthreads();                                    // the next band re-uses every LDS area
    }
}

// ---------------------------------------------------------------------------------------------
// pil_wave_kernel (r03; 128-wide NHWC 16-bit targets = the ReID input of plain StrongSORT / BoT-SORT / Deep-OC-SORT): Pillow's resample with the
// structure of crop_wave3_kernel -- wavefronts that never meet after set-up, source rows prefetched two mini-bands ahead by inline-asm loads behind a
// hand-placed `s_waitcnt vmcnt(4)`, a RING of 8-bit rows so that every source row goes through the horizontal pass once per wavefront, the mini-band's
// contiguous 3 KB output block assembled in the dead staging rows and written as whole cache lines -- and the tap loops specialised on the wave-uniform
// tap counts (an up-scaled axis has exactly 2 taps per output pixel; pil_crop_kernel always ran 3 and 5). pil_crop_kernel (workgroup barriers between
// the staging, horizontal and vertical phases of a 32-row band) measured 0.32-0.35 of the HBM peak.
//   workgroup = (crop, 128 output rows): wave 0 the horizontal coefficient rows (fp64, one IEEE division per tap: Pillow's own arithmetic), waves 1-2 the
//   vertical ones of the 128 rows, wave 3 the (u8 -> normalised T) table; ONE barrier; then every wavefront owns 8 mini-bands of 4 output rows.
// Arithmetic = Resample.c's, bit for bit: 22-bit coefficients, uint8 rounding between the passes; the results of both passes cannot leave [0, 255]
// (non-negative weights whose rounded sum exceeds 2^22 by at most 3), so the clip is a no-op and is not executed.
// ---------------------------------------------------------------------------------------------
constexpr int PWV_SRC = 8;                            // ring slots = staged rows per mini-band at most
constexpr int PWV_NL = 4;                             // 16-byte loads per lane and fetch
constexpr int PWV_PLANE = 128 * 3;                    // bytes per ring row: planar [channel][x]
constexpr int PWV_WAVE_LDS = PWV_SRC * CS_ROW_BYTES + PWV_SRC * PWV_PLANE;
struct PilTab { int4 a, b; };                         // a = (first tap: byte offset | ring slot << 12 ... see users, k0, k1, k2), b = (k3, k4, taps, 0)

// (inlined with ROLLED loops: as a call its 130-register frame became the register count of pil_wave_kernel -- a callee's need is the caller's -- and
// cost the kernel its fourth wavefront per SIMD; rolled and inline it stays below the fast path's own 124)
template <typename T>
__device__ __forceinline__ void pil_direct_unit_nhwc(const unsigned char *__restrict__ base, int W, int cw, int ch, int OH, int OW, int y, int x_base,
                                                  float m0, float m1, float m2, float d0, float d1, float d2, int swap_rb, T *__restrict__ out, size_t slot)
{
    // pil_direct_px: recompute per output pixel, in Pillow's pass order for this crop (the direct branch of pil_crop_kernel)
    const PilAxis ax = pil_axis(cw, OW), ay = pil_axis(ch, OH);
    const float mean[3] = {m0, m1, m2}, stdv[3] = {d0, d1, d2};
    T *o = out + (((size_t)slot * OH + y) * OW + x_base) * 3;
#pragma nounroll
    for (int k = 0; k < 8; ++k) {                        // (rolled on purpose: a rare path must not set the register budget of the kernel that calls it)
        int s[3];
        pil_direct_px(base, W, ax, ay, cw, ch, OH, y, x_base + k, s);
        T px[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float f = (float)s[c] / 255.0f; f = f - mean[c]; f = f / stdv[c];
            px[c] = cvt<T>(f);
        }
        if (swap_rb) { const T t0 = px[0]; px[0] = px[2]; px[2] = t0; }
        o[k * 3] = px[0]; o[k * 3 + 1] = px[1]; o[k * 3 + 2] = px[2];
    }
}

// P16: the frames' row pitch is a multiple of 16 bytes -> constant lane offsets (SGPR-base loads) and constant tap-window shifts, as in crop_wave3_kernel
template <typename T, bool P16>
__global__ void __launch_bounds__(BLOCK) pil_wave_kernel(const unsigned char *__restrict__ frames, int B, int H, int W,
                                                        const double *__restrict__ boxes, int box_stride, const int *__restrict__ counts, int max_n,
                                                        int OH, float m0, float m1, float m2, float d0, float d1, float d2,
                                                        T *__restrict__ out, int swap_rb, int nwg)
{
    static_assert(sizeof(T) == 2, "16-bit element types");
    constexpr int OW = 128, GROUPS = OW / 8, CHUNK_ROWS = CF_BANDS * CS_BAND;
    __shared__ PilTab s_xt[OW];                          // horizontal coefficient rows
    __shared__ PilTab s_yt[CHUNK_ROWS];                  // vertical coefficient rows of this workgroup's output rows
    __shared__ T s_lut[3][256];
    __shared__ int s_nt[3];                              // tap counts: horizontal; vertical (two halves of the chunk)
    __shared__ __attribute__((aligned(16))) unsigned char s_wave[NWAVES * PWV_WAVE_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wg;
    {
        const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int chunks = (OH + CHUNK_ROWS - 1) / CHUNK_ROWS;
    const int slot = wg / chunks, chunk = wg - slot * chunks;
    const int b = slot / max_n, i = slot - b * max_n;
    if (i >= counts[b]) return;                         // padding slot: left untouched
    const int row0 = chunk * CHUNK_ROWS, rows_chunk = min(CHUNK_ROWS, OH - row0);
    int x1, y1, x2, y2;
    ssort_crop_box(boxes + ((size_t)b * max_n + i) * box_stride, W, H, x1, y1, x2, y2);
    const bool valid = (x2 > x1) && (y2 > y1);
    const int cw = x2 - x1, ch = y2 - y1;
    const PilAxis ax = pil_axis(valid ? cw : 1, OW), ay = pil_axis(valid ? ch : 1, OH);
    const bool tabs_ok = valid && cw * 3 + STAGE_PAD <= CS_ROW_BYTES && ax.ksize <= PIL_KMAX && ay.ksize <= PIL_KMAX &&
                         !pil_vertical_first(cw, ch, OH);     // (Pillow resizes such a crop vertically first: direct path)
    auto wave_max = [](int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        return v;
    };
    if (wv == 0) {
        int nt = 0;
        if (tabs_ok)
            for (int x = lane; x < OW; x += WAVE) {
                int xmin, xmax;
                pil_bounds(ax, cw, x, xmin, xmax);
                const double ww = pil_wsum(ax, x, xmin, xmax);
                int k[PIL_KMAX];
#pragma unroll
                for (int t = 0; t < PIL_KMAX; ++t) k[t] = t < xmax ? pil_fixed(ax, x, xmin, t, ww) : 0;
                s_xt[x].a = make_int4(xmin * 3, k[0], k[1], k[2]);
                s_xt[x].b = make_int4(k[3], k[4], xmax, 0);
                nt = max(nt, xmax);
            }
        nt = wave_max(nt);
        if (lane == 0) s_nt[0] = nt;
    } else if (wv < 3) {
        int nt = 0;
        const int row = (wv - 1) * WAVE + lane, y = row0 + row;
        if (tabs_ok && row < rows_chunk) {
            int ymin, ymax;
            pil_bounds(ay, ch, y, ymin, ymax);
            const double ww = pil_wsum(ay, y, ymin, ymax);
            int k[PIL_KMAX];
#pragma unroll
            for (int t = 0; t < PIL_KMAX; ++t) k[t] = t < ymax ? pil_fixed(ay, y, ymin, t, ww) : 0;
            s_yt[row].a = make_int4(ymin | ((ymin % PWV_SRC) << 12) | ((ymin + ymax - 1) << 16), k[0], k[1], k[2]);      // first row (< 2048), its ring slot, last row
            s_yt[row].b = make_int4(k[3], k[4], ymax, 0);
            nt = ymax;
        }
        nt = wave_max(nt);
        if (lane == 0) s_nt[wv] = nt;
    } else {
        const float mean[3] = {m0, m1, m2}, stdv[3] = {d0, d1, d2};
        for (int v = lane; v < 256; v += WAVE)
#pragma unroll
            for (int c = 0; c < 3; ++c) { float f = (float)v / 255.0f; f = f - mean[c]; f = f / stdv[c]; s_lut[c][v] = cvt<T>(f); }
    }
    __syncthreads();
    // ---- no workgroup barrier below this line
    const int nth = s_nt[0], ntv = max(s_nt[1], s_nt[2]);
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const size_t frame_off = (size_t)b * H * W * 3;
    const int mbh = ch * 5 <= OH * 6 ? WV_ROWS : 2;      // 4-row mini-bands up to a vertical scale of 1.2 (<= 8 source rows), 2-row ones up to 2
    const int n_mb = (rows_chunk + mbh - 1) / mbh, mb_per_wave = (n_mb + NWAVES - 1) / NWAVES;
    const int mb_lo = wv * mb_per_wave, mb_hi = min(n_mb, mb_lo + mb_per_wave);
    const int cmax = (cw * 3 + 30) >> 4;
    auto first_row = [&](int row) { return s_yt[row].a.x & 0x7ff; };
    auto last_row = [&](int row) { return (int)(((unsigned int)s_yt[row].a.x >> 16) & 0x7ffu); };
    bool fast = tabs_ok;
    if (fast) {
        fast = frames + frame_off + ((size_t)(y1 + ch - 1) * W + x1) * 3 + 34 * 16 <= gend;
        for (int mb = mb_lo + lane; mb < mb_hi; mb += WAVE) {
            const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
            const int n = last_row(rb) - first_row(ra) + 1;
            if (n > PWV_SRC || n * cmax > PWV_NL * WAVE) fast = false;
        }
        fast = __all(fast);
    }
    if (!fast) {
        for (int mb = mb_lo; mb < mb_hi; ++mb)
            for (int u = lane; u < mbh * GROUPS; u += WAVE) {
                const int ry = u >> 4, x_base = (u & (GROUPS - 1)) * 8, y = row0 + mb * mbh + ry;
                if (y >= OH || mb * mbh + ry >= rows_chunk) continue;
                if (valid) pil_direct_unit_nhwc<T>(frames + frame_off + ((size_t)y1 * W + x1) * 3, W, cw, ch, OH, OW, y, x_base, m0, m1, m2, d0, d1, d2, swap_rb, out, (size_t)slot);
                else
                    for (int k = 0; k < 24; ++k) out[(((size_t)slot * OH + y) * OW + x_base) * 3 + k] = cvt<T>(0.f);
            }
        return;
    }
    unsigned char *s_rows = s_wave + (size_t)wv * PWV_WAVE_LDS;          // this wavefront's staging rows ...
    unsigned char *s_h = s_rows + PWV_SRC * CS_ROW_BYTES;                // ... and its ring of 8-bit rows
    const unsigned char *crop0 = frames + frame_off + ((size_t)y1 * W + x1) * 3;
    const unsigned int W3 = (unsigned int)W * 3u, a_step = W3 & 15u;
    struct RowRegs { tlk_u32x4 v[PWV_NL]; int r_lo, nrows; };
    RowRegs X, Y;
#pragma unroll
    for (int q = 0; q < PWV_NL; ++q) { X.v[q] = tlk_u32x4{0, 0, 0, 0}; Y.v[q] = tlk_u32x4{0, 0, 0, 0}; }
    X.r_lo = X.nrows = Y.r_lo = Y.nrows = 0;
    int sl_rr[PWV_NL], sl_c[PWV_NL], sl_lds[PWV_NL];
    unsigned int sl_goff[PWV_NL];
#pragma unroll
    for (int q = 0; q < PWV_NL; ++q) {
        const int idx = lane + q * WAVE;
        sl_rr[q] = idx / cmax; sl_c[q] = idx - sl_rr[q] * cmax;
        sl_lds[q] = sl_rr[q] * CS_ROW_BYTES + sl_c[q] * 16;
        sl_goff[q] = (unsigned int)sl_rr[q] * W3;
    }
    const int cw3 = cw * 3;
    const int mis0 = (int)((uintptr_t)crop0 & 15);       // P16: the misalignment of EVERY source row of this crop
    unsigned int sl_off[PWV_NL];
#pragma unroll
    for (int q = 0; q < PWV_NL; ++q) sl_off[q] = sl_c[q] < ((mis0 + cw3 + 15) >> 4) ? sl_goff[q] + (unsigned int)sl_c[q] * 16u : 0u;
    auto fetch = [&](int mb_req, RowRegs &R) {
        const int mb = min(mb_req, mb_hi - 1);           // past the last mini-band: that one again -- every wait has its PWV_NL younger loads
        const int ra = mb * mbh, rb = min(ra + mbh, rows_chunk) - 1;
        const int r_first = __builtin_amdgcn_readfirstlane(first_row(ra));
        const int r_last = __builtin_amdgcn_readfirstlane(last_row(rb));
        const int done = mb > mb_lo ? __builtin_amdgcn_readfirstlane(last_row(ra - 1)) : -1;       // sliding window: rows the previous mini-band already put into the ring
        const int r_lo_n = max(r_first, done + 1);
        const int nrows_n = max(0, r_last - r_lo_n + 1);
        const unsigned char *rowp = crop0 + (size_t)min(r_lo_n, ch - 1) * W3;      // (no new row: r_lo_n may be one past the crop -- never address it)
        if constexpr (P16) {
            const unsigned char *base = rowp - mis0;     // SGPR pair, 16-byte aligned
#pragma unroll
            for (int q = 0; q < PWV_NL; ++q) {
                const unsigned int o = sl_rr[q] < nrows_n ? sl_off[q] : 0u;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(R.v[q]) : "v"(o), "s"(base));
            }
        } else {
#pragma unroll
            for (int q = 0; q < PWV_NL; ++q) {
                const bool in = sl_rr[q] < nrows_n;
                const unsigned char *g0 = rowp + (in ? sl_goff[q] : 0u);
                const int mis = (int)((uintptr_t)g0 & 15);
                const unsigned char *pp = g0 - mis + (size_t)((in && sl_c[q] < ((mis + cw3 + 15) >> 4)) ? sl_c[q] : 0) * 16;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.v[q]) : "v"(pp));
            }
        }
        R.r_lo = r_lo_n; R.nrows = nrows_n;
    };
    auto wait_rows = [&](RowRegs &R) {
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3]));
    };
    int st_r_lo = 0, st_nrows = 0;
    auto stage = [&](const RowRegs &R) {
        const int nrows = R.nrows;
#pragma unroll
        for (int q = 0; q < PWV_NL; ++q)
            if (sl_rr[q] < nrows) *reinterpret_cast<tlk_u32x4 *>(s_rows + sl_lds[q]) = R.v[q];
        st_r_lo = R.r_lo; st_nrows = nrows;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // this lane's two adjacent output columns (2 lane, 2 lane + 1): first-tap byte offset and the five 22-bit coefficients of each
    const PilTab xa = s_xt[2 * lane], xb = s_xt[2 * lane + 1];
    const int oA = xa.a.x, oB = xb.a.x;
    const int kA[PIL_KMAX] = {xa.a.y, xa.a.z, xa.a.w, xa.b.x, xa.b.y}, kB[PIL_KMAX] = {xb.a.y, xb.a.z, xb.a.w, xb.b.x, xb.b.y};
    // horizontal pass of the staged (= new) rows into the ring; NT taps (wave-uniform): 3 NT source bytes per column from NW + 1 aligned dwords
    auto hpass = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value, NW = (3 * NT + 3) / 4;
        const int r_lo = st_r_lo, nrows = st_nrows;
        const unsigned int a_lo = (unsigned int)(uintptr_t)(crop0 + (size_t)r_lo * W3) & 15u;
        for (int rr = 0; rr < nrows; ++rr) {
            // (P16: a_step == 0 and a_lo == mis0 for every row, so the aligned offsets and byte shifts below are loop invariants the compiler hoists)
            const int base = rr * CS_ROW_BYTES + (P16 ? mis0 : (int)((a_lo + (unsigned int)rr * a_step) & 15u));
            unsigned int wa[NW], wb[NW];
            {
                const int addr = base + oA;
                const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                unsigned int d[NW + 1];
#pragma unroll
                for (int j = 0; j <= NW; ++j) d[j] = q[j];
#pragma unroll
                for (int j = 0; j < NW; ++j) wa[j] = __builtin_amdgcn_alignbyte(d[j + 1], d[j], (unsigned int)addr & 3u);
            }
            {
                const int addr = base + oB;
                const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                unsigned int d[NW + 1];
#pragma unroll
                for (int j = 0; j <= NW; ++j) d[j] = q[j];
#pragma unroll
                for (int j = 0; j < NW; ++j) wb[j] = __builtin_amdgcn_alignbyte(d[j + 1], d[j], (unsigned int)addr & 3u);
            }
            unsigned char *o = s_h + ((r_lo + rr) % PWV_SRC) * PWV_PLANE + 2 * lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int sa = 1 << (PIL_BITS - 1), sb = 1 << (PIL_BITS - 1);
#pragma unroll
                for (int t = 0; t < NT; ++t) { sa += __mul24(byte_of(wa, 3 * t + c), kA[t]); sb += __mul24(byte_of(wb, 3 * t + c), kB[t]); }
                *reinterpret_cast<unsigned short *>(o + c * OW) = (unsigned short)(((unsigned int)sa >> PIL_BITS) | (((unsigned int)sb >> PIL_BITS) << 8));
            }
        }
    };
    auto mini_band = [&](int mb, RowRegs &N) {
        if (nth <= 2) hpass(std::integral_constant<int, 2>{});
        else if (nth <= 3) hpass(std::integral_constant<int, 3>{});
        else hpass(std::integral_constant<int, 5>{});
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ry = lane >> 4, x_base = (lane & (GROUPS - 1)) * 8;
        const int row = mb * mbh + ry;
        const bool act = ry < mbh && row < rows_chunk;
        const bool block = mbh == WV_ROWS && (mb + 1) * mbh <= rows_chunk;        // wave-uniform: one contiguous output block
        T px[8][3];
        if (act) {
            const PilTab yt = s_yt[row];
            const int slot0 = (yt.a.x >> 12) & 7;
            const int kv[PIL_KMAX] = {yt.a.y, yt.a.z, yt.a.w, yt.b.x, yt.b.y};
            int acc[24];
#pragma unroll
            for (int q = 0; q < 24; ++q) acc[q] = 1 << (PIL_BITS - 1);
#pragma unroll
            for (int t = 0; t < PIL_KMAX; ++t)
                if (t < 2 || t < ntv) {                  // (wave-uniform; taps past a row's own support weigh 0 and read a stale ring row)
                    const unsigned char *p = s_h + ((slot0 + t) & (PWV_SRC - 1)) * PWV_PLANE + x_base;
#pragma unroll
                    for (int co = 0; co < 3; ++co) {         // output channel co <- source plane (R/B swap: a wave-uniform choice, no register shuffle)
                        const uint2 u = *reinterpret_cast<const uint2 *>(p + (swap_rb ? 2 - co : co) * OW);
                        const unsigned int w[2] = {u.x, u.y};
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[co * 8 + k] += __mul24(byte_of(w, k), kv[t]);
                    }
                }
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                const T *lut_c = s_lut[swap_rb ? 2 - co : co];
#pragma unroll
                for (int k = 0; k < 8; ++k) px[k][co] = lut_c[(unsigned int)acc[co * 8 + k] >> PIL_BITS];
            }
        }
        uint4 blk0 = make_uint4(0, 0, 0, 0), blk1 = blk0, blk2 = blk0;
        if (block) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            T *o = reinterpret_cast<T *>(s_rows) + ((size_t)ry * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
            blk0 = l4[lane]; blk1 = l4[WAVE + lane]; blk2 = l4[2 * WAVE + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // staging rows are dead from here (the ring lives on)
        __builtin_amdgcn_wave_barrier();
        if (mb + 1 < mb_hi) {
            wait_rows(N);
            stage(N);
            fetch(mb + 3, N);
        }
        if (block) {
            uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)slot * OH + row0 + mb * mbh) * OW * 3);
            stream_store(g + lane, blk0); stream_store(g + WAVE + lane, blk1); stream_store(g + 2 * WAVE + lane, blk2);
        } else if (act) {
            T *o = out + (((size_t)slot * OH + row0 + row) * OW + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        }
    };
    if (mb_lo < mb_hi) {
        fetch(mb_lo, X);
        fetch(mb_lo + 1, Y);
        wait_rows(X);
        stage(X);
        fetch(mb_lo + 2, X);
    }
    for (int mb = mb_lo; mb < mb_hi; mb += 2) {
        mini_band(mb, Y);
        if (mb + 1 < mb_hi) mini_band(mb + 1, X);
    }
}

// ---- letterbox: workgroup = (frame, LB_BAND output rows); stages the source rows with non-zero weight
constexpr int LB_BAND = 4;

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) letterbox_lds_kernel(const unsigned char *__restrict__ frames, int B, int H, int W, int S,
                                                              int rh, int rw, T *__restrict__ out, int row_bytes_lds, int max_rows, int out_off, int swap_rb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_y0[LB_BAND], s_y1[LB_BAND], s_yw0[LB_BAND], s_yw1[LB_BAND], s_sh0[LB_BAND], s_sh1[LB_BAND];
    __shared__ int s_src[2 * LB_BAND], s_dst[2 * LB_BAND], s_nsrc;
    const int tid = threadIdx.x;
    const int bands = S / LB_BAND;
    const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
    const int y_base = band * LB_BAND;
    const unsigned char *img = frames + (size_t)b * H * W * 3;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    // x coefficient table lives behind the staged rows: (byte offset of tap 0, w0, w1, tap-1 step), k-major
    // [k * groups + xg] so that the 64 lanes of a wave read consecutive 16-byte entries
    int4 *s_xc = reinterpret_cast<int4 *>(s_dyn + (size_t)max_rows * row_bytes_lds);
    const int groups_per_row = S / 8;
    const bool any_real = y_base < rh;
    if (any_real) {
        if (tid == 0) {
            int n = 0;
            for (int ry = 0; ry < LB_BAND; ++ry) {
                const int y = y_base + ry;
                if (y >= rh) break;
                const Coef cy = cv_coef(y, H, rh, false);
                const int r0 = clampi(cy.s, 0, H - 1), r1 = clampi(cy.s + 1, 0, H - 1);
                const unsigned char *g0 = img + (size_t)r0 * W * 3, *g1 = img + (size_t)r1 * W * 3;
                s_y0[ry] = n; s_y1[ry] = cy.w1 != 0 ? n + 1 : n;      // staged slots are handed out compactly (host sized the LDS for it)
                s_yw0[ry] = cy.w0; s_yw1[ry] = cy.w1;
                s_sh0[ry] = (int)((uintptr_t)g0 & 15); s_sh1[ry] = cy.w1 != 0 ? (int)((uintptr_t)g1 & 15) : (int)((uintptr_t)g0 & 15);
                s_src[n] = r0; s_dst[n] = n; ++n;
                if (cy.w1 != 0) { s_src[n] = r1; s_dst[n] = n; ++n; }
            }
            s_nsrc = n;
        }
        for (int x = tid; x < rw; x += BLOCK) {
            const Coef cx = cv_coef(x, W, rw, true);
            s_xc[(x & 7) * groups_per_row + (x >> 3)] = make_int4(cx.s * 3, cx.w0, cx.w1, (cx.s + 1 < W ? 3 : 0));
        }
        __syncthreads();
        // every (source row, 16-byte chunk) pair of the band in one flat sweep, 4 loads in flight per thread before the
        // LDS stores (row-after-row staging serialised up to 2*LB_BAND global round trips per workgroup)
        const int cmax = (W * 3 + 30) >> 4;
        const int total = s_nsrc * cmax;
        auto fetch = [&](int idx, uint4 &v, int &dst) {
            dst = -1;
            if (idx >= total) return;
            const int j = idx / cmax, c = idx - j * cmax;
            const unsigned char *g0 = img + (size_t)s_src[j] * W * 3;
            const int mis = (int)((uintptr_t)g0 & 15);          // pointer ARITHMETIC keeps the global address space (an integer round trip makes the loads flat_load, which also count against lgkmcnt)
            const int chunks = (mis + W * 3 + 15) >> 4;
            if (c >= chunks) return;
            const unsigned char *p = g0 - mis + (size_t)c * 16;
            const int d = s_dst[j] * row_bytes_lds + c * 16;
            if (p + 16 <= gend) { v = *reinterpret_cast<const uint4 *>(p); dst = d; }
            else for (int k = 0; k < 16 && p + k < gend; ++k) s_dyn[d + k] = p[k];
        };
        for (int base = tid; base < total; base += 4 * BLOCK) {
            uint4 v0, v1, v2, v3;           // named registers (an indexed array of these went to scratch memory)
            int d0, d1, d2, d3;
            fetch(base, v0, d0); fetch(base + BLOCK, v1, d1); fetch(base + 2 * BLOCK, v2, d2); fetch(base + 3 * BLOCK, v3, d3);
            if (d0 >= 0) *reinterpret_cast<uint4 *>(s_dyn + d0) = v0;
            if (d1 >= 0) *reinterpret_cast<uint4 *>(s_dyn + d1) = v1;
            if (d2 >= 0) *reinterpret_cast<uint4 *>(s_dyn + d2) = v2;
            if (d3 >= 0) *reinterpret_cast<uint4 *>(s_dyn + d3) = v3;
        }
    }
    __syncthreads();
    constexpr int ROWS = (LAYOUT == LAYOUT_FOCUS_NHWC) ? 2 : 1;
    // Focus layout with 2-byte elements (the detector's input): the 96 contiguous bytes a thread produces are 96 bytes apart from
    // its neighbour's, so direct 16-byte stores touch 64 different 128-byte lines per instruction and the L2 wrote partial
    // lines back (WRITE_SIZE 1.6x the tensor). The units go through LDS instead and leave as fully coalesced 16-byte stores:
    // the band's output is one contiguous block. out_off < 0: direct stores; the staging area may alias the source rows
    // (out_off == 0, all units computed in one sweep) -> barrier between the last tap read and the first staged write.
    constexpr bool LDS_STORE = (LAYOUT == LAYOUT_FOCUS_NHWC) && sizeof(T) == 2;
    constexpr int OUT_STRIDE_W = 28;            // words per unit in LDS: 96 B payload, 16-byte aligned, 2-way write conflicts at most
    const int units_total = (LB_BAND / ROWS) * groups_per_row;
    const bool lds_store = LDS_STORE && out_off >= 0;
    unsigned int *s_out = reinterpret_cast<unsigned int *>(s_dyn + (out_off > 0 ? out_off : 0));
    for (int unit0 = 0; unit0 < units_total; unit0 += BLOCK) {
        const int unit = unit0 + tid;
        const bool act = unit < units_total;
        const int ru = unit / groups_per_row, x_base = (unit - ru * groups_per_row) * 8;
        T px[ROWS][8][3];
        if (act) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int ry = ru * ROWS + r, y = y_base + ry;
            const bool yin = y < rh;
            const unsigned char *p0 = s_dyn, *p1 = s_dyn;
            int b0 = 0, b1 = 0;
            if (yin) {
                p0 = s_dyn + (size_t)s_y0[ry] * row_bytes_lds + s_sh0[ry];
                p1 = s_dyn + (size_t)s_y1[ry] * row_bytes_lds + s_sh1[ry];
                b0 = s_yw0[ry]; b1 = s_yw1[ry];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int x = x_base + k;
                if (yin && x < rw) {
                    const int4 cx = s_xc[k * groups_per_row + (x_base >> 3)];
                    const int o0 = cx.x, o1 = o0 + cx.w, a0 = cx.y, a1 = cx.z;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        int S0 = (int)p0[o0 + c] * a0;
                        if (a1) S0 += (int)p0[o1 + c] * a1;
                        int S1 = 0;
                        if (b1) { S1 = (int)p1[o0 + c] * a0; if (a1) S1 += (int)p1[o1 + c] * a1; }
                        const int v = clampi((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, 0, 255);
                        px[r][k][c] = cvt<T>((float)v);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) px[r][k][c] = cvt<T>(114.f);
                }
                if (swap_rb) { const T t0 = px[r][k][0]; px[r][k][0] = px[r][k][2]; px[r][k][2] = t0; }
            }
        }
        }
        if (lds_store && out_off == 0) __syncthreads();
        if (!act) continue;
        if (LAYOUT == LAYOUT_NCHW) {
            const int y = y_base + ru;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Pack<T, 8> p;
#pragma unroll
                for (int k = 0; k < 8; ++k) p.v[k] = px[0][k][c];
                *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)b * 3 + c) * S + y) * S + x_base) = p;
            }
        } else if (LAYOUT == LAYOUT_NHWC) {
            const int y = y_base + ru;
            T *o = out + (((size_t)b * S + y) * S + x_base) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[0][idx / 3][idx % 3]; }
                *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        } else {
            const int S2 = S / 2, yu = y_base / 2 + ru;
            T *o = out + (((size_t)b * S2 + yu) * S2 + x_base / 2) * 12;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                Pack<T, 8> p;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = k * 8 + e;
                    const int fp = idx / 12, chn = idx % 12;
                    const int grp = chn / 3, c = chn % 3;
                    const int xo = grp >> 1, yo = grp & 1;
                    p.v[e] = px[ROWS > 1 ? yo : 0][fp * 2 + xo][c];
                }
                if (LDS_STORE && lds_store) *reinterpret_cast<Pack<T, 8> *>(s_out + unit * OUT_STRIDE_W + k * 4) = p;
                else *reinterpret_cast<Pack<T, 8> *>(o + k * 8) = p;
            }
        }
    }
    if (LDS_STORE && lds_store) {
        __syncthreads();
        const int S2 = S / 2;
        uint4 *gout = reinterpret_cast<uint4 *>(out + ((size_t)b * S2 + y_base / 2) * S2 * 12);
        for (int c = tid; c < units_total * 6; c += BLOCK) {
            const int u = c / 6, piece = c - u * 6;
            gout[c] = *reinterpret_cast<const uint4 *>(s_out + u * OUT_STRIDE_W + piece * 4);       // (the streaming hint makes no difference here: 56 us either way)
        }
    }
}

// letterbox_wave_kernel (r03; FOCUS layout, 16-bit elements = the detector's input): the recipe that took the crop kernels from 0.32 to 0.53 of the
// HBM peak, applied to the letterbox -- wavefronts that never meet at a barrier after set-up, source rows prefetched TWO items ahead into registers by
// inline-asm loads behind a hand-placed `s_waitcnt vmcnt(N)` (gfx950 counts loads and stores on one counter: a compiler-placed wait drains the
// stores just issued, see crop_wave2_kernel), the item's contiguous output block assembled in the dead staging rows and written as whole cache lines.
// letterbox_lds_kernel (workgroup = 4 output rows, two workgroup barriers, x table of the whole row per workgroup, 160 of 256 threads computing, byte
// LDS taps) measured 0.32: every workgroup's load -> compute -> store phases were serialised behind its barriers.
//   item  = (frame, range of LW_XR = 256 output columns, output row PAIR = one row of the Focus tensor): 3072 contiguous output bytes
//   wave  = LW_PPW consecutive row pairs of one (frame, range); workgroup = 4 wavefronts sharing only the x table of the range (one barrier)
//   ROWS  = staged source rows per pair: 2 when every vertical tap-1 weight of the launch is 0 (1080p -> 640: ratio exactly 1/3), else 4
// arithmetic = letterbox_lds_kernel's (cv2 INTER_LINEAR fixed point, 114 padding), bit for bit.
constexpr int LW_XR = 256, LW_PPW = 4, LW_CPL = 3, LW_RB = LW_CPL * 64 * 16 + 16;        // 256 columns = 64 units of 4: one unit per lane
template <typename T, int ROWS>
__global__ void __launch_bounds__(BLOCK, (ROWS == 2 ? 4 : 2)) letterbox_wave_kernel(const unsigned char *__restrict__ frames, int B, int H, int W, int S, int rh, int rw,
                                                               T *__restrict__ out, int swap_rb)
{
    static_assert(sizeof(T) == 2 && LW_XR == 4 * WAVE, "16-bit elements; one unit of 4 columns per lane");
    constexpr int NL = ROWS * LW_CPL;                    // 16-byte loads per lane and item
    __shared__ __attribute__((aligned(16))) int2 s_xc[LW_XR];
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[NWAVES * ROWS * LW_RB];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_xr = (S + LW_XR - 1) / LW_XR, pairs = S >> 1, S2 = S >> 1;
    // heavy (real) row groups of every frame first: group-major workgroup order
    const int wg = blockIdx.x, grp = wg / (B * n_xr), rem = wg - grp * (B * n_xr), b = rem / n_xr, xr = rem - b * n_xr;
    const int xs = xr * LW_XR, xe = min(S, xs + LW_XR);
    const int units = (xe - xs) >> 2, npieces = units * 3;                       // unit = 4 x of both rows = 48 output bytes
    const int nreal_x = max(0, min(rw, xe) - xs);
    const int p_lo = (grp * NWAVES + wv) * LW_PPW, p_hi = min(pairs, p_lo + LW_PPW);
    const int p_real_hi = nreal_x > 0 ? min(p_hi, (rh + 1) >> 1) : p_lo;
    uint4 *gout0 = reinterpret_cast<uint4 *>(out + ((size_t)b * S2 * S2 + (size_t)(xs >> 1)) * 12);      // + pair * S2 * 12 elements
    const size_t pair_stride16 = (size_t)S2 * 12 * sizeof(T) / 16;
    auto fill_pairs = [&](int pa, int pb) {              // padding rows / columns: 114 everywhere
        const T v114 = cvt<T>(114.f);
        Pack<T, 8> pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk.v[e] = v114;
        const uint4 u = __builtin_bit_cast(uint4, pk);
        for (int p = pa; p < pb; ++p)
            for (int j = lane; j < npieces; j += WAVE) gout0[(size_t)p * pair_stride16 + j] = u;
    };
    const bool wg_real = nreal_x > 0 && 2 * (grp * NWAVES * LW_PPW) < rh;       // workgroup-uniform
    if (!wg_real) { fill_pairs(p_lo, p_hi); return; }
    const int xb0 = cv_coef(xs, W, rw, true).s * 3;
    const int nbytes = cv_coef(xs + nreal_x - 1, W, rw, true).s * 3 + 6 - xb0;
    for (int xl = tid; xl < xe - xs; xl += BLOCK) {
        int2 e = make_int2(0, 0);
        if (xl < nreal_x) {
            const Coef cx = cv_coef(xs + xl, W, rw, true);
            e = make_int2((cx.s * 3 - xb0) | ((cx.s + 1 < W ? 3 : 0) << 16), (cx.w0 & 0xffff) | (cx.w1 << 16));
        }
        s_xc[xl] = e;
    }
    __syncthreads();
    // ---- no workgroup barrier below this line
    if (p_real_hi <= p_lo) { fill_pairs(p_lo, p_hi); return; }
    unsigned char *s_rows = s_stage + (size_t)wv * ROWS * LW_RB;
    const unsigned char *img = frames + (size_t)b * H * W * 3;
    const unsigned char *gend = frames + (size_t)B * H * W * 3;
    const unsigned int W3 = (unsigned int)W * 3u;
    struct Item { tlk_u32x4 v[NL]; int sr[ROWS]; int bw[4]; int pad1; unsigned int tail; };
    Item X, Y;
#pragma unroll
    for (int q = 0; q < NL; ++q) { X.v[q] = tlk_u32x4{0, 0, 0, 0}; Y.v[q] = tlk_u32x4{0, 0, 0, 0}; }
    auto fetch = [&](int p_req, Item &R) {
        const int p = min(p_req, p_real_hi - 1);         // past the wavefront's last real pair: that one again (L2 hits) -- every wait then has NL younger loads
        const int y0 = 2 * p, y1 = 2 * p + 1;
        const Coef c0 = cv_coef(y0, H, rh, false);
        const bool pad1 = y1 >= rh;
        const Coef c1 = pad1 ? c0 : cv_coef(y1, H, rh, false);
        R.pad1 = pad1 ? 1 : 0;
        R.bw[0] = __builtin_amdgcn_readfirstlane(c0.w0); R.bw[1] = __builtin_amdgcn_readfirstlane(c0.w1);
        R.bw[2] = __builtin_amdgcn_readfirstlane(c1.w0); R.bw[3] = __builtin_amdgcn_readfirstlane(c1.w1);
        if constexpr (ROWS == 2) {
            R.sr[0] = __builtin_amdgcn_readfirstlane(clampi(c0.s, 0, H - 1)); R.sr[1] = __builtin_amdgcn_readfirstlane(clampi(c1.s, 0, H - 1));
        } else {
            R.sr[0] = __builtin_amdgcn_readfirstlane(clampi(c0.s, 0, H - 1)); R.sr[1] = __builtin_amdgcn_readfirstlane(clampi(c0.s + 1, 0, H - 1));
            R.sr[ROWS - 2] = __builtin_amdgcn_readfirstlane(clampi(c1.s, 0, H - 1)); R.sr[ROWS - 1] = __builtin_amdgcn_readfirstlane(clampi(c1.s + 1, 0, H - 1));
        }
        unsigned int tail = 0;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const unsigned char *g0 = img + (size_t)R.sr[k] * W3 + xb0;           // wave-uniform
            const int mis = (int)((uintptr_t)g0 & 15);
            const int nch = (mis + nbytes + 15) >> 4;
#pragma unroll
            for (int q = 0; q < LW_CPL; ++q) {
                const int c = lane + q * WAVE;
                const unsigned char *pp = g0 - mis + (size_t)(c < nch ? c : 0) * 16;
                if (pp + 16 > gend) { tail |= 1u << (k * LW_CPL + q); pp = frames; }      // the last bytes of the last frame: fixed up byte-wise in stage()
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(R.v[k * LW_CPL + q]) : "v"(pp));
            }
        }
        R.tail = tail;
    };
    auto wait_rows = [&](Item &R) {                      // ONE form of the wait (see crop_wave2_kernel)
        if constexpr (ROWS == 2)
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3]), "+v"(R.v[4]), "+v"(R.v[5]));
        else
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3]), "+v"(R.v[4]), "+v"(R.v[5]),
                         "+v"(R.v[6]), "+v"(R.v[7]), "+v"(R.v[8]), "+v"(R.v[9]), "+v"(R.v[10]), "+v"(R.v[11]));
    };
    int st_mis[ROWS], st_bw[4] = {0, 0, 0, 0}, st_pad1 = 0;
    auto stage = [&](const Item &R) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const unsigned char *g0 = img + (size_t)R.sr[k] * W3 + xb0;
            const int mis = (int)((uintptr_t)g0 & 15);
            const int nch = (mis + nbytes + 15) >> 4;
            st_mis[k] = mis;
#pragma unroll
            for (int q = 0; q < LW_CPL; ++q) {
                const int c = lane + q * WAVE;
                if (c < nch) *reinterpret_cast<tlk_u32x4 *>(s_rows + k * LW_RB + c * 16) = R.v[k * LW_CPL + q];
            }
        }
        if (__builtin_expect(__any(R.tail != 0), 0)) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const unsigned char *g0 = img + (size_t)R.sr[k] * W3 + xb0;
                const int mis = (int)((uintptr_t)g0 & 15);
#pragma unroll
                for (int q = 0; q < LW_CPL; ++q)
                    if (R.tail & (1u << (k * LW_CPL + q))) {
                        const int c = lane + q * WAVE;
                        const unsigned char *pp = g0 - mis + (size_t)c * 16;
                        for (int e = 0; e < 16 && pp + e < gend; ++e) s_rows[k * LW_RB + c * 16 + e] = pp[e];
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st_bw[k] = (k >= 2 && R.pad1) ? 0 : R.bw[k];
        st_pad1 = R.pad1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // one unit: 4 x of both rows of the pair -> 24 elements in Focus order (per x pair: top-left, bottom-left, top-right, bottom-right channel triples)
    auto unit = [&](int u, Pack<T, 8> (&res)[3]) {
        const int4 e01 = *reinterpret_cast<const int4 *>(&s_xc[u * 4]), e23 = *reinterpret_cast<const int4 *>(&s_xc[u * 4 + 2]);
        const int ex[4] = {e01.x, e01.z, e23.x, e23.z}, ew[4] = {e01.y, e01.w, e23.y, e23.w};
        T val[2][4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o0 = ex[j] & 0xffff;
            const unsigned int sel = 0x0c000c00u | ((unsigned int)(ex[j] >> 16) << 16);
            const us2_t wx = __builtin_bit_cast(us2_t, ew[j]);
            // padding is arithmetic, not control flow: a padded column has zero weights in the x table, a padded bottom row zero vertical weights
            // (stage()), so t = 0 there and the rounding constant carries the 114: ((0 + 2 + 4 * 114) >> 2) = 114
            const bool xpad = u * 4 + j >= nreal_x;
            const unsigned int rnd[2] = {xpad ? 2u + 4u * 114u : 2u, (xpad || st_pad1) ? 2u + 4u * 114u : 2u};
            unsigned int hs[ROWS][3];                    // horizontally interpolated, >> 4
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int addr = k * LW_RB + st_mis[k] + o0;
                const unsigned int *q = reinterpret_cast<const unsigned int *>(s_rows + (addr & ~3));
                const unsigned int d0 = q[0], d1 = q[1], d2 = q[2];
                const unsigned int lo = __builtin_amdgcn_alignbyte(d1, d0, (unsigned int)addr & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, (unsigned int)addr & 3u);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    hs[k][c] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hi, lo, sel + 0x00010001u * c)), wx, 0u, false) >> 4;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    unsigned int t;
                    if constexpr (ROWS == 2) t = __umul24((unsigned int)st_bw[r * 2], hs[r][c]) >> 16;
                    else t = (__umul24((unsigned int)st_bw[r * 2], hs[r * 2][c]) >> 16) + (__umul24((unsigned int)st_bw[r * 2 + 1], hs[r * 2 + 1][c]) >> 16);
                    const int v = (int)((t + rnd[r]) >> 2);   // t <= 1020: v <= 255 always
                    val[r][j][c] = cvt<T>((float)v);
                }
            if (swap_rb) {
#pragma unroll
                for (int r = 0; r < 2; ++r) { const T t0 = val[r][j][0]; val[r][j][0] = val[r][j][2]; val[r][j][2] = t0; }
            }
            __builtin_amdgcn_sched_barrier(0);           // one x at a time: interleaving all four costs 50 more registers (and with them a wavefront per SIMD)
        }
#pragma unroll
        for (int idx = 0; idx < 24; ++idx) {
            const int fp = idx / 12, chn = idx % 12, g = chn / 3, c = chn % 3, xo = g >> 1, yo = g & 1;
            res[idx >> 3].v[idx & 7] = val[yo][fp * 2 + xo][c];
        }
    };
    auto pair_step = [&](int p, Item &N) {
        Pack<T, 8> r0[3];
        unit(lane < units ? lane : 0, r0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // every tap of the pair has been read: the staging rows become the output block
        __builtin_amdgcn_wave_barrier();
        if (lane < units) {
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<Pack<T, 8> *>(s_rows + (size_t)lane * 48 + k * 16) = r0[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (named registers: as an array these went to scratch memory -- whose loads and stores count on vmcnt like everything else)
        const uint4 *l4 = reinterpret_cast<const uint4 *>(s_rows);
        const uint4 blk0 = l4[min(lane, npieces - 1)], blk1 = l4[min(WAVE + lane, npieces - 1)], blk2 = l4[min(2 * WAVE + lane, npieces - 1)];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // staging rows are dead from here
        __builtin_amdgcn_wave_barrier();
        if (p + 1 < p_real_hi) {
            wait_rows(N);
            stage(N);
            fetch(p + 3, N);
        }
        uint4 *g = gout0 + (size_t)p * pair_stride16;
        if (lane < npieces) g[lane] = blk0;
        if (WAVE + lane < npieces) g[WAVE + lane] = blk1;
        if (2 * WAVE + lane < npieces) g[2 * WAVE + lane] = blk2;
    };
    fetch(p_lo, X);
    fetch(p_lo + 1, Y);
    wait_rows(X);
    stage(X);
    fetch(p_lo + 2, X);
    for (int p = p_lo; p < p_real_hi; p += 2) {
        pair_step(p, Y);
        if (p + 1 < p_real_hi) pair_step(p + 1, X);
    }
    // the clamped prefetches of the last pairs are still in flight: keep their registers allocated until they have landed
    if constexpr (ROWS == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(X.v[0]), "+v"(X.v[1]), "+v"(X.v[2]), "+v"(X.v[3]), "+v"(X.v[4]), "+v"(X.v[5]),
                     "+v"(Y.v[0]), "+v"(Y.v[1]), "+v"(Y.v[2]), "+v"(Y.v[3]), "+v"(Y.v[4]), "+v"(Y.v[5]));
    else {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(X.v[0]), "+v"(X.v[1]), "+v"(X.v[2]), "+v"(X.v[3]), "+v"(X.v[4]), "+v"(X.v[5]),
                     "+v"(X.v[6]), "+v"(X.v[7]), "+v"(X.v[8]), "+v"(X.v[9]), "+v"(X.v[10]), "+v"(X.v[11]));
        asm volatile("" : "+v"(Y.v[0]), "+v"(Y.v[1]), "+v"(Y.v[2]), "+v"(Y.v[3]), "+v"(Y.v[4]), "+v"(Y.v[5]),
                     "+v"(Y.v[6]), "+v"(Y.v[7]), "+v"(Y.v[8]), "+v"(Y.v[9]), "+v"(Y.v[10]), "+v"(Y.v[11]));
    }
    fill_pairs(p_real_hi, p_hi);
}

// ---------------------------------------------------------------------------------------------
// YOLOX decode + per-class greedy NMS (rtmlib YOLOX.postprocess / multiclass_nms / nms), one workgroup
// per frame. Candidates (score = obj*cls > score_thr) are compacted and sorted by (score desc, anchor desc): up to 1024 of them by RANK
// (every candidate counts the keys above its own -- m*m/256 comparisons per lane, no barrier ladder), more by a bitonic network in LDS.
// Greedy NMS (r04, VERDICT r03 #4: 163 -> ~25 us per 24-frame launch): candidates are taken 64 at a time; all four wavefronts first build
// the block's suppression bitmask rows in parallel (row i, word w = __ballot over 64 later candidates of "IoU(i, j) > thr"), then ONE
// wavefront walks the 64 rows in order with the alive words in registers (lane = word): alive &= ~row -- the sequential part is one LDS
// read and one AND per candidate instead of an IoU sweep.  Same kept set and order as the sweep: a row is only applied when its candidate
// is still alive at its turn.  fp32 arithmetic in the reference's operation order ("+1" pixel convention, ovr <= thr keeps).
// ---------------------------------------------------------------------------------------------
constexpr int NMS_CAP = 4096;      // max candidates per (frame, class)
constexpr int NMS_RANK_CAP = 1024; // up to here the candidates are sorted by rank, beyond by the bitonic network

__global__ void __launch_bounds__(BLOCK) yolox_decode_nms_kernel(const float *__restrict__ pred_all, int S, int C, float ratio,
                                                                 float nms_thr, float score_thr, int img_w, int img_h,
                                                                 int max_out, float *__restrict__ ltwh_out,
                                                                 float *__restrict__ xyxy_out, float *__restrict__ score_out,
                                                                 int *__restrict__ cls_out, int *__restrict__ count_out,
                                                                 double *__restrict__ trk_in, long long id_base, double category_id)
{
    __shared__ unsigned long long key[NMS_CAP];
    __shared__ float bx[NMS_CAP][4];
    __shared__ float barea[NMS_CAP];
    __shared__ unsigned long long alive[NMS_CAP / 64];
    __shared__ unsigned long long rowmask[64][NMS_CAP / 64];     // suppression rows of the current block of 64 candidates (32 KB)
    __shared__ unsigned long long srt[NMS_RANK_CAP];            // rank-sort destination; afterwards the list of kept candidates (u16)
    __shared__ int s_cnt, s_out, s_err, s_kept, s_emit;
    unsigned short *kept = reinterpret_cast<unsigned short *>(srt);              // NMS_CAP entries = 8 KB = sizeof(srt)
    static_assert(sizeof(srt) >= NMS_CAP * sizeof(unsigned short), "kept list aliases the sort buffer");
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n8 = (S / 8) * (S / 8), n16 = (S / 16) * (S / 16), n32 = (S / 32) * (S / 32);
    const int A = n8 + n16 + n32, F = 5 + C;
    const float *pred = pred_all + (size_t)b * A * F;
    if (tid == 0) { s_out = 0; s_err = 0; }
    for (int c = 0; c < C; ++c) {
        if (tid == 0) { s_cnt = 0; s_kept = 0; s_emit = 0; }
        __syncthreads();
        // 1. candidates: appended in arrival order (the sort below orders them: the anchor is part of the key, keys are distinct)
        // (the loads of 16 anchors per lane are issued together: one round trip to memory per 4096 anchors instead of one per 256)
        for (int a0 = tid; a0 < A; a0 += 16 * BLOCK) {
            float so[16], sc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int a = a0 + u * BLOCK;
                const size_t o = (size_t)(a < A ? a : 0) * F;
                so[u] = pred[o + 4]; sc[u] = pred[o + 5 + c];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int a = a0 + u * BLOCK;
                const float s = so[u] * sc[u];
                if (a < A && s > score_thr) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < NMS_CAP) key[pos] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned int)a;
                }
            }
        }
        __syncthreads();
        const int n = s_cnt;
        if (n > NMS_CAP) { if (tid == 0) s_err = 1; }
        const int m = n < NMS_CAP ? n : NMS_CAP;
        // 2. sort, descending by (score bits, anchor)
        if (m <= NMS_RANK_CAP) {
            for (int i = tid; i < m; i += BLOCK) {
                const unsigned long long mine = key[i];
                int above = 0;
#pragma unroll 8
                for (int j = 0; j < m; ++j) above += key[j] > mine ? 1 : 0;      // every lane reads the same word: LDS broadcast
                srt[above] = mine;
            }
            __syncthreads();
            for (int i = tid; i < m; i += BLOCK) key[i] = srt[i];
            __syncthreads();
        } else {
            int p2 = 1;
            while (p2 < m) p2 <<= 1;
            for (int k = m + tid; k < p2; k += BLOCK) key[k] = 0ull;      // pad: sorts last (scores > 0)
            __syncthreads();
            for (int k = 2; k <= p2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < p2; i += BLOCK) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = key[i], bb = key[ixj];
                            const bool desc = (i & k) == 0;
                            if (desc ? (a < bb) : (a > bb)) { key[i] = bb; key[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
        }
        // 3. decode the candidates (rtmlib: (xy+grid)*stride, exp(wh)*stride, /ratio)
        for (int k = tid; k < m; k += BLOCK) {
            const int a = (int)(key[k] & 0xffffffffull);
            int st, loc, ws;
            if (a < n8) { st = 8; loc = a; ws = S / 8; }
            else if (a < n8 + n16) { st = 16; loc = a - n8; ws = S / 16; }
            else { st = 32; loc = a - n8 - n16; ws = S / 32; }
            const int gy = loc / ws, gx = loc - gy * ws;
            const float *p = pred + (size_t)a * F;
            const float fs = (float)st;
            const float cx = (p[0] + (float)gx) * fs, cy = (p[1] + (float)gy) * fs;
            const float w = expf(p[2]) * fs, h = expf(p[3]) * fs;
            float x1 = cx - w / 2.f, y1 = cy - h / 2.f, x2 = cx + w / 2.f, y2 = cy + h / 2.f;
            x1 /= ratio; y1 /= ratio; x2 /= ratio; y2 /= ratio;
            bx[k][0] = x1; bx[k][1] = y1; bx[k][2] = x2; bx[k][3] = y2;
            barea[k] = (x2 - x1 + 1) * (y2 - y1 + 1);
        }
        for (int k = tid; k < NMS_CAP / 64; k += BLOCK) {
            const int lo = k * 64;
            alive[k] = (lo + 64 <= m) ? ~0ull : (lo >= m ? 0ull : ((1ull << (m - lo)) - 1ull));
        }
        __syncthreads();
        // 4. greedy NMS, 64 candidates (one block of rows) at a time
        const int nwords = (m + 63) >> 6;
        for (int blk = 0; blk < nwords; ++blk) {
            // 4a. all wavefronts: suppression rows of the block's candidates that are still alive, against every later candidate
            const unsigned long long alive_blk = alive[blk];
            for (int r = wave; r < 64; r += NWAVES) {
                const int i = blk * 64 + r;
                if (i >= m || !((alive_blk >> r) & 1ull)) continue;           // uniform per wavefront (a row killed later inside the block is built in vain, never applied)
                const float ix1 = bx[i][0], iy1 = bx[i][1], ix2 = bx[i][2], iy2 = bx[i][3], ia = barea[i];
                for (int w = blk; w < nwords; ++w) {
                    const int j = w * 64 + lane;
                    bool kill = false;
                    if (j > i && j < m) {
                        const float xx1 = fmaxf(ix1, bx[j][0]), yy1 = fmaxf(iy1, bx[j][1]);
                        const float xx2 = fminf(ix2, bx[j][2]), yy2 = fminf(iy2, bx[j][3]);
                        const float w_ = fmaxf(0.0f, xx2 - xx1 + 1), h_ = fmaxf(0.0f, yy2 - yy1 + 1);
                        const float inter = w_ * h_;
                        const float ovr = inter / (ia + barea[j] - inter);
                        kill = !(ovr <= nms_thr);
                    }
                    const unsigned long long km = __ballot(kill);
                    if (lane == 0) rowmask[r][w] = km;
                }
            }
            __syncthreads();
            // 4b. wavefront 0.  Lane r holds the block-diagonal word of row r, so who survives INSIDE the block is resolved with scalar
            // bit operations and readlane (no memory access in the sequential chain); the survivors' rows are then OR-ed into every
            // later word in parallel (lane = word) and the survivors are appended to the kept list.
            if (tid < WAVE) {
                const unsigned long long diag = rowmask[lane][blk];             // (rows not rebuilt this block are never selected below)
                const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
                unsigned long long a0 = alive_blk, keep = 0ull;
                while (a0) {                                                    // uniform: a0 is the same in every lane
                    const int r = __builtin_ctzll(a0);
                    keep |= 1ull << r;
                    a0 &= ~(1ull << r);
                    const unsigned long long row = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, r) << 32) |
                                                   (unsigned int)__builtin_amdgcn_readlane((int)dlo, r);
                    a0 &= ~row;
                }
                const int w = blk + 1 + lane;                                   // later words
                if (w < nwords) {
                    unsigned long long aw = alive[w], kk = keep;
                    while (kk) {
                        const int r = __builtin_ctzll(kk);
                        kk &= kk - 1;
                        aw &= ~rowmask[r][w];
                    }
                    alive[w] = aw;
                }
                const int base = s_kept;
                if ((keep >> lane) & 1ull) kept[base + __builtin_popcountll(keep & ((1ull << lane) - 1ull))] = (unsigned short)(blk * 64 + lane);
                if (lane == 0) s_kept = base + __builtin_popcountll(keep);
            }
            __syncthreads();
        }
        // 5. emission, in parallel: the kept candidates with a final score above 0.3 (rtmlib: final_scores > 0.3) are a prefix of the kept
        // list (scores descend along it)
        const int nk = s_kept, out0 = s_out;
        for (int k = tid; k < nk; k += BLOCK)
            if (__uint_as_float((unsigned int)(key[kept[k]] >> 32)) > 0.3f) atomicMax(&s_emit, k + 1);
        __syncthreads();
        const int ne = s_emit;
        for (int k = tid; k < ne; k += BLOCK) {
            const int i = kept[k], n_out = out0 + k;
            if (n_out >= max_out) continue;
            const size_t o = (size_t)b * max_out + n_out;
            float l = bx[i][0], t = bx[i][1], r = bx[i][2], bt = bx[i][3];
            xyxy_out[o * 4] = l; xyxy_out[o * 4 + 1] = t; xyxy_out[o * 4 + 2] = r; xyxy_out[o * 4 + 3] = bt;
            // RTMLibDetector: ltrb_to_ltwh(bbox, (W,H)) -> sanitize_bbox_ltrb (coordinates.py:270-295,318-328), float32
            l = fmaxf(0.f, fminf(l, (float)(img_w - 2))); t = fmaxf(0.f, fminf(t, (float)(img_h - 2)));
            r = fmaxf(1.f, fminf(r, (float)(img_w - 1))); bt = fmaxf(1.f, fminf(bt, (float)(img_h - 1)));
            ltwh_out[o * 4] = l; ltwh_out[o * 4 + 1] = t; ltwh_out[o * 4 + 2] = r - l; ltwh_out[o * 4 + 3] = bt - t;
            score_out[o] = __uint_as_float((unsigned int)(key[i] >> 32)); cls_out[o] = c;
            if (trk_in) {   // row the tracker wrapper would build (oc_sort_api.py:37-45): float32 ltwh -> ltrb,
                            // bbox_conf = 1.0 and category_id as set by RTMLibDetector (rtmlib_api.py:36-41)
                double *q = trk_in + o * 7;
                const float w = r - l, h = bt - t;
                q[0] = (double)l; q[1] = (double)t; q[2] = (double)(l + w); q[3] = (double)(t + h);
                q[4] = 1.0; q[5] = category_id; q[6] = (double)(id_base + (long long)b * max_out + n_out);
            }
        }
        __syncthreads();
        if (tid == 0) s_out = out0 + ne;
        __syncthreads();
    }
    if (tid == 0) count_out[b] = s_err ? TLK_ECAPACITY : (s_out > max_out ? TLK_ECAPACITY : s_out);
}

template <typename T>
int launch_letterbox(const unsigned char *frames, int B, int H, int W, int S, int rh, int rw, int layout, void *out, hipStream_t st, int swap_rb)
{
    const int row_bytes = ((W * 3 + STAGE_PAD) + 15) & ~15;
    // staged source rows per band: one per output row plus one more where the vertical tap-1 weight is non-zero (same
    // cv_coef as the kernel, evaluated here so the LDS allocation -- and with it the workgroups per CU -- is what the ratio needs)
    int max_rows = 1;
    if (S % LB_BAND == 0)
        for (int y0 = 0; y0 < rh; y0 += LB_BAND) {
            int n = 0;
            for (int y = y0; y < y0 + LB_BAND && y < rh; ++y) n += cv_coef(y, H, rh, false).w1 != 0 ? 2 : 1;
            if (n > max_rows) max_rows = n;
        }
    size_t smem = (size_t)max_rows * row_bytes + (size_t)4 * S * sizeof(int);
    // focus layout, 2-byte elements: output units leave through an LDS staging area (28 words per 96-byte unit); it aliases the
    // source rows when one sweep computes every unit of the band, else it is appended
    int out_off = -1;
    if (layout == LAYOUT_FOCUS_NHWC && sizeof(T) == 2) {
        const int units = (LB_BAND / 2) * (S / 8);
        const size_t need = (size_t)units * 28 * 4;
        if (units <= BLOCK && (size_t)max_rows * row_bytes >= need) out_off = 0;
        else { out_off = (int)smem; smem += need; }
    }
    // r03: free-running wavefronts for the detector's input format (Focus layout, 16-bit elements); TLK_LETTERBOX_WAVE=0: letterbox_lds_kernel
    if constexpr (sizeof(T) == 2) {
        static const int wave = [] { const char *e = getenv("TLK_LETTERBOX_WAVE"); return e ? atoi(e) : 1; }();
        if (wave && layout == LAYOUT_FOCUS_NHWC && S % 8 == 0 && rh >= 1 && rw >= 1) {
            bool ok = true, two_rows = true;
            for (int xs = 0; xs < S && xs < rw && ok; xs += LW_XR) {
                const int nreal = (rw < xs + LW_XR ? rw : xs + LW_XR) - xs;
                const int nbytes = cv_coef(xs + nreal - 1, W, rw, true).s * 3 + 6 - cv_coef(xs, W, rw, true).s * 3;
                if (((15 + nbytes + 15) >> 4) > LW_CPL * 64) ok = false;
            }
            for (int y = 0; y < rh && two_rows; ++y) two_rows = cv_coef(y, H, rh, false).w1 == 0;
            if (ok) {
                const int groups = (S / 2 + NWAVES * LW_PPW - 1) / (NWAVES * LW_PPW), n_xr = (S + LW_XR - 1) / LW_XR;
                const dim3 grid((unsigned)(groups * B * n_xr));
                if (two_rows) hipLaunchKernelGGL((letterbox_wave_kernel<T, 2>), grid, dim3(BLOCK), 0, st, frames, B, H, W, S, rh, rw, (T *)out, swap_rb);
                else hipLaunchKernelGGL((letterbox_wave_kernel<T, 4>), grid, dim3(BLOCK), 0, st, frames, B, H, W, S, rh, rw, (T *)out, swap_rb);
                return TLK_OK;
            }
        }
    }
    if (smem <= 64 * 1024 && S % LB_BAND == 0) {          // LDS-staged fast path
        const dim3 grid((unsigned)(B * (S / LB_BAND)));
        if (layout == LAYOUT_NCHW) hipLaunchKernelGGL((letterbox_lds_kernel<T, LAYOUT_NCHW>), grid, dim3(BLOCK), smem, st, frames, B, H, W, S, rh, rw, (T *)out, row_bytes, max_rows, out_off, swap_rb);
        else if (layout == LAYOUT_NHWC) hipLaunchKernelGGL((letterbox_lds_kernel<T, LAYOUT_NHWC>), grid, dim3(BLOCK), smem, st, frames, B, H, W, S, rh, rw, (T *)out, row_bytes, max_rows, out_off, swap_rb);
        else hipLaunchKernelGGL((letterbox_lds_kernel<T, LAYOUT_FOCUS_NHWC>), grid, dim3(BLOCK), smem, st, frames, B, H, W, S, rh, rw, (T *)out, row_bytes, max_rows, out_off, swap_rb);
        return TLK_OK;
    }
    const long long units = (long long)B * (S / 8) * (layout == LAYOUT_FOCUS_NHWC ? S / 2 : S);
    const dim3 grid((unsigned)((units + BLOCK - 1) / BLOCK));
    if (layout == LAYOUT_NCHW) hipLaunchKernelGGL((letterbox_kernel<T, LAYOUT_NCHW>), grid, dim3(BLOCK), 0, st, frames, B, H, W, S, rh, rw, (T *)out, swap_rb);
    else if (layout == LAYOUT_NHWC) hipLaunchKernelGGL((letterbox_kernel<T, LAYOUT_NHWC>), grid, dim3(BLOCK), 0, st, frames, B, H, W, S, rh, rw, (T *)out, swap_rb);
    else hipLaunchKernelGGL((letterbox_kernel<T, LAYOUT_FOCUS_NHWC>), grid, dim3(BLOCK), 0, st, frames, B, H, W, S, rh, rw, (T *)out, swap_rb);
    return TLK_OK;
}

// the normalisation table of crop_sep_kernel: built once per (device, statistics, element type) by crop_lut_kernel and kept for the
// life of the process (3 x 1024 elements)
struct CropLutKey { int dev, elem; float v[6]; bool operator<(const CropLutKey &o) const { return memcmp(this, &o, sizeof(*this)) < 0; } };
template <typename T>
const void *crop_lut(float m0, float m1, float m2, float d0, float d1, float d2)
{
    static std::mutex mu;
    static std::map<CropLutKey, void *> cache;
    CropLutKey k;
    memset(&k, 0, sizeof(k));
    if (hipGetDevice(&k.dev) != hipSuccess) return nullptr;
    k.elem = (int)sizeof(T) * 4 + (std::is_same<T, bf16_t>::value ? 1 : 0);
    k.v[0] = m0; k.v[1] = m1; k.v[2] = m2; k.v[3] = d0; k.v[4] = d1; k.v[5] = d2;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(k);
    if (it != cache.end()) return it->second;
    void *p = nullptr;
    if (hipMalloc(&p, 3 * CS_LUT_N * sizeof(T)) != hipSuccess) { set_error("tlk_roi_crop_resize_norm: hipMalloc of the normalisation table failed"); return nullptr; }
    hipLaunchKernelGGL((crop_lut_kernel<T>), dim3(CS_LUT_N / 256), dim3(256), 0, 0, (T *)p, m0, m1, m2, d0, d1, d2);
    if (hipDeviceSynchronize() != hipSuccess) { hipFree(p); set_error("tlk_roi_crop_resize_norm: building the normalisation table failed"); return nullptr; }
    cache[k] = p;
    return p;
}

template <typename T>
int launch_crop(const unsigned char *frames, int B, int H, int W, const float *boxes, const int *counts, int max_n, int OH, int OW,
                const float *mean, const float *stdv, int layout, void *out, hipStream_t st, int swap_rb)
{
    const long long units = (long long)B * max_n * OH * (OW / 8);
    const dim3 grid((unsigned)((units + BLOCK - 1) / BLOCK));
    // swap_rb: output channel c is source channel 2 - c; the kernels normalise per SOURCE channel and exchange 0 <-> 2 at the store
    const int sw0 = swap_rb ? 2 : 0, sw2 = swap_rb ? 0 : 2;
    const float m0 = mean[sw0] * 255.f, m1 = mean[1] * 255.f, m2 = mean[sw2] * 255.f;
    const float d0 = 1.0f / (stdv[sw0] * 255.f), d1 = 1.0f / (stdv[1] * 255.f), d2 = 1.0f / (stdv[sw2] * 255.f);
    static const int variant = [] { const char *e = getenv("TLK_CROP_KERNEL"); return e ? atoi(e) : 2; }();     // 2 separable (default), 1 round-1 LDS kernel, 0 direct
    if (variant == 2 && OW <= 256) {                       // separable fast path: workgroup = (slot, band of CS_BAND rows)
        const T *lut = (const T *)crop_lut<T>(m0, m1, m2, d0, d1, d2);
        if (!lut) return TLK_EHIP;
        const int nwg = (int)((long long)B * max_n * ((OH + CS_BAND - 1) / CS_BAND));
        const size_t smem = crop_sep_lds_bytes(OW, sizeof(T));
#define CROP_SEP_LAUNCH(LAY, OWC) hipLaunchKernelGGL((crop_sep_kernel<T, LAY, OWC>), dim3(nwg), dim3(BLOCK), smem, st, frames, B, H, W, boxes, counts, \
                                                     max_n, OH, OW, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg)
        static const int fat = [] { const char *e = getenv("TLK_CROP_FAT"); return e ? atoi(e) : 1; }();       // 0: a workgroup per band (crop_sep_kernel)
        if (OW == 128 && fat) {                            // the ReID shape: a workgroup per CF_BANDS bands of a crop
            const int bands = (OH + CS_BAND - 1) / CS_BAND, chunks = (bands + CF_BANDS - 1) / CF_BANDS;
            const int nwg2 = (int)((long long)B * max_n * chunks);
            const size_t smem2 = (size_t)CS_ROWS * CS_ROW_BYTES + (size_t)CS_ROWS * 128 * 6 + 128 * 8 + (size_t)3 * CS_LUT_N * sizeof(T);
            static const int wave = [] { const char *e = getenv("TLK_CROP_WAVE"); return e ? atoi(e) : 3; }();       // 3 (default; 16-bit outputs): crop_wave3_kernel (sliding window); 4: crop_pw_kernel (persistent wavefronts, measured equal); 2: crop_wave2_kernel (hand-placed waits; fp32 outputs); 1: crop_wave_kernel; 0: crop_fat_kernel
            const size_t smem3 = (size_t)128 * 8 + ((3 * CS_LUT_N * sizeof(T) + 15) & ~(size_t)15) + (size_t)NWAVES * WV_WAVE_LDS;
            if constexpr (sizeof(T) == 2) {
                if (wave >= 4 && OH % WV_ROWS == 0) {       // persistent wavefronts: 2 workgroups of 8 per CU walk the valid crops
                    static int n_cu = 0;
                    if (n_cu == 0) {
                        int dev = 0, v = 0;
                        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
                        n_cu = v;
                    }
                    const size_t smem4 = ((3 * CS_LUT_N * sizeof(T) + 15) & ~(size_t)15) + (size_t)PW_WAVES * PW_WAVE_LDS;
                    static bool attr_nchw = false, attr_nhwc = false;
                    if (layout == LAYOUT_NCHW) {
                        if (!attr_nchw) { if (hipFuncSetAttribute((const void *)crop_pw_kernel<T, LAYOUT_NCHW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4) != hipSuccess) return TLK_EHIP; attr_nchw = true; }
                        hipLaunchKernelGGL((crop_pw_kernel<T, LAYOUT_NCHW>), dim3(2 * n_cu), dim3(PW_BLOCK), smem4, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb);
                    } else {
                        if (!attr_nhwc) { if (hipFuncSetAttribute((const void *)crop_pw_kernel<T, LAYOUT_NHWC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4) != hipSuccess) return TLK_EHIP; attr_nhwc = true; }
                        hipLaunchKernelGGL((crop_pw_kernel<T, LAYOUT_NHWC>), dim3(2 * n_cu), dim3(PW_BLOCK), smem4, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb);
                    }
                    return TLK_OK;
                }
                if (wave >= 3) {
                    static const int p16_on = [] { const char *e = getenv("TLK_CROP_P16"); return e ? atoi(e) : 1; }();     // 0: the general-pitch code also for 16-byte-multiple row pitches
                    const bool p16 = p16_on && ((long long)W * 3) % 16 == 0;
                    static const int depth3 = [] { const char *e = getenv("TLK_CROP_DEPTH"); return e ? atoi(e) : 2; }() == 3;       // 3: three mini-bands of source rows in flight (NHWC, 16-byte pitch)
                    static const int nt_off = [] { const char *e = getenv("TLK_CROP_NT"); return e ? (atoi(e) == 0 ? 2 : 0) : 0; }();
#define TLK_CW3(LAY, P, D3) hipLaunchKernelGGL((crop_wave3_kernel<T, LAY, P, D3>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, (swap_rb ? 1 : 0) | nt_off, nwg2)
                    if (layout == LAYOUT_NCHW) { if (p16) TLK_CW3(LAYOUT_NCHW, true, false); else TLK_CW3(LAYOUT_NCHW, false, false); }
                    else if (p16 && depth3) TLK_CW3(LAYOUT_NHWC, true, true);
                    else { if (p16) TLK_CW3(LAYOUT_NHWC, true, false); else TLK_CW3(LAYOUT_NHWC, false, false); }
#undef TLK_CW3
                    return TLK_OK;
                }
            }
            if (wave >= 2 && layout == LAYOUT_NCHW)
                hipLaunchKernelGGL((crop_wave2_kernel<T, LAYOUT_NCHW>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2);
            else if (wave >= 2)
                hipLaunchKernelGGL((crop_wave2_kernel<T, LAYOUT_NHWC>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2);
            else if (wave && layout == LAYOUT_NCHW)
                hipLaunchKernelGGL((crop_wave_kernel<T, LAYOUT_NCHW>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2);
            else if (wave)
                hipLaunchKernelGGL((crop_wave_kernel<T, LAYOUT_NHWC>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2);
            else if (layout == LAYOUT_NCHW)
                hipLaunchKernelGGL((crop_fat_kernel<T, LAYOUT_NCHW>), dim3(nwg2), dim3(BLOCK), smem2, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2);
            else
                hipLaunchKernelGGL((crop_fat_kernel<T, LAYOUT_NHWC>), dim3(nwg2), dim3(BLOCK), smem2, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2);
            return TLK_OK;
        }
        if (layout == LAYOUT_NCHW) { if (OW == 128) CROP_SEP_LAUNCH(LAYOUT_NCHW, 128); else CROP_SEP_LAUNCH(LAYOUT_NCHW, 0); }
        else { if (OW == 128) CROP_SEP_LAUNCH(LAYOUT_NHWC, 128); else CROP_SEP_LAUNCH(LAYOUT_NHWC, 0); }
#undef CROP_SEP_LAUNCH
        return TLK_OK;
    }
    if (variant >= 1 && OW <= 256) {                       // round-1 LDS-staged path: workgroup = (slot, band of rows)
        const dim3 g2((unsigned)((long long)B * max_n * ((OH + CROP_BAND - 1) / CROP_BAND)));
        if (layout == LAYOUT_NCHW)
            hipLaunchKernelGGL((crop_lds_kernel<T, LAYOUT_NCHW>), g2, dim3(BLOCK), 0, st, frames, B, H, W, boxes, counts, max_n, OH, OW, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb);
        else
            hipLaunchKernelGGL((crop_lds_kernel<T, LAYOUT_NHWC>), g2, dim3(BLOCK), 0, st, frames, B, H, W, boxes, counts, max_n, OH, OW, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb);
        return TLK_OK;
    }
    if (layout == LAYOUT_NCHW)
        hipLaunchKernelGGL((crop_kernel<T, LAYOUT_NCHW>), grid, dim3(BLOCK), 0, st, frames, B, H, W, boxes, counts, max_n, OH, OW, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb);
    else
        hipLaunchKernelGGL((crop_kernel<T, LAYOUT_NHWC>), grid, dim3(BLOCK), 0, st, frames, B, H, W, boxes, counts, max_n, OH, OW, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb);
    return TLK_OK;
}

}  // namespace

extern "C" int tlk_letterbox_u8(const uint8_t *frames_dev, int batch, int h, int w, int size, int layout, int dtype,
                                void *out_dev, double *ratio_out, void *hip_stream)
{
    if (batch < 0 || h <= 0 || w <= 0 || size <= 0) return fail(TLK_EINVAL, "tlk_letterbox_u8: bad size");
    if (size % 16 != 0) return fail(TLK_EINVAL, "tlk_letterbox_u8: size must be a multiple of 16");
    const int swap_rb = (layout & TLK_SWAP_RB) ? 1 : 0;
    layout &= ~TLK_SWAP_RB;
    if (layout < 0 || layout > 2 || dtype < 0 || dtype > 2) return fail(TLK_EINVAL, "tlk_letterbox_u8: bad layout/dtype");
    const double ratio = std::min((double)size / h, (double)size / w);      // rtmlib YOLOX.preprocess
    if (ratio_out) *ratio_out = ratio;
    if (batch == 0) return TLK_OK;
    if (!frames_dev || !out_dev) return fail(TLK_EINVAL, "tlk_letterbox_u8: null pointer");
    const int rw = (int)(w * ratio), rh = (int)(h * ratio);
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == 0) launch_letterbox<float>(frames_dev, batch, h, w, size, rh, rw, layout, out_dev, st, swap_rb);
    else if (dtype == 1) launch_letterbox<__half>(frames_dev, batch, h, w, size, rh, rw, layout, out_dev, st, swap_rb);
    else launch_letterbox<bf16_t>(frames_dev, batch, h, w, size, rh, rw, layout, out_dev, st, swap_rb);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_roi_crop_resize_norm(const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev,
                                        const int32_t *counts_dev, int max_n, int out_h, int out_w, const float *mean3,
                                        const float *std3, int layout, int dtype, void *out_dev, void *hip_stream)
{
    if (batch < 0 || h <= 0 || w <= 0 || max_n < 0 || out_h <= 0 || out_w <= 0) return fail(TLK_EINVAL, "tlk_roi_crop_resize_norm: bad size");
    if (out_w % 8 != 0) return fail(TLK_EINVAL, "tlk_roi_crop_resize_norm: out_w must be a multiple of 8");
    const int swap_rb = (layout & TLK_SWAP_RB) ? 1 : 0;
    layout &= ~TLK_SWAP_RB;
    if (layout < 0 || layout > 1 || dtype < 0 || dtype > 2) return fail(TLK_EINVAL, "tlk_roi_crop_resize_norm: bad layout/dtype");
    if (batch == 0 || max_n == 0) return TLK_OK;
    if (!frames_dev || !boxes_ltwh_dev || !counts_dev || !mean3 || !std3 || !out_dev) return fail(TLK_EINVAL, "tlk_roi_crop_resize_norm: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == 0) launch_crop<float>(frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb);
    else if (dtype == 1) launch_crop<__half>(frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb);
    else launch_crop<bf16_t>(frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

template <typename T>
int launch_pil_crop(const unsigned char *frames, int B, int H, int W, const double *boxes, int box_stride, const int *counts, int max_n,
                    int OH, int OW, const float *mean, const float *stdv, int layout, void *out, hipStream_t st, int swap_rb)
{
    const int sw0 = swap_rb ? 2 : 0, sw2 = swap_rb ? 0 : 2;
    // r03: free-running wavefronts for the ReID input format (128 wide, NHWC, 16-bit elements); TLK_PIL_WAVE=0: pil_crop_kernel
    if constexpr (sizeof(T) == 2) {
        static const int wave = [] { const char *e = getenv("TLK_PIL_WAVE"); return e ? atoi(e) : 1; }();
        if (wave && layout == LAYOUT_NHWC && OW == 128) {
            const int chunks = (OH + CF_BANDS * CS_BAND - 1) / (CF_BANDS * CS_BAND);
            const int nwg = B * max_n * chunks;
            static const int p16_on = [] { const char *e = getenv("TLK_CROP_P16"); return e ? atoi(e) : 1; }();
            if (p16_on && ((long long)W * 3) % 16 == 0)
                hipLaunchKernelGGL((pil_wave_kernel<T, true>), dim3((unsigned)nwg), dim3(BLOCK), 0, st, frames, B, H, W, boxes, box_stride, counts, max_n, OH,
                                   mean[sw0], mean[1], mean[sw2], stdv[sw0], stdv[1], stdv[sw2], (T *)out, swap_rb, nwg);
            else
                hipLaunchKernelGGL((pil_wave_kernel<T, false>), dim3((unsigned)nwg), dim3(BLOCK), 0, st, frames, B, H, W, boxes, box_stride, counts, max_n, OH,
                                   mean[sw0], mean[1], mean[sw2], stdv[sw0], stdv[1], stdv[sw2], (T *)out, swap_rb, nwg);
            return TLK_OK;
        }
    }
    const int pil_bands = (OH + PIL_BAND - 1) / PIL_BAND;
    const dim3 grid((unsigned)((long long)B * max_n * ((pil_bands + PIL_BPW - 1) / PIL_BPW)));
#define TLK_PIL_LAUNCH(LAY, OWC) hipLaunchKernelGGL((pil_crop_kernel<T, LAY, OWC>), grid, dim3(BLOCK), 0, st, frames, B, H, W, boxes, box_stride, counts, max_n, OH, OW, \
                                                  mean[sw0], mean[1], mean[sw2], stdv[sw0], stdv[1], stdv[sw2], (T *)out, swap_rb)
    if (layout == LAYOUT_NCHW) { if (OW == 128) TLK_PIL_LAUNCH(LAYOUT_NCHW, 128); else TLK_PIL_LAUNCH(LAYOUT_NCHW, 0); }
    else { if (OW == 128) TLK_PIL_LAUNCH(LAYOUT_NHWC, 128); else TLK_PIL_LAUNCH(LAYOUT_NHWC, 0); }
#undef TLK_PIL_LAUNCH
    return TLK_OK;
}

extern "C" int tlk_roi_crop_pil_resize_norm(const uint8_t *frames_dev, int batch, int h, int w, const double *boxes_xyxy_dev, int box_stride,
                                            const int32_t *counts_dev, int max_n, int out_h, int out_w, const float *mean3,
                                            const float *std3, int layout, int dtype, void *out_dev, void *hip_stream)
{
    if (batch < 0 || h <= 0 || w <= 0 || max_n < 0 || out_h <= 0 || out_w <= 0 || box_stride < 4) return fail(TLK_EINVAL, "tlk_roi_crop_pil_resize_norm: bad size");
    if (out_w % 8 != 0) return fail(TLK_EINVAL, "tlk_roi_crop_pil_resize_norm: out_w must be a multiple of 8");
    const int swap_rb = (layout & TLK_SWAP_RB) ? 1 : 0;
    layout &= ~TLK_SWAP_RB;
    if (layout < 0 || layout > 1 || dtype < 0 || dtype > 2) return fail(TLK_EINVAL, "tlk_roi_crop_pil_resize_norm: bad layout/dtype");
    if (batch == 0 || max_n == 0) return TLK_OK;
    if (!frames_dev || !boxes_xyxy_dev || !counts_dev || !mean3 || !std3 || !out_dev) return fail(TLK_EINVAL, "tlk_roi_crop_pil_resize_norm: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == 0) launch_pil_crop<float>(frames_dev, batch, h, w, boxes_xyxy_dev, box_stride, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb);
    else if (dtype == 1) launch_pil_crop<__half>(frames_dev, batch, h, w, boxes_xyxy_dev, box_stride, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb);
    else launch_pil_crop<bf16_t>(frames_dev, batch, h, w, boxes_xyxy_dev, box_stride, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_yolox_decode_nms(const float *pred_dev, int batch, int size, int num_classes, float ratio, float nms_thr,
                                    float score_thr, int img_w, int img_h, int max_out, float *ltwh_dev, float *xyxy_dev,
                                    float *scores_dev, int32_t *cls_dev, int32_t *counts_dev, double *trk_in_dev,
                                    int64_t det_id_base, double category_id, void *hip_stream)
{
    if (batch < 0 || size <= 0 || size % 32 != 0 || num_classes < 1 || max_out < 0) return fail(TLK_EINVAL, "tlk_yolox_decode_nms: bad size");
    if (batch == 0) return TLK_OK;
    if (!pred_dev || !ltwh_dev || !xyxy_dev || !scores_dev || !cls_dev || !counts_dev) return fail(TLK_EINVAL, "tlk_yolox_decode_nms: null pointer");
    hipLaunchKernelGGL(yolox_decode_nms_kernel, dim3(batch), dim3(BLOCK), 0, (hipStream_t)hip_stream, pred_dev, size, num_classes,
                       ratio, nms_thr, score_thr, img_w, img_h, max_out, ltwh_dev, xyxy_dev, scores_dev, (int *)cls_dev, (int *)counts_dev,
                       trk_in_dev, (long long)det_id_base, category_id);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
