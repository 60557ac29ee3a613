// tlk_image.hip -- detector / ReID pre- and post-processing on gfx950:
//   * batched letterbox (rtmlib YOLOX.preprocess semantics, cv2 INTER_LINEAR fixed-point bilinear)
//   * ROI crop -> resize -> normalize for ReID patches (KPReId.preprocess semantics)
//   * YOLOX grid decode + per-class greedy NMS with wavefront ballot bitmasks
// HBM-bound byte movers: each thread produces 8 consecutive output pixels so every store is a
// 16-byte-per-lane coalesced global_store_dwordx4; source rows are re-used through L1/L2.
#include <hip/hip_fp16.h>

#include "tlk_common.hpp"
#include "tlk_image_common.hpp"

#include <map>
#include <mutex>
#include <type_traits>

using namespace tlk;

namespace {
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
                                                     T *__restrict__ out, int swap_rb, const int *__restrict__ slot_base)
{
    const int groups_per_row = OW / 8;
    const int per_crop = groups_per_row * OH;
    const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= (long long)per_crop * B * max_n) return;
    const int slot = (int)(gid / per_crop);
    const int rem = (int)(gid - (long long)slot * per_crop);
    const int y = rem / groups_per_row, x_base = (rem - y * groups_per_row) * 8;
    const int b = slot / max_n, i = slot - b * max_n;
    const int oslot = slot_base ? slot_base[b] + i : slot;      // compact output (tlk_roi_crop_resize_norm_compact): crop i of frame b lands at slot_base[b] + i
    if (slot_base && i >= counts[b]) return;                       // (compact form: a padding slot has no place to be zero-filled)
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
            *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)oslot * 3 + c) * OH + y) * OW + x_base) = p;
        }
    } else {
        T *o = out + (((size_t)oslot * OH + y) * OW + x_base) * 3;
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

struct XCoef { short off; short w0, w1, pad; };       // byte offset of tap 0 inside the staged row, weights; tap 1 = off + step
static_assert(sizeof(XCoef) == 8, "XCoef");




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

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
template <typename T> struct LutBits;
template <> struct LutBits<float> { using type = unsigned int; };
template <> struct LutBits<__half> { using type = unsigned short; };
template <> struct LutBits<bf16_t> { using type = unsigned short; };

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


// a workgroup of the wave kernels owns CF_BANDS consecutive bands (CS_BAND output rows each) of ONE crop

// rare path kept out of line so that its registers do not count against the main loop's occupancy
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
                                                          float d0, float d1, float d2, T *__restrict__ out, int swap_rb, int nwg, const int *__restrict__ slot_base)
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
    const int oslot = slot_base ? slot_base[b] + i : slot;      // compact output (tlk_roi_crop_resize_norm_compact): crop i of frame b lands at slot_base[b] + i
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
                                                       swap_rb, out, (size_t)oslot);
                else
                    for (int k = 0; k < 8; ++k)
                        for (int c = 0; c < 3; ++c) {
                            if (LAYOUT == LAYOUT_NCHW) out[(((size_t)oslot * 3 + c) * OH + y) * OW + x_base + k] = cvt<T>(0.f);
                            else out[(((size_t)oslot * OH + y) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
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
            uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)oslot * OH + row0 + mb * mbh) * OW * 3);
#pragma unroll
            for (int k = 0; k < NST; ++k) stream_store(g + k * WAVE + lane, blk[k]);
        } else if (act) {
            if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Pack<T, 8> p;
#pragma unroll
                    for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
                    *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)oslot * 3 + c) * OH + y) * OW + x_base) = p;
                }
            } else {
                T *o = out + (((size_t)oslot * OH + y) * OW + x_base) * 3;
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
                                                          float d0, float d1, float d2, T *__restrict__ out, int swap_flags, int nwg, const int *__restrict__ slot_base)
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
    const int oslot = slot_base ? slot_base[b] + i : slot;      // compact output (tlk_roi_crop_resize_norm_compact): crop i of frame b lands at slot_base[b] + i
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
                                                       swap_rb, out, (size_t)oslot);
                else
                    for (int k = 0; k < 8; ++k)
                        for (int c = 0; c < 3; ++c) {
                            if (LAYOUT == LAYOUT_NCHW) out[(((size_t)oslot * 3 + c) * OH + y) * OW + x_base + k] = cvt<T>(0.f);
                            else out[(((size_t)oslot * OH + y) * OW + x_base + k) * 3 + c] = cvt<T>(0.f);
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
            uint4 *g = reinterpret_cast<uint4 *>(out + ((size_t)oslot * OH + row0 + mb * mbh) * OW * 3);
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
                    *reinterpret_cast<Pack<T, 8> *>(out + (((size_t)oslot * 3 + c) * OH + y) * OW + x_base) = p;
                }
            } else {
                T *o = out + (((size_t)oslot * OH + y) * OW + x_base) * 3;
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
                const float *mean, const float *stdv, int layout, void *out, hipStream_t st, int swap_rb, const int *slot_base = nullptr)
{
    const long long units = (long long)B * max_n * OH * (OW / 8);
    const dim3 grid((unsigned)((units + BLOCK - 1) / BLOCK));
    // swap_rb: output channel c is source channel 2 - c; the kernels normalise per SOURCE channel and exchange 0 <-> 2 at the store
    const int sw0 = swap_rb ? 2 : 0, sw2 = swap_rb ? 0 : 2;
    const float m0 = mean[sw0] * 255.f, m1 = mean[1] * 255.f, m2 = mean[sw2] * 255.f;
    const float d0 = 1.0f / (stdv[sw0] * 255.f), d1 = 1.0f / (stdv[1] * 255.f), d2 = 1.0f / (stdv[sw2] * 255.f);
    // Two routes (r04: the five superseded generations -- crop_lds / crop_sep / crop_fat / crop_wave / crop_pw -- are gone):
    //   * the ReID shape, 128 output columns: free-running wavefronts over mini-bands of a crop, separable passes through LDS, normalisation
    //     from a table -- crop_wave3_kernel for 16-bit outputs (sliding source window), crop_wave2_kernel for fp32 (the reference-precision run);
    //   * every other shape: crop_kernel, the direct form (one lane = 8 output pixels, taps straight from global memory), any size and pitch.
    if (OW == 128) {
        const T *lut = (const T *)crop_lut<T>(m0, m1, m2, d0, d1, d2);
        if (!lut) return TLK_EHIP;
        const int bands = (OH + CS_BAND - 1) / CS_BAND, chunks = (bands + CF_BANDS - 1) / CF_BANDS;
        const int nwg2 = (int)((long long)B * max_n * chunks);
        const size_t smem3 = (size_t)128 * 8 + ((3 * CS_LUT_N * sizeof(T) + 15) & ~(size_t)15) + (size_t)NWAVES * WV_WAVE_LDS;
        if constexpr (sizeof(T) == 2) {
            static const int p16_on = [] { const char *e = getenv("TLK_CROP_P16"); return e ? atoi(e) : 1; }();     // 0: the general-pitch code also for 16-byte-multiple row pitches (tests)
            const bool p16 = p16_on && ((long long)W * 3) % 16 == 0;
#define TLK_CW3(LAY, P) hipLaunchKernelGGL((crop_wave3_kernel<T, LAY, P, false>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, (swap_rb ? 1 : 0), nwg2, slot_base)
            if (layout == LAYOUT_NCHW) { if (p16) TLK_CW3(LAYOUT_NCHW, true); else TLK_CW3(LAYOUT_NCHW, false); }
            else { if (p16) TLK_CW3(LAYOUT_NHWC, true); else TLK_CW3(LAYOUT_NHWC, false); }
#undef TLK_CW3
        } else if (layout == LAYOUT_NCHW)
            hipLaunchKernelGGL((crop_wave2_kernel<T, LAYOUT_NCHW>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2, slot_base);
        else
            hipLaunchKernelGGL((crop_wave2_kernel<T, LAYOUT_NHWC>), dim3(nwg2), dim3(BLOCK), smem3, st, frames, B, H, W, boxes, counts, max_n, OH, lut, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, nwg2, slot_base);
        return TLK_OK;
    }
    if (layout == LAYOUT_NCHW)
        hipLaunchKernelGGL((crop_kernel<T, LAYOUT_NCHW>), grid, dim3(BLOCK), 0, st, frames, B, H, W, boxes, counts, max_n, OH, OW, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, slot_base);
    else
        hipLaunchKernelGGL((crop_kernel<T, LAYOUT_NHWC>), grid, dim3(BLOCK), 0, st, frames, B, H, W, boxes, counts, max_n, OH, OW, m0, m1, m2, d0, d1, d2, (T *)out, swap_rb, slot_base);
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

namespace {
int roi_crop_common(const char *fn, const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev, const int32_t *counts_dev, int max_n,
                    int out_h, int out_w, const float *mean3, const float *std3, int layout, int dtype, void *out_dev, const int32_t *slot_base_dev,
                    void *hip_stream);

// slot bookkeeping of the compact crop batch, one wavefront: base[b] = counts[0] + ... + counts[b - 1] (counts clamped to [0, max_n]),
// total[0] = their sum, slot_of[b * max_n + i] = base[b] + i for i < counts[b] (position of crop i of frame b in the compact batch) and,
// for the padding slots, total - 1 clamped to 0 (any valid row: consumers gather through slot_of and ignore i >= counts[b])
__global__ void __launch_bounds__(WAVE) crop_slot_bases_kernel(const int *__restrict__ counts, int batch, int max_n, int *__restrict__ base,
                                                              int *__restrict__ total, long long *__restrict__ slot_of)
{
    const int lane = threadIdx.x;
    int run = 0;
    for (int b0 = 0; b0 < batch; b0 += WAVE) {
        const int b = b0 + lane;
        int c = b < batch ? counts[b] : 0;
        c = c < 0 ? 0 : (c > max_n ? max_n : c);
        int incl = c;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (b < batch) base[b] = run + incl - c;
        run += __shfl(incl, WAVE - 1);
    }
    if (lane == 0) total[0] = run;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (slot_of) {
        const int last = run > 0 ? run - 1 : 0;
        for (int k = lane; k < batch * max_n; k += WAVE) {
            const int b = k / max_n, i = k - b * max_n;
            int c = counts[b];
            c = c < 0 ? 0 : (c > max_n ? max_n : c);
            slot_of[k] = i < c ? base[b] + i : last;
        }
    }
}
}  // namespace

extern "C" int tlk_crop_slot_bases(const int32_t *counts_dev, int batch, int max_n, int32_t *base_dev, int32_t *total_dev, int64_t *slot_of_dev, void *hip_stream)
{
    if (batch < 0 || max_n < 0) return fail(TLK_EINVAL, "tlk_crop_slot_bases: negative size");
    if (!counts_dev || !base_dev || !total_dev) return fail(TLK_EINVAL, "tlk_crop_slot_bases: null pointer");
    hipLaunchKernelGGL(crop_slot_bases_kernel, dim3(1), dim3(WAVE), 0, (hipStream_t)hip_stream, counts_dev, batch, max_n, base_dev, total_dev, (long long *)slot_of_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_roi_crop_resize_norm_compact(const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev,
                                                const int32_t *counts_dev, const int32_t *slot_base_dev, int max_n, int out_h, int out_w, const float *mean3,
                                                const float *std3, int layout, int dtype, void *out_dev, void *hip_stream)
{
    if (!slot_base_dev) return fail(TLK_EINVAL, "tlk_roi_crop_resize_norm_compact: null slot_base_dev");
    return roi_crop_common("tlk_roi_crop_resize_norm_compact", frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, dtype,
                           out_dev, slot_base_dev, hip_stream);
}

extern "C" int tlk_roi_crop_resize_norm(const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev,
                                        const int32_t *counts_dev, int max_n, int out_h, int out_w, const float *mean3,
                                        const float *std3, int layout, int dtype, void *out_dev, void *hip_stream)
{
    return roi_crop_common("tlk_roi_crop_resize_norm", frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, dtype, out_dev,
                           nullptr, hip_stream);
}

namespace {
int roi_crop_common(const char *fn, const uint8_t *frames_dev, int batch, int h, int w, const float *boxes_ltwh_dev, const int32_t *counts_dev, int max_n,
                    int out_h, int out_w, const float *mean3, const float *std3, int layout, int dtype, void *out_dev, const int32_t *slot_base_dev,
                    void *hip_stream)
{
    if (batch < 0 || h <= 0 || w <= 0 || max_n < 0 || out_h <= 0 || out_w <= 0) return fail(TLK_EINVAL, std::string(fn) + ": bad size");
    if (out_w % 8 != 0) return fail(TLK_EINVAL, std::string(fn) + ": out_w must be a multiple of 8");
    const int swap_rb = (layout & TLK_SWAP_RB) ? 1 : 0;
    layout &= ~TLK_SWAP_RB;
    if (layout < 0 || layout > 1 || dtype < 0 || dtype > 2) return fail(TLK_EINVAL, std::string(fn) + ": bad layout/dtype");
    if (batch == 0 || max_n == 0) return TLK_OK;
    if (!frames_dev || !boxes_ltwh_dev || !counts_dev || !mean3 || !std3 || !out_dev) return fail(TLK_EINVAL, std::string(fn) + ": null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == 0) launch_crop<float>(frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb, slot_base_dev);
    else if (dtype == 1) launch_crop<__half>(frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb, slot_base_dev);
    else launch_crop<bf16_t>(frames_dev, batch, h, w, boxes_ltwh_dev, counts_dev, max_n, out_h, out_w, mean3, std3, layout, out_dev, st, swap_rb, slot_base_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
}  // namespace
