// tlk_pil.hip -- the ReID input of plain StrongSORT / BoT-SORT / Deep-OC-SORT: crop + Pillow Image.resize(BILINEAR) +
// ToTensor + Normalize, bit for bit Pillow's Resample.c (r04: moved out of tlk_image.hip).
#include "tlk_image_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
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



}  // namespace

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


