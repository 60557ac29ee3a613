// tlk_conv16.hip -- the backbones' convolutions on the 16-bit MFMA (v_mfma_f32_32x32x16_f16, 16x the rate of the fp32-input MFMA), two modes:
//
//   MODE_F16   activations / weights f16, fp32 accumulation, epilogue (bias, residual, ReLU / SiLU) in fp32, output f16: the f16 leg, with the
//              epilogue INSIDE the convolution (r01-r03 ran MIOpen / CK / hipBLASLt + a separate bias_act pass for the 3x3 ones).
//   MODE_SPLIT every fp32 value x travels as TWO f16 numbers, hi = f16(x) and lo = f16((x - hi) * 2^11): x = hi + lo * 2^-11 to a relative 2^-22
//              (an f16 significand is 11 bits, so the pair holds 22; the scale keeps lo a NORMAL f16 whenever hi is; for |x| < 2^-14, where
//              hi is subnormal, the pair is exact to an absolute 2^-35).  A product a*b is
//              a_hi*b_hi + 2^-11 (a_hi*b_lo + a_lo*b_hi) + O(2^-22 |ab|): THREE MFMAs per operand pair, two fp32 accumulators (the head and
//              the cross terms), merged as acc_hh + 2^-11 acc_x in the epilogue.  Every f16 x f16 product is exact in fp32 (22 bits) and the
//              sums are fp32, so the result carries fp32-class error -- |err| <= ~3 * 2^-22 * sum|a||b| representation error plus the same
//              accumulation round-off as any fp32 kernel -- at a third of the f16 MFMA rate = ~5x the fp32-input MFMA peak.  gfx950 has no
//              TF32 / xf32 path; this is how fp32-grade convolutions get onto the fast matrix pipe.  Range: |x| <= 65504 (values beyond
//              saturate to f16 infinity, as they would in the f16 leg); the tests compare with fp64 at the SAME bound as the exact-fp32 kernel.
//              Inputs, outputs and residuals are (hi, lo) plane pairs of the same NHWC shape; OUT_F32 writes plain fp32 instead.
//
// Kernel structure = tlk_conv.hip's, byte for byte where it can be: implicit GEMM, 128-byte K slices per LDS row (64 f16 of one plane, or
// 32 hi | 32 lo), rows padded to 144 B (conflict-free ds_read_b128 / ds_write_b128), buffer loads with hardware zero-fill so a K step is one
// basic block, LDS double-buffered, scheduling barriers between MFMA chunks.  The 16-bit MFMA is 2-5x shorter per K step than the fp32 one, so
// the global loads run TWO steps ahead in two register sets (issued in step s-1, written to LDS in the second half of step s, read in step s+1).
#include "tlk_conv16.hpp"

using namespace tlk;
using namespace tlk::c16;

namespace {

constexpr int LDB = ROW_BYTES + 16;      // LDS row of the register-staged kernel: 128 data bytes + 16 pad

template <int TM, int TN, int WGM, int WGN, int ACT, bool RES, int MODE, bool OUT_F32>
__global__ void __launch_bounds__(64 * WGM * WGN) conv16_mfma_kernel(const Conv16Args p)
{
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int PLANES = MODE == MODE_SPLIT ? 2 : 1;
    constexpr int BKE = MODE == MODE_SPLIT ? 32 : 64;                 // K elements per step
    constexpr int CH = ROW_BYTES / 16 / PLANES;                       // 16-byte chunks per row and plane: 8 or 4
    constexpr int ROWS_PER_PASS = NT / CH;
    constexpr int PA = BM / ROWS_PER_PASS, PB = BN / ROWS_PER_PASS;   // passes per plane
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile rows must be a multiple of the loader pass");
    constexpr int NL = PLANES * (PA + PB);                            // 16-byte loads per lane and K step
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *As = lds;                                          // [2][BM][LDB]
    unsigned char *Bs = lds + 2 * BM * LDB;                           // [2][BN][LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const long long M = live_rows(p);                      // dynamic batch: the XCD-aware order is laid over the LIVE tiles (see tlk_conv16x.hip)
    const long long live_tiles = ((M + BM - 1) / BM) * p.tiles_n;
    if ((long long)blockIdx.x >= live_tiles) return;
    const long long tile = xcd_tile(live_tiles);
    const long long m0 = (tile / p.tiles_n) * BM;
    const int n0 = (int)(tile % p.tiles_n) * BN;

    // ---- loader geometry (see tlk_conv.hip): lane = chunk `lc` of row `lr + pass * ROWS_PER_PASS` of one plane
    const int lr = tid / CH, lc = tid % CH;
    constexpr int OOB = (int)0x80000000;
    long long base_pix;
    {
        const unsigned hw = (unsigned)(p.Ho * p.Wo), n = (unsigned)m0 / hw;
        const int rem = (int)((unsigned)m0 - n * hw), ho = rem / p.Wo;
        const int hi = ho * p.stride - p.pad;
        base_pix = (long long)n * p.H * p.W + (long long)(hi > 0 ? hi : 0) * p.W;
    }
    const long long total_pix = (long long)((unsigned)p.M / (unsigned)(p.Ho * p.Wo)) * p.H * p.W;
    long long a_bytes = ((total_pix - base_pix - 1) * p.x_pix + p.Cin) * 2;
    if (a_bytes > 0x7ffffff0LL) a_bytes = 0x7ffffff0LL;
    const int w_bytes = (int)((long long)p.Cout * p.K * 2);
    __amdgpu_buffer_rsrc_t rs_a[PLANES], rs_b[PLANES];
    rs_a[0] = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + base_pix * p.x_pix), 0, (int)a_bytes, 0x00020000);
    rs_b[0] = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, w_bytes, 0x00020000);
    if (MODE == MODE_SPLIT) {
        rs_a[PLANES - 1] = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x_lo + base_pix * p.x_pix), 0, (int)a_bytes, 0x00020000);
        rs_b[PLANES - 1] = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_lo, 0, w_bytes, 0x00020000);
    }
    int a_hi0[PA], a_wi0[PA], a_rel[PA];
    bool a_ok[PA];
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) {
        const long long m = m0 + lr + ps * ROWS_PER_PASS;
        a_ok[ps] = m < M;
        const long long mm = a_ok[ps] ? m : m0;
        const unsigned n = (unsigned)mm / (unsigned)(p.Ho * p.Wo);
        const int rem = (int)((unsigned)mm - n * (unsigned)(p.Ho * p.Wo));
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_hi0[ps] = ho * p.stride - p.pad; a_wi0[ps] = wo * p.stride - p.pad;
        a_rel[ps] = (int)((long long)n * p.H * p.W - base_pix) + a_hi0[ps] * p.W + a_wi0[ps];
    }
    int b_off[PB];
#pragma unroll
    for (int ps = 0; ps < PB; ++ps) {
        const int co = n0 + lr + ps * ROWS_PER_PASS;
        b_off[ps] = co < p.Cout ? co * p.K * 2 : OOB;
    }

    i32x4 rg[2][NL];                                       // two staging register sets: loads run two K steps ahead of the MFMAs
    // Tap of the step being loaded.  The 16-bit MFMA leaves only a few hundred cycles per K step, so the integer work per load matters: when
    // Cin is a multiple of the step (every layer but an RGB stem) the whole step lies inside ONE tap, (kh, kw, ci0) advance by increments
    // (wave-uniform, scalar unit) and a load's offset is one vector add; otherwise the general form divides.
    const bool tap_uniform = p.Cin % BKE == 0;
    int u_kh = 0, u_kw = 0, u_ci0 = 0, u_k0 = 0;           // uniform: tap and first channel of the step (valid when tap_uniform)
    int t_kh = 0, t_kw = 0, t_ci = 0, t_k = 0;
    bool t_in = false;
    bool first_tap = true;
    auto set_tap = [&](int k0) {                           // called with k0 = 0, BKE, 2 BKE, ... in order
        if (tap_uniform) {
            if (!first_tap) { u_k0 += BKE; u_ci0 += BKE; if (u_ci0 >= p.Cin) { u_ci0 = 0; if (++u_kw == p.KW) { u_kw = 0; ++u_kh; } } }
            first_tap = false;
            t_k = u_k0 + lc * 8; t_in = u_k0 < p.K; t_kh = u_kh; t_kw = u_kw; t_ci = u_ci0 + lc * 8;
        } else {
            t_k = k0 + lc * 8;
            t_in = t_k < p.K;
            t_kh = 0; t_kw = 0; t_ci = t_k;
            if (p.KH * p.KW != 1) { const int tap = t_k / p.Cin; t_ci = t_k - tap * p.Cin; t_kh = tap / p.KW; t_kw = tap - t_kh * p.KW; }
        }
    };
    // load i of a step: i = plane * (PA + PB) + (pass of A | PA + pass of B)
    auto issue_load = [&](int set, int i) {
        const int plane = i / (PA + PB), j = i % (PA + PB);
        if (j < PA) {
            const int hi = a_hi0[j] + t_kh, wi = a_wi0[j] + t_kw;
            const bool ok = t_in && a_ok[j] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int off = ((a_rel[j] + t_kh * p.W + t_kw) * p.x_pix + t_ci) * 2;
            rg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a[plane], ok ? off : OOB, 0, 0);
        } else {
            const int bo = b_off[j - PA];
            rg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b[plane], (t_in && bo != OOB) ? bo + t_k * 2 : OOB, 0, 0);
        }
    };
    auto issue_store = [&](int set, int i, int buf) {
        const int plane = i / (PA + PB), j = i % (PA + PB);
        unsigned char *dst = j < PA ? As + (buf * BM + lr + j * ROWS_PER_PASS) * LDB : Bs + (buf * BN + lr + (j - PA) * ROWS_PER_PASS) * LDB;
        *reinterpret_cast<i32x4 *>(dst + plane * (ROW_BYTES / 2) + lc * 16) = rg[set][i];
    };

    constexpr int NACC = MODE == MODE_SPLIT ? 2 : 1;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;

    const int steps = (p.K + BKE - 1) / BKE;
    // prologue: step 0 straight into LDS, step 1 into register set 1
    set_tap(0);
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_load(0, i);
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_store(0, i, 0);
    set_tap(BKE);
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_load(1, i);
    __syncthreads();

    // fragments: lane l holds row (l & 31), 8 consecutive k starting at 8 * (l >> 5) of a 16-wide slice
    const int frag_off = (lane & 31) * LDB + (lane >> 5) * 16;
    const unsigned char *a_frag = As + (wm * TM * 32) * LDB + frag_off, *b_frag = Bs + (wn * TN * 32) * LDB + frag_off;
    constexpr int NJ = BKE / 16;                           // 16-wide slices per step: 4 (f16) or 2 (split)
    h16x8 fa[2][PLANES][TM], fb[2][PLANES][TN];
    auto read_frags = [&](int buf, int j, int set) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[set][pl][i] = *reinterpret_cast<const h16x8 *>(a_frag + (buf * BM + i * 32) * LDB + pl * (ROW_BYTES / 2) + j * 32);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[set][pl][i] = *reinterpret_cast<const h16x8 *>(b_frag + (buf * BN + i * 32) * LDB + pl * (ROW_BYTES / 2) + j * 32);
        }
    };
    read_frags(0, 0, 0);

    // One K step.  P = parity of the step = LDS buffer it reads = register set it REFILLS (with the loads of step s + 2) in its first half;
    // in its second half it writes the other set (loaded during step s - 1, for step s + 1) to the other LDS buffer.
    auto k_step = [&](auto parity, int s) {
        constexpr int P = decltype(parity)::value;
        set_tap((s + 2) * BKE);                            // beyond K near the end: the loads return zeros and land in a dead buffer
        constexpr int NCH = 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = MODE == MODE_SPLIT ? (c >> 1) : c, fset = j & 1;
            const bool first_of_j = MODE == MODE_SPLIT ? (c & 1) == 0 : true;
            if (first_of_j && j + 1 < NJ) read_frags(P, j + 1, fset ^ 1);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                if (c < NCH / 2 && (i * (NCH / 2)) / NL == c) issue_load(P, i);
                if (c >= NCH / 2 && (i * (NCH / 2)) / NL == c - NCH / 2) issue_store(P ^ 1, i, P ^ 1);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    if (MODE == MODE_F16) {
                        acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fset][0][i], fb[fset][0][jj], acc[0][i][jj], 0, 0, 0);
                    } else if ((c & 1) == 0) {
                        acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fset][0][i], fb[fset][0][jj], acc[0][i][jj], 0, 0, 0);
                    } else {
                        acc[NACC - 1][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fset][0][i], fb[fset][PLANES - 1][jj], acc[NACC - 1][i][jj], 0, 0, 0);
                        acc[NACC - 1][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fset][PLANES - 1][i], fb[fset][0][jj], acc[NACC - 1][i][jj], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        read_frags(P ^ 1, 0, 0);
    };
    for (int s = 0; s < steps; s += 2) {
        k_step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < steps) k_step(std::integral_constant<int, 1>{}, s + 1);
    }
    __syncthreads();

    // ---- epilogue through LDS (fp32 tile), 8 output elements per lane
    constexpr int LDC = BN + 4;
    float *Cs = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            float *c = Cs + ((wm * TM + i) * 32 + 4 * (lane >> 5)) * LDC + (wn * TN + jj) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[0][i][jj][r];
                if (MODE == MODE_SPLIT) v = v + acc[NACC - 1][i][jj][r] * LO_INV;
                c[((r & 3) + 8 * (r >> 2)) * LDC] = v;
            }
        }
    __syncthreads();
    constexpr int V_PER_ROW = BN / 8, NVEC = BM * V_PER_ROW, ITS = (NVEC + NT - 1) / NT;
    const bool scaled = MODE == MODE_SPLIT && (p.s_in || p.s_res || p.s_out);      // wave-uniform: the unscaled call runs the r05 epilogue, instruction for instruction
    const SplitScales sc = load_scales(p);
    float amax = 0.f;
    for (int it = 0; it < ITS; ++it) {
        const int idx = it * NT + tid;
        const int row = idx / V_PER_ROW, ec = (idx - row * V_PER_ROW) * 8;
        const long long m = m0 + row;
        const int co = n0 + ec;
        if ((NVEC % NT != 0 && idx >= NVEC) || m >= M || co >= p.Cout) continue;      // Cout % 8 == 0 (checked by the host side)
        float v[8];
        {
            const float4 c0 = *reinterpret_cast<const float4 *>(Cs + row * LDC + ec), c1 = *reinterpret_cast<const float4 *>(Cs + row * LDC + ec + 4);
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (p.bias) { b0 = *reinterpret_cast<const float4 *>(p.bias + co); b1 = *reinterpret_cast<const float4 *>(p.bias + co + 4); }
            if (MODE == MODE_SPLIT && scaled) {      // scaled planes (r06): the accumulators hold the sum over x / s_in
                v[0] = c0.x * sc.in + b0.x; v[1] = c0.y * sc.in + b0.y; v[2] = c0.z * sc.in + b0.z; v[3] = c0.w * sc.in + b0.w;
                v[4] = c1.x * sc.in + b1.x; v[5] = c1.y * sc.in + b1.y; v[6] = c1.z * sc.in + b1.z; v[7] = c1.w * sc.in + b1.w;
            } else {
                v[0] = c0.x + b0.x; v[1] = c0.y + b0.y; v[2] = c0.z + b0.z; v[3] = c0.w + b0.w;
                v[4] = c1.x + b1.x; v[5] = c1.y + b1.y; v[6] = c1.z + b1.z; v[7] = c1.w + b1.w;
            }
        }
        float rv[8];
        if (RES) {
            const h16x8 rh = *reinterpret_cast<const h16x8 *>(p.res + m * p.r_pix + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (float)rh[e];
            if (MODE == MODE_SPLIT) {
                const h16x8 rl = *reinterpret_cast<const h16x8 *>(p.res_lo + m * p.r_pix + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] += (float)rl[e] * LO_INV;
                if (scaled) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] *= sc.res;
                }
            }
            if (!p.res_post) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act16<ACT>(v[e]);
        if (RES && p.res_post) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (OUT_F32) {
            float *o = p.y32 + m * p.y_pix + co;
            *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (MODE == MODE_SPLIT) {
            h16x8 oh, ol;
            if (scaled) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { amax = fmaxf(amax, fabsf(v[e])); v[e] *= sc.out_inv; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f32(v[e], h, l); oh[e] = h; ol[e] = l; }
            *reinterpret_cast<h16x8 *>(p.y + m * p.y_pix + co) = oh;
            *reinterpret_cast<h16x8 *>(p.y_lo + m * p.y_pix + co) = ol;
        } else {
            h16x8 oh;
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[e] = (_Float16)v[e];
            *reinterpret_cast<h16x8 *>(p.y + m * p.y_pix + co) = oh;
        }
    }
    if (MODE == MODE_SPLIT && !OUT_F32 && scaled) note_amax(p, amax);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant (r04b): the layers whose input-channel count is a multiple of the K step (every layer but an RGB stem).
// `global_load_lds_dwordx4` moves 16 bytes per lane from global memory straight into LDS -- no staging registers, no ds_write pass -- but
// its destination is wave-uniform base + lane * 16, so the LDS rows are UNPADDED (128 B) and the bank spreading comes from an XOR swizzle:
// logical 16-byte chunk q of row i is stored at chunk position q ^ ((i >> 1) & 7).  The swizzle is applied on the SOURCE side (a lane
// fetches the logical chunk that belongs at its position) and again when the fragments are read.  One wavefront-instruction fills 8 rows.
// 128 x 128 tile, 4 wavefronts of 64 x 64, two LDS stages (64 KB -> two workgroups per CU in both modes, which the register-staged split
// kernel could not have): the loads of step s + 1 are in flight while step s computes; masked lanes read a 16-byte zero page.
template <int ACT, bool RES, int MODE, bool OUT_F32>
__global__ void __launch_bounds__(256, 2) conv16_glds_kernel(const Conv16Args p, const unsigned char *__restrict__ zeros)
{
    constexpr int TM = 2, TN = 2, WGN = 2, NT = 256, BM = 128, BN = 128;
    constexpr int PLANES = MODE == MODE_SPLIT ? 2 : 1;
    constexpr int BKE = MODE == MODE_SPLIT ? 32 : 64;
    constexpr int STAGE = (BM + BN) * ROW_BYTES;                       // 32 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const long long M = live_rows(p);                      // dynamic batch: the XCD-aware order is laid over the LIVE tiles (see tlk_conv16x.hip)
    const long long live_tiles = ((M + BM - 1) / BM) * p.tiles_n;
    if ((long long)blockIdx.x >= live_tiles) return;
    const long long tile = xcd_tile(live_tiles);
    const long long m0 = (tile / p.tiles_n) * BM;
    const int n0 = (int)(tile % p.tiles_n) * BN;

    // ---- loader: instruction q (0..3) of this wavefront fills rows [(q * 4 + wave) * 8, + 8) of A and of B; lane = (row r8 = lane >> 3, position pc)
    const int r8 = lane >> 3, pc = lane & 7;
    const unsigned char *a_ptr[4][PLANES > 1 ? 1 : 1];
    const unsigned char *a_row[4], *b_row[4];          // global address of the lane's LOGICAL chunk at tap (0, 0) / k = 0; nullptr = row masked
    int a_hi0[4], a_wi0[4];
    (void)a_ptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (q * 4 + wave) * 8 + r8;
        const int lcq = pc ^ ((row >> 1) & 7);                          // logical chunk stored at this lane's position
        const int plane = MODE == MODE_SPLIT ? (lcq >> 2) : 0, kc = MODE == MODE_SPLIT ? (lcq & 3) : lcq;
        const long long m = m0 + row;
        a_row[q] = nullptr; a_hi0[q] = 0; a_wi0[q] = 0;
        if (m < M) {
            const unsigned n = (unsigned)m / (unsigned)(p.Ho * p.Wo);
            const int rem = (int)((unsigned)m - n * (unsigned)(p.Ho * p.Wo));
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            a_hi0[q] = ho * p.stride - p.pad; a_wi0[q] = wo * p.stride - p.pad;
            const _Float16 *base = plane ? p.x_lo : p.x;
            a_row[q] = reinterpret_cast<const unsigned char *>(base + ((long long)n * p.H * p.W + (long long)a_hi0[q] * p.W + a_wi0[q]) * p.x_pix + kc * 8);
        }
        const int co = n0 + row;
        b_row[q] = co < p.Cout ? reinterpret_cast<const unsigned char *>((plane ? p.w_lo : p.w) + (long long)co * p.K + kc * 8) : nullptr;
    }
    const bool no_halo = p.KH == 1 && p.KW == 1 && p.pad == 0;
    int u_kh = 0, u_kw = 0, u_ci0 = 0, u_k0 = 0;                       // tap of the step being loaded (wave-uniform)
    auto issue_stage = [&](int buf) {
        const long long a_add = ((long long)(u_kh * p.W + u_kw) * p.x_pix + u_ci0) * 2, b_add = (long long)u_k0 * 2;
        unsigned char *la = lds + buf * STAGE, *lb = la + BM * ROW_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bool ok = a_row[q] != nullptr;
            if (!no_halo) { const int hi = a_hi0[q] + u_kh, wi = a_wi0[q] + u_kw; ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W; }
            const unsigned char *ga = ok ? a_row[q] + a_add : zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ga,
                                             (__attribute__((address_space(3))) void *)(la + (q * 4 + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned char *gb = b_row[q] ? b_row[q] + b_add : zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gb,
                                             (__attribute__((address_space(3))) void *)(lb + (q * 4 + wave) * 1024), 16, 0, 0);
        }
        u_k0 += BKE; u_ci0 += BKE;
        if (u_ci0 >= p.Cin) { u_ci0 = 0; if (++u_kw == p.KW) { u_kw = 0; ++u_kh; } }
    };

    constexpr int NACC = MODE == MODE_SPLIT ? 2 : 1;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;

    // fragment addressing: lane l reads row (l & 31) (+ 32 per tile: the swizzle term (row >> 1) & 7 is the same), logical chunk 2 j + (l >> 5)
    // (+ 4 for the lo plane), stored at that chunk XOR the row's swizzle
    const int sw = ((lane & 31) >> 1) & 7, hsel = lane >> 5;
    const int row_a = (wm * TM * 32 + (lane & 31)) * ROW_BYTES, row_b = BM * ROW_BYTES + (wn * TN * 32 + (lane & 31)) * ROW_BYTES;
    constexpr int NJ = BKE / 16;
    auto chunk_off = [&](int q) { return ((q ^ sw) << 4); };

    const int steps = p.K / BKE;
    issue_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (s + 1 < steps) issue_stage(cur ^ 1);
        const unsigned char *sa = lds + cur * STAGE + row_a, *sb = lds + cur * STAGE + row_b;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            h16x8 fa[PLANES][TM], fb[PLANES][TN];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                const int off = chunk_off(pl * 4 + 2 * j + hsel);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const h16x8 *>(sa + i * 32 * ROW_BYTES + off);
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[pl][i] = *reinterpret_cast<const h16x8 *>(sb + i * 32 * ROW_BYTES + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][jj], acc[0][i][jj], 0, 0, 0);
                    if (MODE == MODE_SPLIT) {
                        acc[NACC - 1][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[PLANES - 1][jj], acc[NACC - 1][i][jj], 0, 0, 0);
                        acc[NACC - 1][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[PLANES - 1][i], fb[0][jj], acc[NACC - 1][i][jj], 0, 0, 0);
                    }
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the next stage has landed (it had the whole step to do so)
        __syncthreads();
    }

    // ---- epilogue (as conv16_mfma_kernel): fp32 tile through LDS, 8 output elements per lane
    constexpr int LDC = BN + 4;
    float *Cs = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            float *c = Cs + ((wm * TM + i) * 32 + 4 * (lane >> 5)) * LDC + (wn * TN + jj) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[0][i][jj][r];
                if (MODE == MODE_SPLIT) v = v + acc[NACC - 1][i][jj][r] * LO_INV;
                c[((r & 3) + 8 * (r >> 2)) * LDC] = v;
            }
        }
    __syncthreads();
    constexpr int V_PER_ROW = BN / 8, NVEC = BM * V_PER_ROW, ITS = NVEC / NT;
    const bool scaled = MODE == MODE_SPLIT && (p.s_in || p.s_res || p.s_out);      // wave-uniform: the unscaled call runs the r05 epilogue, instruction for instruction
    const SplitScales sc = load_scales(p);
    float amax = 0.f;
#pragma unroll 2
    for (int it = 0; it < ITS; ++it) {
        const int idx = it * NT + tid;
        const int row = idx / V_PER_ROW, ec = (idx - row * V_PER_ROW) * 8;
        const long long m = m0 + row;
        const int co = n0 + ec;
        if (m >= M || co >= p.Cout) continue;
        float v[8];
        {
            const float4 c0 = *reinterpret_cast<const float4 *>(Cs + row * LDC + ec), c1 = *reinterpret_cast<const float4 *>(Cs + row * LDC + ec + 4);
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (p.bias) { b0 = *reinterpret_cast<const float4 *>(p.bias + co); b1 = *reinterpret_cast<const float4 *>(p.bias + co + 4); }
            if (MODE == MODE_SPLIT && scaled) {      // scaled planes (r06): the accumulators hold the sum over x / s_in
                v[0] = c0.x * sc.in + b0.x; v[1] = c0.y * sc.in + b0.y; v[2] = c0.z * sc.in + b0.z; v[3] = c0.w * sc.in + b0.w;
                v[4] = c1.x * sc.in + b1.x; v[5] = c1.y * sc.in + b1.y; v[6] = c1.z * sc.in + b1.z; v[7] = c1.w * sc.in + b1.w;
            } else {
                v[0] = c0.x + b0.x; v[1] = c0.y + b0.y; v[2] = c0.z + b0.z; v[3] = c0.w + b0.w;
                v[4] = c1.x + b1.x; v[5] = c1.y + b1.y; v[6] = c1.z + b1.z; v[7] = c1.w + b1.w;
            }
        }
        float rv[8];
        if (RES) {
            const h16x8 rh = *reinterpret_cast<const h16x8 *>(p.res + m * p.r_pix + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (float)rh[e];
            if (MODE == MODE_SPLIT) {
                const h16x8 rl = *reinterpret_cast<const h16x8 *>(p.res_lo + m * p.r_pix + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] += (float)rl[e] * LO_INV;
                if (scaled) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] *= sc.res;
                }
            }
            if (!p.res_post) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act16<ACT>(v[e]);
        if (RES && p.res_post) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (OUT_F32) {
            float *o = p.y32 + m * p.y_pix + co;
            *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (MODE == MODE_SPLIT) {
            h16x8 oh, ol;
            if (scaled) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { amax = fmaxf(amax, fabsf(v[e])); v[e] *= sc.out_inv; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f32(v[e], h, l); oh[e] = h; ol[e] = l; }
            *reinterpret_cast<h16x8 *>(p.y + m * p.y_pix + co) = oh;
            *reinterpret_cast<h16x8 *>(p.y_lo + m * p.y_pix + co) = ol;
        } else {
            h16x8 oh;
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[e] = (_Float16)v[e];
            *reinterpret_cast<h16x8 *>(p.y + m * p.y_pix + co) = oh;
        }
    }
    if (MODE == MODE_SPLIT && !OUT_F32 && scaled) note_amax(p, amax);
}

}  // namespace

const unsigned char *tlk::c16::zero_page()
{
    static unsigned char *z[16] = {nullptr};          // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!z[dev]) {
        if (hipMalloc((void **)&z[dev], 256) != hipSuccess) return nullptr;
        if (hipMemset(z[dev], 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { hipFree(z[dev]); z[dev] = nullptr; return nullptr; }
    }
    return z[dev];
}

namespace {

int g_glds = -1;          // -1: read TLK_CONV16_GLDS once (default on); tlk_conv16_set_glds for A/B runs in one process
int g_cfg16x = 0;         // tlk_conv16_set_config: -1 = r04 kernels only, 0 = the large-tile kernels where their heuristic takes the shape, > 0 = force one

template <int MODE, bool OUT_F32> int launch16_glds(Conv16Args &a, int act, hipStream_t st)
{
    constexpr size_t LDS_STAGE = (size_t)2 * 256 * ROW_BYTES, LDS_C = (size_t)128 * 132 * sizeof(float);
    constexpr size_t LDS_BYTES = LDS_STAGE > LDS_C ? LDS_STAGE : LDS_C;
    const unsigned char *z = zero_page();
    if (!z) return fail(TLK_EHIP, "tlk_conv2d_nhwc_16: cannot allocate the zero page");
    a.tiles_n = (a.Cout + 127) / 128;
    a.tiles = ((a.M + 127) / 128) * a.tiles_n;
    if (a.tiles > 0x7fffffffLL || a.M > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: more than 2^31 - 1 output pixels in one launch");
    const bool res = a.res != nullptr;
#define TLK_G16_LAUNCH(A, R)                                                                                                               \
    do {                                                                                                                                   \
        auto kern = conv16_glds_kernel<A, R, MODE, OUT_F32>;                                                                               \
        static bool attr_set = false;                                                                                                      \
        if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles), dim3(256), LDS_BYTES, st, a, z);                                                 \
    } while (0)
    if (res) { if (act == 0) TLK_G16_LAUNCH(ACT_NONE, true); else if (act == 1) TLK_G16_LAUNCH(ACT_RELU, true); else TLK_G16_LAUNCH(ACT_SILU, true); }
    else { if (act == 0) TLK_G16_LAUNCH(ACT_NONE, false); else if (act == 1) TLK_G16_LAUNCH(ACT_RELU, false); else TLK_G16_LAUNCH(ACT_SILU, false); }
#undef TLK_G16_LAUNCH
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

template <int TM, int TN, int WGM, int WGN, int MODE, bool OUT_F32> int launch16(Conv16Args &a, int act, hipStream_t st)
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, NT = 64 * WGM * WGN;
    constexpr size_t LDS_STAGE = (size_t)2 * (BM + BN) * LDB, LDS_C = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t LDS_BYTES = LDS_STAGE > LDS_C ? LDS_STAGE : LDS_C;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
    if (a.tiles > 0x7fffffffLL || a.M > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: more than 2^31 - 1 output pixels in one launch");
    const bool res = a.res != nullptr;
#define TLK_C16_LAUNCH(A, R)                                                                                                               \
    do {                                                                                                                                   \
        auto kern = conv16_mfma_kernel<TM, TN, WGM, WGN, A, R, MODE, OUT_F32>;                                                             \
        static bool attr_set = false;                                                                                                      \
        if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles), dim3(NT), LDS_BYTES, st, a);                                                     \
    } while (0)
    if (res) { if (act == 0) TLK_C16_LAUNCH(ACT_NONE, true); else if (act == 1) TLK_C16_LAUNCH(ACT_RELU, true); else TLK_C16_LAUNCH(ACT_SILU, true); }
    else { if (act == 0) TLK_C16_LAUNCH(ACT_NONE, false); else if (act == 1) TLK_C16_LAUNCH(ACT_RELU, false); else TLK_C16_LAUNCH(ACT_SILU, false); }
#undef TLK_C16_LAUNCH
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// Tile choice.  Both modes move 4 (split) or 2 (f16) bytes per element through L2 -> LDS at 5x / 16x the fp32 MFMA rate, so the 128 x 128 tile
// is the smallest that keeps the CU's L2 traffic (~40-60 B/clk) below what the L2 delivers; the split mode carries two accumulators per
// tile, so it spreads the 128 x 128 tile over EIGHT wavefronts of 32 x 64 (one workgroup per CU, two wavefronts per SIMD) where the f16 mode
// uses four of 64 x 64 (two workgroups per CU).
template <int MODE, bool OUT_F32> int dispatch16(Conv16Args &a, int act, hipStream_t st)
{
    if (g_glds < 0) { const char *e = getenv("TLK_CONV16_GLDS"); g_glds = e ? atoi(e) : 1; }
    g_last_cfg16x = -1;
    if (g_cfg16x >= 0) {
        const int r = launch16x(a, MODE == MODE_SPLIT, OUT_F32, act, g_cfg16x, st);
        if (r != 1) return r;                                                            // 1 = "not a shape for the large tiles"
    }
    if (g_glds && a.Cout > 64 && a.Cin % (MODE == MODE_SPLIT ? 32 : 64) == 0) return launch16_glds<MODE, OUT_F32>(a, act, st);
    if (MODE == MODE_SPLIT) {
        if (a.Cout > 64) return launch16<1, 2, 4, 2, MODE, OUT_F32>(a, act, st);         // 128 x 128, 8 wavefronts
        return launch16<1, 2, 4, 1, MODE, OUT_F32>(a, act, st);                          // 128 x 64, 4 wavefronts
    }
    if (a.Cout > 64) return launch16<2, 2, 2, 2, MODE, OUT_F32>(a, act, st);             // 128 x 128, 4 wavefronts
    return launch16<2, 1, 2, 2, MODE, OUT_F32>(a, act, st);                              // 128 x 64
}

// fp32 NHWC (c_in channels) -> (hi, lo) f16 planes with c_out >= c_in channels (zero padded): the entry of a split-precision network
__global__ void __launch_bounds__(BLOCK) split_planes_kernel(const float *__restrict__ x, long long pixels, int c_in, int x_pix, int c_out,
                                                             _Float16 *__restrict__ hi, _Float16 *__restrict__ lo, float *__restrict__ state, const int *__restrict__ n_dyn,
                                                             long long pix_per_image)
{
    long long n = pixels * c_out;
    if (n_dyn) { const long long nd = (long long)n_dyn[0] * pix_per_image * c_out; n = nd < n ? (nd < 0 ? 0 : nd) : n; }      // dynamic batch: the live images only
    const float inv = state ? 1.f / state[0] : 1.f;
    float am = 0.f;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) {
        const long long px = i / c_out;
        const int c = (int)(i - px * c_out);
        _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
        if (c < c_in) { const float v = x[px * x_pix + c]; am = fmaxf(am, fabsf(v)); split_f32(v * inv, h, l); }
        hi[i] = h; lo[i] = l;
    }
    if (state) record_amax(state + 1, am);
}

__global__ void __launch_bounds__(BLOCK) merge_planes_kernel(const _Float16 *__restrict__ hi, const _Float16 *__restrict__ lo, long long n, float *__restrict__ y,
                                                             const float *__restrict__ scale)
{
    const float s = scale ? scale[0] : 1.f;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) y[i] = ((float)hi[i] + (float)lo[i] * LO_INV) * s;
}

// One thread per plane state {scale, bits of the largest |value| recorded since the last update}: the scale for the NEXT forward.
//   target: largest value / scale <= 2^14 (a factor 4 below float16's 65504); the scale only ever is a power of two >= 1 -- a tensor that fits float16
//   keeps scale 1 and the planes of r05, bit for bit; it GROWS at once when the recorded maximum asks for it and SHRINKS (to twice the needed
//   value) only when the maximum fell a factor 8 below what the current scale was chosen for.
//   `changed` counts the states whose scale grew or whose maximum was not finite (an overflow upstream: the calibration loop runs again).
__global__ void split_scale_update_kernel(float *__restrict__ states, int n, int *__restrict__ changed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float scale = states[2 * i];
    const float am = states[2 * i + 1];
    if (!(scale >= 1.f)) scale = 1.f;
    bool ch = false;
    if (am != am || am > 3.0e38f) ch = true;
    else if (am > 0.f) {
        int e = 0;
        (void)frexpf(am, &e);                              // am = f * 2^e, 0.5 <= f < 1: am < 2^e
        const int need = e - 14 > 0 ? (e - 14 > 100 ? 100 : e - 14) : 0;
        const float s_need = ldexpf(1.f, need);
        if (s_need > scale) { scale = s_need; ch = true; }
        else if (s_need * 8.f <= scale) scale = s_need * 2.f;
    }
    states[2 * i] = scale;
    states[2 * i + 1] = 0.f;
    if (ch && changed) atomicAdd(changed, 1);
}

}  // namespace

static int conv16_entry(const void *x_dev, const void *x_lo_dev, const void *w_dev, const void *w_lo_dev, const float *bias_dev,
                        const void *res_dev, const void *res_lo_dev, void *y_dev, void *y_lo_dev, float *y_f32_dev,
                        int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                        int x_pix_stride, int y_pix_stride, int res_pix_stride, const float *in_scale_dev, const float *res_scale_dev, float *out_state_dev,
                        void *hip_stream)
{
    if (n < 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: bad shape");
    if (cin % 8 != 0 || cout % 8 != 0) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: Cin and Cout must be multiples of 8 (pad with zero channels)");
    const int res_post = (act_kind & TLK_ACT_RES_AFTER) ? 1 : 0;
    act_kind &= ~TLK_ACT_RES_AFTER;
    if (act_kind < 0 || act_kind > 2) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: act_kind is 0 (none), 1 (ReLU) or 2 (SiLU), optionally | TLK_ACT_RES_AFTER");
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: empty output");
    if (n == 0) return TLK_OK;
    const bool split = x_lo_dev != nullptr;
    if (!x_dev || !w_dev || (split && !w_lo_dev) || (!split && w_lo_dev)) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: x / w planes do not match");
    if (res_dev && split != (res_lo_dev != nullptr)) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: residual planes do not match the mode");
    const bool out32 = y_f32_dev != nullptr;
    if (!out32 && (!y_dev || (split && !y_lo_dev))) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: no output");
    Conv16Args a;
    a.x = (const _Float16 *)x_dev; a.x_lo = (const _Float16 *)x_lo_dev; a.w = (const _Float16 *)w_dev; a.w_lo = (const _Float16 *)w_lo_dev;
    a.res = (const _Float16 *)res_dev; a.res_lo = (const _Float16 *)res_lo_dev; a.bias = bias_dev;
    a.y = (_Float16 *)y_dev; a.y_lo = (_Float16 *)y_lo_dev; a.y32 = y_f32_dev;
    a.M = (long long)n * ho * wo;
    a.H = h; a.W = w; a.Cin = cin; a.Ho = ho; a.Wo = wo; a.Cout = cout; a.KH = kh; a.KW = kw; a.stride = stride; a.pad = pad; a.K = kh * kw * cin;
    a.x_pix = x_pix_stride > 0 ? x_pix_stride : cin;
    a.y_pix = y_pix_stride > 0 ? y_pix_stride : cout;
    a.r_pix = res_pix_stride > 0 ? res_pix_stride : cout;
    a.res_post = res_post;
    a.n_dyn = conv_dynamic_batch();
    if ((in_scale_dev || res_scale_dev || out_state_dev) && !split) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16s: plane scales belong to the split mode");
    if (out_state_dev && out32) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16s: an fp32 output carries no scale (out_state must be NULL)");
    a.s_in = in_scale_dev; a.s_res = res_dev ? res_scale_dev : nullptr; a.s_out = out_state_dev;
    if (a.x_pix < cin || a.y_pix < cout || a.r_pix < cout || (a.x_pix | a.y_pix | a.r_pix) % 8 != 0)
        return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: pixel strides must cover the channels and be multiples of 8");
    if (((uintptr_t)x_dev | (uintptr_t)x_lo_dev | (uintptr_t)w_dev | (uintptr_t)w_lo_dev | (uintptr_t)res_dev | (uintptr_t)res_lo_dev | (uintptr_t)y_dev |
         (uintptr_t)y_lo_dev | (uintptr_t)y_f32_dev | (uintptr_t)bias_dev) & 15)
        return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: every pointer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)hip_stream;
    if (split) return out32 ? dispatch16<MODE_SPLIT, true>(a, act_kind, st) : dispatch16<MODE_SPLIT, false>(a, act_kind, st);
    return out32 ? dispatch16<MODE_F16, true>(a, act_kind, st) : dispatch16<MODE_F16, false>(a, act_kind, st);
}

extern "C" int tlk_conv2d_nhwc_16(const void *x_dev, const void *x_lo_dev, const void *w_dev, const void *w_lo_dev, const float *bias_dev,
                                  const void *res_dev, const void *res_lo_dev, void *y_dev, void *y_lo_dev, float *y_f32_dev,
                                  int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                                  int x_pix_stride, int y_pix_stride, int res_pix_stride, void *hip_stream)
{
    return conv16_entry(x_dev, x_lo_dev, w_dev, w_lo_dev, bias_dev, res_dev, res_lo_dev, y_dev, y_lo_dev, y_f32_dev, n, h, w, cin, cout, kh, kw, stride, pad, act_kind,
                        x_pix_stride, y_pix_stride, res_pix_stride, nullptr, nullptr, nullptr, hip_stream);
}

extern "C" int tlk_conv2d_nhwc_16s(const void *x_dev, const void *x_lo_dev, const void *w_dev, const void *w_lo_dev, const float *bias_dev,
                                   const void *res_dev, const void *res_lo_dev, void *y_dev, void *y_lo_dev, float *y_f32_dev,
                                   int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                                   int x_pix_stride, int y_pix_stride, int res_pix_stride, const float *in_scale_dev, const float *res_scale_dev,
                                   float *out_state_dev, void *hip_stream)
{
    return conv16_entry(x_dev, x_lo_dev, w_dev, w_lo_dev, bias_dev, res_dev, res_lo_dev, y_dev, y_lo_dev, y_f32_dev, n, h, w, cin, cout, kh, kw, stride, pad, act_kind,
                        x_pix_stride, y_pix_stride, res_pix_stride, in_scale_dev, res_scale_dev, out_state_dev, hip_stream);
}

extern "C" int tlk_conv16_set_glds(int on) { g_glds = on ? 1 : 0; return TLK_OK; }

extern "C" int tlk_conv16_set_config(int cfg)
{
    if (cfg < -1 || cfg > 26) return fail(TLK_EINVAL, "tlk_conv16_set_config: cfg is -1 (r04 kernels only), 0 (heuristic) or a tile configuration 1..26");
    g_cfg16x = cfg;
    return TLK_OK;
}

// the tile configuration the most recent tlk_conv2d_nhwc_16 / _16s call launched (1..26 f16, 1..12 split), -1 = the r04 kernels
extern "C" int tlk_conv16_last_config(void) { return c16::g_last_cfg16x; }

static int split_planes_entry(const float *x_dev, long long pixels, int c_in, int x_pix_stride, int c_out, void *hi_dev, void *lo_dev, float *state_dev,
                              long long pixels_per_image, void *hip_stream)
{
    if (pixels < 0 || c_in <= 0 || c_out < c_in) return fail(TLK_EINVAL, "tlk_split_f32_planes: bad shape");
    if (pixels == 0) return TLK_OK;
    if (!x_dev || !hi_dev || !lo_dev) return fail(TLK_EINVAL, "tlk_split_f32_planes: null pointer");
    const long long n = pixels * c_out;
    long long blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)hip_stream, x_dev, pixels, c_in,
                       x_pix_stride > 0 ? x_pix_stride : c_in, c_out, (_Float16 *)hi_dev, (_Float16 *)lo_dev, state_dev,
                       pixels_per_image > 0 ? conv_dynamic_batch() : nullptr, pixels_per_image);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_split_f32_planes(const float *x_dev, long long pixels, int c_in, int x_pix_stride, int c_out, void *hi_dev, void *lo_dev, void *hip_stream)
{
    return split_planes_entry(x_dev, pixels, c_in, x_pix_stride, c_out, hi_dev, lo_dev, nullptr, 0, hip_stream);
}

extern "C" int tlk_split_f32_planes_s(const float *x_dev, long long pixels, int c_in, int x_pix_stride, int c_out, void *hi_dev, void *lo_dev, float *state_dev,
                                      long long pixels_per_image, void *hip_stream)
{
    return split_planes_entry(x_dev, pixels, c_in, x_pix_stride, c_out, hi_dev, lo_dev, state_dev, pixels_per_image, hip_stream);
}

static int merge_planes_entry(const void *hi_dev, const void *lo_dev, long long n, float *y_dev, const float *scale_dev, void *hip_stream)
{
    if (n < 0) return fail(TLK_EINVAL, "tlk_merge_planes_f32: bad size");
    if (n == 0) return TLK_OK;
    if (!hi_dev || !lo_dev || !y_dev) return fail(TLK_EINVAL, "tlk_merge_planes_f32: null pointer");
    long long blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(merge_planes_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)hip_stream, (const _Float16 *)hi_dev, (const _Float16 *)lo_dev, n, y_dev,
                       scale_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_merge_planes_f32(const void *hi_dev, const void *lo_dev, long long n, float *y_dev, void *hip_stream)
{
    return merge_planes_entry(hi_dev, lo_dev, n, y_dev, nullptr, hip_stream);
}

extern "C" int tlk_merge_planes_f32_s(const void *hi_dev, const void *lo_dev, long long n, const float *scale_dev, float *y_dev, void *hip_stream)
{
    return merge_planes_entry(hi_dev, lo_dev, n, y_dev, scale_dev, hip_stream);
}

extern "C" int tlk_split_scale_update(float *states_dev, int n_states, int *changed_dev, void *hip_stream)
{
    if (n_states < 0) return fail(TLK_EINVAL, "tlk_split_scale_update: bad count");
    if (n_states == 0) return TLK_OK;
    if (!states_dev) return fail(TLK_EINVAL, "tlk_split_scale_update: null pointer");
    hipLaunchKernelGGL(split_scale_update_kernel, dim3((unsigned)((n_states + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream, states_dev, n_states, changed_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
