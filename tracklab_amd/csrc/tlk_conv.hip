// tlk_conv.hip -- fp32 convolution of the backbones at the REFERENCE's precision, hand-written for gfx950:
// implicit GEMM on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain, no reduced-precision
// path exists on this chip) with the convolution epilogue -- bias, residual add, ReLU / SiLU -- fused into the store.
//
// The reference runs its detector / ReID / pose networks in fp32 (tracklab/configs/modules/track/strong_sort.yaml:10 `fp16: false`;
// ONNXRuntime fp32 models behind wrappers/bbox_detector/rtmlib_api.py:21 and wrappers/pose_estimator/rtmlib_api.py:21; torchreid fp32
// behind wrappers/reid/kpreid_api.py:147-182).  In fp32 the library route of r01-r03 (MIOpen igemm / CK grouped conv + separate
// bias / activation / residual passes) ran the config-3 step at 375 ms; this kernel is what the fp32 step runs on from r04.
//
// GEMM view (channels-last activations, weights (Cout, KH, KW, Cin) = torch's channels_last weight):
//     Y[m][co] = act( sum_k A[m][k] * Wt[co][k] + bias[co] (+ R[m][co]) ),   m = (n, ho, wo),  k = (kh, kw, ci)
// A is gathered on the fly (zero outside the image / beyond K / beyond M), never materialised.
//   * workgroup tile BM x BN = (WGM*TM*32) x (WGN*TN*32), K step 32; 4 wavefronts; each wavefront owns TM x TN MFMA tiles of 32x32;
//   * global -> registers -> LDS, 16 B per lane both ways; the LDS rows are padded to 36 floats so that the ds_read_b128 fragment reads
//     (16-lane groups of the wide read) and the ds_write_b128 staging writes touch every bank once;
//   * LDS double-buffered: the global loads of step s+1 are in flight while step s runs its 16 * TM * TN MFMAs (64 cycles each), one
//     barrier per step;
//   * one ds_read_b128 feeds FOUR MFMAs: lane l holds k = 8j + 4*(l>>5) + r for r = 0..3, so MFMA r of group j multiplies the k pair
//     (8j + r, 8j + 4 + r) -- the summation order over k is therefore 0,4,1,5,2,6,3,7 within every group of 8, groups ascending.  That
//     order is part of the contract: oracle/src/conv.c walks the same chain with fmaf and the results are BIT-IDENTICAL for every tile
//     configuration (no split-K, one accumulator chain per output element);
//   * blockIdx -> tile mapping is XCD-aware: consecutive tiles (same rows of A, neighbouring column tiles of W) land on one XCD's L2.
#include "tlk_common.hpp"
#include "tlk_conv16.hpp"

using namespace tlk;

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
constexpr int BK = 32;            // K step (floats)
constexpr int LDK = BK + 4;       // padded LDS row: 36 floats = 144 B

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *x, *w, *bias, *res;
    float *y;
    long long M;                  // N * Ho * Wo
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, K;
    int x_pix, y_pix, r_pix;      // floats between two pixels of x / y / residual (>= channel count: lets a call read or write a channel
                                  // slice of a wider tensor, e.g. write straight into its part of a concatenation)
    int tiles_n;
    long long tiles;
    int res_post;                 // 1: the residual is added AFTER the activation (y = act(conv + bias) + r: CSPNeXt's identity add)
    const int *n_dyn;             // not NULL: the number of images is read from device memory (<= the n of the call): tlk_conv_set_dynamic_batch
};

template <int ACT> __device__ __forceinline__ float act_f32(float v)
{
    if (ACT == ACT_RELU) return v < 0.f ? 0.f : v;          // (this form lets NaN through, like torch.relu: an overflow upstream must stay visible, r05)
    if (ACT == ACT_SILU) return v / (1.f + __expf(-v));
    return v;
}

constexpr int epi_rows(int tm, int wgm, int rows_max)        // MFMA tile rows of every wavefront per epilogue pass: the most that fit the LDS at hand
{
    int e = tm;
    while (e > 1 && (tm % e != 0 || wgm * 32 * e > rows_max)) --e;
    return e;
}

// NST = LDS stages: 2 = double-buffered K loop (the compute-bound layers), 1 = ONE stage and an epilogue in passes -- half the LDS, so a CU
// holds twice the workgroups: what the memory-bound layers want (r05; the 16-bit kernels' probe showed residency beating an in-workgroup pipeline)
template <int TM, int TN, int WGM, int WGN, int ACT, bool RES, int NST = 2>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_f32_mfma_kernel(const ConvArgs p)
{
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int ROWS_PER_PASS = NT / 8;                 // 8 lanes x 16 B cover one 32-float row slice
    constexpr int PA = BM / ROWS_PER_PASS, PB = BN / ROWS_PER_PASS;
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile rows must be a multiple of the loader pass");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                                       // [NST][BM][LDK]
    float *Bs = lds + NST * BM * LDK;                      // [NST][BN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // dynamic batch (r05): the launch is sized for the n of the call, the rows that exist are n_dyn[0] images' worth; workgroups beyond the live
    // tiles leave at once (before any barrier), a tile across the edge masks its rows exactly like the last tile of a static launch
    long long M = p.M;
    if (p.n_dyn) { const long long md = (long long)p.n_dyn[0] * p.Ho * p.Wo; M = md < M ? (md < 0 ? 0 : md) : M; }
    const long long live_tiles = ((M + BM - 1) / BM) * p.tiles_n;
    if ((long long)blockIdx.x >= live_tiles) return;
    // XCD-aware tile order: hardware deals consecutive workgroups round-robin to the 8 XCDs; give each XCD a contiguous run of the LIVE tiles
    long long tile;
    {
        const long long b = blockIdx.x, q = live_tiles >> 3;
        const int r = (int)(live_tiles & 7), xcd = (int)(b & 7);
        tile = (long long)xcd * q + (xcd < r ? xcd : r) + (b >> 3);
    }
    const long long m0 = (tile / p.tiles_n) * BM;
    const int n0 = (int)(tile % p.tiles_n) * BN;

    // ---- loader geometry: this lane moves chunk `lc` (4 floats of k) of row `lr + pass * ROWS_PER_PASS`.
    // Loads are BUFFER loads (raw buffer descriptor, 32-bit byte offsets): a lane whose chunk is outside the image / beyond K / beyond M
    // uses an offset past num_records and the hardware returns zeros -- no branch, no select, so the whole K step is one basic block the
    // scheduler barriers below can shape.  The A descriptor is re-based per workgroup (a tensor may exceed the 4 GB a 32-bit offset spans;
    // the rows of one tile never do): base = first input row any tap of the tile's first output row can touch.
    const int lr = tid >> 3, lc = tid & 7;
    constexpr int OOB = (int)0x80000000;                   // beyond every num_records below (< 2^31)
    long long base_pix;
    {
        const unsigned hw = (unsigned)(p.Ho * p.Wo), n = (unsigned)m0 / hw;         // M < 2^31 (checked by the host side): 32-bit divisions
        const int rem = (int)((unsigned)m0 - n * hw), ho = rem / p.Wo;
        const int hi = ho * p.stride - p.pad;
        base_pix = n * (long long)p.H * p.W + (long long)(hi > 0 ? hi : 0) * p.W;
    }
    const long long total_pix = (long long)((unsigned)p.M / (unsigned)(p.Ho * p.Wo)) * p.H * p.W;
    long long a_bytes = ((total_pix - base_pix - 1) * p.x_pix + p.Cin) * 4;
    if (a_bytes > 0x7ffffff0LL) a_bytes = 0x7ffffff0LL;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + base_pix * p.x_pix), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)((long long)p.Cout * p.K * 4), 0x00020000);
    int a_hi0[PA], a_wi0[PA], a_rel[PA];                   // a_rel: pixel index of tap (0,0) relative to base_pix (may be negative: masked then)
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
        a_rel[ps] = (int)(n * (long long)p.H * p.W - base_pix) + a_hi0[ps] * p.W + a_wi0[ps];
    }
    int b_off[PB];                                         // byte offset of (co, k = 0), OOB for rows beyond Cout
#pragma unroll
    for (int ps = 0; ps < PB; ++ps) {
        const int co = n0 + lr + ps * ROWS_PER_PASS;
        b_off[ps] = co < p.Cout ? co * p.K * 4 : OOB;
    }

    constexpr int NL = PA + PB;                            // 16-byte loads per lane and K step
    i32x4 rg[NL];                                          // staging registers: global -> LDS
    int t_kh = 0, t_kw = 0, t_ci = 0, t_k = 0;             // tap of this lane's chunk in the step being loaded
    bool t_in = false;
    auto set_tap = [&](int k0) {
        t_k = k0 + lc * 4;
        t_in = t_k < p.K;
        t_kh = 0; t_kw = 0; t_ci = t_k;
        if (p.KH * p.KW != 1) { const int tap = t_k / p.Cin; t_ci = t_k - tap * p.Cin; t_kh = tap / p.KW; t_kw = tap - t_kh * p.KW; }
    };
    auto issue_load = [&](int i) {
        if (i < PA) {
            const int hi = a_hi0[i] + t_kh, wi = a_wi0[i] + t_kw;
            const bool ok = t_in && a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int off = ((a_rel[i] + t_kh * p.W + t_kw) * p.x_pix + t_ci) * 4;
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? off : OOB, 0, 0);
        } else {
            const int bo = b_off[i - PA];
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (t_in && bo != OOB) ? bo + t_k * 4 : OOB, 0, 0);
        }
    };
    auto issue_store = [&](int i, int buf) {
        float *dst = i < PA ? As + buf * BM * LDK + (lr + i * ROWS_PER_PASS) * LDK + lc * 4
                            : Bs + buf * BN * LDK + (lr + (i - PA) * ROWS_PER_PASS) * LDK + lc * 4;
        *reinterpret_cast<i32x4 *>(dst) = rg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The residual of this lane's output vectors is fetched NOW and waits in registers: the 1x1 expansions that carry it are short in K
    // (2-16 steps), their epilogue used to sit on HBM latency with four loads in flight; here all of them ride under the main loop.
    // epilogue geometry: the tile leaves through LDS EPI tile rows of every wavefront at a time (all of it in one pass when the LDS of the
    // stages holds it: the two-stage configurations)
    constexpr int LDC = BN + 4;
    constexpr int STAGE_FLOATS = (BM + BN) * LDK;
    constexpr int LDS_AVAIL = (NST * STAGE_FLOATS > WGM * 32 * LDC) ? NST * STAGE_FLOATS : WGM * 32 * LDC;      // floats
    constexpr int EPI = epi_rows(TM, WGM, LDS_AVAIL / LDC);
    static_assert(WGM * 32 * EPI * LDC <= LDS_AVAIL, "epilogue pass does not fit the LDS");
    constexpr int PROWS = WGM * 32 * EPI, NPASS = TM / EPI;
    constexpr int EV_PER_ROW = BN / 4, ENVEC = PROWS * EV_PER_ROW, EITS = (ENVEC + NT - 1) / NT;
    const bool evec = ((p.Cout | p.y_pix | p.r_pix) & 3) == 0 && (((uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.bias) & 15) == 0;
    auto pass_row = [&](int ps, int prow) {        // row of the workgroup tile that row `prow` of pass `ps` holds
        const int pw = prow / (32 * EPI), within = prow - pw * (32 * EPI);
        return pw * (TM * 32) + ps * (EPI * 32) + within;
    };
    float4 rres[RES ? NPASS : 1][RES ? EITS : 1];
    if (RES && evec) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int it = 0; it < EITS; ++it) {
                const int idx = it * NT + tid;
                const int prow = idx / EV_PER_ROW, ec = (idx - prow * EV_PER_ROW) * 4;
                const long long m = m0 + pass_row(ps, prow);
                const int co = n0 + ec;
                const bool ok = !(ENVEC % NT != 0 && idx >= ENVEC) && m < M && co < p.Cout;
                rres[ps][it] = ok ? *reinterpret_cast<const float4 *>(p.res + m * p.r_pix + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    const int steps = (p.K + BK - 1) / BK;
    const int frag_off = (lane & 31) * LDK + (lane >> 5) * 4;
    const float *a_frag = As + (wm * TM * 32) * LDK + frag_off, *b_frag = Bs + (wn * TN * 32) * LDK + frag_off;
    float4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int buf, int j, int set) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const float4 *>(a_frag + buf * BM * LDK + i * 32 * LDK + j * 8);
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[set][i] = *reinterpret_cast<const float4 *>(b_frag + buf * BN * LDK + i * 32 * LDK + j * 8);
    };
    if (NST == 1) {
        // ONE stage: load, store, multiply.  Same fmaf chain (groups of 8 ascending, 0,4,1,5,2,6,3,7 inside), so the result is the same bits.
        for (int s = 0; s < steps; ++s) {
            set_tap(s * BK);
#pragma unroll
            for (int i = 0; i < NL; ++i) issue_load(i);
#pragma unroll
            for (int i = 0; i < NL; ++i) issue_store(i, 0);
            __syncthreads();
            read_frags(0, 0, 0);
#pragma unroll
            for (int j = 0; j < BK / 8; ++j) {
                const int set = j & 1;
                if (j + 1 < BK / 8) read_frags(0, j + 1, set ^ 1);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj) {
                            const float av = r == 0 ? fa[set][i].x : r == 1 ? fa[set][i].y : r == 2 ? fa[set][i].z : fa[set][i].w;
                            const float bv = r == 0 ? fb[set][jj].x : r == 1 ? fb[set][jj].y : r == 2 ? fb[set][jj].z : fb[set][jj].w;
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][jj], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    } else {
    set_tap(0);
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_load(i);
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_store(i, 0);
    __syncthreads();
    read_frags(0, 0, 0);
    // One K step = 16 chunks of TM*TN MFMAs (64 cycles each).  The step is software-pipelined INSIDE the wavefront, chunk by chunk, with
    // scheduling barriers between the chunks so the order below is the order issued: the loads of the next step (address arithmetic + buffer
    // load) ride under chunks 0-7, their LDS stores under chunks 8-15 (>= 2048 cycles after the load), the fragment reads of group j+1 under
    // the first chunk of group j.  A wavefront keeps the MFMA pipe busy on its own, whatever its neighbour on the SIMD does.
    constexpr int NCH = 4 * (BK / 8);
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        set_tap((s + 1) * BK);                             // beyond K on the last step: every load returns zeros, the stores hit a dead buffer
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = c >> 2, r = c & 3, set = j & 1;
            if (r == 0 && j + 1 < BK / 8) read_frags(cur, j + 1, set ^ 1);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                if (c < NCH / 2 && (i * (NCH / 2)) / NL == c) issue_load(i);
                if (c >= NCH / 2 && (i * (NCH / 2)) / NL == c - NCH / 2) issue_store(i, cur ^ 1);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    const float av = r == 0 ? fa[set][i].x : r == 1 ? fa[set][i].y : r == 2 ? fa[set][i].z : fa[set][i].w;
                    const float bv = r == 0 ? fb[set][jj].x : r == 1 ? fb[set][jj].y : r == 2 ? fb[set][jj].z : fb[set][jj].w;
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][jj], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        read_frags(cur ^ 1, 0, 0);                         // (after the last step: reads the dead buffer, unused)
    }
    __syncthreads();                                       // nobody still reads fragments when the tile is staged below
    }

    // ---- epilogue.  C/D map of the 32x32 tile: column (= cout) = lane & 31, row (= pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // The tile goes through LDS (free after the last barrier) so that every lane then moves 16 contiguous bytes: rows of BN floats leave as
    // 512-byte runs, the residual arrives the same way, instead of 64 scalar stores per lane.
    float *Cs = lds;                                       // [PROWS][LDC]
    const float *__restrict__ resp = p.res;
    float *__restrict__ yp = p.y;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int ii = 0; ii < EPI; ++ii)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                float *c = Cs + ((wm * EPI + ii) * 32 + 4 * (lane >> 5)) * LDC + (wn * TN + jj) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2)) * LDC] = acc[ps * EPI + ii][jj][r];
            }
        __syncthreads();
        if (evec) {
#pragma unroll
            for (int it = 0; it < EITS; ++it) {
                const int idx = it * NT + tid;
                const int prow = idx / EV_PER_ROW, ec = (idx - prow * EV_PER_ROW) * 4;
                const long long m = m0 + pass_row(ps, prow);
                const int co = n0 + ec;
                if ((ENVEC % NT != 0 && idx >= ENVEC) || m >= M || co >= p.Cout) continue;
                float4 v = *reinterpret_cast<const float4 *>(Cs + prow * LDC + ec);
                if (p.bias) { const float4 bv = *reinterpret_cast<const float4 *>(p.bias + co); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                else { v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f; }
                if (RES && !p.res_post) { const float4 rv = rres[RES ? ps : 0][RES ? it : 0]; v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                v.x = act_f32<ACT>(v.x); v.y = act_f32<ACT>(v.y); v.z = act_f32<ACT>(v.z); v.w = act_f32<ACT>(v.w);
                if (RES && p.res_post) { const float4 rv = rres[RES ? ps : 0][RES ? it : 0]; v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                *reinterpret_cast<float4 *>(yp + m * p.y_pix + co) = v;
            }
        } else {
            for (int it = 0; it < EITS; ++it) {
                const int idx = it * NT + tid;
                const int prow = idx / EV_PER_ROW, ec = (idx - prow * EV_PER_ROW) * 4;
                const long long m = m0 + pass_row(ps, prow);
                const int co = n0 + ec;
                if ((ENVEC % NT != 0 && idx >= ENVEC) || m >= M) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (co + e >= p.Cout) break;
                    float v = Cs[prow * LDC + ec + e] + (p.bias ? p.bias[co + e] : 0.f);
                    if (RES && !p.res_post) v += resp[m * p.r_pix + co + e];
                    v = act_f32<ACT>(v);
                    if (RES && p.res_post) v += resp[m * p.r_pix + co + e];
                    yp[m * p.y_pix + co + e] = v;
                }
            }
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
}

template <int TM, int TN, int WGM, int WGN, int NST = 2> int launch_cfg(ConvArgs &a, int act, hipStream_t st)
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, NT = 64 * WGM * WGN;
    constexpr size_t LDS_STAGE = (size_t)NST * (BM + BN) * LDK * sizeof(float), LDS_C = (size_t)WGM * 32 * (BN + 4) * sizeof(float);
    constexpr size_t LDS_BYTES = LDS_STAGE > LDS_C ? LDS_STAGE : LDS_C;      // (the kernel sizes its epilogue passes to this)
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
    if (a.tiles > 0x7fffffffLL || a.M > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: more than 2^31 - 1 output pixels in one launch");
    const bool res = a.res != nullptr;
#define TLK_CONV_LAUNCH(A, R)                                                                                                              \
    do {                                                                                                                                   \
        auto kern = conv_f32_mfma_kernel<TM, TN, WGM, WGN, A, R, NST>;                                                                       \
        static bool attr_set = false;                                                                                                      \
        if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles), dim3(NT), LDS_BYTES, st, a);                                                     \
    } while (0)
    if (res) { if (act == 0) TLK_CONV_LAUNCH(ACT_NONE, true); else if (act == 1) TLK_CONV_LAUNCH(ACT_RELU, true); else TLK_CONV_LAUNCH(ACT_SILU, true); }
    else { if (act == 0) TLK_CONV_LAUNCH(ACT_NONE, false); else if (act == 1) TLK_CONV_LAUNCH(ACT_RELU, false); else TLK_CONV_LAUNCH(ACT_SILU, false); }
#undef TLK_CONV_LAUNCH
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// which layers the direct-to-LDS kernels take by default (0 = none): filled in from the probe's table (profiles/r05_conv_f32_shapes.txt)
int pick_x32(const ConvArgs &a, int cout, bool has_res)
{
    // measured (tools/micro/conv32_probe: 2400 crops of ResNet-50, 2211 of HRNet-W32; profiles/r05_conv_f32_shapes.txt)
    const bool big = a.M >= 256 * 1024;
    //  * 3 x 3 / stride 1 on 32 channels (HRNet's high-resolution branch): the PATCH kernel -- the tile's input rows land in LDS once instead of
    //    once per tap: 1.07 vs 1.82 ms (117 vs 69 TFLOP/s), 1.13 vs 2.02 ms with a residual
    if (a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 32 && a.H == a.Ho && a.W == a.Wo && a.Wo >= 8 && a.Wo <= 64 && 256 % a.Wo == 0 &&
        ((long long)a.Ho * a.Wo) % 256 == 0 && a.M >= 64 * 1024)
        return 11;
    //  * r06: the same on 64 channels (two K steps per pixel: the patch is two regions) -- HRNet's 64-channel branch 1.01 vs 1.11 ms (124 vs 113 TFLOP/s),
    //    1.04 vs 1.18 with a residual; ResNet's layer 1 4.04 vs 4.28 ms (profiles/r06_conv_f32_patch64.txt)
    if (a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.H == a.Ho && a.W == a.Wo && a.Wo >= 8 && a.Wo <= 64 && 128 % a.Wo == 0 &&
        ((long long)a.Ho * a.Wo) % 128 == 0 && a.M >= 64 * 1024)
        return 14;
    // r06: the rules below were re-derived from a sweep of EVERY distinct convolution of the six networks under every configuration
    // (tools/sweep_conv_f32.py; profiles/r06_conv_f32_sweep.txt holds the tables before and after)
    const long long tiles128 = ((a.M + 127) / 128) * ((cout + 127) / 128);
    //  * 32 wide: the 256 x 32 tile, one LDS stage on a short K loop (1 x 1 64 > 32: 0.14 vs 0.21 ms), two stages from K = 256 (HRNet's fuse
    //    layers 256 > 32: 3 x 3 9.31 vs 9.97 ms, 1 x 1 0.028 vs 0.031)
    if (cout == 32 && a.M >= 64 * 1024) return a.K >= 256 ? 8 : 7;
    //  * 96 wide = three of those tiles exactly -- YOLOX-m's / CSPNeXt-m's 96-channel stage on the 256 x 96 register-staged tile 0.065 / 0.315 /
    //    0.219 ms, here 0.047 / 0.257 / 0.175 (1 x 1, 3 x 3 + residual, 1 x 1 at 160 x 160)
    if (cout == 96 && a.M >= 64 * 1024) return 7;
    //  * 64 wide: the one-stage 256 x 64 tile on the 1 x 1 layers of the largest maps (ResNet's layer 1, 6.8 M pixels: 0.92 vs 1.06 ms; at 0.4-0.6 M
    //    pixels the register-staged 128 x 64 tile is 5-9 % ahead) and the 3 x 3 ones (1.08 vs 1.11 ms; stride 2: 0.63 vs 0.68)
    if (cout == 64 && ((a.KH == 1 && !has_res && a.K <= 256 && a.M >= 1024 * 1024) || (a.KH == 3 && a.K <= 576 && big))) return 3;
    //  * the short-K 1 x 1 expansions (K <= 128): the one-stage 64 x 128 tile with the residual prefetched (4.05 vs 4.5 ms, 2.87 vs 3.13 ms)
    if (a.KH == 1 && a.KW == 1 && a.stride == 1 && has_res && a.K <= 128 && cout % 128 == 0 && big) return 5;
    //  * 3 x 3 on 128-multiple widths, K >= 1152 (with or without residual, any stride), enough tiles to fill the chip four times: the
    //    two-stage 128 x 128 direct-to-LDS tile -- ResNet's layer-3 / layer-4 3 x 3 3.65 vs 3.77 ms and 14.35 vs 14.90 (139.7 TFLOP/s = 0.89 of
    //    the peak), HRNet's 256-channel blocks 0.99 vs 1.03, CSPNeXt's 4.19 vs 4.35; basic blocks with residual as in r05 (0.95 vs 0.98 ms)
    if (a.KH == 3 && cout % 128 == 0 && a.K >= 1152 && tiles128 >= 1024) return 4;
    //  * the same layers on FEW tiles (the detectors' deep stages, the ReID net on one frame's crops): widths of 3 x 128 / 6 x 128 stay on that
    //    tile (0.437 vs 0.488 ms, 0.427 vs 0.481); powers of two are 20-60 % faster on 64 x 128 tiles while the launch is small (256 > 256 on 19 K
    //    pixels 0.235 vs 0.284, 128 > 128 on 13 K 0.051 vs 0.085)
    if (a.KH == 3 && cout % 128 == 0 && a.K >= 1152 && a.M >= 8 * 1024) {
        if ((cout / 128) % 3 == 0) return 4;
        if (a.M <= 24 * 1024) return 5;
    }
    // every other layer is as fast or faster on conv_f32_mfma_kernel
    return 0;
}

int g_force_cfg = -1;     // tlk_conv2d_set_config (probes / tests): -1 = heuristic
thread_local const int *g_dyn_batch = nullptr;      // tlk_conv_set_dynamic_batch: per HOST THREAD (a process-global pointer silently truncated the batches
                                                    // of any other thread's / pipeline's launches while it was set)
int g_last_cfg = -1;      // configuration of the most recent launch (tlk_conv2d_last_config: bench.py groups its event timings by kernel instantiation)

}  // namespace

extern "C" int tlk_conv2d_set_config(int cfg)
{
    if (cfg < -1 || (cfg > 9 && (cfg < 21 || cfg > 39))) return fail(TLK_EINVAL, "tlk_conv2d_set_config: cfg must be -1 (heuristic), 0..9, or 21..39 (the direct-to-LDS kernels of tlk_conv16x.hip on fp32 tensors)");
    g_force_cfg = cfg;
    return TLK_OK;
}

extern "C" int tlk_conv2d_last_config(void) { return g_last_cfg; }

const int *tlk::conv_dynamic_batch() { return g_dyn_batch; }

// Dynamic batch for every convolution launched from now on (tlk_conv2d_nhwc_f32, tlk_conv2d_nhwc_16) until it is cleared with NULL:
// the kernels take the number of images from n_images_dev[0] (device memory, read when the kernel RUNS -- so a captured hipGraph follows it)
// and treat the `n` of the call as the capacity.  Rows beyond are neither read nor written.
extern "C" int tlk_conv_set_dynamic_batch(const int32_t *n_images_dev)
{
    g_dyn_batch = n_images_dev;
    return TLK_OK;
}

extern "C" int tlk_conv2d_nhwc_f32(const float *x_dev, const float *w_dev, const float *bias_dev, const float *residual_dev, float *y_dev,
                                   int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int act_kind,
                                   int x_pix_stride, int y_pix_stride, int res_pix_stride, void *hip_stream)
{
    if (n < 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: bad shape");
    const int res_post = (act_kind & TLK_ACT_RES_AFTER) ? 1 : 0;
    act_kind &= ~TLK_ACT_RES_AFTER;
    if (cin == 3 && !residual_dev && act_kind >= 0 && act_kind <= 2 && n > 0 && x_dev && w_dev && y_dev) {
        // RGB stems: the direct kernel of tlk_conv_stem.hip (7 x 7 / 3 x 3, stride 2, Cout <= 64) reads the 3-channel image as it is
        const int ho_ = (h + 2 * pad - kh) / stride + 1, wo_ = (w + 2 * pad - kw) / stride + 1;
        const int xp = x_pix_stride > 0 ? x_pix_stride : 3, yp = y_pix_stride > 0 ? y_pix_stride : cout;
        if (ho_ > 0 && wo_ > 0 && xp >= 3 && yp >= cout) {
            const int r = conv_stem3_f32(x_dev, w_dev, bias_dev, y_dev, n, h, w, cout, kh, kw, stride, pad, act_kind, xp, yp, (hipStream_t)hip_stream);
            if (r != 1) { if (r == TLK_OK) g_last_cfg = 15; return r; }
        }
    }
    if (cin % 4 != 0) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: Cin must be a multiple of 4 (pad the input channels with zeros), or 3 for a 7x7 / 3x3 stride-2 stem with Cout <= 64");
    if (act_kind < 0 || act_kind > 2) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: act_kind is 0 (none), 1 (ReLU) or 2 (SiLU), optionally | TLK_ACT_RES_AFTER");
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: empty output");
    if (n == 0) return TLK_OK;
    if (!x_dev || !w_dev || !y_dev) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: null pointer");
    ConvArgs a;
    a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.res = residual_dev; a.y = y_dev;
    a.M = (long long)n * ho * wo;
    a.H = h; a.W = w; a.Cin = cin; a.Ho = ho; a.Wo = wo; a.Cout = cout; a.KH = kh; a.KW = kw; a.stride = stride; a.pad = pad;
    a.K = kh * kw * cin;
    a.x_pix = x_pix_stride > 0 ? x_pix_stride : cin;
    a.y_pix = y_pix_stride > 0 ? y_pix_stride : cout;
    a.r_pix = res_pix_stride > 0 ? res_pix_stride : cout;
    a.res_post = res_post;
    a.n_dyn = g_dyn_batch;
    if (a.x_pix < cin || a.y_pix < cout || a.r_pix < cout || a.x_pix % 4 != 0)
        return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: pixel strides must cover the channels (x stride a multiple of 4)");
    if (((uintptr_t)x_dev | (uintptr_t)w_dev) & 15) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: x and w must be 16-byte aligned");
    hipStream_t st = (hipStream_t)hip_stream;
    // tile configuration (measured on MI355X, profiles/r04_conv_f32_shapes.md): 128 x 128 wherever Cout fills it (130-135 TFLOP/s on the
    // K >= 512 layers); 128 x 64 for Cout <= 64 and the other widths that are not multiples of 128 (two workgroups per CU fit, which the
    // 256 x 64 tile's 92 KB of LDS do not); 256 x 96 for odd multiples of 96 (YOLOX-m widths); 64 x 128 when 128-row tiles would leave CUs idle
    int cfg = g_force_cfg;
    // the direct-to-LDS kernels of tlk_conv16x.hip on fp32 tensors (MODE_F32: same fmaf chain, bit-identical results): the memory-bound layers
    auto x32 = [&](int xcfg) {
        c16::Conv16Args b;
        b.x = (const _Float16 *)x_dev; b.x_lo = nullptr; b.w = (const _Float16 *)w_dev; b.w_lo = nullptr; b.res = (const _Float16 *)residual_dev; b.res_lo = nullptr;
        b.bias = bias_dev; b.y = nullptr; b.y_lo = nullptr; b.y32 = y_dev;
        b.M = a.M; b.H = h; b.W = w; b.Cin = cin; b.Ho = ho; b.Wo = wo; b.Cout = cout; b.KH = kh; b.KW = kw; b.stride = stride; b.pad = pad; b.K = a.K;
        b.x_pix = a.x_pix; b.y_pix = a.y_pix; b.r_pix = a.r_pix; b.res_post = res_post; b.n_dyn = g_dyn_batch;
        return c16::launch32x(b, act_kind, xcfg, st);
    };
    const bool x32_ok = cin % 32 == 0 && cout % 4 == 0 && ((a.y_pix | a.r_pix) & 3) == 0 && (((uintptr_t)y_dev | (uintptr_t)residual_dev | (uintptr_t)bias_dev) & 15) == 0;
    if (cfg >= 21) {
        if (!x32_ok) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: this shape cannot take the direct-to-LDS kernels (Cin % 32, Cout % 4, 16-byte aligned rows)");
        g_last_cfg = cfg;
        return x32(cfg - 20);
    }
    if (cfg < 0 && x32_ok) {
        const int pick = pick_x32(a, cout, residual_dev != nullptr);
        if (pick > 0) { const int r = x32(pick); if (r != 1) { if (r == TLK_OK) g_last_cfg = 20 + pick; return r; } }
    }
    if (cfg < 0) {
        const int c = cout;
        if (c <= 32) cfg = 4;
        else if (c % 128 == 0) cfg = 0;
        else if (c % 96 == 0 && (c / 96) % 2 == 1 && a.M >= 64 * 1024) cfg = 3;
        else cfg = 2;
        if (cfg == 0 && ((a.M + 127) / 128) * (cout / 128) < 1024) cfg = 5;
        // r06 (same sweep): the 128 x 64 tile with ONE LDS stage (28 KB: three workgroups per CU) wherever the K loop is short or the launch is
        // large -- YOLOX-m's 48- / 192-wide layers 0.093 > 0.074 ms (1 x 1, K 48), 0.546 > 0.435 (the Focus stem), 0.136 > 0.121, 0.924 > 0.886
        if (cfg == 2 && ((a.K <= 192 && a.M >= 16 * 1024) || (a.M >= 128 * 1024 && a.K <= 1728))) cfg = 8;
        // r06: a 1 x 1 layer with K <= 64 that widens to a multiple of 128 (ResNet's / HRNet's 64 > 256 projection): 128 x 128 with ONE stage,
        // 2.79 vs 3.13 ms
        if (cfg == 0 && kh == 1 && kw == 1 && a.K <= 64 && residual_dev == nullptr && a.M >= 256 * 1024) cfg = 7;
    }
    g_last_cfg = cfg;
    switch (cfg) {
    case 0: return launch_cfg<2, 2, 2, 2>(a, act_kind, st);    // 128 x 128
    case 1: return launch_cfg<2, 2, 4, 1>(a, act_kind, st);    // 256 x 64
    case 2: return launch_cfg<2, 1, 2, 2>(a, act_kind, st);    // 128 x 64
    case 3: return launch_cfg<2, 3, 4, 1>(a, act_kind, st);    // 256 x 96
    case 4: return launch_cfg<2, 1, 4, 1>(a, act_kind, st);    // 256 x 32
    case 6: return launch_cfg<1, 2, 4, 1>(a, act_kind, st);    // 128 x 64, wavefronts stacked in M
    case 7: return launch_cfg<2, 2, 2, 2, 1>(a, act_kind, st); // 128 x 128, ONE LDS stage (37 KB: the memory-bound layers)
    case 8: return launch_cfg<2, 1, 2, 2, 1>(a, act_kind, st); // 128 x 64, ONE stage (28 KB)
    case 9: return launch_cfg<1, 2, 2, 2, 1>(a, act_kind, st); // 64 x 128, ONE stage
    default: return launch_cfg<1, 2, 2, 2>(a, act_kind, st);   // 64 x 128 (small M)
    }
}
