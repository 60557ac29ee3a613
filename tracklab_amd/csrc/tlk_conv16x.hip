// tlk_conv16x.hip -- the LARGE-TILE 16-bit MFMA convolution kernels (r05): the compute-bound layers of the backbones (K >= 256, Cout a
// multiple of 64) in f16 mode and in split-fp32 mode (see tlk_conv16.hip for the two modes' arithmetic, which is unchanged here).
//
// Why another kernel.  r04's conv16 kernels compute a 128 x 128 tile with four wavefronts of 64 x 64: every ds_read_b128 feeds two
// v_mfma_f32_32x32x16_f16 on average, the LDS spends as many cycles delivering fragments (plus absorbing the direct-to-LDS stream) as the
// matrix pipes spend multiplying, and the kernel sits at 0.18 (ReID forward) / 0.34 (largest layers) of the 2.5 PFLOP/s peak
// (profiles/r04_conv_kernels_pmc.md: 5.75 cycles of LDS wait per MFMA).  The fix is operand reuse per LDS byte:
//
//   * f16: 256 x 256 tile, EIGHT wavefronts (2 in M x 4 in N), each 128 x 64 = 4 x 2 MFMA tiles: 6 fragment reads feed 8 MFMAs per 16-wide
//     k slice (0.75 reads / MFMA against 1.0), the tile's operand bytes per flop halve against 128 x 128; 128 accumulator registers per lane,
//     two wavefronts per SIMD.  256 x 128 and 512 x 64 tiles of the same construction cover Cout = 128 and 64.
//   * split: every product is three MFMAs on two accumulator sets, so a wavefront takes 64 x 64 (2 x 2 tiles x 2 sets = 128 registers): 8
//     reads feed 12 MFMAs per slice (r04: 6 reads / 6 MFMAs on a 32 x 64 wavefront tile).  The hi and lo planes are SEPARATE LDS regions with
//     64-byte rows and their own XOR swizzle -- r04's shared 128-byte rows (32 hi | 32 lo) read with 2.69 bank conflicts per MFMA.
//   * K step = 128 bytes per row (64 f16 of a plane or 32 + 32), two LDS stages: stage s + 1 streams in (direct-to-LDS, 16 B per lane, no
//     staging registers, no ds_write pass) while stage s is multiplied; one barrier per step of 32 (f16) / 24 (split) MFMAs per wavefront.
//   * LDS image: unpadded rows, 16-byte chunk q of row i stored at position q ^ ((i >> 1) & 7) (128-byte rows) or q ^ ((i >> 2) & 3)
//     (64-byte rows): the 16-lane groups of ds_read_b128 {0-3, 12-15, 20-27} ... then touch 16 distinct 16-byte slots of the 256-byte bank
//     row -- conflict-free; a direct-to-LDS load lands lane-linear, so the swizzle is applied to the SOURCE address each lane fetches.
//   * addressing: raw buffer loads with 32-bit byte offsets off a descriptor re-based per workgroup; a tap outside the image / a row beyond
//     M / a column beyond Cout uses an out-of-range offset and the hardware writes zeros (USE_BUF), or -- the r04 form, kept selectable
//     until the A/B on the GPU is in (tlk_conv16_set_config) -- a 64-bit pointer that falls back to a zero page.
//   * epilogue: the fp32 tile leaves through the (now idle) LDS stages one MFMA tile row at a time, 8 channels (16 bytes of f16) per lane.
// Activation kind, residual and output format are run-time (wave-uniform) switches here, not template arguments: the epilogue is a few
// percent of a compute-bound launch and the instantiation count stays small.
//
// Later in r05 the same kernel grew three more uses (template arguments NST, RESPF, MODE_F32, PATCH):
//   * NST = 1: ONE LDS stage (33 KB for 128 x 128), 3-4 workgroups per CU -- residency hides the latencies better than a pipeline inside the
//     workgroup on every memory-bound layer and most large ones; NST = 3 / 4: a ring with counted waits for the small launches of the
//     online step (64 x 64 tiles); RESPF: the residual of the lane's output vectors prefetched into registers before the main loop;
//   * MODE_F32: fp32 tensors through the same loader (128-byte rows = 32 floats) on v_mfma_f32_32x32x2_f32, the k order of tlk_conv.hip's
//     contract (bit-identical results): the memory-bound and narrow layers of the fp32 networks (tlk_conv2d_nhwc_f32 configurations 21-29);
//   * PATCH: 3 x 3 / stride 1 layers whose Cin is exactly one K step (32 floats / 64 halfs) and whose tiles are whole image rows -- the
//     tile's input rows + halo land in LDS ONCE and the nine taps are nine shifted fragment reads, instead of nine passes over the input
//     through an L2 that no longer holds it (fp32 configurations 30-33: HRNet's 32-channel branch, 117 vs 69 TFLOP/s; f16 17 / 18:
//     ResNet's layer 1, 716 vs 625).
#include "tlk_conv16.hpp"

using namespace tlk;
using namespace tlk::c16;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000;                      // beyond every num_records (< 2^31): the hardware returns zeros

// scheduling recipe for one chunk of the K loop: NPAIR x (one instruction of class MASK, one MFMA), then REST MFMAs
// (LLVM SchedGroupMask: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read)
template <int NPAIR, int MASK, int REST> __device__ __forceinline__ void interleave()
{
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
        __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    if (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
}

constexpr int pick_epi(int tm, int wgm, int rows_max)
{
    int e = tm;
    while (e > 1 && (tm % e != 0 || wgm * 32 * e > rows_max)) --e;
    return e;
}

// ROWB (r06): bytes of one K step per tile row in LDS -- 128 everywhere but the f16 HALF-STEP tiles (64: K step = 32 halfs, the geometry of one plane of
// the split mode), for layers whose Cin is a multiple of 32 but not of 64 (YOLOX-m's 96-channel layers: they sat on the r04 register-staged kernel,
// whose loader divides per load when a K step straddles taps: 7 % of the one-frame f16 step)
// CPP (r06, PATCH only): K steps per input pixel -- Cin = CPP * (one K step): the patch is CPP regions of 128-byte rows, one region per channel part
template <int WGM, int WGN, int TM, int TN, int MODE, int NST, bool RESPF, bool PATCH = false, int ROWB = ROW_BYTES, int CPP = 1>
__global__ void __launch_bounds__(64 * WGM * WGN) conv16x_kernel(const Conv16Args p, const int act)
{
    static_assert(ROWB == ROW_BYTES || (ROWB == 64 && MODE == MODE_F16 && !PATCH), "half-step rows: f16 mode only");
    static_assert(CPP == 1 || PATCH, "channel parts: patch mode only");
    constexpr bool USE_BUF = true;                        // (r05 A/B on the GPU: buffer loads with hardware zero fill >= 64-bit pointers + zero page on every layer)
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int PLANES = MODE == MODE_SPLIT ? 2 : 1;
    constexpr int ES = MODE == MODE_F32 ? 4 : 2;          // bytes per element
    constexpr int EPC = 16 / ES;                          // elements per 16-byte chunk: 8 / 4
    constexpr int BKE = ROWB / PLANES / ES;               // K elements per step: 64 (f16), 32 (split: 32 hi + 32 lo), 32 (fp32), 32 (f16 half step)
    constexpr int RB = ROWB / PLANES;                     // bytes of one tile row in one plane's LDS region
    constexpr int CPR = RB / 16;                          // 16-byte chunks per row: 8 / 4
    constexpr int RPI = 64 / CPR;                         // rows one wavefront-instruction fills: 8 / 16
    constexpr int SWS = RB == 128 ? 1 : 2;                // chunk q of row i sits at position q ^ ((i >> SWS) & (CPR - 1))
    constexpr int QA = BM / (RPI * NW), QB = BN / (RPI * NW);      // load instructions per wavefront, plane and stage
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile rows must be a multiple of the loader pass");
    constexpr int A_REGION = BM * RB, B_REGION = BN * RB;
    constexpr int STAGE = PLANES * (A_REGION + B_REGION);          // (BM + BN) * 128 bytes
    constexpr int NJ = ROWB / PLANES / 32;                // slices per step (two chunks = one fragment pair each): 4 / 2 / 4 / 2
    constexpr int NACC = MODE == MODE_SPLIT ? 2 : 1;
    constexpr int NR = PLANES * (TM + TN);                // fragment reads per slice
    constexpr int NM = TM * TN * (MODE == MODE_SPLIT ? 3 : MODE == MODE_F32 ? 4 : 1);     // MFMAs per slice
    constexpr int NLD = PLANES * (QA + QB);               // direct-to-LDS loads per wavefront and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    // dynamic batch: the launch is sized for the capacity, the tiles that exist are the live rows'; the XCD-aware order is laid over THOSE (a
    // contiguous run per XCD of the live tiles -- laid over the capacity it would hand all the dead tiles to the last XCDs and leave the work
    // of the others unchanged), the workgroups beyond them leave before any barrier
    const long long M = live_rows(p);
    const long long live_tiles = ((M + BM - 1) / BM) * p.tiles_n;
    if ((long long)blockIdx.x >= live_tiles) return;
    const long long tile = xcd_tile(live_tiles);
    const long long m0 = (tile / p.tiles_n) * BM;
    const int n0 = (int)(tile % p.tiles_n) * BN;

    // ---- loader.  Instruction q of this wavefront fills rows [(q * NW + wave) * RPI, + RPI) of A (and of B); lane = (row lrow, position pc) and
    // fetches the LOGICAL chunk that belongs at its position.  Offsets are bytes relative to the workgroup's base pixel (A) / the weights (B).
    const int lrow = lane / CPR, pc = lane % CPR;
    long long base_pix;
    {
        const unsigned hw = (unsigned)(p.Ho * p.Wo), n = (unsigned)m0 / hw;
        const int rem = (int)((unsigned)m0 - n * hw), ho = rem / p.Wo;
        const int hi = ho * p.stride - p.pad;
        base_pix = (long long)n * p.H * p.W + (long long)(hi > 0 ? hi : 0) * p.W;
    }
    const unsigned char *abase[PLANES], *wbase[PLANES];   // (MODE_F32: the _Float16 * members of the argument block hold float pointers)
    abase[0] = reinterpret_cast<const unsigned char *>(p.x) + base_pix * p.x_pix * ES;
    if (MODE == MODE_SPLIT) abase[PLANES - 1] = reinterpret_cast<const unsigned char *>(p.x_lo) + base_pix * p.x_pix * ES;
    wbase[0] = reinterpret_cast<const unsigned char *>(p.w);
    if (MODE == MODE_SPLIT) wbase[PLANES - 1] = reinterpret_cast<const unsigned char *>(p.w_lo);
    __amdgpu_buffer_rsrc_t rs_a[PLANES], rs_b[PLANES];
    {
        const long long total_pix = (long long)((unsigned)p.M / (unsigned)(p.Ho * p.Wo)) * p.H * p.W;
        long long a_bytes = ((total_pix - base_pix - 1) * p.x_pix + p.Cin) * ES;
        if (a_bytes > 0x7ffffff0LL) a_bytes = 0x7ffffff0LL;
        const int w_bytes = (int)((long long)p.Cout * p.K * ES);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            rs_a[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)abase[pl], 0, (int)a_bytes, 0x00020000);
            rs_b[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)wbase[pl], 0, w_bytes, 0x00020000);
        }
    }
    int a_off0[QA], a_hi0[QA], a_wi0[QA], b_off0[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int row = (q * NW + wave) * RPI + lrow;
        const int lcq = pc ^ ((row >> SWS) & (CPR - 1));
        const long long m = m0 + row;
        a_off0[q] = OOB; a_hi0[q] = 0; a_wi0[q] = 0;
        if (m < M) {
            const unsigned n = (unsigned)m / (unsigned)(p.Ho * p.Wo);
            const int rem = (int)((unsigned)m - n * (unsigned)(p.Ho * p.Wo));
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            a_hi0[q] = ho * p.stride - p.pad; a_wi0[q] = wo * p.stride - p.pad;
            const int rel = (int)((long long)n * p.H * p.W - base_pix) + a_hi0[q] * p.W + a_wi0[q];
            a_off0[q] = (rel * p.x_pix + lcq * EPC) * ES;
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int row = (q * NW + wave) * RPI + lrow;
        const int lcq = pc ^ ((row >> SWS) & (CPR - 1));
        const int co = n0 + row;
        b_off0[q] = co < p.Cout ? (co * p.K + lcq * EPC) * ES : OOB;
    }
    int u_kh = 0, u_kw = 0, u_ci0 = 0, u_k0 = 0;          // tap of the step being loaded (wave-uniform)
    auto load16 = [&](const __amdgpu_buffer_rsrc_t &rs, const unsigned char *, int off, unsigned char *dst) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)dst, 16, off, 0, 0, 0);
    };
    auto issue_stage = [&](int buf) {                      // a stage beyond K is all zeros (it lands in a buffer nobody multiplies)
        const bool in_k = u_k0 < p.K;
        // (weights stay below 2^30 bytes -- host check -- so "valid offset + 2^30" is out of range for them: a scalar select, no branch)
        const int a_add = ((u_kh * p.W + u_kw) * p.x_pix + u_ci0) * ES, b_add = (USE_BUF && !in_k) ? 0x40000000 : u_k0 * ES;
        unsigned char *st = lds + buf * STAGE;
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            int off = a_off0[q] + a_add;                  // (an OOB row stays beyond 2^31: a_add is a small positive number)
            const int hi = a_hi0[q] + u_kh, wi = a_wi0[q] + u_kw;      // (branch-free on purpose: the step stays one basic block)
            off = (in_k & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W)) ? off : OOB;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) load16(rs_a[pl], abase[pl], off, st + pl * A_REGION + (q * NW + wave) * 1024);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            int off = b_off0[q] + b_add;
            if (!USE_BUF) off = in_k ? off : OOB;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) load16(rs_b[pl], wbase[pl], off, st + PLANES * A_REGION + pl * B_REGION + (q * NW + wave) * 1024);
        }
        u_k0 += BKE; u_ci0 += BKE;
        const bool wrap_c = u_ci0 >= p.Cin;
        u_ci0 = wrap_c ? 0 : u_ci0; u_kw += wrap_c ? 1 : 0;
        const bool wrap_w = u_kw == p.KW;
        u_kw = wrap_w ? 0 : u_kw; u_kh += wrap_w ? 1 : 0;
    };

    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;

    // fragment addressing: lane l reads row (l & 31) of an MFMA tile (+ 32 per tile: the swizzle term is the same for every tile), logical
    // chunk 2 j + (l >> 5) of the step's slice j
    const int sw = ((lane & 31) >> SWS) & (CPR - 1), hsel = lane >> 5;
    const int a_lane = (wm * TM * 32 + (lane & 31)) * RB;
    const int b_lane = PLANES * A_REGION + (wn * TN * 32 + (lane & 31)) * RB;
    int coff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) coff[j] = ((2 * j + hsel) ^ sw) << 4;
    i32x4 fa[2][PLANES][TM], fb[2][PLANES][TN];           // two fragment sets (16 bytes each: 8 f16 or 4 floats): slice j + 1 is read while slice j is multiplied
    auto read_frags = [&](int buf, int j, int set) {
        const unsigned char *sa = lds + buf * STAGE + a_lane + coff[j], *sb = lds + buf * STAGE + b_lane + coff[j];
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[set][pl][i] = *reinterpret_cast<const i32x4 *>(sa + pl * A_REGION + i * 32 * RB);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[set][pl][i] = *reinterpret_cast<const i32x4 *>(sb + pl * B_REGION + i * 32 * RB);
        }
    };
    auto mfmas = [&](int set) {
        if (MODE == MODE_F32) {
            // lane l holds k = 8 j + 4 (l >> 5) + r, r = 0..3: MFMA r multiplies the k pair (8 j + r, 8 j + 4 + r) -- the fmaf chain
            // 0,4,1,5,2,6,3,7 within every group of 8 that oracle/src/conv.c walks (tlk_conv.hip's contract, bit for bit).  r is the OUTER
            // loop: consecutive MFMAs write different accumulators (each accumulator still sees r = 0, 1, 2, 3 in order)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) {
                        const f32x4 af = __builtin_bit_cast(f32x4, fa[set][0][i]), bf = __builtin_bit_cast(f32x4, fb[set][0][jj]);
                        const float av = r == 0 ? af.x : r == 1 ? af.y : r == 2 ? af.z : af.w, bv = r == 0 ? bf.x : r == 1 ? bf.y : r == 2 ? bf.z : bf.w;
                        acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0][i][jj], 0, 0, 0);
                    }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                if (MODE == MODE_F32) {
                    // lane l holds k = 8 j + 4 (l >> 5) + r, r = 0..3: MFMA r multiplies the k pair (8 j + r, 8 j + 4 + r) -- the fmaf chain
                    // 0,4,1,5,2,6,3,7 within every group of 8 that oracle/src/conv.c walks (tlk_conv.hip's contract, bit for bit)
                    const f32x4 af = __builtin_bit_cast(f32x4, fa[set][0][i]), bf = __builtin_bit_cast(f32x4, fb[set][0][jj]);
                    acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[0][i][jj], 0, 0, 0);
                    acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[0][i][jj], 0, 0, 0);
                    acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[0][i][jj], 0, 0, 0);
                    acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[0][i][jj], 0, 0, 0);
                } else {
                    acc[0][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fa[set][0][i]), __builtin_bit_cast(h16x8, fb[set][0][jj]), acc[0][i][jj], 0, 0, 0);
                    if (MODE == MODE_SPLIT) {
                        acc[NACC - 1][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fa[set][0][i]), __builtin_bit_cast(h16x8, fb[set][PLANES - 1][jj]),
                                                                                      acc[NACC - 1][i][jj], 0, 0, 0);
                        acc[NACC - 1][i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fa[set][PLANES - 1][i]), __builtin_bit_cast(h16x8, fb[set][0][jj]),
                                                                                      acc[NACC - 1][i][jj], 0, 0, 0);
                    }
                }
            }
    };

    // Main loop, software-pipelined across the barrier: the LAST slice of a step is multiplied AFTER the step's barrier, behind the reads of the
    // next step's first fragments and the issue of the stage after next -- so the first thing a wavefront does when the barrier opens is start
    // memory work, and it has TM * TN (x 3) MFMAs in hand to cover that work's latency.  Scheduling barriers keep the chunks in this order.
    //   stage s lives in buffer s & 1;  it is loaded during step s - 1 (issued right after barrier s - 2), waited for before barrier s - 1.
    static_assert(NJ % 2 == 0, "fragment set parity assumes an even number of slices per step");
    const int steps = p.K / BKE;
    // The residual of this lane's output vectors is fetched NOW and waits in registers (RESPF: the tile configurations used for the short-K
    // 1 x 1 expansions, which are memory-bound: their epilogue used to sit on HBM latency)
    constexpr int LDC = BN + 4;
    constexpr int LDS_AVAIL = (NST * STAGE > WGM * 32 * LDC * 4) ? NST * STAGE : WGM * 32 * LDC * 4;
    constexpr int EPI = pick_epi(TM, WGM, LDS_AVAIL / (LDC * 4));
    static_assert(WGM * 32 * EPI * LDC * 4 <= LDS_AVAIL, "epilogue pass does not fit the LDS");
    constexpr int PROWS = WGM * 32 * EPI;                 // tile rows per pass
    constexpr int VW = MODE == MODE_F32 ? 4 : 8;          // output channels per lane and pass: 16 bytes of the tensors' element type
    constexpr int V_PER_ROW = BN / VW, NVEC = PROWS * V_PER_ROW, ITS = (NVEC + NT - 1) / NT, NPASS = TM / EPI;
    const bool has_res = p.res != nullptr, out32 = p.y32 != nullptr;
    i32x4 rpre[RESPF ? NPASS : 1][RESPF ? ITS : 1];        // 16 bytes of residual per vector: 8 f16 / 4 floats
    if (RESPF && has_res) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int idx = it * NT + tid;
                const int prow = idx / V_PER_ROW, ec = (idx - prow * V_PER_ROW) * VW;
                const int pw = prow / (32 * EPI), within = prow - pw * (32 * EPI);
                const long long m = m0 + pw * (TM * 32) + ps * EPI * 32 + within;
                const int co = n0 + ec;
                const bool ok = !(NVEC % NT != 0 && idx >= NVEC) && m < M && co < p.Cout;
                const i32x4 z = {0, 0, 0, 0};
                rpre[ps][it] = ok ? *reinterpret_cast<const i32x4 *>(reinterpret_cast<const unsigned char *>(p.res) + (m * p.r_pix + co) * ES) : z;
            }
    }
    if (PATCH) {
        // PATCH mode (3 x 3, stride 1, pad 1, Cin == one K step, whole image rows per tile): the implicit GEMM re-fetches every input pixel once
        // per tap -- nine times -- and on the 32-channel layers that stream IS the bound (HRNet's high-resolution branch: 6.3 TB/s through the
        // L2 at 100 TFLOP/s).  Here the tile's input rows + halo land in LDS ONCE, one 128-byte row per input pixel, and tap (kh, kw) of output
        // pixel (r, c) is patch row (r + kh) * (Wo + 2) + c + kw: the nine taps are nine shifted fragment reads of the same image.  The XOR
        // swizzle is keyed on the PATCH row, so the reader recomputes its term per tap (three integer operations per 16 MFMAs).  The k order (kh, kw, ci) and the fmaf chain within it are the implicit GEMM's: same bits.
        const int Wp = p.Wo + 2;
        const int prows = (BM / p.Wo + 2) * Wp;
        const int n_ai = (prows + 7) >> 3;                  // wavefront-instructions that fill one region of the patch (8 rows of 128 bytes each)
        const int preg = n_ai * 1024;                       // bytes of one region = one channel part (r06: CPP parts per pixel, Cin = CPP K steps)
        unsigned char *bs = lds + CPP * preg;
        const int h0 = (int)(((unsigned)m0 % (unsigned)(p.Ho * p.Wo)) / (unsigned)p.Wo);
        const int hb = h0 > 0 ? h0 - 1 : 0;                 // base_pix's image row
        for (int i = wave; i < n_ai * CPP; i += NW) {
            const int part = i / n_ai, ii = i - part * n_ai;
            const int row = ii * 8 + lrow;
            const int pr = row / Wp, px = row - pr * Wp;
            const int hi = h0 - 1 + pr, wi = px - 1;
            const int lcq = pc ^ ((row >> 1) & 7);
            int off = (((hi - hb) * p.W + wi) * p.x_pix + part * BKE + lcq * EPC) * ES;
            off = (row < prows && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) ? off : OOB;
            load16(rs_a[0], abase[0], off, lds + part * preg + ii * 1024);
        }
        // the weights stream through a ring of two 128-byte-row tiles, one K step (tap, channel part) each: step s + 1 is issued right behind the
        // barrier that opens step s (everybody is then done with step s - 1, whose buffer it overwrites) and has the step's MFMAs to land.
        // 52 KB in all on the 32-wide layers: three workgroups per CU -- with all nine taps resident (81 KB, two per CU) the load and epilogue phases showed.
        auto issue_b = [&](int sidx) {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int col = (q * NW + wave) * 8 + lrow;
                const int lcq = pc ^ ((col >> 1) & 7);
                const int co = n0 + col;
                load16(rs_b[0], wbase[0], co < p.Cout ? (co * p.K + sidx * BKE + lcq * EPC) * ES : OOB, bs + (sidx & 1) * B_REGION + (q * NW + wave) * 1024);
            }
        };
        issue_b(0);
        int prow0[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = (wm * TM + i) * 32 + (lane & 31), orow = ml / p.Wo;
            prow0[i] = orow * Wp + (ml - orow * p.Wo);
        }
        const int bp_lane = (wn * TN * 32 + (lane & 31)) * RB;
        auto read_patch = [&](int toff, int aoff, int sidx, int j, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pr = prow0[i] + toff;
                fa[set][0][i] = *reinterpret_cast<const i32x4 *>(lds + aoff + (pr << 7) + (((2 * j + hsel) ^ ((pr >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[set][0][i] = *reinterpret_cast<const i32x4 *>(bs + (sidx & 1) * B_REGION + bp_lane + i * 32 * RB + coff[j]);
        };
        constexpr int NSTEP = 9 * CPP;
#pragma unroll
        for (int sidx = 0; sidx < NSTEP; ++sidx) {
            const int t = sidx / CPP, part = sidx - t * CPP;
            const int toff = (t / 3) * Wp + t % 3, aoff = part * preg;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (sidx + 1 < NSTEP) issue_b(sidx + 1);
            read_patch(toff, aoff, sidx, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) read_patch(toff, aoff, sidx, j + 1, (j + 1) & 1);
                mfmas(j & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    } else if (NST == 1) {
        // one LDS stage (the short-K layers: K <= 128, one or two steps): load, wait, multiply -- what hides the latencies is the number of
        // workgroups a CU holds at 32 KB each, not a pipeline inside one of them
        for (int s = 0; s < steps; ++s) {
            issue_stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            read_frags(0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) read_frags(0, j + 1, (j + 1) & 1);
                mfmas(j & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    } else {
        // Ring of NST stages, NST - 1 of them in flight: stage s + NST - 1 is issued the moment stage s - 1's buffer is free and has NST - 1 steps
        // to land.  The wait is COUNTED -- "all but the youngest (NST - 2) stages' loads have landed" -- and the barrier is the raw s_barrier:
        // __syncthreads() with a direct-to-LDS load in flight drains the whole queue (it lowers to vmcnt(0)).  NST = 2 is the plain double
        // buffer; NST = 3 / 4 is what the SMALL launches want (a 64 x 64 tile multiplies for ~130 cycles per step: without several stages in
        // flight every step costs a full memory round trip).
        constexpr int INFLIGHT = (NST - 2) * NLD;         // loads of the younger stages that may still be outstanding at the wait
        static_assert(INFLIGHT <= 60, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int k = 0; k < NST - 1; ++k) issue_stage(k);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(0, 0, 0);
        issue_stage(NST - 1);
        int cur = 0;
        for (int s = 0; s < steps; ++s) {
            const int nxt = cur + 1 == NST ? 0 : cur + 1;
#pragma unroll
            for (int j = 0; j + 1 < NJ; ++j) {
                read_frags(cur, j + 1, (j + 1) & 1);
                mfmas(j & 1);
                interleave<NR < NM ? NR : NM, 0x100, NM - (NR < NM ? NR : NM)>();      // fragment read, MFMA, fragment read, MFMA, ... MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");      // stage s + 1 has landed; this wavefront's reads of stage s are done
            __builtin_amdgcn_s_barrier();                                 // ... for everyone
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, 0, 0);
            issue_stage(cur);                                             // stage s + NST overwrites stage s
            mfmas(1);                                                     // slice NJ - 1 of step s
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);           // first the reads, then one direct-to-LDS load per MFMA
            interleave<NLD < NM ? NLD : NM, 0x020, NM - (NLD < NM ? NLD : NM)>();
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the stages beyond K: zeros in flight towards LDS)
        __syncthreads();
    }

    // ---- epilogue: EPI MFMA tile rows of every wavefront at a time through LDS as fp32, then 8 output channels (16 bytes of f16) per lane.
    // C/D map of the 32x32 tile: column (= cout) = lane & 31, row (= pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    float *Cs = reinterpret_cast<float *>(lds);
    const bool scaled = MODE == MODE_SPLIT && (p.s_in || p.s_res || p.s_out);      // wave-uniform: the unscaled call runs the r05 epilogue, instruction for instruction
    SplitScales sc = {1.f, 1.f, 1.f};
    if (scaled) sc = load_scales(p);
    float amax = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += EPI) {
#pragma unroll
        for (int ii = 0; ii < EPI; ++ii)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                float *c = Cs + ((wm * EPI + ii) * 32 + 4 * (lane >> 5)) * LDC + (wn * TN + jj) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[0][i0 + ii][jj][r];
                    if (MODE == MODE_SPLIT) v = v + acc[NACC - 1][i0 + ii][jj][r] * LO_INV;
                    c[((r & 3) + 8 * (r >> 2)) * LDC] = v;
                }
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int idx = it * NT + tid;
            const int prow = idx / V_PER_ROW, ec = (idx - prow * V_PER_ROW) * VW;
            const int pw = prow / (32 * EPI), within = prow - pw * (32 * EPI);
            const long long m = m0 + pw * (TM * 32) + i0 * 32 + within;
            const int co = n0 + ec;
            if ((NVEC % NT != 0 && idx >= NVEC) || m >= M || co >= p.Cout) continue;      // Cout % 8 == 0 (checked by the host side; % 4 for fp32)
            float v[VW];
#pragma unroll
            for (int q4 = 0; q4 < VW / 4; ++q4) {
                const float4 c0 = *reinterpret_cast<const float4 *>(Cs + prow * LDC + ec + 4 * q4);
                v[4 * q4] = c0.x; v[4 * q4 + 1] = c0.y; v[4 * q4 + 2] = c0.z; v[4 * q4 + 3] = c0.w;
            }
            if (MODE == MODE_SPLIT && scaled) {      // scaled planes (r06): the accumulators hold the sum over x / s_in
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] *= sc.in;
            }
            if (p.bias) {
#pragma unroll
                for (int q4 = 0; q4 < VW / 4; ++q4) {
                    const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + co + 4 * q4);
                    v[4 * q4] += b0.x; v[4 * q4 + 1] += b0.y; v[4 * q4 + 2] += b0.z; v[4 * q4 + 3] += b0.w;
                }
            }
            float rv[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) rv[e] = 0.f;
            if (has_res) {
                const i32x4 rraw = RESPF ? rpre[RESPF ? i0 / EPI : 0][RESPF ? it : 0]
                                         : *reinterpret_cast<const i32x4 *>(reinterpret_cast<const unsigned char *>(p.res) + (m * p.r_pix + co) * ES);
                if (MODE == MODE_F32) {
                    const f32x4 rf = __builtin_bit_cast(f32x4, rraw);      // (whole-vector cast: a per-element bit_cast inside an unrolled loop read element 0 four times)
                    rv[0] = rf.x; rv[1] = rf.y; rv[2] = rf.z; rv[3] = rf.w;
                } else {
                    const h16x8 rh = __builtin_bit_cast(h16x8, rraw);
#pragma unroll
                    for (int e = 0; e < VW; ++e) rv[e] = (float)rh[e & 7];
                    if (MODE == MODE_SPLIT) {
                        const h16x8 rl = *reinterpret_cast<const h16x8 *>(p.res_lo + m * p.r_pix + co);
#pragma unroll
                        for (int e = 0; e < VW; ++e) rv[e] += (float)rl[e & 7] * LO_INV;
                        if (scaled) {
#pragma unroll
                            for (int e = 0; e < VW; ++e) rv[e] *= sc.res;
                        }
                    }
                }
                if (!p.res_post) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) v[e] += rv[e];
                }
            }
            if (act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] = act16<ACT_RELU>(v[e]);
            } else if (act == ACT_SILU) {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] = act16<ACT_SILU>(v[e]);
            }
            if (has_res && p.res_post) {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] += rv[e];
            }
            if (out32 || MODE == MODE_F32) {
                float *o = p.y32 + m * p.y_pix + co;
#pragma unroll
                for (int q4 = 0; q4 < VW / 4; ++q4) *reinterpret_cast<float4 *>(o + 4 * q4) = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
            } else if (MODE == MODE_SPLIT) {
                h16x8 oh, ol;
                if (scaled) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) { amax = fmaxf(amax, fabsf(v[e])); v[e] *= sc.out_inv; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f32(v[e % VW], h, l); oh[e] = h; ol[e] = l; }
                *reinterpret_cast<h16x8 *>(p.y + m * p.y_pix + co) = oh;
                *reinterpret_cast<h16x8 *>(p.y_lo + m * p.y_pix + co) = ol;
            } else {
                h16x8 oh;
#pragma unroll
                for (int e = 0; e < 8; ++e) oh[e] = (_Float16)v[e % VW];
                *reinterpret_cast<h16x8 *>(p.y + m * p.y_pix + co) = oh;
            }
        }
        if (i0 + EPI < TM) __syncthreads();
    }
    if (MODE == MODE_SPLIT && scaled) note_amax(p, amax);
}

template <int WGM, int WGN, int TM, int TN, int MODE, int NST, bool RESPF, int ROWB = ROW_BYTES> int launch_x(Conv16Args &a, int act, hipStream_t st)
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, NT = 64 * WGM * WGN;
    constexpr size_t STAGES = (size_t)NST * (BM + BN) * ROWB, EPI_MIN = (size_t)WGM * 32 * (BN + 4) * 4;
    constexpr size_t LDS_BYTES = STAGES > EPI_MIN ? STAGES : EPI_MIN;
    static_assert(LDS_BYTES <= 160 * 1024, "the stages must fit the CU's LDS");
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
    if (a.tiles > 0x7fffffffLL || a.M > 0x7fffffffLL) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: more than 2^31 - 1 output pixels in one launch");
    auto kern = conv16x_kernel<WGM, WGN, TM, TN, MODE, NST, RESPF, false, ROWB>;
    static bool attr_set = false;
    if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles), dim3(NT), LDS_BYTES, st, a, act);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// PATCH mode: 3 x 3 / stride 1 / pad 1 layers whose Cin is exactly CPP K steps and whose tiles are whole image rows
template <int WGM, int WGN, int TM, int TN, int MODE, bool RESPF, int CPP = 1> int launch_patch(Conv16Args &a, int act, hipStream_t st, const char *who)
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, NT = 64 * WGM * WGN;
    constexpr int BKE = MODE == MODE_F32 ? 32 : 64;
    static_assert(MODE != MODE_SPLIT, "patch mode: one plane");
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin != BKE * CPP || a.H != a.Ho || a.W != a.Wo || a.Wo < 8 || a.Wo > 64 || BM % a.Wo != 0 ||
        ((long long)a.Ho * a.Wo) % BM != 0)
        return fail(TLK_EINVAL, std::string(who) + ": the patch kernel takes 3 x 3 / stride 1 / pad 1 layers with Cin == " + std::to_string(BKE * CPP) +
                                    " (its K steps per pixel) and tiles of whole image rows");
    const int prows = (BM / a.Wo + 2) * (a.Wo + 2);
    const size_t lds_bytes = (size_t)CPP * ((prows + 7) / 8) * 1024 + (size_t)2 * BN * ROW_BYTES;
    constexpr size_t LDS_MAX = (size_t)CPP * ((BM / 8 + 2) * (8 + 2) > (BM / 64 + 2) * (64 + 2) ? (BM / 8 + 2) * (8 + 2) + 8 : (BM / 64 + 2) * (64 + 2) + 8) * ROW_BYTES +
                               (size_t)2 * BN * ROW_BYTES;
    static_assert(LDS_MAX <= 160 * 1024, "patch + weights must fit the CU's LDS");
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
    if (a.tiles > 0x7fffffffLL || a.M > 0x7fffffffLL) return fail(TLK_EINVAL, std::string(who) + ": more than 2^31 - 1 output pixels in one launch");
    auto kern = conv16x_kernel<WGM, WGN, TM, TN, MODE, 1, RESPF, true, ROW_BYTES, CPP>;
    static bool attr_set = false;
    if (!attr_set) { TLK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX)); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles), dim3(NT), lds_bytes, st, a, act);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// tile configurations (tlk_conv16_set_config numbers them; the heuristic below picks by shape)
int launch_cfg_x(Conv16Args &a, bool split, int act, int cfg, hipStream_t st)
{
    if (!split) {
        switch (cfg) {
        case 1: return launch_x<2, 4, 4, 2, MODE_F16, 2, false>(a, act, st);     // 256 x 256, 8 wavefronts of 128 x 64: the compute-bound layers
        case 2: return launch_x<4, 2, 2, 2, MODE_F16, 2, true>(a, act, st);      // 256 x 128, 8 wavefronts of 64 x 64
        case 3: return launch_x<2, 2, 2, 2, MODE_F16, 1, true>(a, act, st);      // 128 x 128, 4 wavefronts, ONE stage (K <= 128), residual prefetched
        case 4: return launch_x<8, 1, 2, 2, MODE_F16, 2, false>(a, act, st);     // 512 x 64, 8 wavefronts of 64 x 64
        case 5: return launch_x<4, 1, 2, 2, MODE_F16, 2, true>(a, act, st);      // 256 x 64, 4 wavefronts of 64 x 64
        case 6: return launch_x<2, 2, 2, 2, MODE_F16, 2, true>(a, act, st);      // 128 x 128, 4 wavefronts of 64 x 64, residual prefetched
        case 7: return launch_x<4, 1, 2, 2, MODE_F16, 1, true>(a, act, st);      // 256 x 64, ONE stage
        case 8: return launch_x<2, 4, 2, 2, MODE_F16, 2, true>(a, act, st);      // 128 x 256, 8 wavefronts of 64 x 64
        case 9: return launch_x<2, 2, 2, 2, MODE_F16, 1, false>(a, act, st);     // 128 x 128, ONE stage, residual read in the epilogue (fewer registers)
        case 10: return launch_x<4, 1, 2, 2, MODE_F16, 1, false>(a, act, st);    // 256 x 64, ONE stage, no residual prefetch
        case 11: return launch_x<2, 1, 2, 2, MODE_F16, 1, true>(a, act, st);     // 128 x 64, TWO wavefronts, ONE stage
        case 12: return launch_x<2, 2, 1, 1, MODE_F16, 4, true>(a, act, st);     // 64 x 64, 4 wavefronts of 32 x 32, FOUR stages: the small launches
        case 13: return launch_x<2, 2, 1, 2, MODE_F16, 3, true>(a, act, st);     // 64 x 128, three stages
        case 14: return launch_x<2, 2, 2, 2, MODE_F16, 3, true>(a, act, st);     // 128 x 128, three stages
        case 15: return launch_x<4, 2, 2, 2, MODE_F16, 3, false>(a, act, st);    // 256 x 128, 8 wavefronts of 64 x 64, three stages
        case 16: return launch_x<2, 2, 1, 1, MODE_F16, 2, true>(a, act, st);     // 64 x 64, two stages
        case 17: return launch_patch<4, 1, 2, 2, MODE_F16, true>(a, act, st, "tlk_conv2d_nhwc_16");      // 256 x 64 PATCH: 3 x 3 / 1 on 64 channels, input rows resident (see the fp32 form)
        case 18: return launch_patch<4, 1, 1, 2, MODE_F16, true>(a, act, st, "tlk_conv2d_nhwc_16");      // 128 x 64 PATCH
        // r06: HALF-STEP tiles (K step = 32 halfs): Cin a multiple of 32 but not of 64
        case 19: return launch_x<2, 2, 1, 1, MODE_F16, 4, true, 64>(a, act, st);     // 64 x 64, four stages: the small launches
        case 20: return launch_x<2, 2, 2, 2, MODE_F16, 1, true, 64>(a, act, st);     // 128 x 128, ONE stage, residual prefetched
        case 21: return launch_x<2, 2, 1, 2, MODE_F16, 3, true, 64>(a, act, st);     // 64 x 128, three stages
        case 22: return launch_x<2, 2, 2, 2, MODE_F16, 2, true, 64>(a, act, st);     // 128 x 128, two stages
        // r06: 32-COLUMN tiles (outputs of <= 32 channels: HRNet-W32's high-resolution branch and the exchange paths into it), as in split mode
        case 23: return launch_x<2, 1, 2, 1, MODE_F16, 1, true, 64>(a, act, st);     // half step: 128 x 32, two wavefronts of 64 x 32, ONE stage
        case 24: return launch_x<2, 1, 2, 1, MODE_F16, 2, true, 64>(a, act, st);     // half step: 128 x 32, two stages
        case 25: return launch_x<2, 1, 2, 1, MODE_F16, 1, true>(a, act, st);         // full step: 128 x 32, two wavefronts, ONE stage
        case 26: return launch_x<4, 1, 2, 1, MODE_F16, 1, true>(a, act, st);         // full step: 256 x 32, four wavefronts of 64 x 32, ONE stage
        default: return fail(TLK_EINVAL, "tlk_conv16_set_config: f16 configurations are 1..26");
        }
    }
    switch (cfg) {
    case 1: return launch_x<2, 4, 2, 2, MODE_SPLIT, 2, false>(a, act, st);       // 128 x 256, wavefront 64 x 64 (x 2 accumulator sets)
    case 2: return launch_x<4, 2, 2, 2, MODE_SPLIT, 2, false>(a, act, st);       // 256 x 128
    case 3: return launch_x<2, 2, 2, 2, MODE_SPLIT, 2, false>(a, act, st);       // 128 x 128, four wavefronts
    case 4: return launch_x<4, 1, 2, 2, MODE_SPLIT, 2, false>(a, act, st);       // 256 x 64, four wavefronts of 64 x 64
    case 5: return launch_x<2, 2, 2, 2, MODE_SPLIT, 1, false>(a, act, st);       // 128 x 128, ONE stage
    case 6: return launch_x<4, 2, 1, 2, MODE_SPLIT, 1, false>(a, act, st);       // 128 x 128, 8 wavefronts of 32 x 64, ONE stage
    case 7: return launch_x<4, 2, 1, 2, MODE_SPLIT, 2, false>(a, act, st);       // 128 x 128, 8 wavefronts of 32 x 64, two stages
    // r06, 32-wide outputs (HRNet-W32's high-resolution branch: 64 of the split forward's 305 launches, 91 of its 245 ms on 64-wide tiles that
    // multiply 50 % padding): tiles of 32 columns.  A split-mode loader instruction fills 16 rows, so a 32-row B tile allows two wavefronts
    case 8: return launch_x<2, 1, 4, 1, MODE_SPLIT, 2, false>(a, act, st);       // 256 x 32, two wavefronts of 128 x 32, two stages
    case 9: return launch_x<2, 1, 4, 1, MODE_SPLIT, 1, false>(a, act, st);       // 256 x 32, ONE stage
    case 10: return launch_x<2, 1, 2, 1, MODE_SPLIT, 2, false>(a, act, st);      // 128 x 32, two wavefronts of 64 x 32, two stages
    case 11: return launch_x<2, 1, 2, 1, MODE_SPLIT, 1, false>(a, act, st);      // 128 x 32, ONE stage
    // (64-column tiles of two wavefronts -- 128 x 64 in one / two stages, 256 x 64 in one -- were measured on HRNet's 64-channel branch and gained
    //  nothing over 4 / the r04 kernel: 68.2 vs 69.5 ms over its 64-wide layers; not kept)
    case 12: return launch_x<2, 1, 2, 1, MODE_SPLIT, 3, false>(a, act, st);      // 128 x 32, THREE stages: the long K loops (3 x 3 on 256 channels into 32: 8.24 -> 6.91 ms)
    // (one-wavefront 64 x 32 / 128 x 32 tiles were measured too: 5-25 % behind 11 on the 3 x 3 / 32 layers; not kept)
    default: return fail(TLK_EINVAL, "tlk_conv16_set_config: split configurations are 1..12");
    }
}

// fp32 tensors on the same kernels (MODE_F32): the memory-bound layers of the fp32 networks
int launch_cfg_x32(Conv16Args &a, int act, int cfg, hipStream_t st)
{
    switch (cfg) {
    case 1: return launch_x<2, 2, 2, 2, MODE_F32, 1, true>(a, act, st);      // 128 x 128, ONE stage, residual prefetched (16 floats of it per lane and pass)
    case 2: return launch_x<2, 2, 2, 2, MODE_F32, 1, false>(a, act, st);     // 128 x 128, ONE stage
    case 3: return launch_x<4, 1, 2, 2, MODE_F32, 1, false>(a, act, st);     // 256 x 64, ONE stage
    case 4: return launch_x<2, 2, 2, 2, MODE_F32, 2, true>(a, act, st);      // 128 x 128, two stages
    case 5: return launch_x<2, 2, 1, 2, MODE_F32, 1, true>(a, act, st);      // 64 x 128, ONE stage
    case 6: return launch_x<4, 1, 2, 2, MODE_F32, 1, true>(a, act, st);      // 256 x 64, ONE stage, residual prefetched
    case 7: return launch_x<4, 1, 2, 1, MODE_F32, 1, true>(a, act, st);      // 256 x 32, ONE stage: the 32-wide layers (HRNet's high-resolution branch)
    case 8: return launch_x<4, 1, 2, 1, MODE_F32, 2, true>(a, act, st);      // 256 x 32, two stages
    case 9: return launch_x<4, 1, 4, 1, MODE_F32, 1, true>(a, act, st);      // 512 x 32, 128 x 32 per wavefront, ONE stage
    case 10: return launch_patch<4, 1, 2, 1, MODE_F32, true>(a, act, st, "tlk_conv2d_nhwc_f32");      // 256 x 32 PATCH: 3 x 3 on 32 channels, input rows resident
    case 11: return launch_patch<4, 1, 2, 1, MODE_F32, false>(a, act, st, "tlk_conv2d_nhwc_f32");
    case 12: return launch_patch<4, 1, 1, 1, MODE_F32, true>(a, act, st, "tlk_conv2d_nhwc_f32");      // 128 x 32 PATCH (34 KB: four workgroups per CU)
    case 13: return launch_patch<2, 1, 2, 1, MODE_F32, true>(a, act, st, "tlk_conv2d_nhwc_f32");      // 128 x 32 PATCH, two wavefronts of 64 x 32
    // r06: PATCH on 64 channels (two K steps per pixel: the patch is two regions) -- ResNet's layer 1 and HRNet's 64-channel branch
    case 14: return launch_patch<4, 1, 1, 2, MODE_F32, true, 2>(a, act, st, "tlk_conv2d_nhwc_f32");   // 128 x 64 PATCH, four wavefronts of 32 x 64
    case 15: return launch_patch<4, 1, 2, 2, MODE_F32, true, 2>(a, act, st, "tlk_conv2d_nhwc_f32");   // 256 x 64 PATCH, four wavefronts of 64 x 64
    case 16: return launch_patch<2, 1, 2, 2, MODE_F32, true, 2>(a, act, st, "tlk_conv2d_nhwc_f32");   // 128 x 64 PATCH, two wavefronts of 64 x 64
    case 17: return launch_patch<4, 1, 1, 2, MODE_F32, false, 2>(a, act, st, "tlk_conv2d_nhwc_f32");  // 128 x 64 PATCH, residual read in the epilogue
    // r06, from the with-residual 1 x 1 expansions of ResNet's layers 3 / 4 (K 256 / 512 > 1024 / 2048 channels): the residual read in the epilogue
    case 18: return launch_x<2, 2, 2, 2, MODE_F32, 2, false>(a, act, st);    // 128 x 128, two stages, no residual prefetch (fewer registers)
    case 19: return launch_x<4, 2, 2, 2, MODE_F32, 2, false>(a, act, st);    // 256 x 128, 8 wavefronts of 64 x 64, two stages
    default: return fail(TLK_EINVAL, "tlk_conv2d_set_config: the direct-to-LDS fp32 configurations are 21..39");
    }
}

}  // namespace

namespace tlk {
namespace c16 {

int launch32x(Conv16Args &a, int act, int cfg, hipStream_t st)
{
    if (a.Cin % 32 != 0 || a.K % 32 != 0 || a.Cout % 4 != 0) return cfg > 0 ? fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: the direct-to-LDS kernels need Cin % 32 == 0") : 1;
    {
        const long long span_rows = 512 / (a.Wo > 0 ? a.Wo : 1) + a.KH + 2;
        const long long a_span = (span_rows * a.stride + a.KH) * (long long)a.W * a.x_pix * 4 + (long long)a.H * a.W * a.x_pix * 4;
        if (a_span >= 0x7fffffffLL || (long long)a.Cout * a.K * 4 >= 0x3fffffffLL) return cfg > 0 ? fail(TLK_EINVAL, "tlk_conv2d_nhwc_f32: tensor rows too long for 32-bit tile offsets") : 1;
    }
    if (cfg <= 0) return 1;        // (the caller's own heuristic decides when these kernels take a layer)
    return launch_cfg_x32(a, act, cfg, st);
}

int g_last_cfg16x = -1;      // tlk_conv16_last_config: the tile configuration of the most recent large-tile launch (-1: the call went to the r04 kernels)

int launch16x(Conv16Args &a, bool split, bool out32, int act, int cfg, hipStream_t st)
{
    (void)out32;
    const bool half_cfg = !split && cfg >= 19 && cfg <= 24;
    // f16 layers whose Cin is a multiple of 32 but not of 64 take the half-step tiles (19..24); a forced full-step configuration on such a layer is refused
    const bool half_step = !split && a.Cin % 32 == 0 && a.Cin % 64 != 0;
    const int bke = (split || half_step || half_cfg) ? 32 : 64;
    if (a.Cin % bke != 0 || a.K % bke != 0) return cfg > 0 ? fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: the large-tile kernels need Cin to be a multiple of the K step") : 1;
    if (cfg > 0 && !split && half_step && !half_cfg) return fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: Cin is a multiple of 32 but not of 64: half-step tile configurations 19..24 only");
    // every offset the loader forms stays below 2^31: the rows of one tile + their halo, and the weights
    {
        const long long span_rows = 512 / (a.Wo > 0 ? a.Wo : 1) + a.KH + 2;
        const long long a_span = (span_rows * a.stride + a.KH) * (long long)a.W * a.x_pix * 2 + (long long)a.H * a.W * a.x_pix * 2;
        if (a_span >= 0x7fffffffLL || (long long)a.Cout * a.K * 2 >= 0x3fffffffLL) return cfg > 0 ? fail(TLK_EINVAL, "tlk_conv2d_nhwc_16: tensor rows too long for 32-bit tile offsets") : 1;
    }
    if (cfg <= 0) {
        // heuristic, from the per-layer table of the standalone probe (profiles/r05_conv16x_shapes.txt):
        //   * what a CU holds matters more than a pipeline inside one workgroup: the ONE-stage 128 x 128 tile (33 KB of LDS, three workgroups per
        //     CU) with the residual prefetched wins every memory-bound layer (the 1 x 1 expansions run at 5.2 TB/s, r04: 3.4) and most others;
        //   * the 256 x 256 tile wins the compute-bound layers -- Cout a multiple of 256, K >= 512, no residual, at least two rounds of tiles
        //     over the chip (1100 TFLOP/s on the 3 x 3 / 512 layers, r04: 830);
        //   * Cout <= 64: the 256 x 64 one-stage tile.
        const long long tiles256 = ((a.M + 255) / 256) * ((a.Cout + 255) / 256);
        if (!split && half_step) {
            const long long t128h = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
            // r06 (tools/sweep_conv16.py, profiles/r06_conv16_sweep.txt): narrow outputs on the 64 x 64 tile whatever the launch size -- a 128-wide tile
            // multiplies 75 % padding on 32 channels (HRNet's high-resolution branch, 64 launches per forward: 0.83 -> 0.67 ms, with residual 0.91 -> 0.76;
            // YOLOX-m's / CSPNeXt's 96 > 48 layers 0.84 -> 0.60)
            // r06, second sweep (profiles/r06_conv16_sweep_hrnet_f16_32wide.txt): <= 32 outputs on 32-COLUMN tiles of two wavefronts where the launch has
            // at least two of them per CU -- HRNet's 3 x 3 / 32 at 96 x 32: 0.67 -> 0.35 ms, with residual 0.75 -> 0.40 (64 launches per forward)
            if (a.Cout <= 32 && (a.M + 127) / 128 >= 512) cfg = 23;
            else if (a.Cout <= 64) cfg = 19;
            else if (t128h >= 384) cfg = 20;
            else if (t128h >= 256) cfg = 22;
            else if (((a.M + 63) / 64) * ((a.Cout + 127) / 128) >= 192) cfg = 21;
            else cfg = 19;
        } else if (!split) {
            const long long t128 = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
            //   * 3 x 3 / stride 1 on exactly 64 channels with whole image rows per tile (ResNet's layer 1): the PATCH kernel -- the tile's input
            //     rows land in LDS once instead of once per tap (716 vs 625 TFLOP/s at 2400 crops, 611 vs 510 at 100)
            const bool patch_ok = a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.H == a.Ho && a.W == a.Wo && a.Wo >= 8 && a.Wo <= 64 &&
                                  128 % a.Wo == 0 && ((long long)a.Ho * a.Wo) % 256 == 0 && a.Cout % 64 == 0;
            if (patch_ok) cfg = ((a.M + 255) / 256 >= 768) ? 17 : 18;
            else if (a.Cout <= 32 && a.K < 1024 && (a.M + 127) / 128 >= 512) cfg = 25;      // (the exchange paths' 1 x 1 into 32 channels: 0.121 -> 0.069 ms)
            else if (a.Cout <= 32) cfg = 16;                // r06 sweep: 32 wide -> 64 x 64 tiles (HRNet's 256 > 32 fuse layers 4.06 -> 2.90 ms)
            else if (a.Cout <= 64) cfg = ((a.M + 255) / 256 >= 768) ? (a.res ? 7 : 10) : 12;
            else if (a.Cout % 256 == 0 && a.K >= 1024 && !a.res && tiles256 >= 512) cfg = 1;
            // r06 sweep: odd multiples of 64 (192: CSPNeXt-m, YOLOX-m) are three 64-wide tiles exactly -- the 256 x 64 one-stage tile, 8-14 % ahead
            else if (t128 >= 384 && a.Cout % 128 == 64) cfg = 10;
            // (without a residual to prefetch the tile needs fewer registers: four workgroups per CU; r06 sweep: a long K loop -- basic blocks' 3 x 3
            //  with residual, HRNet -- hides the residual read by itself: 0.149 -> 0.131 ms)
            else if (t128 >= 384) cfg = (a.res && a.K < 1152) ? 3 : 9;
            //   * SMALL launches (the online step: one frame, ~100 crops): fewer than 1.5 128 x 128 tiles per CU -> no neighbours to hide a
            //     workgroup's memory round trips, so the pipeline goes back INSIDE the workgroup (two stages); fewer than one tile per CU ->
            //     smaller tiles, so that more CUs work, with three / four stages in flight (a 64 x 64 tile multiplies for ~130 cycles per step)
            else if (t128 >= 256) cfg = 6;
            else if (((a.M + 63) / 64) * ((a.Cout + 127) / 128) >= 192) cfg = 13;
            else cfg = 12;
        } else {
            // r06 (profiles/r06_conv16_sweep_hrnet_split.txt): outputs of <= 32 channels on 32-column tiles -- a 64-wide tile multiplies 50 % padding there
            // (HRNet-W32's high-resolution branch, 3 x 3 on 32 channels at 96 x 32: 1.28 -> 0.74 ms, with residual 1.58 -> 0.88, 64 launches per forward)
            if (a.Cout <= 32) cfg = a.K >= 1024 ? 12 : 11;
            else if (a.Cout % 256 == 0 && a.K >= 256 && !a.res && ((a.M + 127) / 128) * (a.Cout / 256) >= 512) cfg = 1;
            else if (a.Cout % 128 == 0 && a.K >= 256 && !a.res && ((a.M + 255) / 256) * (a.Cout / 128) >= 512) cfg = 2;
            // r06: the 1 x 1 expansions WITH residual (ResNet's c3: 44 of the split ReID forward's 120 ms sat on the r04 kernel).  128 x 128 tiles of
            // EIGHT wavefronts of 32 x 64 -- 64 accumulator registers per lane, two workgroups' worth of wavefronts per SIMD, residual planes read in the
            // epilogue: one stage for the short-K layers (K <= 128: 3.41 vs 3.69 ms on 64 -> 256, 2.12 vs 2.16 on 128 -> 512), two stages beyond
            // (1.40 vs 1.45 on 256 -> 1024, 4.08 vs 4.37 on 512 -> 2048; profiles/r06_split_res_probe.txt)
            else if (a.res && a.Cout % 128 == 0 && ((a.M + 127) / 128) * (a.Cout / 128) >= 512) cfg = a.K <= 128 ? 6 : 7;
            // ... and the 3 x 3 layers on 64 channels (ResNet's layer 1): 256 x 64 tiles of four 64 x 64 wavefronts, 2.57 vs 2.80 ms (profiles/r06_split_l1_probe.txt)
            // (stride 2 with a short K loop -- HRNet's 32 > 64 and 64 > 64 down-sampling steps -- stays on the r04 kernel: 0.45 vs 0.53 ms)
            else if (!a.res && a.KH == 3 && a.Cout == 64 && (a.M + 255) / 256 >= 768 && (a.stride == 1 || a.K >= 1024)) cfg = 4;
            // r06 (tools/sweep_conv16.py, profiles/r06_conv16_sweep.txt): the detector's layers in split mode -- a few hundred tiles, widths of 96 ... 768 --
            // all sat on the r04 kernel; the eight-wavefront 128 x 128 tile with two stages is 8-40 % faster on every one of them (YOLOX-m, 24 frames:
            // 10.2 -> 9.0 ms per forward) and within 5-12 % of the best configuration of each
            else if (a.Cout >= 96 && a.M <= 256 * 1024) cfg = 7;
            else return 1;
        }
    }
    g_last_cfg16x = cfg;
    return launch_cfg_x(a, split, act, cfg, st);
}

}  // namespace c16
}  // namespace tlk

