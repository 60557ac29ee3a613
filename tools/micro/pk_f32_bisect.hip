// pk_f32_bisect.hip -- r03 bisect of the r02 finding (profiles/r02_pk_f32_overlap.md): the 21-instruction dx/dy update of the LK kernel
// (two int64 -> f64 -> f32 conversions feeding v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 with operand-select / negate modifiers) gave
// lane-dependent results on wave-uniform inputs while MFMA kernels of another stream shared the compute units.  Every variant below is the same
// arithmetic; a pass is BAD when any lane's six outputs differ from the values the scalar C++ twin computes.  Bad passes are logged (inputs,
// per-output lane masks, the first wrong lane's values, HW_ID) so that the failing instruction can be identified offline
// (tools/micro/pk_f32_bisect.py).  Registers are hard-coded: the instruction stream is ours, not the compiler's.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -shared -fPIC -o tools/micro/libpk_f32_bisect.so tools/micro/pk_f32_bisect.hip
#include <hip/hip_runtime.h>

struct Event {                         // 48 dwords
    unsigned var, wave, pass, lane;    // first wrong lane
    unsigned in[4];                    // s1lo s1hi s2lo s2hi
    unsigned coef[6];                  // a12 a22 a11 dinv nx ny (float bits)
    unsigned got[6], exp[6];           // dx dy nx' ny' sq0 sq1
    unsigned mask[12];                 // per output: ballot of lanes that are WRONG (lo, hi)
    unsigned hwid, xcc, pad[6];
};

#define HEAD                                                                                                        \
    "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"      \
    "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"          \
    "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"          \
    "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n"
#define HEAD_HYGIENE                   /* every register the sequence can possibly read is initialised, nothing in flight */ \
    "v_readfirstlane_b32 s41, %[scale]\n v_readfirstlane_b32 s43, %[half]\n"                                        \
    "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n" \
    "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n" \
    "s_waitcnt vmcnt(0) lgkmcnt(0) expcnt(0)\n"
#define GAP "s_nop 4\n"
// conversion of (s1hi:s1lo) -> v119 and (s2hi:s2lo) -> v118, as the compiler wrote them (the first conversion's destination v119 is the upper
// half of its own source pair v[118:119]); NOP1 / NOP2 = what follows each v_cvt_f32_f64
#define CVT(NOP1, NOP2)                                 \
    "v_cvt_f64_i32_e32 v[104:105], v101\n"              \
    "v_cvt_f64_u32_e32 v[118:119], v100\n"              \
    "v_ldexp_f64 v[104:105], v[104:105], 32\n"          \
    "v_add_f64 v[118:119], v[104:105], v[118:119]\n"    \
    "v_cvt_f32_f64_e32 v119, v[118:119]\n" NOP1         \
    "v_cvt_f64_i32_e32 v[106:107], v103\n"              \
    "v_cvt_f64_u32_e32 v[104:105], v102\n"              \
    "v_ldexp_f64 v[106:107], v[106:107], 32\n"          \
    "v_add_f64 v[104:105], v[106:107], v[104:105]\n"    \
    "v_cvt_f32_f64_e32 v118, v[104:105]\n" NOP2
#define CVT_NO_OVERLAP                                  \
    "v_cvt_f64_i32_e32 v[104:105], v101\n"              \
    "v_cvt_f64_u32_e32 v[108:109], v100\n"              \
    "v_ldexp_f64 v[104:105], v[104:105], 32\n"          \
    "v_add_f64 v[108:109], v[104:105], v[108:109]\n"    \
    "v_cvt_f32_f64_e32 v119, v[108:109]\n"              \
    "v_cvt_f64_i32_e32 v[106:107], v103\n"              \
    "v_cvt_f64_u32_e32 v[104:105], v102\n"              \
    "v_ldexp_f64 v[106:107], v[106:107], 32\n"          \
    "v_add_f64 v[104:105], v[106:107], v[104:105]\n"    \
    "v_cvt_f32_f64_e32 v118, v[104:105]\n"
#define CVT_MOV "v_mov_b32 v119, %[b1f]\n v_mov_b32 v118, %[b2f]\n"
// the packed part as the compiler wrote it; N = wait states between the instructions ("s_nop 0" is the compiler's own)
#define PK(N0, N1)                                                                                  \
    "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n" N0                            \
    "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n" N0             \
    "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n" N0 \
    "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n" N1                                          \
    "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n" N1                              \
    "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n" N1                                          \
    "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
// SGPR operands replaced by VGPR pairs, the lane-crossing operand select and the negate forms kept
#define PK_NO_SGPR                                                                                  \
    "v_mov_b32 v126, s40\n v_mov_b32 v127, s40\n v_mov_b32 v130, s42\n v_mov_b32 v131, s42\n"       \
    "v_pk_mul_f32 v[118:119], v[118:119], v[126:127]\n s_nop 0\n"                                   \
    "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 0\n"      \
    "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n s_nop 0\n" \
    "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"                                             \
    "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"                                 \
    "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"                                             \
    "v_pk_add_f32 v[118:119], v[120:121], v[130:131]\n"
// SGPR operand forms kept, the half-swapping operand select replaced by a pre-swizzled register pair, negate by v_xor
#define PK_NO_SWAP                                                                                  \
    "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n s_nop 0\n"                     \
    "v_mov_b32 v128, v119\n v_mov_b32 v129, v118\n s_nop 1\n"                                       \
    "v_pk_mul_f32 v[120:121], v[112:113], v[128:129]\n s_nop 0\n"                                   \
    "v_xor_b32 v120, 0x80000000, v120\n v_xor_b32 v121, 0x80000000, v121\n s_nop 1\n"               \
    "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121]\n s_nop 0\n"                       \
    "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"                                             \
    "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"                                 \
    "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"                                             \
    "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
// A: the half-swapping operand select kept, the negate modifiers replaced by v_xor
#define PK_SWAP_ONLY                                                                                \
    "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n s_nop 0\n"                     \
    "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 0\n"      \
    "v_xor_b32 v120, 0x80000000, v120\n v_xor_b32 v121, 0x80000000, v121\n s_nop 1\n"               \
    "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121]\n s_nop 0\n"                       \
    "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"                                             \
    "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"                                 \
    "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"                                             \
    "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
// B: the negate modifiers kept, the half-swapping operand select replaced by a pre-swizzled register pair
#define PK_NEG_ONLY                                                                                 \
    "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n s_nop 0\n"                     \
    "v_mov_b32 v128, v119\n v_mov_b32 v129, v118\n s_nop 1\n"                                       \
    "v_pk_mul_f32 v[120:121], v[112:113], v[128:129]\n s_nop 0\n"                                   \
    "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n s_nop 0\n" \
    "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"                                             \
    "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"                                 \
    "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"                                             \
    "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
// no operand-select / negate / SGPR form at all
#define PK_PLAIN                                                                                    \
    "v_mov_b32 v126, s40\n v_mov_b32 v127, s40\n v_mov_b32 v130, s42\n v_mov_b32 v131, s42\n s_nop 1\n" \
    "v_pk_mul_f32 v[118:119], v[118:119], v[126:127]\n s_nop 0\n"                                   \
    "v_mov_b32 v128, v119\n v_mov_b32 v129, v118\n s_nop 1\n"                                       \
    "v_pk_mul_f32 v[120:121], v[112:113], v[128:129]\n s_nop 0\n"                                   \
    "v_xor_b32 v120, 0x80000000, v120\n v_xor_b32 v121, 0x80000000, v121\n s_nop 1\n"               \
    "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121]\n s_nop 0\n"                       \
    "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"                                             \
    "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"                                 \
    "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"                                             \
    "v_pk_add_f32 v[118:119], v[120:121], v[130:131]\n"
#define SCALAR                                                                                      \
    "v_mul_f32 v118, s40, v118\n v_mul_f32 v119, s40, v119\n"                                       \
    "v_mul_f32 v120, v112, v119\n v_mul_f32 v121, v113, v118\n"                                     \
    "v_fma_f32 v118, v110, v118, -v120\n v_fma_f32 v119, v111, v119, -v121\n"                       \
    "v_mul_f32 v122, v114, v118\n v_mul_f32 v123, v115, v119\n"                                     \
    "v_fma_f32 v120, v114, v118, v116\n v_fma_f32 v121, v115, v119, v117\n"                         \
    "v_mul_f32 v124, v122, v122\n v_mul_f32 v125, v123, v123\n"                                     \
    "v_add_f32 v118, s42, v120\n v_add_f32 v119, s42, v121\n"
#define TAIL                                                                                        \
    "s_nop 4\n"                                                                                     \
    "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
#define OPERANDS                                                                                                                                   \
    : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)                     \
    : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),     \
      [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)                                              \
    : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",       \
      "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42",  \
      "s43", "memory"
#define SEQ(BODY) asm volatile(BODY OPERANDS)

constexpr int N_VAR = 16;
template <int VAR>
__global__ void __launch_bounds__(256) bisect_kernel(int passes, unsigned *__restrict__ cnt, Event *__restrict__ log, int log_cap)
{
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned bad = 0;
    unsigned h = (unsigned)wave * 2654435761u + 12345u;
    for (int p = 0; p < passes; ++p) {
        h = h * 1664525u + 1013904223u;
        int s1lo = (int)(h >> 3), s1hi = ((h >> 9) & 1) ? -1 : 0;              // window sums of a few million, either sign
        s1lo = s1hi ? -(s1lo & 0x3fffff) : (s1lo & 0x3fffff);
        h = h * 1664525u + 1013904223u;
        int s2lo = (int)(h >> 5), s2hi = ((h >> 11) & 1) ? -1 : 0;
        s2lo = s2hi ? -(s2lo & 0x3fffff) : (s2lo & 0x3fffff);
        float a12 = 0.25f + (float)((h >> 8) & 255) * 0.01f, a22 = 3.0f + (float)((h >> 16) & 255) * 0.02f, a11 = 2.5f + (float)((h >> 24) & 255) * 0.015f;
        float dinv = 1.0f / (a11 * a22 - a12 * a12), nx = 100.25f + (float)(h & 63), ny = 57.5f + (float)((h >> 6) & 63);
        float scale = 9.5367431640625e-07f, half = 10.0f;
        const float b1f = (float)(double)(((long long)s1hi << 32) | (unsigned)s1lo), b2f = (float)(double)(((long long)s2hi << 32) | (unsigned)s2lo);
        float o_dx, o_dy, o_nx, o_ny, o_sq0, o_sq1;
        if constexpr (VAR == 0) SEQ(HEAD GAP CVT("", "") PK("s_nop 0\n", "") TAIL);                       // the compiler's sequence
        if constexpr (VAR == 1) SEQ(HEAD GAP CVT("s_nop 1\n", "s_nop 1\n") PK("s_nop 0\n", "") TAIL);     // wait states after both conversions
        if constexpr (VAR == 2) SEQ(HEAD GAP CVT("s_nop 1\n", "s_nop 1\n") PK("s_nop 1\n", "s_nop 1\n") TAIL);   // ... and between all packed
        if constexpr (VAR == 3) SEQ(HEAD GAP CVT_MOV PK("s_nop 0\n", "") TAIL);                           // no f64 producer at all
        if constexpr (VAR == 4) SEQ(HEAD GAP CVT("", "") PK_PLAIN TAIL);                                  // plain packed forms only
        if constexpr (VAR == 5) SEQ(HEAD GAP CVT("", "") SCALAR TAIL);                                    // scalar twins
        if constexpr (VAR == 6) SEQ(HEAD GAP CVT("", "s_nop 1\n") PK("s_nop 0\n", "") TAIL);              // wait only after the ADJACENT conversion (v118)
        if constexpr (VAR == 7) SEQ(HEAD GAP CVT("s_nop 1\n", "") PK("s_nop 0\n", "") TAIL);              // wait only after the first conversion (v119)
        if constexpr (VAR == 8) SEQ(HEAD GAP CVT_NO_OVERLAP PK("s_nop 0\n", "") TAIL);                    // cvt destination outside its source pair
        if constexpr (VAR == 9) SEQ(HEAD HEAD_HYGIENE GAP CVT("", "") PK("s_nop 0\n", "") TAIL);          // reproducer hygiene: everything initialised + waited
        if constexpr (VAR == 10) SEQ(HEAD GAP CVT("", "") PK_NO_SGPR TAIL);                               // no SGPR-pair operand
        if constexpr (VAR == 11) SEQ(HEAD GAP CVT("", "") PK_NO_SWAP TAIL);                               // no half-swapping op_sel, no neg
        if constexpr (VAR == 12) SEQ(HEAD GAP CVT("", "") PK("", "") TAIL);                               // the compiler's s_nop 0 removed
        if constexpr (VAR == 14) SEQ(HEAD GAP CVT("", "") PK_SWAP_ONLY TAIL);                             // A: swap kept, neg replaced
        if constexpr (VAR == 15) SEQ(HEAD GAP CVT("", "") PK_NEG_ONLY TAIL);                              // B: neg kept, swap replaced
        if constexpr (VAR == 13) SEQ(HEAD GAP CVT("", "v_nop\n v_nop\n") PK("s_nop 0\n", "") TAIL);       // v_nop instead of s_nop after the adjacent cvt
        // expected values: the scalar arithmetic, in C++
        const float b1 = b1f * scale, b2 = b2f * scale;
        const float m0 = a22 * b1, m1 = a11 * b2;
        const float f0 = __builtin_fmaf(a12, b2, -m0), f1 = __builtin_fmaf(a12, b1, -m1);
        const float e[6] = {dinv * f0, dinv * f1, __builtin_fmaf(dinv, f0, nx) + half, __builtin_fmaf(dinv, f1, ny) + half, (dinv * f0) * (dinv * f0), (dinv * f1) * (dinv * f1)};
        const float g[6] = {o_dx, o_dy, o_nx, o_ny, o_sq0, o_sq1};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 6; ++k) ok &= (__float_as_int(e[k]) == __float_as_int(g[k]));
        if (!__all(ok)) {
            ++bad;
            unsigned long long wrong[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wrong[k] = __ballot(__float_as_int(e[k]) != __float_as_int(g[k]));
            const unsigned long long any = __ballot(!ok);
            const int first = __builtin_ctzll(any);
            if (lane == first) {
                const unsigned slot = atomicAdd(cnt + 2, 1u);
                if ((int)slot < log_cap) {
                    Event &ev = log[slot];
                    ev.var = VAR; ev.wave = wave; ev.pass = p; ev.lane = lane;
                    ev.in[0] = s1lo; ev.in[1] = s1hi; ev.in[2] = s2lo; ev.in[3] = s2hi;
                    ev.coef[0] = __float_as_int(a12); ev.coef[1] = __float_as_int(a22); ev.coef[2] = __float_as_int(a11);
                    ev.coef[3] = __float_as_int(dinv); ev.coef[4] = __float_as_int(nx); ev.coef[5] = __float_as_int(ny);
                    for (int k = 0; k < 6; ++k) { ev.got[k] = __float_as_int(g[k]); ev.exp[k] = __float_as_int(e[k]);
                                                  ev.mask[2 * k] = (unsigned)wrong[k]; ev.mask[2 * k + 1] = (unsigned)(wrong[k] >> 32); }
                    unsigned hw, xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
                    ev.hwid = hw; ev.xcc = xcc;
                }
            }
        }
    }
    if (lane == 0) { atomicAdd(cnt, (unsigned)passes); atomicAdd(cnt + 1, bad); }
}

template <int V>
static void launch(int var, int blocks, int passes, unsigned *cnt, Event *log, int cap, hipStream_t st)
{
    if (var == V) hipLaunchKernelGGL(bisect_kernel<V>, dim3(blocks), dim3(256), 0, st, passes, cnt, log, cap);
    if constexpr (V + 1 < N_VAR) launch<V + 1>(var, blocks, passes, cnt, log, cap, st);
}
extern "C" int bisect_n_variants() { return N_VAR; }
extern "C" int bisect_event_dwords() { return (int)(sizeof(Event) / 4); }
extern "C" int bisect_launch(int var, int blocks, int passes, unsigned *cnt, void *log, int log_cap, void *stream)
{
    if (var < 0 || var >= N_VAR) return -1;
    launch<0>(var, blocks, passes, cnt, (Event *)log, log_cap, (hipStream_t)stream);
    return (int)hipGetLastError();
}
