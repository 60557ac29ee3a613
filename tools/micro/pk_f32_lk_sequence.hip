// pk_f32_lk_sequence.hip -- the EXACT instruction sequence the compiler emitted for the LK kernel's dx/dy update (r02, before the side-stream
// sources were built without packed FP32): two int64 -> f64 -> f32 conversions feeding v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 with the same
// operand-select / negate modifiers and the same s_nop placement, on wave-uniform inputs; a pass whose results are not identical in all 64 lanes
// is counted.  Registers are hard-coded so that the instruction stream is the one of profiles/r02_pk_f32_overlap.md, not the compiler's choice.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libpk_f32_lk_sequence.so tools/micro/pk_f32_lk_sequence.hip
#include <hip/hip_runtime.h>

// VAR 0: the compiler's sequence.  Bisect variants (next round's first GPU call; `python tools/micro/pk_f32_overlap.py --bisect`):
//  1: s_nop 1 after each v_cvt_f32_f64          2: s_nop 1 between ALL packed instructions (and after the conversions)
//  3: conversions replaced by v_mov of precomputed floats (packed part unchanged, incl. operand-select forms)
//  4: conversions kept, the operand-select / negate forms replaced by plain forms on pre-swizzled registers
//  5: conversions kept, packed instructions replaced by their scalar twins (v_mul_f32 / v_fma_f32 / v_add_f32)
#define CVT_TAIL(V) ((V) == 1 || (V) == 2 ? "s_nop 1\n" : "")
template <int VAR>
__global__ void __launch_bounds__(256) lkseq_kernel(int passes, unsigned *__restrict__ cnt)
{
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned bad = 0;
    unsigned h = (unsigned)wave * 2654435761u + 12345u;
    for (int p = 0; p < passes; ++p) {
        h = h * 1664525u + 1013904223u;
        int s1lo = (int)(h >> 3), s1hi = ((h >> 9) & 1) ? -1 : 0;              // sums of a few million, either sign
        s1lo = s1hi ? -(s1lo & 0x3fffff) : (s1lo & 0x3fffff);
        h = h * 1664525u + 1013904223u;
        int s2lo = (int)(h >> 5), s2hi = ((h >> 11) & 1) ? -1 : 0;
        s2lo = s2hi ? -(s2lo & 0x3fffff) : (s2lo & 0x3fffff);
        float a12 = 0.25f + (float)((h >> 8) & 255) * 0.01f, a22 = 3.0f + (float)((h >> 16) & 255) * 0.02f, a11 = 2.5f + (float)((h >> 24) & 255) * 0.015f;
        float dinv = 1.0f / (a11 * a22 - a12 * a12), nx = 100.25f + (float)(h & 63), ny = 57.5f + (float)((h >> 6) & 63);
        float scale = 9.5367431640625e-07f, half = 10.0f;
        const float b1f = (float)(double)(((long long)s1hi << 32) | (unsigned)s1lo), b2f = (float)(double)(((long long)s2hi << 32) | (unsigned)s2lo);
        float o_dx, o_dy, o_nx, o_ny, o_sq0, o_sq1;
        if constexpr (VAR == 0) {
        asm volatile(
                "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"
                "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"
                "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"
                "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n s_nop 4\n"
                "v_cvt_f64_i32_e32 v[104:105], v101\n"
                "v_cvt_f64_u32_e32 v[118:119], v100\n"
                "v_ldexp_f64 v[104:105], v[104:105], 32\n"
                "v_add_f64 v[118:119], v[104:105], v[118:119]\n"
                "v_cvt_f32_f64_e32 v119, v[118:119]\n"
                "v_cvt_f64_i32_e32 v[106:107], v103\n"
                "v_cvt_f64_u32_e32 v[104:105], v102\n"
                "v_ldexp_f64 v[106:107], v[106:107], 32\n"
                "v_add_f64 v[104:105], v[106:107], v[104:105]\n"
                "v_cvt_f32_f64_e32 v118, v[104:105]\n"
                "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "s_nop 0\n"
                "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"
                "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"
                "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"
                "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
                "s_nop 4\n"
                "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
                : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)
                : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),
                  [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42", "s43", "memory");
    }
        if constexpr (VAR == 1) {
        asm volatile(
                "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"
                "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"
                "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"
                "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n s_nop 4\n"
                "v_cvt_f64_i32_e32 v[104:105], v101\n"
                "v_cvt_f64_u32_e32 v[118:119], v100\n"
                "v_ldexp_f64 v[104:105], v[104:105], 32\n"
                "v_add_f64 v[118:119], v[104:105], v[118:119]\n"
                "v_cvt_f32_f64_e32 v119, v[118:119]\n s_nop 1\n"
                "v_cvt_f64_i32_e32 v[106:107], v103\n"
                "v_cvt_f64_u32_e32 v[104:105], v102\n"
                "v_ldexp_f64 v[106:107], v[106:107], 32\n"
                "v_add_f64 v[104:105], v[106:107], v[104:105]\n"
                "v_cvt_f32_f64_e32 v118, v[104:105]\n s_nop 1\n"
                "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "s_nop 0\n"
                "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"
                "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"
                "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"
                "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
                "s_nop 4\n"
                "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
                : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)
                : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),
                  [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42", "s43", "memory");
    }
        if constexpr (VAR == 2) {
        asm volatile(
                "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"
                "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"
                "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"
                "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n s_nop 4\n"
                "v_cvt_f64_i32_e32 v[104:105], v101\n"
                "v_cvt_f64_u32_e32 v[118:119], v100\n"
                "v_ldexp_f64 v[104:105], v[104:105], 32\n"
                "v_add_f64 v[118:119], v[104:105], v[118:119]\n"
                "v_cvt_f32_f64_e32 v119, v[118:119]\n s_nop 1\n"
                "v_cvt_f64_i32_e32 v[106:107], v103\n"
                "v_cvt_f64_u32_e32 v[104:105], v102\n"
                "v_ldexp_f64 v[106:107], v[106:107], 32\n"
                "v_add_f64 v[104:105], v[106:107], v[104:105]\n"
                "v_cvt_f32_f64_e32 v118, v[104:105]\n s_nop 1\n"
                "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n"
                "s_nop 1\n"
                "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "s_nop 1\n"
                "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                "s_nop 1\n"
                "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n s_nop 1\n"
                "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n s_nop 1\n"
                "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n s_nop 1\n"
                "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
                "s_nop 4\n"
                "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
                : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)
                : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),
                  [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42", "s43", "memory");
    }
        if constexpr (VAR == 3) {
        asm volatile(
                "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"
                "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"
                "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"
                "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n s_nop 4\n"
                "v_mov_b32 v119, %[b1f]\n v_mov_b32 v118, %[b2f]\n"
                "v_pk_mul_f32 v[118:119], v[118:119], s[40:41] op_sel_hi:[1,0]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[120:121], v[112:113], v[118:119] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "s_nop 0\n"
                "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"
                "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"
                "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"
                "v_pk_add_f32 v[118:119], v[120:121], s[42:43] op_sel_hi:[1,0]\n"
                "s_nop 4\n"
                "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
                : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)
                : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),
                  [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42", "s43", "memory");
    }
        if constexpr (VAR == 4) {
        asm volatile(
                "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"
                "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"
                "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"
                "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n s_nop 4\n"
                "v_cvt_f64_i32_e32 v[104:105], v101\n"
                "v_cvt_f64_u32_e32 v[118:119], v100\n"
                "v_ldexp_f64 v[104:105], v[104:105], 32\n"
                "v_add_f64 v[118:119], v[104:105], v[118:119]\n"
                "v_cvt_f32_f64_e32 v119, v[118:119]\n"
                "v_cvt_f64_i32_e32 v[106:107], v103\n"
                "v_cvt_f64_u32_e32 v[104:105], v102\n"
                "v_ldexp_f64 v[106:107], v[106:107], 32\n"
                "v_add_f64 v[104:105], v[106:107], v[104:105]\n"
                "v_cvt_f32_f64_e32 v118, v[104:105]\n"
                "v_mov_b32 v126, s40\n v_mov_b32 v127, s40\n v_mov_b32 v130, s42\n v_mov_b32 v131, s42\n s_nop 1\n"
                "v_pk_mul_f32 v[118:119], v[118:119], v[126:127]\n"
                "s_nop 0\n"
                "v_mov_b32 v128, v119\n v_mov_b32 v129, v118\n s_nop 1\n"
                "v_pk_mul_f32 v[120:121], v[112:113], v[128:129]\n"
                "s_nop 0\n"
                "v_xor_b32 v120, 0x80000000, v120\n v_xor_b32 v121, 0x80000000, v121\n s_nop 1\n"
                "v_pk_fma_f32 v[118:119], v[110:111], v[118:119], v[120:121]\n"
                "s_nop 0\n"
                "v_pk_mul_f32 v[122:123], v[114:115], v[118:119]\n"
                "v_pk_fma_f32 v[120:121], v[114:115], v[118:119], v[116:117]\n"
                "v_pk_mul_f32 v[124:125], v[122:123], v[122:123]\n"
                "v_pk_add_f32 v[118:119], v[120:121], v[130:131]\n"
                "s_nop 4\n"
                "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
                : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)
                : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),
                  [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42", "s43", "memory");
    }
        if constexpr (VAR == 5) {
        asm volatile(
                "v_mov_b32 v100, %[s1lo]\n v_mov_b32 v101, %[s1hi]\n v_mov_b32 v102, %[s2lo]\n v_mov_b32 v103, %[s2hi]\n"
                "v_mov_b32 v110, %[a12]\n v_mov_b32 v111, %[a12]\n v_mov_b32 v112, %[a22]\n v_mov_b32 v113, %[a11]\n"
                "v_mov_b32 v114, %[dinv]\n v_mov_b32 v115, %[dinv]\n v_mov_b32 v116, %[nx]\n v_mov_b32 v117, %[ny]\n"
                "v_readfirstlane_b32 s40, %[scale]\n v_readfirstlane_b32 s42, %[half]\n s_nop 4\n"
                "v_cvt_f64_i32_e32 v[104:105], v101\n"
                "v_cvt_f64_u32_e32 v[118:119], v100\n"
                "v_ldexp_f64 v[104:105], v[104:105], 32\n"
                "v_add_f64 v[118:119], v[104:105], v[118:119]\n"
                "v_cvt_f32_f64_e32 v119, v[118:119]\n"
                "v_cvt_f64_i32_e32 v[106:107], v103\n"
                "v_cvt_f64_u32_e32 v[104:105], v102\n"
                "v_ldexp_f64 v[106:107], v[106:107], 32\n"
                "v_add_f64 v[104:105], v[106:107], v[104:105]\n"
                "v_cvt_f32_f64_e32 v118, v[104:105]\n"
                "v_mul_f32 v118, s40, v118\n v_mul_f32 v119, s40, v119\n"
                "v_mul_f32 v120, v112, v119\n v_mul_f32 v121, v113, v118\n"
                "v_fma_f32 v118, v110, v118, -v120\n v_fma_f32 v119, v111, v119, -v121\n"
                "v_mul_f32 v122, v114, v118\n v_mul_f32 v123, v115, v119\n"
                "v_fma_f32 v120, v114, v118, v116\n v_fma_f32 v121, v115, v119, v117\n"
                "v_mul_f32 v124, v122, v122\n v_mul_f32 v125, v123, v123\n"
                "v_add_f32 v118, s42, v120\n v_add_f32 v119, s42, v121\n"
                "s_nop 4\n"
                "v_mov_b32 %[o_dx], v122\n v_mov_b32 %[o_dy], v123\n v_mov_b32 %[o_nx], v118\n v_mov_b32 %[o_ny], v119\n v_mov_b32 %[o_sq0], v124\n v_mov_b32 %[o_sq1], v125\n"
                : [o_dx] "=&v"(o_dx), [o_dy] "=&v"(o_dy), [o_nx] "=&v"(o_nx), [o_ny] "=&v"(o_ny), [o_sq0] "=&v"(o_sq0), [o_sq1] "=&v"(o_sq1)
                : [s1lo] "v"(s1lo), [s1hi] "v"(s1hi), [s2lo] "v"(s2lo), [s2hi] "v"(s2hi), [a12] "v"(a12), [a22] "v"(a22), [a11] "v"(a11), [dinv] "v"(dinv),
                  [nx] "v"(nx), [ny] "v"(ny), [scale] "v"(scale), [half] "v"(half), [b1f] "v"(b1f), [b2f] "v"(b2f)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                  "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s40", "s41", "s42", "s43", "memory");
    }
        auto uni = [&](float v) { const int b = __float_as_int(v); return __all(b == __builtin_amdgcn_readfirstlane(b)); };
        bad += !(uni(o_dx) && uni(o_dy) && uni(o_nx) && uni(o_ny) && uni(o_sq0) && uni(o_sq1));
    }
    if (lane == 0) { atomicAdd(cnt, (unsigned)passes); atomicAdd(cnt + 1, bad); }
}
extern "C" int lkseq_launch_var(int var, int blocks, int passes, unsigned *cnt, void *stream)
{
    switch (var) {
        case 1: hipLaunchKernelGGL(lkseq_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt); break;
        case 2: hipLaunchKernelGGL(lkseq_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt); break;
        case 3: hipLaunchKernelGGL(lkseq_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt); break;
        case 4: hipLaunchKernelGGL(lkseq_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt); break;
        case 5: hipLaunchKernelGGL(lkseq_kernel<5>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt); break;
        default: hipLaunchKernelGGL(lkseq_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt); break;
    }
    return (int)hipGetLastError();
}
extern "C" int lkseq_launch(int blocks, int passes, unsigned *cnt, void *stream)
{
    hipLaunchKernelGGL(lkseq_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, cnt);
    return (int)hipGetLastError();
}
