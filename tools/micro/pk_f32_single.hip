// pk_f32_single.hip -- r03: ONE packed-FP32 instruction per pass on wave-uniform operands held in fixed registers, isolated by s_nop 4 on both
// sides, checked in every lane against the scalar arithmetic.  Tells which instruction FORM (operand select, negate, dependent pair) gives wrong
// lanes while MFMA kernels of another stream share the compute units (tools/micro/pk_f32_bisect.hip narrowed the LK dx/dy sequence down to the
// half-swapping operand select / negate pair).  Operands: a = v[100:101], b = v[102:103], c = v[104:105]; results d = v[106:107], e = v[108:109],
// both preset to a sentinel so that "not written" and "written as zero" can be told apart.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -shared -fPIC -o tools/micro/libpk_f32_single.so tools/micro/pk_f32_single.hip
#include <hip/hip_runtime.h>

struct Event {                      // 32 dwords
    unsigned op, wave, pass, lane;
    unsigned in[6];                 // a0 a1 b0 b1 c0 c1
    unsigned got[4], exp[4];        // d.lo d.hi e.lo e.hi
    unsigned mask[8];               // per output: wrong lanes (lo, hi dword)
    unsigned hwid, xcc, pad[4];
};
#define HEAD                                                                                                              \
    "v_mov_b32 v100, %[a0]\n v_mov_b32 v101, %[a1]\n v_mov_b32 v102, %[b0]\n v_mov_b32 v103, %[b1]\n"                    \
    "v_mov_b32 v104, %[c0]\n v_mov_b32 v105, %[c1]\n v_mov_b32 v106, %[sent]\n v_mov_b32 v107, %[sent]\n"                 \
    "v_mov_b32 v108, %[sent]\n v_mov_b32 v109, %[sent]\n s_nop 4\n"
#define TAIL "s_nop 4\n v_mov_b32 %[d0], v106\n v_mov_b32 %[d1], v107\n v_mov_b32 %[e0], v108\n v_mov_b32 %[e1], v109\n"
#define OPERANDS                                                                                                          \
    : [d0] "=&v"(d0), [d1] "=&v"(d1), [e0] "=&v"(e0), [e1] "=&v"(e1)                                                      \
    : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c0), [c1] "v"(c1), [sent] "v"(sent)                \
    : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory"
#define RUN(BODY) asm volatile(HEAD BODY TAIL OPERANDS)

constexpr int N_OP = 16;
template <int OP>
__global__ void __launch_bounds__(256) single_kernel(int passes, unsigned *__restrict__ cnt, Event *__restrict__ log, int log_cap)
{
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned bad = 0, h = (unsigned)wave * 2654435761u + 977u;
    const float sent = 12345.0f;
    for (int p = 0; p < passes; ++p) {
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { h = h * 1664525u + 1013904223u; v[k] = ((float)((h >> 8) & 0xffff) - 32768.0f) * 0.000244140625f + 0.0078125f; }
        const float a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3], c0 = v[4], c1 = v[5];
        float d0, d1, e0, e1, x[4] = {sent, sent, sent, sent};
        if constexpr (OP == 0) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103]\n"); x[0] = a0 * b0; x[1] = a1 * b1; }
        if constexpr (OP == 1) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n"); x[0] = a0 * b1; x[1] = a1 * b0; }
        if constexpr (OP == 2) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103] op_sel:[0,1]\n"); x[0] = a0 * b1; x[1] = a1 * b1; }
        if constexpr (OP == 3) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103] op_sel_hi:[1,0]\n"); x[0] = a0 * b0; x[1] = a1 * b0; }
        if constexpr (OP == 4) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103] op_sel:[1,0] op_sel_hi:[0,1]\n"); x[0] = a1 * b0; x[1] = a0 * b1; }
        if constexpr (OP == 5) { RUN("v_pk_fma_f32 v[106:107], v[100:101], v[102:103], v[104:105] neg_lo:[0,0,1] neg_hi:[0,0,1]\n");
                                 x[0] = __builtin_fmaf(a0, b0, -c0); x[1] = __builtin_fmaf(a1, b1, -c1); }
        if constexpr (OP == 6) { RUN("v_pk_fma_f32 v[106:107], v[100:101], v[102:103], v[104:105]\n");
                                 x[0] = __builtin_fmaf(a0, b0, c0); x[1] = __builtin_fmaf(a1, b1, c1); }
        if constexpr (OP == 7) { RUN("v_pk_add_f32 v[106:107], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n"); x[0] = a0 + b1; x[1] = a1 + b0; }
        if constexpr (OP == 8) { RUN("v_pk_fma_f32 v[106:107], v[100:101], v[102:103], v[104:105] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n");
                                 x[0] = __builtin_fmaf(a0, b0, c1); x[1] = __builtin_fmaf(a1, b1, c0); }
        // the pair of the LK sequence: swapped product, then the fma that subtracts it (dependent, the compiler's s_nop 0 between)
        if constexpr (OP == 9) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 0\n"
                                     "v_pk_fma_f32 v[108:109], v[104:105], v[102:103], v[106:107] neg_lo:[0,0,1] neg_hi:[0,0,1]\n");
                                 x[0] = a0 * b1; x[1] = a1 * b0; x[2] = __builtin_fmaf(c0, b0, -x[0]); x[3] = __builtin_fmaf(c1, b1, -x[1]); }
        // swapped operand produced by the instruction right before (as v[118:119] in the LK sequence)
        if constexpr (OP == 10) { RUN("v_pk_mul_f32 v[110:111], v[102:103], v[104:105]\n s_nop 0\n"
                                      "v_pk_mul_f32 v[106:107], v[100:101], v[110:111] op_sel:[0,1] op_sel_hi:[1,0]\n");
                                  x[0] = a0 * (b1 * c1); x[1] = a1 * (b0 * c0); }
        if constexpr (OP == 11) { RUN("v_pk_mov_b32 v[106:107], v[102:103], v[102:103] op_sel:[1,0]\n"); x[0] = b1; x[1] = b0; }
        if constexpr (OP == 12) { RUN("v_pk_mul_f32 v[106:107], v[100:101], v[102:103] neg_lo:[0,1] neg_hi:[0,1]\n"); x[0] = a0 * -b0; x[1] = a1 * -b1; }
        if constexpr (OP == 13) { RUN("v_pk_fma_f32 v[106:107], v[100:101], v[102:103], v[104:105] neg_lo:[0,0,1]\n");
                                  x[0] = __builtin_fmaf(a0, b0, -c0); x[1] = __builtin_fmaf(a1, b1, c1); }
        // scalar twins of OP 9 (control)
        if constexpr (OP == 14) { RUN("v_mul_f32 v106, v100, v103\n v_mul_f32 v107, v101, v102\n"
                                      "v_fma_f32 v108, v104, v102, -v106\n v_fma_f32 v109, v105, v103, -v107\n");
                                  x[0] = a0 * b1; x[1] = a1 * b0; x[2] = __builtin_fmaf(c0, b0, -x[0]); x[3] = __builtin_fmaf(c1, b1, -x[1]); }
        // OP 9 with plain forms on pre-swizzled / pre-negated registers (control: packed, but no modifiers)
        if constexpr (OP == 15) { RUN("v_mov_b32 v110, v103\n v_mov_b32 v111, v102\n s_nop 1\n"
                                      "v_pk_mul_f32 v[106:107], v[100:101], v[110:111]\n s_nop 0\n"
                                      "v_xor_b32 v110, 0x80000000, v106\n v_xor_b32 v111, 0x80000000, v107\n s_nop 1\n"
                                      "v_pk_fma_f32 v[108:109], v[104:105], v[102:103], v[110:111]\n");
                                  x[0] = a0 * b1; x[1] = a1 * b0; x[2] = __builtin_fmaf(c0, b0, -x[0]); x[3] = __builtin_fmaf(c1, b1, -x[1]); }
        const float g[4] = {d0, d1, e0, e1};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) ok &= (__float_as_int(x[k]) == __float_as_int(g[k]));
        if (!__all(ok)) {
            ++bad;
            unsigned long long wrong[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wrong[k] = __ballot(__float_as_int(x[k]) != __float_as_int(g[k]));
            const int first = __builtin_ctzll(__ballot(!ok));
            if (lane == first) {
                const unsigned slot = atomicAdd(cnt + 2, 1u);
                if ((int)slot < log_cap) {
                    Event &ev = log[slot];
                    ev.op = OP; ev.wave = wave; ev.pass = p; ev.lane = lane;
                    for (int k = 0; k < 6; ++k) ev.in[k] = __float_as_int(v[k]);
                    for (int k = 0; k < 4; ++k) { ev.got[k] = __float_as_int(g[k]); ev.exp[k] = __float_as_int(x[k]);
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
static void launch(int op, int blocks, int passes, unsigned *cnt, Event *log, int cap, hipStream_t st)
{
    if (op == V) hipLaunchKernelGGL(single_kernel<V>, dim3(blocks), dim3(256), 0, st, passes, cnt, log, cap);
    if constexpr (V + 1 < N_OP) launch<V + 1>(op, blocks, passes, cnt, log, cap, st);
}
extern "C" int single_n_ops() { return N_OP; }
extern "C" int single_event_dwords() { return (int)(sizeof(Event) / 4); }
extern "C" int single_launch(int op, int blocks, int passes, unsigned *cnt, void *log, int log_cap, void *stream)
{
    if (op < 0 || op >= N_OP) return -1;
    launch<0>(op, blocks, passes, cnt, (Event *)log, log_cap, (hipStream_t)stream);
    return (int)hipGetLastError();
}
