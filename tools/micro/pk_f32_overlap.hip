// pk_f32_overlap.hip -- do packed-FP32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) stay exact while MFMA-heavy kernels of
// ANOTHER stream share the compute units?  Every lane of a wavefront runs the same recurrence on the same (wave-uniform) inputs, once with packed
// and once with scalar instructions; a pass whose result is not identical in all 64 lanes, or differs between the two forms, is counted.
// Found while chasing tlk_cmc.hip's LK kernel (profiles/r02_pk_f32_overlap.md).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libpk_f32_overlap.so tools/micro/pk_f32_overlap.hip
#include <hip/hip_runtime.h>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) pk_kernel(int passes, int inner, unsigned *__restrict__ cnt)
{
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned bad_pk = 0, bad_sc = 0, bad_cmp = 0;
    for (int p = 0; p < passes; ++p) {
        float seed = 1.0f + (float)((wave * 131 + p * 17) & 1023) * 0.0009765625f;
        asm volatile("" : "+v"(seed));                                   // a VGPR copy per lane, identical in all lanes
        float2v a = {seed, seed * 0.75f}, b = {0.5f + seed * 0.125f, 0.96875f - seed * 0.0625f}, c = {0.001f * seed, -0.002f * seed};
        float sa0 = a.x, sa1 = a.y;
        for (int k = 0; k < inner; ++k) {
            float2v t;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(t), "v"(c));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
            float u0, u1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(sa0), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(sa1), "v"(b.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sa0) : "v"(u0), "v"(c.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sa1) : "v"(u1), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sa0) : "v"(sa0), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sa1) : "v"(sa1), "v"(b.y), "v"(c.y));
        }
        const int p0 = __float_as_int(a.x), p1 = __float_as_int(a.y), s0 = __float_as_int(sa0), s1 = __float_as_int(sa1);
        const bool uni_pk = __all(p0 == __builtin_amdgcn_readfirstlane(p0)) && __all(p1 == __builtin_amdgcn_readfirstlane(p1));
        const bool uni_sc = __all(s0 == __builtin_amdgcn_readfirstlane(s0)) && __all(s1 == __builtin_amdgcn_readfirstlane(s1));
        const bool same = __all(p0 == s0 && p1 == s1);
        bad_pk += !uni_pk; bad_sc += !uni_sc; bad_cmp += !same;
    }
    if (lane == 0) { atomicAdd(cnt, (unsigned)passes); atomicAdd(cnt + 1, bad_pk); atomicAdd(cnt + 2, bad_sc); atomicAdd(cnt + 3, bad_cmp); }
}
extern "C" int pk_launch(int blocks, int passes, int inner, unsigned *cnt, void *stream)
{
    hipLaunchKernelGGL(pk_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, passes, inner, cnt);
    return (int)hipGetLastError();
}
