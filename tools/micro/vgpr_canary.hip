// vgpr_canary.hip -- does a co-running kernel on another stream disturb the registers of a resident wavefront?
// Each wavefront parks NREG known values in VGPRs (opaque to the compiler), idles ~1 ms in s_sleep steps, then checks them.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libvgpr_canary.so tools/micro/vgpr_canary.hip
#include <hip/hip_runtime.h>
template <int NREG>
__global__ void __launch_bounds__(256) canary_kernel(int iters, unsigned *__restrict__ n_bad, unsigned *__restrict__ first_bad)
{
    unsigned r[NREG];
    const unsigned seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
#pragma unroll
    for (int k = 0; k < NREG; ++k) { r[k] = seed + (unsigned)k * 40503u; asm volatile("" : "+v"(r[k])); }
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_sleep(32);
#pragma unroll
        for (int k = 0; k < NREG; ++k) asm volatile("" : "+v"(r[k]));
    }
    unsigned bad = 0, fk = 0, fx = 0;
#pragma unroll
    for (int k = 0; k < NREG; ++k) { const unsigned e = seed + (unsigned)k * 40503u; if (r[k] != e) { if (!bad) { fk = k; fx = r[k] ^ e; } ++bad; } }
    if (bad) { const unsigned slot = atomicAdd(n_bad, 1u); if (slot < 64) { first_bad[4 * slot] = blockIdx.x * 256u + threadIdx.x; first_bad[4 * slot + 1] = fk; first_bad[4 * slot + 2] = fx; first_bad[4 * slot + 3] = bad; } }
}
extern "C" int canary_launch(int nreg, int blocks, int iters, unsigned *n_bad, unsigned *first_bad, void *stream)
{
    if (nreg == 40) hipLaunchKernelGGL(canary_kernel<40>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, n_bad, first_bad);
    else hipLaunchKernelGGL(canary_kernel<96>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, n_bad, first_bad);
    return (int)hipGetLastError();
}

// active canary: every wavefront re-reads a fixed 21x21 byte window of a static image (byte loads, 7 per lane, like the LK kernel's
// window) and compares the weighted sum with the one of its first pass; 60 values parked in SGPRs are checked at the end
__global__ void __launch_bounds__(256) active_kernel(const unsigned char *__restrict__ img, int w, int h, int iters, unsigned *__restrict__ n_bad, unsigned *__restrict__ first_bad)
{
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned s[60];
#pragma unroll
    for (int k = 0; k < 60; ++k) { s[k] = (unsigned)wave * 2246822519u + (unsigned)k * 3266489917u; asm volatile("" : "+s"(s[k])); }
    const int x0 = (wave * 37) % (w - 24), y0 = (wave * 101) % (h - 24);
    long long first = 0; unsigned bad = 0, badit = 0;
    for (int it = 0; it < iters; ++it) {
        long long sum = 0;
        int z = 0; asm volatile("" : "+v"(z));          // opaque zero: the loads stay inside the loop
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int k = q * 64 + lane;
            if (k < 441) { const int y = k / 21, x = k - y * 21; sum += (long long)img[(size_t)(y0 + y) * w + x0 + x + z] * (k + 1); }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
#pragma unroll
        for (int k = 0; k < 60; ++k) asm volatile("" : "+s"(s[k]));
        if (it == 0) first = sum;
        else if (sum != first) { if (!bad) badit = it; ++bad; }
    }
    unsigned sbad = 0;
#pragma unroll
    for (int k = 0; k < 60; ++k) sbad += s[k] != (unsigned)wave * 2246822519u + (unsigned)k * 3266489917u;
    if ((bad || sbad) && lane == 0) { const unsigned slot = atomicAdd(n_bad, 1u); if (slot < 64) { first_bad[4 * slot] = wave; first_bad[4 * slot + 1] = bad; first_bad[4 * slot + 2] = badit; first_bad[4 * slot + 3] = sbad; } }
}
extern "C" int active_launch(int blocks, int iters, const unsigned char *img, int w, int h, unsigned *n_bad, unsigned *first_bad, void *stream)
{
    hipLaunchKernelGGL(active_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, w, h, iters, n_bad, first_bad);
    return (int)hipGetLastError();
}
