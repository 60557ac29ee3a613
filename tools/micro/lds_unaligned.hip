// Micro-benchmark (not product): cost of reading 6-8 consecutive bytes per lane from LDS at byte-granular addresses on gfx950:
//   0: six ds_read_u8      1: one unaligned ds_read_b64      2: aligned ds_read2_b32 + ds_read_b32 + 2 x v_alignbyte
//   3: one unaligned ds_read_b128 (pil_crop_kernel's tap read)   4: aligned 5 dwords (read_b128 at 16B-aligned? no: read2+read2+read) + alignbyte
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip ; run: ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(const int *off, unsigned *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char s[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    int o = off[threadIdx.x];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char *p = s + o;
        if (MODE == 0) {
            acc += p[0] + p[1] * 3 + p[2] * 5 + p[3] * 7 + p[4] * 11 + p[5] * 13;
        } else if (MODE == 1) {
            unsigned w[2]; __builtin_memcpy(w, p, 8);
            acc += (w[0] & 0xff) + ((w[0] >> 8) & 0xff) * 3 + ((w[0] >> 16) & 0xff) * 5 + (w[0] >> 24) * 7 + (w[1] & 0xff) * 11 + ((w[1] >> 8) & 0xff) * 13;
        } else if (MODE == 2) {
            const unsigned *q = reinterpret_cast<const unsigned *>(s + (o & ~3));
            const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
            const unsigned sh = (unsigned)o & 3u;
            const unsigned w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            acc += (w0 & 0xff) + ((w0 >> 8) & 0xff) * 3 + ((w0 >> 16) & 0xff) * 5 + (w0 >> 24) * 7 + (w1 & 0xff) * 11 + ((w1 >> 8) & 0xff) * 13;
        } else if (MODE == 3) {
            unsigned w[4]; __builtin_memcpy(w, p, 16);
            acc += (w[0] & 0xff) + ((w[0] >> 8) & 0xff) * 3 + ((w[1] >> 16) & 0xff) * 5 + (w[2] >> 24) * 7 + (w[3] & 0xff) * 11 + ((w[3] >> 8) & 0xff) * 13;
        } else {
            const unsigned *q = reinterpret_cast<const unsigned *>(s + (o & ~3));
            const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
            const unsigned sh = (unsigned)o & 3u;
            const unsigned w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh), w2 = __builtin_amdgcn_alignbyte(d3, d2, sh),
                           w3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
            acc += (w0 & 0xff) + ((w0 >> 8) & 0xff) * 3 + ((w1 >> 16) & 0xff) * 5 + (w2 >> 24) * 7 + (w3 & 0xff) * 11 + ((w3 >> 8) & 0xff) * 13;
        }
        o = (o + 512 + (acc & 0)) & 8191;        // next row, dependent on acc so the loop cannot be collapsed
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// look-up table reads: 0: u16 entries, index 0..1023   1: u32 entries, index 0..1023   2: u32 entries, index 0..255   3: u16, index 0..255
template <int MODE>
__global__ void __launch_bounds__(256) klut(unsigned *out, int iters, int spread)
{
    __shared__ __attribute__((aligned(16))) unsigned s32[3 * 1024];
    unsigned short *s16 = reinterpret_cast<unsigned short *>(s32);
    for (int i = threadIdx.x; i < 3 * 1024; i += 256) s32[i] = i * 2654435761u;
    __syncthreads();
    unsigned acc = 0, idx = threadIdx.x * 37u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            // smooth-image-like indices: a per-lane base plus a small per-value spread
            const unsigned t = (idx + q * spread) & ((MODE == 2 || MODE == 3) ? 255u : 1023u);
            if (MODE == 0 || MODE == 3) acc += s16[(q % 3) * 1024 + t];
            else acc += s32[(q % 3) * 1024 + t];
        }
        idx = idx * 1664525u + 1013904223u + (acc & 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> float runlut(unsigned *dout, int iters, int spread)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(klut<MODE>, dim3(2048), dim3(256), 0, 0, dout, iters, spread);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(klut<MODE>, dim3(2048), dim3(256), 0, 0, dout, iters, spread);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
template <int MODE> float run(const int *doff, unsigned *dout, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, doff, dout, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, doff, dout, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    int *doff; unsigned *dout; hipMalloc(&doff, 1024); hipMalloc(&dout, 2048 * 256 * 4);
    const int iters = 2000;
    for (int pat = 0; pat < 3; ++pat) {
        std::vector<int> off(256);
        for (int t = 0; t < 256; ++t) off[t] = pat == 0 ? ((t & 127) * 2 + (t & 127) / 3) * 1 + 5 + (t >> 7) * 544      // crop-like: ~2.3 bytes per lane, odd base
                                           : pat == 1 ? (t & 127) * 3 + 7 + (t >> 7) * 544                               // pil-like: 3 bytes per lane
                                                      : (t & 127) * 8 + (t >> 7) * 1024;                                // aligned reference
        hipMemcpy(doff, off.data(), 1024, hipMemcpyHostToDevice);
        float t0 = run<0>(doff, dout, iters), t1 = run<1>(doff, dout, iters), t2 = run<2>(doff, dout, iters), t3 = run<3>(doff, dout, iters), t4 = run<4>(doff, dout, iters);
        // cycles per wave-iteration per CU: 2048 WGs x 4 waves x iters over 256 CUs at 2.4 GHz
        const double f = 2.4e9 * 1e-3 / (2048.0 * 4 * iters / 256.0);
        printf("pattern %d: 6xu8 %.3f ms (%.1f cyc/wave-iter/CU)  unaligned b64 %.3f (%.1f)  aligned 3 dw + alignbyte %.3f (%.1f)  unaligned b128 %.3f (%.1f)  aligned 5 dw %.3f (%.1f)\n",
               pat, t0, t0 * f, t1, t1 * f, t2, t2 * f, t3, t3 * f, t4, t4 * f);
    }
    for (int spread : {1, 17}) {
        const int it2 = 500;
        const double f = 2.4e9 * 1e-3 / (2048.0 * 4 * it2 * 24 / 256.0);
        float a = runlut<0>(dout, it2, spread), b = runlut<1>(dout, it2, spread), c = runlut<2>(dout, it2, spread), d = runlut<3>(dout, it2, spread);
        printf("lut spread %d: u16[1024] %.3f ms (%.1f cyc/read-instr/CU)  u32[1024] %.3f (%.1f)  u32[256] %.3f (%.1f)  u16[256] %.3f (%.1f)\n", spread, a, a * f, b, b * f, c, c * f, d, d * f);
    }
    return 0;
}
