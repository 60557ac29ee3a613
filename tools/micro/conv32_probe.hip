// conv32_probe -- standalone probe of libtlk's fp32 convolution kernels:
//   * the direct RGB stem kernel (tlk_conv_stem.hip) against the implicit-GEMM kernel on the 4-channel-padded problem: BIT-identical output
//     required (same fmaf chain), milliseconds of both;
//   * the ResNet-50 1x1 expansions with residual (the memory-bound quarter of the fp32 step): milliseconds, TFLOP/s, GB/s on the algorithmic
//     bytes -- the launches tools/pmc_conv32.sh puts FETCH_SIZE / WRITE_SIZE counters on.
//   run: tools/micro/conv32_probe [crops=2400] [what=all|stem|exp] [iters=5] [cfg]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tlk.h"

extern "C" int tlk_conv2d_set_config(int cfg);

#define CK(x)                                                                                            \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define TK(x)                                                                                            \
    do {                                                                                                 \
        int r_ = (x);                                                                                    \
        if (r_ != 0) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, tlk_last_error()); exit(3); } \
    } while (0)

__device__ __forceinline__ float hash_uniform(unsigned long long i, unsigned seed)
{
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;
}
__global__ void fill_kernel(float *p, long long n, float scale, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = scale * hash_uniform((unsigned long long)i, seed);
}
// (pixels, 3) -> (pixels, 4) with a zero 4th channel
__global__ void pad4_kernel(const float *a, float *b, long long pixels)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels * 4; i += (long long)gridDim.x * blockDim.x) {
        const long long px = i >> 2; const int c = (int)(i & 3);
        b[i] = c < 3 ? a[px * 3 + c] : 0.f;
    }
}
__global__ void show_diff_kernel(const float *a, const float *b, long long n, int cout, unsigned long long *slot, const float *x, const float *w, const float *bias,
                                 const float *res, int cin)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (a[i] != b[i] && atomicAdd(slot, 1ull) < 6) {
            const long long row = i / cout; const int col = (int)(i % cout);
            double acc = 0, half0 = 0;
            for (int k = 0; k < cin; ++k) { acc += (double)x[row * cin + k] * w[(long long)col * cin + k]; if (k < 32) half0 = acc; }
            const double pre = acc + bias[col] + (res ? res[i] : 0.0);
            printf("      row %lld col %d: this %.9g, cfg 0 %.9g | naive relu(dot + bias + res) = %.9g (dot %.6g, first 32 k %.6g, bias %.4g, res %.4g)\n", row, col, a[i], b[i],
                   pre > 0 ? pre : 0.0, acc, half0, bias[col], res ? res[i] : 0.f);
        }
}
__global__ void count_diff_kernel(const unsigned *a, const unsigned *b, long long n, unsigned long long *out)
{
    unsigned long long c = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        c += (a[i] != b[i]) && !((a[i] << 1) == 0 && (b[i] << 1) == 0);        // (+0 / -0 count as equal)
    if (c) atomicAdd(out, c);
}

static hipEvent_t e0, e1;
template <class F> float time_ms(F run, int iters)
{
    run(); run();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0, 0)); run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char **argv)
{
    const int crops = argc > 1 ? atoi(argv[1]) : 2400;
    const char *what = argc > 2 ? argv[2] : "all";
    const int iters = argc > 3 ? atoi(argv[3]) : 5;
    const int cfg_only = argc > 4 ? atoi(argv[4]) : -2;       // expansions: only this tile configuration (-2: all)
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *bias; CK(hipMalloc(&bias, 4096 * 4));
    fill_kernel<<<16, 256>>>(bias, 4096, 0.5f, 5);
    unsigned long long *dcount; CK(hipMalloc(&dcount, 8));
    printf("# conv32_probe: %d crops\n", crops);
    if (!strcmp(what, "all") || !strcmp(what, "stem")) {
        struct Stem { const char *name; int H, W, Cout, k, pad; } stems[] = {{"ResNet-50 stem 7x7 s2 3>64 (384x128)", 384, 128, 64, 7, 3}, {"RTMPose stem 3x3 s2 3>32 (256x192)", 256, 192, 32, 3, 1},
                                                                              {"3x3 s2 3>48, ragged 250x190", 250, 190, 48, 3, 1}};
        for (const Stem &S : stems) {
            const int Ho = (S.H + 2 * S.pad - S.k) / 2 + 1, Wo = (S.W + 2 * S.pad - S.k) / 2 + 1;
            const long long px = (long long)crops * S.H * S.W, ny = (long long)crops * Ho * Wo * S.Cout, nw = (long long)S.Cout * S.k * S.k;
            float *x3, *x4, *w3, *w4, *ya, *yb;
            CK(hipMalloc(&x3, px * 3 * 4)); CK(hipMalloc(&x4, px * 4 * 4)); CK(hipMalloc(&w3, nw * 3 * 4)); CK(hipMalloc(&w4, nw * 4 * 4));
            CK(hipMalloc(&ya, ny * 4)); CK(hipMalloc(&yb, ny * 4));
            fill_kernel<<<2048, 256>>>(x3, px * 3, 2.0f, 21);
            fill_kernel<<<64, 256>>>(w3, nw * 3, 0.1f, 22);
            pad4_kernel<<<2048, 256>>>(x3, x4, px);
            pad4_kernel<<<64, 256>>>(w3, w4, nw);
            CK(hipMemset(ya, 0xff, ny * 4)); CK(hipMemset(yb, 0xee, ny * 4));
            auto run_gemm = [&] { TK(tlk_conv2d_nhwc_f32(x4, w4, bias, nullptr, ya, crops, S.H, S.W, 4, S.Cout, S.k, S.k, 2, S.pad, 1, 0, 0, 0, nullptr)); };
            auto run_stem = [&] { TK(tlk_conv2d_nhwc_f32(x3, w3, bias, nullptr, yb, crops, S.H, S.W, 3, S.Cout, S.k, S.k, 2, S.pad, 1, 0, 0, 0, nullptr)); };
            const float ms_g = time_ms(run_gemm, iters), ms_s = time_ms(run_stem, iters);
            CK(hipMemset(dcount, 0, 8));
            count_diff_kernel<<<1024, 256>>>((const unsigned *)ya, (const unsigned *)yb, ny, dcount);
            unsigned long long nd; CK(hipMemcpy(&nd, dcount, 8, hipMemcpyDeviceToHost));
            const double flops = 2.0 * crops * Ho * Wo * (double)S.Cout * S.k * S.k * 3;
            printf("  %-40s implicit GEMM on 4 padded channels %8.3f ms (%6.1f TFLOP/s) | direct stem kernel %8.3f ms (%6.1f TFLOP/s, %5.0f GB/s) | differing outputs: %llu of %lld %s\n",
                   S.name, ms_g, flops / ms_g / 1e9, ms_s, flops / ms_s / 1e9, (px * 3 + ny) * 4.0 / ms_s / 1e6, nd, ny, nd ? " <-- MISMATCH" : "");
            fflush(stdout);
            CK(hipFree(x3)); CK(hipFree(x4)); CK(hipFree(w3)); CK(hipFree(w4)); CK(hipFree(ya)); CK(hipFree(yb));
        }
    }
    if (!strcmp(what, "all") || !strcmp(what, "exp") || !strcmp(what, "hrnet")) {
        struct Exp { const char *name; int H, W, Cin, Cout, res, k; };
        const Exp hr[] = {{"hr 3x3 32>32 @96x32", 96, 32, 32, 32, 0, 3}, {"hr 3x3 32>32 +res", 96, 32, 32, 32, 1, 3}, {"hr 3x3 64>64 @48x16", 48, 16, 64, 64, 0, 3},
                          {"hr 3x3 64>64 +res", 48, 16, 64, 64, 1, 3}, {"hr 3x3 128>128 @24x8", 24, 8, 128, 128, 1, 3}, {"hr 3x3 256>256 @12x4", 12, 4, 256, 256, 1, 3},
                          {"hr 1x1 64>32 @48x16", 48, 16, 64, 32, 0, 1}, {"hr 3x3 s2 32>64", 96, 32, 32, 64, 0, 3}, {"l1 3x3 64>64 @96x32", 96, 32, 64, 64, 0, 3}};
        const Exp exps_[] = {
            {"l1 1x1 64>256 +res", 96, 32, 64, 256, 1}, {"l2 1x1 128>512 +res", 48, 16, 128, 512, 1}, {"l3 1x1 256>1024 +res", 24, 8, 256, 1024, 1},
            {"l4 1x1 512>2048 +res", 24, 8, 512, 2048, 1}, {"l1 1x1 256>64", 96, 32, 256, 64, 0}, {"l2 1x1 512>128", 48, 16, 512, 128, 0},
            {"l1 1x1 64>64", 96, 32, 64, 64, 0}, {"l2 1x1 256>128", 96, 32, 256, 128, 0}, {"l3 1x1 1024>256", 24, 8, 1024, 256, 0}, {"l4 1x1 2048>512", 24, 8, 2048, 512, 0}};
        const bool hrnet = !strcmp(what, "hrnet");
        const Exp *exps = hrnet ? hr : exps_;
        const int nexp = hrnet ? (int)(sizeof(hr) / sizeof(hr[0])) : (int)(sizeof(exps_) / sizeof(exps_[0]));
        for (int ei = 0; ei < nexp; ++ei) {
            Exp E = exps[ei];
            if (!hrnet) E.k = 1;
            const long long M = (long long)crops * E.H * E.W;
            float *x, *w, *r, *y;
            CK(hipMalloc(&x, M * E.Cin * 4)); CK(hipMalloc(&w, (size_t)E.Cout * E.Cin * E.k * E.k * 4)); CK(hipMalloc(&r, M * E.Cout * 4)); CK(hipMalloc(&y, M * E.Cout * 4));
            fill_kernel<<<2048, 256>>>(x, M * E.Cin, 1.0f, 31); fill_kernel<<<64, 256>>>(w, (long long)E.Cout * E.Cin * E.k * E.k, 0.1f / E.k, 32); fill_kernel<<<2048, 256>>>(r, M * E.Cout, 1.0f, 33);
            const int stride = strstr(E.name, " s2 ") ? 2 : 1, pad = E.k / 2;
            const int Ho = (E.H + 2 * pad - E.k) / stride + 1, Wo = (E.W + 2 * pad - E.k) / stride + 1;
            const long long Mo = (long long)crops * Ho * Wo;
            auto run = [&] { TK(tlk_conv2d_nhwc_f32(x, w, bias, E.res ? r : nullptr, y, crops, E.H, E.W, E.Cin, E.Cout, E.k, E.k, stride, pad, 1, 0, 0, 0, nullptr)); };
            const double flops = 2.0 * Mo * E.Cout * E.Cin * E.k * E.k, bytes = (double)(M * E.Cin + Mo * E.Cout * (E.res ? 2 : 1) + E.Cout * E.Cin * E.k * E.k) * 4;
            const int cfgs[] = {-1, 0, 2, 4, 9, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37};      // heuristic; two-stage tiles; 64 x 128 one-stage; the direct-to-LDS kernels
            float *yref; CK(hipMalloc(&yref, M * E.Cout * 4));
            TK(tlk_conv2d_set_config(0)); run(); CK(hipMemcpy(yref, y, M * E.Cout * 4, hipMemcpyDeviceToDevice));
            for (int cfg : cfgs) {
                if (cfg_only >= -1 && cfg != cfg_only) continue;
                TK(tlk_conv2d_set_config(cfg));
                CK(hipMemset(y, 0xff, M * E.Cout * 4));
                if (tlk_conv2d_nhwc_f32(x, w, bias, E.res ? r : nullptr, y, crops, E.H, E.W, E.Cin, E.Cout, E.k, E.k, stride, pad, 1, 0, 0, 0, nullptr) != 0) {
                    printf("  %-26s cfg %2d  not applicable (%s)\n", E.name, cfg, tlk_last_error()); continue;
                }
                const float ms = time_ms(run, iters);
                CK(hipMemset(dcount, 0, 8));
                count_diff_kernel<<<1024, 256>>>((const unsigned *)y, (const unsigned *)yref, Mo * E.Cout, dcount);
                unsigned long long nd; CK(hipMemcpy(&nd, dcount, 8, hipMemcpyDeviceToHost));
                printf("  %-26s cfg %2d  %8.3f ms  %6.1f TFLOP/s  %6.0f GB/s on %.2f GB algorithmic   bits differing from cfg 0: %llu%s\n", E.name, cfg, ms, flops / ms / 1e9,
                       bytes / ms / 1e6, bytes / 1e9, nd, nd ? "  <-- MISMATCH" : "");
                fflush(stdout);
                if (nd && E.k == 1) { CK(hipMemset(dcount, 0, 8)); show_diff_kernel<<<64, 256>>>(y, yref, M * E.Cout, E.Cout, dcount, x, w, bias, E.res ? r : nullptr, E.Cin); CK(hipDeviceSynchronize()); }
            }
            CK(hipFree(yref));
            TK(tlk_conv2d_set_config(-1));
            CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(r)); CK(hipFree(y));
        }
    }
    return 0;
}
