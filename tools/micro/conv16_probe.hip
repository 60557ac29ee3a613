// conv16_probe -- standalone (no Python, no torch) probe of libtlk's 16-bit MFMA convolution kernels on the ReID ResNet-50 layer shapes:
// every tile configuration of tlk_conv16x.hip against the r04 kernels, per layer: milliseconds, TFLOP/s on the algorithmic flops, GB/s on
// the algorithmic bytes (input + weights + residual + output once), and the largest deviation from a naive fp32-accumulating reference
// convolution on a small batch of the same shape.
//   build:  tools/micro/build_conv16_probe.sh conv16_probe      run:  tools/micro/conv16_probe [crops=2400] [mode=f16|split] [cfgs=auto|c1,c2,...] [filter] [yolox]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tlk.h"

extern "C" int tlk_conv16_set_config(int cfg);

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
    return (float)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;            // [-1, 1)
}

// value v = scale * u; f16 mode stores f16(v) in hi; split mode stores the (hi, lo) pair of the fp32 value
__global__ void fill_kernel(_Float16 *hi, _Float16 *lo, long long n, float scale, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = scale * hash_uniform((unsigned long long)i, seed);
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        if (lo) lo[i] = (_Float16)((v - (float)h) * 2048.f);
    }
}

__global__ void fill_f32_kernel(float *p, long long n, float scale, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = scale * hash_uniform((unsigned long long)i, seed);
}

// naive reference: one thread per (pixel, cout); double accumulation of the exact operand values (hi + lo / 2048 in split mode)
__global__ void ref_conv_kernel(const _Float16 *x, const _Float16 *xl, const _Float16 *w, const _Float16 *wl, const float *bias, const _Float16 *res,
                                const _Float16 *resl, double *y, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int act)
{
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const long long total = (long long)n * Ho * Wo * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const long long m = i / Cout;
        const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
        double acc = 0.0;
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) {
                const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
                if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                const long long xo = (((long long)b * H + hi) * W + wi) * Cin, wo_ = (((long long)co * KH + kh) * KW + kw) * Cin;
                for (int c = 0; c < Cin; ++c) {
                    double a = (double)(float)x[xo + c], bb = (double)(float)w[wo_ + c];
                    if (xl) { a += (double)(float)xl[xo + c] / 2048.0; bb += (double)(float)wl[wo_ + c] / 2048.0; }
                    acc += a * bb;
                }
            }
        acc += bias ? (double)bias[co] : 0.0;
        if (res) { double r = (double)(float)res[m * Cout + co]; if (resl) r += (double)(float)resl[m * Cout + co] / 2048.0; acc += r; }
        if (act == 1) acc = acc > 0 ? acc : 0;
        y[i] = acc;
    }
}

__global__ void diff_kernel(const _Float16 *y, const _Float16 *yl, const double *ref, long long n, double *out /* [0] max |d| / (|ref| + 1), [1] max |ref| */)
{
    double md = 0, mr = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double v = (double)(float)y[i];
        if (yl) v += (double)(float)yl[i] / 2048.0;
        const double r = ref[i], d = fabs(v - r) / (fabs(r) + 1.0);
        md = d > md || d != d ? d : md; mr = fabs(r) > mr ? fabs(r) : mr;
    }
    // (racy max through atomics on the bit pattern of non-negative doubles: monotone)
    atomicMax((unsigned long long *)&out[0], (unsigned long long)__double_as_longlong(md != md ? 1e300 : md));
    atomicMax((unsigned long long *)&out[1], (unsigned long long)__double_as_longlong(mr));
}

struct Layer { const char *name; int H, W, Cin, Cout, k, stride, pad, res, count; };

int main(int argc, char **argv)
{
    const int crops = argc > 1 ? atoi(argv[1]) : 2400;
    const bool split = argc > 2 && !strcmp(argv[2], "split");
    const char *cfg_arg = argc > 3 ? argv[3] : "auto";
    const char *filter = argc > 4 ? argv[4] : "";
    const int check_crops = 3;
    // ReID ResNet-50 (last stride 1) at 384 x 128 crops: the stem's output after max pooling is 96 x 32
    const Layer layers[] = {
        {"l1 1x1 64>64", 96, 32, 64, 64, 1, 1, 0, 0, 1},        {"l1 1x1 256>64", 96, 32, 256, 64, 1, 1, 0, 0, 2},
        {"l1 3x3 64", 96, 32, 64, 64, 3, 1, 1, 0, 3},           {"l1 1x1 64>256 +res", 96, 32, 64, 256, 1, 1, 0, 1, 4},
        {"l2 1x1 256>128", 96, 32, 256, 128, 1, 1, 0, 0, 1},    {"l2 3x3 128 s2", 96, 32, 128, 128, 3, 2, 1, 0, 1},
        {"l2 down 256>512 s2", 96, 32, 256, 512, 1, 2, 0, 0, 1}, {"l2 1x1 512>128", 48, 16, 512, 128, 1, 1, 0, 0, 3},
        {"l2 3x3 128", 48, 16, 128, 128, 3, 1, 1, 0, 3},        {"l2 1x1 128>512 +res", 48, 16, 128, 512, 1, 1, 0, 1, 4},
        {"l3 1x1 512>256", 48, 16, 512, 256, 1, 1, 0, 0, 1},    {"l3 3x3 256 s2", 48, 16, 256, 256, 3, 2, 1, 0, 1},
        {"l3 down 512>1024 s2", 48, 16, 512, 1024, 1, 2, 0, 0, 1}, {"l3 1x1 1024>256", 24, 8, 1024, 256, 1, 1, 0, 0, 5},
        {"l3 3x3 256", 24, 8, 256, 256, 3, 1, 1, 0, 5},         {"l3 1x1 256>1024 +res", 24, 8, 256, 1024, 1, 1, 0, 1, 6},
        {"l4 1x1 1024>512", 24, 8, 1024, 512, 1, 1, 0, 0, 1},   {"l4 down 1024>2048", 24, 8, 1024, 2048, 1, 1, 0, 0, 1},
        {"l4 1x1 2048>512", 24, 8, 2048, 512, 1, 1, 0, 0, 2},   {"l4 3x3 512", 24, 8, 512, 512, 3, 1, 1, 0, 3},
        {"l4 1x1 512>2048 +res", 24, 8, 512, 2048, 1, 1, 0, 1, 3}, {"reduce 2048>256", 24, 8, 2048, 256, 1, 1, 0, 0, 1},
    };
    // YOLOX-m at 640 x 640 (the layers whose input width is a multiple of the 64-element K step; `crops` = frames): H, W = INPUT size
    const Layer yolox_layers[] = {
        {"y 3x3 192 @80", 80, 80, 192, 192, 3, 1, 1, 0, 4},        {"y 3x3 192 @40", 40, 40, 192, 192, 3, 1, 1, 0, 14},
        {"y 3x3 384 @20", 20, 20, 384, 384, 3, 1, 1, 0, 4},        {"y 3x3 s2 192>384 80>40", 80, 80, 192, 384, 3, 2, 1, 0, 1},
        {"y 3x3 s2 384>768 40>20", 40, 40, 384, 768, 3, 2, 1, 0, 1}, {"y 1x1 192>192 @80", 80, 80, 192, 192, 1, 1, 0, 0, 3},
        {"y 1x1 384>192 @40", 40, 40, 384, 192, 1, 1, 0, 0, 6},    {"y 1x1 384>384 @40", 40, 40, 384, 384, 1, 1, 0, 0, 3},
        {"y 1x1 768>384 @20", 20, 20, 768, 384, 1, 1, 0, 0, 6},    {"y 1x1 192>192 @40", 40, 40, 192, 192, 1, 1, 0, 0, 10},
        {"y 3x3 s2 192 80>40", 80, 80, 192, 192, 3, 2, 1, 0, 1},   {"y 3x3 s2 384 40>20", 40, 40, 384, 384, 3, 2, 1, 0, 1},
        {"y 3x3 192 @20", 20, 20, 192, 192, 3, 1, 1, 0, 4},        {"y 1x1 1536>768 @20", 20, 20, 1536, 768, 1, 1, 0, 0, 1},
        {"y 1x1 768>768 @20", 20, 20, 768, 768, 1, 1, 0, 0, 2},    {"y 1x1 768>192 @40", 40, 40, 768, 192, 1, 1, 0, 0, 2},
        {"y 1x1 384>96 @80", 80, 80, 384, 96, 1, 1, 0, 0, 2},      {"y 1x1 192>96 @80", 80, 80, 192, 96, 1, 1, 0, 0, 2},
    };
    const bool yolox = argc > 5 && !strcmp(argv[5], "yolox");
    const Layer *Ls = yolox ? yolox_layers : layers;
    const int nl = yolox ? (int)(sizeof(yolox_layers) / sizeof(yolox_layers[0])) : (int)(sizeof(layers) / sizeof(layers[0]));
    // buffers sized for the largest layer
    long long max_x = 0, max_y = 0, max_w = 0;
    for (int i = 0; i < nl; ++i) {
        const Layer &L = Ls[i];
        const int Ho = (L.H + 2 * L.pad - L.k) / L.stride + 1, Wo = (L.W + 2 * L.pad - L.k) / L.stride + 1;
        max_x = std::max(max_x, (long long)std::max(crops, 3) * L.H * L.W * L.Cin);
        max_y = std::max(max_y, (long long)std::max(crops, 3) * Ho * Wo * L.Cout);
        max_w = std::max(max_w, (long long)L.Cout * L.k * L.k * L.Cin);
    }
    _Float16 *x, *xl = nullptr, *w, *wl = nullptr, *y, *yl = nullptr, *r, *rl = nullptr;
    float *bias;
    double *ref, *dres;
    CK(hipMalloc(&x, max_x * 2)); CK(hipMalloc(&w, max_w * 2)); CK(hipMalloc(&y, max_y * 2)); CK(hipMalloc(&r, max_y * 2));
    if (split) { CK(hipMalloc(&xl, max_x * 2)); CK(hipMalloc(&wl, max_w * 2)); CK(hipMalloc(&yl, max_y * 2)); CK(hipMalloc(&rl, max_y * 2)); }
    CK(hipMalloc(&bias, 4096 * 4));
    long long max_ref = 0;
    for (int i = 0; i < nl; ++i) {
        const Layer &L = Ls[i];
        const int Ho = (L.H + 2 * L.pad - L.k) / L.stride + 1, Wo = (L.W + 2 * L.pad - L.k) / L.stride + 1;
        max_ref = std::max(max_ref, (long long)check_crops * Ho * Wo * L.Cout);
    }
    CK(hipMalloc(&ref, max_ref * 8)); CK(hipMalloc(&dres, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fill_f32_kernel<<<64, 256>>>(bias, 4096, 0.5f, 77);
    printf("# conv16_probe: %d crops, mode %s, configurations %s (-1 = the r04 kernels)\n", crops, split ? "split" : "f16", cfg_arg);
    printf("# %-22s %3s %6s %4s | %8s %8s %8s | %9s\n", "layer", "cnt", "GFLOP", "cfg", "ms", "TFLOP/s", "GB/s", "max err");
    double tot_ms[16][2] = {{0}};      // [cfg + 1][loader]: sum over the forward of count * ms (best-of kept separately)
    double best_total = 0, r04_total = 0, flops_total = 0;
    for (int li = 0; li < nl; ++li) {
        const Layer &L = Ls[li];
        if (*filter && !strstr(L.name, filter)) continue;
        const int Ho = (L.H + 2 * L.pad - L.k) / L.stride + 1, Wo = (L.W + 2 * L.pad - L.k) / L.stride + 1;
        const long long nx = (long long)crops * L.H * L.W * L.Cin, ny = (long long)crops * Ho * Wo * L.Cout, nw = (long long)L.Cout * L.k * L.k * L.Cin;
        const int K = L.k * L.k * L.Cin;
        const double flops = 2.0 * crops * Ho * Wo * (double)L.Cout * K;
        const double bytes = (double)(nx + nw + ny * (L.res ? 2 : 1)) * (split ? 4 : 2);
        fill_kernel<<<2048, 256>>>(x, xl, nx, 1.0f, 11 + li);
        fill_kernel<<<256, 256>>>(w, wl, nw, 1.0f / sqrtf((float)K), 1000 + li);
        if (L.res) fill_kernel<<<2048, 256>>>(r, rl, ny, 1.0f, 2000 + li);
        CK(hipDeviceSynchronize());
        // candidate configurations: an explicit list, or every tile whose width fits the layer (BN <= Cout, BN >= Cout / 4)
        std::vector<int> cfgs;
        if (strcmp(cfg_arg, "auto")) {
            for (const char *q = cfg_arg; *q;) { cfgs.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
        } else {
            cfgs.push_back(-1);
            cfgs.push_back(0);                          // the library's own choice
            const int bn_f16[] = {0, 256, 128, 128, 64, 64, 128, 64, 256, 128, 64, 64, 64, 128, 128, 128, 64}, bn_split[] = {0, 256, 128, 128, 64, 128, 128, 128};
            const int ncfg = split ? 7 : 16;
            for (int c = 1; c <= ncfg; ++c) {
                const int bn = split ? bn_split[c] : bn_f16[c];
                if (bn <= std::max(L.Cout, 64) && (bn >= 128 || L.Cout <= 128 || c == 12 || c == 16 || L.Cout % 128 != 0)) cfgs.push_back(c);
            }
        }
        double best = 1e30, r04 = 0;
        for (int cfg : cfgs) {
            {
                const int ld = 1;
                TK(tlk_conv16_set_config(cfg));
                auto run = [&](int n) {
                    return tlk_conv2d_nhwc_16(x, xl, w, wl, bias, L.res ? r : nullptr, L.res ? rl : nullptr, y, yl, nullptr, n, L.H, L.W, L.Cin, L.Cout, L.k, L.k, L.stride,
                                              L.pad, 1, 0, 0, 0, nullptr);
                };
                // correctness on a small batch (ragged M: check_crops * Ho * Wo is not a multiple of 256 for every layer)
                CK(hipMemset(y, 0xff, (size_t)check_crops * Ho * Wo * L.Cout * 2));
                if (yl) CK(hipMemset(yl, 0xff, (size_t)check_crops * Ho * Wo * L.Cout * 2));
                const int rc = run(check_crops);
                if (rc != 0) { printf("  %-22s %3d %6.1f %4d | not applicable (%s)\n", L.name, L.count, flops / 1e9, cfg, tlk_last_error()); continue; }
                ref_conv_kernel<<<1024, 256>>>(x, xl, w, wl, bias, L.res ? r : nullptr, L.res ? rl : nullptr, ref, check_crops, L.H, L.W, L.Cin, L.Cout, L.k, L.k, L.stride, L.pad, 1);
                CK(hipMemset(dres, 0, 16));
                diff_kernel<<<256, 256>>>(y, yl, ref, (long long)check_crops * Ho * Wo * L.Cout, dres);
                double hres[2];
                CK(hipMemcpy(hres, dres, 16, hipMemcpyDeviceToHost));
                // timing
                TK(run(crops)); TK(run(crops));
                CK(hipDeviceSynchronize());
                float ms_best = 1e30f;
                for (int it = 0; it < 5; ++it) {
                    CK(hipEventRecord(e0, 0));
                    TK(run(crops));
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    ms_best = std::min(ms_best, ms);
                }
                const double tol = split ? 2e-6 : 1.5e-3;
                printf("  %-22s %3d %6.1f %3d%c | %8.3f %8.1f %8.0f | %9.2e %s\n", L.name, L.count, flops / 1e9, cfg, ' ', ms_best,
                       flops / ms_best / 1e9, bytes / ms_best / 1e6, hres[0], hres[0] <= tol ? "" : "  <-- MISMATCH");
                fflush(stdout);
                if (cfg < 0) r04 = ms_best;
                best = std::min(best, (double)ms_best);
                tot_ms[cfg + 1][ld] += L.count * ms_best;
            }
        }
        best_total += L.count * best; r04_total += L.count * r04; flops_total += L.count * flops;
    }
    printf("# forward (sum of count x ms): r04 kernels %.2f ms = %.1f TFLOP/s; best configuration per layer %.2f ms = %.1f TFLOP/s\n", r04_total,
           flops_total / r04_total / 1e9, best_total, flops_total / best_total / 1e9);
    return 0;
}
