// tlk_cosine.hip -- cosine gallery distance of plain StrongSORT on the f32 matrix cores.
//   cost[t][n] = min over the gallery rows g of track t of  1 - (g/|g|) . (d_n/|d_n|)
// (plugins/track/strong_sort/sort/nn_matching.py:30-50 _cosine_distance, :73-91 _nn_cosine_distance, :144-161 distance).
// The contraction (sum of gallery sizes) x N x D -- up to 10^4 x 100 x 512 = 1 GFLOP per frame -- is the one GEMM-shaped
// piece of the association path; it runs on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, so the result is what an fp32
// dot product gives, cf. numpy sgemm). One wavefront owns (track, 16 detections): it walks the track's gallery in
// chunks of 16 rows, keeps the running minimum in the accumulator layout and reduces it across rows at the end, so no
// atomics and no intermediate (Gtot x N) matrix ever reaches HBM.
#include "tlk_common.hpp"
#include "tlk_cosine.hpp"

using namespace tlk;

namespace {

// one wavefront per row: L2 norm (np.linalg.norm on float32)
__global__ void __launch_bounds__(BLOCK) rownorm_kernel(const float *__restrict__ x, int rows, int D, float *__restrict__ nrm)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * NWAVES + w;
    if (r >= rows) return;
    const float *p = x + (size_t)r * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += WAVE) { const float v = p[d]; ss += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) nrm[r] = sqrtf(ss);
}

// grid.x = ceil(N/16) detection tiles * 1, grid.y = ceil(T / NWAVES): wave w of block handles track blockIdx.y*NWAVES + w
__global__ void __launch_bounds__(BLOCK) cosine_gallery_kernel(const float *__restrict__ gallery, const int *__restrict__ offsets, int T,
                                                               const float *__restrict__ dets, int N, int D,
                                                               const float *__restrict__ gnorm, const float *__restrict__ dnorm,
                                                               double *__restrict__ out)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.y * NWAVES + w;
    if (t >= T) return;
    const int n0 = blockIdx.x * 16;
    const int i = lane & 15, g = lane >> 4;
    const int dn = min(n0 + i, N - 1);
    const float *drow = dets + (size_t)dn * D;
    const float nd = dnorm[dn];
    const int g_lo = offsets[t], g_hi = offsets[t + 1];
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    for (int c0 = g_lo; c0 < g_hi; c0 += 16) {
        const int gr = min(c0 + i, g_hi - 1);
        const float *grow = gallery + (size_t)gr * D;
        const float ng = gnorm[gr];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < D; c += 16) {
            const float4 a = *reinterpret_cast<const float4 *>(grow + c + 4 * g);
            const float4 b = *reinterpret_cast<const float4 *>(drow + c + 4 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x / ng, b.x / nd, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y / ng, b.y / nd, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z / ng, b.z / nd, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w / ng, b.w / nd, acc, 0, 0, 0);
        }
        // C layout: col = lane & 15 (detection), row = (lane >> 4) * 4 + reg (gallery row inside the chunk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = c0 + g * 4 + r < g_hi;
            const float v = 1.f - acc[r];
            if (valid && v < best[r]) best[r] = v;
        }
    }
    float m = fminf(fminf(best[0], best[1]), fminf(best[2], best[3]));
    m = fminf(m, __shfl_xor(m, 16));
    m = fminf(m, __shfl_xor(m, 32));
    if (g == 0 && n0 + i < N) out[(size_t)t * N + n0 + i] = (double)m;
}

// Pipelined variant for D = 16*DS in {64,128,256,512}: workgroup = (track, 16 detections), see tlk_cosine.hpp.
template <int DS>
__global__ void __launch_bounds__(BLOCK) cosine_gallery_kernel_t(const float *__restrict__ gallery, const int *__restrict__ offsets, int T,
                                                                 const float *__restrict__ dets, int N,
                                                                 const float *__restrict__ gnorm, const float *__restrict__ dnorm,
                                                                 double *__restrict__ out)
{
    __shared__ float s_min[NWAVES][16];
    const int t = blockIdx.y;
    cosine_gallery_tile<DS>(gallery, offsets[t], offsets[t + 1], gnorm, dets, N, blockIdx.x * 16, dnorm, out + (size_t)t * N, s_min);
}

}  // namespace

extern "C" int tlk_cosine_gallery_min_f32(const float *gallery_dev, const int32_t *offsets_dev, int T, int gallery_rows,
                                          const float *dets_dev, int N, int D, double *out_dev, void *hip_stream)
{
    if (T < 0 || N < 0 || gallery_rows < 0 || D < 16 || D % 16 != 0) return fail(TLK_EINVAL, "tlk_cosine_gallery_min_f32: bad shape (D % 16 == 0)");
    if (T == 0 || N == 0) return TLK_OK;
    if (!gallery_dev || !offsets_dev || !dets_dev || !out_dev) return fail(TLK_EINVAL, "tlk_cosine_gallery_min_f32: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    float *gn = nullptr, *dn = nullptr;
    TLK_HIP(hipMallocAsync((void **)&gn, sizeof(float) * (size_t)(gallery_rows > 0 ? gallery_rows : 1), st));
    TLK_HIP(hipMallocAsync((void **)&dn, sizeof(float) * (size_t)N, st));
    if (gallery_rows > 0)
        hipLaunchKernelGGL(rownorm_kernel, dim3((gallery_rows + NWAVES - 1) / NWAVES), dim3(BLOCK), 0, st, gallery_dev, gallery_rows, D, gn);
    hipLaunchKernelGGL(rownorm_kernel, dim3((N + NWAVES - 1) / NWAVES), dim3(BLOCK), 0, st, dets_dev, N, D, dn);
    const dim3 grid((N + 15) / 16, (T + NWAVES - 1) / NWAVES), grid_t((N + 15) / 16, T);
#define COS_LAUNCH(DS) hipLaunchKernelGGL((cosine_gallery_kernel_t<DS>), grid_t, dim3(BLOCK), 0, st, gallery_dev, (const int *)offsets_dev, \
                                          T, dets_dev, N, (const float *)gn, (const float *)dn, out_dev)
    if (D == 512) COS_LAUNCH(32);
    else if (D == 256) COS_LAUNCH(16);
    else if (D == 128) COS_LAUNCH(8);
    else if (D == 64) COS_LAUNCH(4);
    else hipLaunchKernelGGL(cosine_gallery_kernel, grid, dim3(BLOCK), 0, st, gallery_dev, (const int *)offsets_dev, T, dets_dev, N, D,
                            (const float *)gn, (const float *)dn, out_dev);
#undef COS_LAUNCH
    TLK_HIP(hipGetLastError());
    TLK_HIP(hipFreeAsync(gn, st));
    TLK_HIP(hipFreeAsync(dn, st));
    return TLK_OK;
}
