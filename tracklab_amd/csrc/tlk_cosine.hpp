// tlk_cosine.hpp -- the (track, 16 detections) tile of the cosine gallery minimum on v_mfma_f32_16x16x4_f32, shared by
// the stateless entry point (tlk_cosine.hip) and the plain-StrongSORT tracker bank (tlk_ssort.hip).
#pragma once
#include "tlk_common.hpp"

namespace {
using namespace tlk;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = 16*DS in {64,128,256,512}. Called by all 4 wavefronts of a 256-thread workgroup for gallery rows [g_lo, g_hi) (row-major,
// D floats each, norms in gnorm[row]) against detections n0..n0+15 (dets row-major, norms dnorm[n]); writes
// min over rows of 1 - cos to out_row[n0 + i] as double. The wave's detection operand (16 dets x D, normalised) stays in VGPRs
// for the whole gallery walk; gallery rows stream through two register buffers of GS k-steps each, the next group's 16-byte
// loads in flight while the current group's 4*GS MFMAs issue; the 4 waves interleave the 16-row chunks.
template <int DS>
__device__ __forceinline__ void cosine_gallery_tile(const float *__restrict__ gallery, int g_lo, int g_hi, const float *__restrict__ gnorm,
                                                    const float *__restrict__ dets, int N, int n0, const float *__restrict__ dnorm,
                                                    double *__restrict__ out_row, float (*s_min)[16])
{
    constexpr int D = DS * 16;
    constexpr int GS = DS >= 16 ? 8 : DS / 2;
    constexpr int GROUPS = DS / GS;               // even by construction
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int dn = min(n0 + i, N - 1);
    float4 breg[DS];
    {
        const float4 *drow = reinterpret_cast<const float4 *>(dets + (size_t)dn * D) + g;
        const float nd = dnorm[dn];
        const float rnd = 1.f / nd;
#pragma unroll
        for (int s = 0; s < DS; ++s) { float4 b = drow[s * 4]; b.x *= rnd; b.y *= rnd; b.z *= rnd; b.w *= rnd; breg[s] = b; }
    }
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    float4 A0[GS], A1[GS];
    auto load = [&](float4 (&buf)[GS], int c0, int grp) {
        const int gr = min(c0 + i, g_hi - 1);
        const float4 *grow = reinterpret_cast<const float4 *>(gallery + (size_t)gr * D) + g + grp * GS * 4;
#pragma unroll
        for (int s = 0; s < GS; ++s) buf[s] = grow[s * 4];
    };
    const int c_first = g_lo + 16 * w;
    if (c_first < g_hi) load(A0, c_first, 0);
    for (int c0 = c_first; c0 < g_hi; c0 += 16 * NWAVES) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int grp = 0; grp < GROUPS; ++grp) {
            // prefetch the next group (or the first group of the next chunk) into the other buffer
            if (grp + 1 < GROUPS) { if (grp & 1) load(A0, c0, grp + 1); else load(A1, c0, grp + 1); }
            else if (c0 + 16 * NWAVES < g_hi) { load(A0, c0 + 16 * NWAVES, 0); }       // GROUPS even -> group 0 always lives in A0
#pragma unroll
            for (int s = 0; s < GS; ++s) {
                const float4 a = (grp & 1) ? A1[s] : A0[s];
                const float4 b = breg[grp * GS + s];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
        // the gallery row's 1/|g| is applied to the finished dot product (one multiply per output instead of a division per
        // element in front of every MFMA, which made the loop VALU-bound); fp32 rounding differs by ~1e-7 relative.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = c0 + g * 4 + r;
            const bool valid = row < g_hi;
            const float v = 1.f - acc[r] * (1.f / gnorm[min(row, g_hi - 1)]);
            if (valid && v < best[r]) best[r] = v;
        }
    }
    float m = fminf(fminf(best[0], best[1]), fminf(best[2], best[3]));
    m = fminf(m, __shfl_xor(m, 16));
    m = fminf(m, __shfl_xor(m, 32));
    if (g == 0) s_min[w][i] = m;
    __syncthreads();
    if (w == 0 && g == 0 && n0 + i < N)
        out_row[n0 + i] = (double)fminf(fminf(s_min[0][i], s_min[1][i]), fminf(s_min[2][i], s_min[3][i]));
}

// r03: the same arithmetic with the loops swapped for the tracker bank -- workgroup = ONE track, every wavefront owns its own 16-detection tiles
// (n0 = 16 w, 16 (w + 4), ...) and walks ALL gallery rows of the track. cosine_gallery_tile's (track, 16 detections) workgroups re-read the
// track's gallery once per detection tile -- 7 times at 100 detections -- and consecutive workgroups go to different XCDs, so none of the
// re-reads hit an L2: a bank of 64 streams moved 9 GB per frame-launch (1.8 ms). Here the four wavefronts of the workgroup read the same rows at
// about the same time (vector-L1 hits), a wavefront's second tile re-reads what its compute unit's L2 just delivered, and the gallery crosses
// HBM once. The minimum over a track's rows is taken over the same set of bit-identical dot products, so results are unchanged bit for bit.
template <int DS>
__device__ __forceinline__ void cosine_gallery_track(const float *__restrict__ gallery, int g_lo, int g_hi, const float *__restrict__ gnorm,
                                                     const float *__restrict__ dets, int N, const float *__restrict__ dnorm, double *__restrict__ out_row)
{
    constexpr int D = DS * 16;
    constexpr int GS = DS >= 16 ? 8 : DS / 2;
    constexpr int GROUPS = DS / GS;               // even by construction
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    for (int n0 = 16 * w; n0 < N; n0 += 16 * NWAVES) {
        const int dn = min(n0 + i, N - 1);
        float4 breg[DS];
        {
            const float4 *drow = reinterpret_cast<const float4 *>(dets + (size_t)dn * D) + g;
            const float nd = dnorm[dn];
            const float rnd = 1.f / nd;
#pragma unroll
            for (int s = 0; s < DS; ++s) { float4 b = drow[s * 4]; b.x *= rnd; b.y *= rnd; b.z *= rnd; b.w *= rnd; breg[s] = b; }
        }
        float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        float4 A0[GS], A1[GS];
        auto load = [&](float4 (&buf)[GS], int c0, int grp) {
            const int gr = min(c0 + i, g_hi - 1);
            const float4 *grow = reinterpret_cast<const float4 *>(gallery + (size_t)gr * D) + g + grp * GS * 4;
#pragma unroll
            for (int s = 0; s < GS; ++s) buf[s] = grow[s * 4];
        };
        load(A0, g_lo, 0);
        for (int c0 = g_lo; c0 < g_hi; c0 += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int grp = 0; grp < GROUPS; ++grp) {
                if (grp + 1 < GROUPS) { if (grp & 1) load(A0, c0, grp + 1); else load(A1, c0, grp + 1); }
                else if (c0 + 16 < g_hi) { load(A0, c0 + 16, 0); }
#pragma unroll
                for (int s = 0; s < GS; ++s) {
                    const float4 a = (grp & 1) ? A1[s] : A0[s];
                    const float4 b = breg[grp * GS + s];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = c0 + g * 4 + r;
                const bool valid = row < g_hi;
                const float v = 1.f - acc[r] * (1.f / gnorm[min(row, g_hi - 1)]);
                if (valid && v < best[r]) best[r] = v;
            }
        }
        float m = fminf(fminf(best[0], best[1]), fminf(best[2], best[3]));
        m = fminf(m, __shfl_xor(m, 16));
        m = fminf(m, __shfl_xor(m, 32));
        if (g == 0 && n0 + i < N) out_row[n0 + i] = (double)m;
    }
}

}  // namespace
