// tlk_eval.hip -- HOTA of one sequence on the device. Restates HOTA.eval_sequence of the TrackEval copy the reference
// vendors (plugins/eval/PoseTrack21/posetrack21/posetrack21/trackeval/metrics/hota.py:30-155; official path: pip trackeval behind
// tracklab/wrappers/eval/trackeval_evaluator.py:28-110), in the form tracklab_amd/hota.py pins on it (<= 1e-12):
//   1. per frame: IoU similarity of ground-truth and tracker boxes, its row / column sums            (hota_sim_kernel, a workgroup per frame)
//   2. per (gt id, tracker id): the global alignment score from the frame-ordered sum of sim / (rowsum + colsum - sim)
//                                                                                                     (hota_potential_kernel, a thread per pair)
//   3. per frame: Hungarian matching on -(alignment * similarity) -- scipy's linear_sum_assignment, i.e. the wavefront solver the
//      trackers use -- then for each of the 19 thresholds TP / FN / FP, the sum of matched similarities and the (gt, tracker) match
//      counts                                                                                         (hota_match_kernel, a wavefront per frame)
//   4. per threshold: AssA / AssRe / AssPr from the match counts, LocA summed over the frames in frame order   (hota_final_kernel)
// Frames are independent in steps 1 and 3, id pairs in step 2: the whole sequence is four launches. Counters are integer atomics;
// every floating-point sum has a fixed order (frame order per id pair and per threshold, a fixed lane mapping elsewhere), so the
// result is reproducible run to run; it agrees with the numpy restatement to ~1e-15 (summation order of the wide reductions).
#include "tlk_common.hpp"

#include <vector>

using namespace tlk;

namespace {

constexpr int NA = 19;
constexpr double HEPS = 2.220446049250313e-16;                   // np.finfo('float').eps

struct HotaIn {
    const int *gid, *tid; const double *gbox, *tbox; const long long *goff, *toff, *moff;     // per-frame offsets into ids / boxes / matrices
    int T, n_gt, n_tr;
};

__global__ void __launch_bounds__(BLOCK) hota_sim_kernel(HotaIn in, double *__restrict__ sim, double *__restrict__ rsum, double *__restrict__ csum,
                                                         int *__restrict__ posg, int *__restrict__ post, int *__restrict__ gcnt, int *__restrict__ tcnt)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const long long g0 = in.goff[f], t0 = in.toff[f];
    const int g = (int)(in.goff[f + 1] - g0), t = (int)(in.toff[f + 1] - t0);
    double *S = sim + in.moff[f];
    for (int i = tid; i < g; i += BLOCK) { posg[(size_t)f * in.n_gt + in.gid[g0 + i]] = i; atomicAdd(&gcnt[in.gid[g0 + i]], 1); }
    for (int j = tid; j < t; j += BLOCK) { post[(size_t)f * in.n_tr + in.tid[t0 + j]] = j; atomicAdd(&tcnt[in.tid[t0 + j]], 1); }
    for (int e = tid; e < g * t; e += BLOCK) {                    // TrackEval _calculate_box_ious (x0y0x1y1)
        const int i = e / t, j = e - i * t;
        const double *a = in.gbox + (g0 + i) * 4, *b = in.tbox + (t0 + j) * 4;
        const double iw = fmax(fmin(a[2], b[2]) - fmax(a[0], b[0]), 0.0), ih = fmax(fmin(a[3], b[3]) - fmax(a[1], b[1]), 0.0);
        const double inter = iw * ih, uni = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter;
        S[e] = uni > 0 + HEPS ? inter / uni : 0.0;
    }
    __syncthreads();
    for (int i = tid; i < g; i += BLOCK) { double s = 0; for (int j = 0; j < t; ++j) s += S[(size_t)i * t + j]; rsum[g0 + i] = s; }
    for (int j = tid; j < t; j += BLOCK) { double s = 0; for (int i = 0; i < g; ++i) s += S[(size_t)i * t + j]; csum[t0 + j] = s; }
}

__global__ void __launch_bounds__(BLOCK) hota_potential_kernel(HotaIn in, const double *__restrict__ sim, const double *__restrict__ rsum,
                                                               const double *__restrict__ csum, const int *__restrict__ posg, const int *__restrict__ post,
                                                               const int *__restrict__ gcnt, const int *__restrict__ tcnt, double *__restrict__ gas)
{
    const long long p = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= (long long)in.n_gt * in.n_tr) return;
    const int G = (int)(p / in.n_tr), Tq = (int)(p - (long long)G * in.n_tr);
    double pot = 0.0;
    for (int f = 0; f < in.T; ++f) {                              // frame order: the order numpy accumulates potential_matches_count in
        const int i = posg[(size_t)f * in.n_gt + G], j = post[(size_t)f * in.n_tr + Tq];
        if (i < 0 || j < 0) continue;
        const int t = (int)(in.toff[f + 1] - in.toff[f]);
        const double s = sim[in.moff[f] + (size_t)i * t + j];
        const double den = csum[in.toff[f] + j] + rsum[in.goff[f] + i] - s;
        if (den > 0 + HEPS) pot += s / den;
    }
    gas[p] = pot / ((double)gcnt[G] + (double)tcnt[Tq] - pot);
}

__global__ void __launch_bounds__(BLOCK) hota_match_kernel(HotaIn in, const double *__restrict__ sim, const double *__restrict__ gas, double *__restrict__ score,
                                                           int *__restrict__ mrows, int *__restrict__ mcols, unsigned long long *__restrict__ cnt3,
                                                           double *__restrict__ loc, int *__restrict__ matches, int *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * NWAVES + w;
    if (f >= in.T) return;
    const long long g0 = in.goff[f], t0 = in.toff[f];
    const int g = (int)(in.goff[f + 1] - g0), t = (int)(in.toff[f + 1] - t0);
    if (lane < NA) loc[(size_t)f * NA + lane] = 0.0;
    if (g == 0 || t == 0) {
        if (lane < NA) atomicAdd(&cnt3[(g == 0 ? 2 : 1) * NA + lane], (unsigned long long)(g == 0 ? t : g));      // FP += len(t) / FN += len(g), every threshold
        return;
    }
    const int mx = g > t ? g : t;
    if (mx > 512) { if (lane == 0) *err = TLK_ECAPACITY; return; }
    const double *S = sim + in.moff[f];
    double *C = score + in.moff[f];
    for (int e = lane; e < g * t; e += WAVE) {
        const int i = e / t, j = e - i * t;
        C[e] = -(gas[(size_t)in.gid[g0 + i] * in.n_tr + in.tid[t0 + j]] * S[e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
    LsaWork W;
    unsigned char *base = smem + (size_t)w * 512 * (sizeof(double) + sizeof(int));
    W.u = (double *)base; W.col4row = (int *)(base + 512 * sizeof(double));
    W.v = W.spc = nullptr; W.path = W.row4col = W.remaining = nullptr; W.SR = W.SC = nullptr;
    int *ro = mrows + (g0 < t0 ? g0 : t0), *co = mcols + (g0 < t0 ? g0 : t0);        // min(g, t) pairs per frame fit the smaller prefix
    const int np = wave_lsa(C, g, t, (size_t)t, (size_t)1, W, ro, co);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
    // the thresholds are numpy's own (np.arange(0.05, 0.99, 0.05)), stored by the host behind the counters
    const double *alphas = reinterpret_cast<const double *>(cnt3 + 3 * NA);
    for (int a = 0; a < NA; ++a) {
        const double thr = alphas[a] - HEPS;
        int n = 0; double s = 0.0;
        for (int k = lane; k < np; k += WAVE) {
            const int r = ro[k], c = co[k];
            const double v = S[(size_t)r * t + c];
            if (v >= thr) { ++n; s += v; atomicAdd(&matches[((size_t)a * in.n_gt + in.gid[g0 + r]) * in.n_tr + in.tid[t0 + c]], 1); }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { n += __shfl_xor(n, off); s += __shfl_xor(s, off); }
        if (lane == 0) {
            atomicAdd(&cnt3[a], (unsigned long long)n); atomicAdd(&cnt3[NA + a], (unsigned long long)(g - n)); atomicAdd(&cnt3[2 * NA + a], (unsigned long long)(t - n));
            loc[(size_t)f * NA + a] = s;
        }
    }
}

__global__ void __launch_bounds__(BLOCK) hota_final_kernel(HotaIn in, const int *__restrict__ matches, const int *__restrict__ gcnt, const int *__restrict__ tcnt,
                                                           const unsigned long long *__restrict__ cnt3, const double *__restrict__ loc, double *__restrict__ out)
{
    __shared__ double s_red[NWAVES][3];
    const int a = blockIdx.x, tid = threadIdx.x;
    double s[3] = {0, 0, 0};
    const long long np = (long long)in.n_gt * in.n_tr;
    for (long long p = tid; p < np; p += BLOCK) {
        const int mc = matches[(size_t)a * np + p];
        if (!mc) continue;
        const int G = (int)(p / in.n_tr), Tq = (int)(p - (long long)G * in.n_tr);
        const double m = (double)mc, gc = (double)gcnt[G], tc = (double)tcnt[Tq];
        s[0] += m * (m / fmax(1.0, gc + tc - m)); s[1] += m * (m / fmax(1.0, gc)); s[2] += m * (m / fmax(1.0, tc));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
    }
    if ((tid & 63) == 0) for (int k = 0; k < 3; ++k) s_red[tid >> 6][k] = s[k];
    __syncthreads();
    if (tid == 0) {
        double tot[3] = {0, 0, 0};
        for (int q = 0; q < NWAVES; ++q) for (int k = 0; k < 3; ++k) tot[k] += s_red[q][k];
        const double tp = (double)cnt3[a], tpm = fmax(1.0, tp);
        double l = 0.0;
        for (int f = 0; f < in.T; ++f) l += loc[(size_t)f * NA + a];
        out[a] = tp; out[NA + a] = (double)cnt3[NA + a]; out[2 * NA + a] = (double)cnt3[2 * NA + a]; out[3 * NA + a] = l;
        out[4 * NA + a] = tot[0] / tpm; out[5 * NA + a] = tot[1] / tpm; out[6 * NA + a] = tot[2] / tpm;
    }
}

}  // namespace

// HOST buffers. gt_ids / tr_ids: per-frame ids re-labelled 0..n-1 (TrackEval's preprocessing), concatenated; *_ltrb (., 4) float64
// x0 y0 x1 y1; *_off (n_frames + 1) offsets of the frames; alphas19: the thresholds (np.arange(0.05, 0.99, 0.05), passed in so that they
// are numpy's own bits). stats (7, 19): HOTA_TP, HOTA_FN, HOTA_FP, LocA sum, AssA, AssRe, AssPr -- the fields of hota.hota_sequence.
extern "C" int tlk_hota_sequence_f64(const int32_t *gt_ids, const double *gt_ltrb, const int64_t *gt_off, const int32_t *tr_ids, const double *tr_ltrb,
                                     const int64_t *tr_off, int n_frames, int n_gt, int n_tr, const double *alphas19, double *stats)
{
    if (n_frames < 0 || n_gt < 0 || n_tr < 0 || !gt_off || !tr_off || !alphas19 || !stats) return fail(TLK_EINVAL, "tlk_hota_sequence_f64: bad argument");
    const long long ng = n_frames ? gt_off[n_frames] : 0, nt = n_frames ? tr_off[n_frames] : 0;
    for (int k = 0; k < 7 * NA; ++k) stats[k] = 0.0;
    if (nt == 0) { for (int a = 0; a < NA; ++a) stats[NA + a] = (double)ng; return TLK_OK; }          // hota.py:51-56
    if (ng == 0) { for (int a = 0; a < NA; ++a) stats[2 * NA + a] = (double)nt; return TLK_OK; }
    if (!gt_ids || !tr_ids || !gt_ltrb || !tr_ltrb) return fail(TLK_EINVAL, "tlk_hota_sequence_f64: null pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_hota_sequence_f64: no HIP device (libtlk has no CPU fallback)");
    std::vector<long long> moff(n_frames + 1, 0);
    for (int f = 0; f < n_frames; ++f) {
        const long long g = gt_off[f + 1] - gt_off[f], t = tr_off[f + 1] - tr_off[f];
        if (g < 0 || t < 0) return fail(TLK_EINVAL, "tlk_hota_sequence_f64: offsets must ascend");
        if (g > 512 || t > 512) return fail(TLK_ECAPACITY, "tlk_hota_sequence_f64: at most 512 boxes per frame and side");
        moff[f + 1] = moff[f] + g * t;
    }
    const long long nm = moff[n_frames] ? moff[n_frames] : 1, npair = (long long)n_gt * n_tr;
    // one allocation, carved
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_gid = carve(sizeof(int) * ng), o_tid = carve(sizeof(int) * nt), o_gb = carve(sizeof(double) * 4 * ng), o_tb = carve(sizeof(double) * 4 * nt);
    const size_t o_goff = carve(sizeof(long long) * (n_frames + 1)), o_toff = carve(sizeof(long long) * (n_frames + 1)), o_moff = carve(sizeof(long long) * (n_frames + 1));
    const size_t o_sim = carve(sizeof(double) * nm), o_score = carve(sizeof(double) * nm), o_rs = carve(sizeof(double) * ng), o_cs = carve(sizeof(double) * nt);
    const size_t o_posg = carve(sizeof(int) * (size_t)n_frames * n_gt), o_post = carve(sizeof(int) * (size_t)n_frames * n_tr);
    const size_t o_gcnt = carve(sizeof(int) * n_gt), o_tcnt = carve(sizeof(int) * n_tr), o_gas = carve(sizeof(double) * npair);
    const size_t o_mr = carve(sizeof(int) * (ng < nt ? ng : nt) + 64), o_mc = carve(sizeof(int) * (ng < nt ? ng : nt) + 64);
    const size_t o_cnt3 = carve(sizeof(unsigned long long) * 3 * NA + sizeof(double) * NA), o_loc = carve(sizeof(double) * (size_t)n_frames * NA);
    const size_t o_match = carve(sizeof(int) * NA * (size_t)npair), o_err = carve(sizeof(int)), o_out = carve(sizeof(double) * 7 * NA);
    unsigned char *d = nullptr;
    TLK_HIP(hipMalloc((void **)&d, off));
    hipError_t e = hipMemset(d + o_posg, 0xff, sizeof(int) * (size_t)n_frames * n_gt);
    if (e == hipSuccess) e = hipMemset(d + o_post, 0xff, sizeof(int) * (size_t)n_frames * n_tr);
    if (e == hipSuccess) e = hipMemset(d + o_gcnt, 0, sizeof(int) * n_gt);
    if (e == hipSuccess) e = hipMemset(d + o_tcnt, 0, sizeof(int) * n_tr);
    if (e == hipSuccess) e = hipMemset(d + o_cnt3, 0, sizeof(unsigned long long) * 3 * NA);
    if (e == hipSuccess) e = hipMemset(d + o_match, 0, sizeof(int) * NA * (size_t)npair);
    if (e == hipSuccess) e = hipMemset(d + o_err, 0, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d + o_gid, gt_ids, sizeof(int) * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_tid, tr_ids, sizeof(int) * nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_gb, gt_ltrb, sizeof(double) * 4 * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_tb, tr_ltrb, sizeof(double) * 4 * nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_goff, gt_off, sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_toff, tr_off, sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_moff, moff.data(), sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_cnt3 + sizeof(unsigned long long) * 3 * NA, alphas19, sizeof(double) * NA, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); return fail(TLK_EHIP, std::string("tlk_hota_sequence_f64: ") + hipGetErrorString(e)); }
    HotaIn in{(const int *)(d + o_gid), (const int *)(d + o_tid), (const double *)(d + o_gb), (const double *)(d + o_tb), (const long long *)(d + o_goff),
              (const long long *)(d + o_toff), (const long long *)(d + o_moff), n_frames, n_gt, n_tr};
    hipLaunchKernelGGL(hota_sim_kernel, dim3(n_frames), dim3(BLOCK), 0, 0, in, (double *)(d + o_sim), (double *)(d + o_rs), (double *)(d + o_cs), (int *)(d + o_posg),
                       (int *)(d + o_post), (int *)(d + o_gcnt), (int *)(d + o_tcnt));
    hipLaunchKernelGGL(hota_potential_kernel, dim3((unsigned)((npair + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, in, (const double *)(d + o_sim), (const double *)(d + o_rs),
                       (const double *)(d + o_cs), (const int *)(d + o_posg), (const int *)(d + o_post), (const int *)(d + o_gcnt), (const int *)(d + o_tcnt), (double *)(d + o_gas));
    hipLaunchKernelGGL(hota_match_kernel, dim3((n_frames + NWAVES - 1) / NWAVES), dim3(BLOCK), NWAVES * 512 * (sizeof(double) + sizeof(int)), 0, in, (const double *)(d + o_sim),
                       (const double *)(d + o_gas), (double *)(d + o_score), (int *)(d + o_mr), (int *)(d + o_mc), (unsigned long long *)(d + o_cnt3), (double *)(d + o_loc),
                       (int *)(d + o_match), (int *)(d + o_err));
    hipLaunchKernelGGL(hota_final_kernel, dim3(NA), dim3(BLOCK), 0, 0, in, (const int *)(d + o_match), (const int *)(d + o_gcnt), (const int *)(d + o_tcnt),
                       (const unsigned long long *)(d + o_cnt3), (const double *)(d + o_loc), (double *)(d + o_out));
    e = hipGetLastError();
    int err = 0;
    if (e == hipSuccess) e = hipMemcpy(&err, d + o_err, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(stats, d + o_out, sizeof(double) * 7 * NA, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_hota_sequence_f64: ") + hipGetErrorString(e));
    if (err) return fail(err, "tlk_hota_sequence_f64: a frame exceeds the solver's capacity");
    return TLK_OK;
}
