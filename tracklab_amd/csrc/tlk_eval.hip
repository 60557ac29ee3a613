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


// ---------------------------------------------------------------------------------------------------------------------------------
// CLEAR-MOT and ID measures of one sequence (py-motmetrics copy vendored by the reference: posetrack21_mot/motmetrics/mot.py:134-345,
// metrics.py:342-728, distances.py:52-129, lap.py:79-130), in the form tracklab_amd/clearmot.py pins on it (<= 1e-12).
// The event accumulator is SEQUENTIAL in the frames (a frame first carries the previous correspondences forward, then assigns the rest), so
// the sequence is ONE workgroup walking the frames: all threads build the frame's distance matrix and the masked cost matrix, wavefront 0 runs
// the order-dependent parts (carry-forward in object order, scipy's assignment, the pair events in row order with the float64 distance sum in
// exactly the accumulator's order). Afterwards: track ratios / fragmentations per object, and the global ID assignment (one Hungarian
// problem of size n_gt + n_tr on the NaN-edged fp + fn matrix).
struct ClearDev {
    const int *gid, *tid; const double *gbox, *tbox; const long long *goff, *toff;      // ids dense 0..n-1 IN SORTED ORDER of the original ids; boxes ltwh
    int T, n_gt, n_tr; double max_iou;
    double *D, *Cm;                         // frame matrices (512 x 512)
    int *m, *res_m, *last_occ, *last_match, *hyp_hist, *ocs, *hcs, *tps, *hits, *prev_ev, *pend, *frag;
    unsigned char *om, *hm;
    int *mrows, *mcols;
    // ID assignment
    double *fpm, *fnm, *idc; LsaWork idw; int *idr, *idcol;
    double *out;                            // SUM_FIELDS order of tracklab_amd/clearmot.py (19 doubles)
    int *err;
};
enum { C_FRAMES, C_MATCHES, C_SWITCHES, C_TRANSFER, C_ASCEND, C_MIGRATE, C_FP, C_MISSES, C_OBJECTS, C_PREDS, C_UNIQUE, C_MT, C_PT, C_ML, C_FRAG, C_SUMD, C_IDTP, C_IDFP, C_IDFN, C_N };

__device__ __forceinline__ bool finite_d(double v) { return v - v == 0.0; }

// lap.lsa_solve_scipy's cost substitution: non-finite entries -> 2 * min(shape) * (max |finite| + 1) + 1 (all finite: as is; none: zeros)
__device__ void masked_cost(const double *D, double *Cm, int nr, int nc, double *s_red, int *s_flag)
{
    const int tid = threadIdx.x, n = nr * nc;
    double mx = 0.0; int nvalid = 0;
    for (int e = tid; e < n; e += BLOCK) { const double v = D[e]; if (finite_d(v)) { ++nvalid; const double a = fabs(v); mx = a > mx ? a : mx; } }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off); mx = o > mx ? o : mx; nvalid += __shfl_xor(nvalid, off); }
    __syncthreads();
    if ((tid & 63) == 0) { s_red[tid >> 6] = mx; s_flag[tid >> 6] = nvalid; }
    __syncthreads();
    mx = 0.0; nvalid = 0;
    for (int w = 0; w < NWAVES; ++w) { mx = s_red[w] > mx ? s_red[w] : mx; nvalid += s_flag[w]; }
    const double big = 2.0 * (double)(nr < nc ? nr : nc) * (mx + 1.0) + 1.0;
    for (int e = tid; e < n; e += BLOCK) { const double v = D[e]; Cm[e] = nvalid == 0 ? 0.0 : (finite_d(v) ? v : big); }
    __threadfence_block();
    __syncthreads();
}

__global__ void __launch_bounds__(BLOCK) clear_seq_kernel(ClearDev A)
{
    __shared__ double s_u[512];
    __shared__ int s_c4r[512];
    __shared__ double s_red[NWAVES];
    __shared__ int s_flag[NWAVES];
    __shared__ double s_cnt[C_N];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < C_N) s_cnt[tid] = 0.0;
    for (int k = tid; k < A.n_gt; k += BLOCK) { A.m[k] = -1; A.last_occ[k] = -1; A.last_match[k] = -1; A.ocs[k] = 0; A.hits[k] = 0; A.prev_ev[k] = -1; A.pend[k] = 0; A.frag[k] = 0; }
    for (int k = tid; k < A.n_tr; k += BLOCK) { A.res_m[k] = -1; A.hyp_hist[k] = -1; A.hcs[k] = 0; }
    for (size_t k = tid; k < (size_t)A.n_gt * A.n_tr; k += BLOCK) A.tps[k] = 0;
    __threadfence_block();
    __syncthreads();
    LsaWork W;
    W.u = s_u; W.col4row = s_c4r; W.v = W.spc = nullptr; W.path = W.row4col = W.remaining = nullptr; W.SR = W.SC = nullptr;
    for (int f = 0; f < A.T; ++f) {
        const long long g0 = A.goff[f], t0 = A.toff[f];
        const int no = (int)(A.goff[f + 1] - g0), nh = (int)(A.toff[f + 1] - t0);
        if (no > 512 || nh > 512) { if (tid == 0) *A.err = TLK_ECAPACITY; return; }
        const int *oid = A.gid + g0, *hid = A.tid + t0;
        // distances.iou_matrix(max_iou): 1 - IoU of (x, y, w, h) rectangles with numpy's operation order, NaN above max_iou; raw events
        for (int e = tid; e < no * nh; e += BLOCK) {
            const int i = e / nh, j = e - i * nh;
            const double *a = A.gbox + (g0 + i) * 4, *b = A.tbox + (t0 + j) * 4;
            const double ax1 = a[0] + a[2], ay1 = a[1] + a[3], bx1 = b[0] + b[2], by1 = b[1] + b[3];
            double dx = (ax1 < bx1 ? ax1 : bx1) - (a[0] > b[0] ? a[0] : b[0]), dy = (ay1 < by1 ? ay1 : by1) - (a[1] > b[1] ? a[1] : b[1]);
            dx = dx > 0.0 ? dx : 0.0; dy = dy > 0.0 ? dy : 0.0;
            const double iv = dx * dy;
            double aw = ax1 - a[0], ah = ay1 - a[1], bw = bx1 - b[0], bh = by1 - b[1];
            aw = aw > 0.0 ? aw : 0.0; ah = ah > 0.0 ? ah : 0.0; bw = bw > 0.0 ? bw : 0.0; bh = bh > 0.0 ? bh : 0.0;
            const double uv = aw * ah + bw * bh - iv;
            const double iou = iv == 0.0 ? 0.0 : iv / uv;
            double d = 1.0 - iou;
            if (d > A.max_iou) d = __builtin_nan("");
            A.D[e] = d;
            if (finite_d(d)) A.tps[(size_t)oid[i] * A.n_tr + hid[j]] += 1;      // (ids are unique within a frame)
        }
        for (int i = tid; i < no; i += BLOCK) { A.ocs[oid[i]] += 1; A.om[i] = 0; }
        for (int j = tid; j < nh; j += BLOCK) { A.hcs[hid[j]] += 1; A.hm[j] = 0; }
        __threadfence_block();
        __syncthreads();
        if (no > 0 && nh > 0) {
            if (wv == 0) {                                   // 1. carry established correspondences forward, in object order
                double sumd = s_cnt[C_SUMD]; int nmatch = 0;
                for (int i = 0; i < no; ++i) {
                    const int o = oid[i], mo = A.m[o];
                    if (mo < 0) continue;
                    int jf = -1;
                    for (int j0 = 0; j0 < nh && jf < 0; j0 += WAVE) {
                        const int j = j0 + lane;
                        const bool hit = j < nh && !A.hm[j] && hid[j] == mo;
                        const unsigned long long b = __ballot(hit);
                        if (b) jf = j0 + __builtin_ctzll(b);
                    }
                    if (jf < 0) continue;
                    const double d = A.D[(size_t)i * nh + jf];
                    if (!finite_d(d)) continue;
                    if (lane == 0) {
                        A.om[i] = 1; A.hm[jf] = 1;
                        A.last_match[o] = f; A.hyp_hist[hid[jf]] = f;
                        A.hits[o] += 1; A.frag[o] += A.pend[o]; A.pend[o] = 0; A.prev_ev[o] = 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    sumd += d; ++nmatch;
                }
                if (lane == 0) { s_cnt[C_SUMD] = sumd; s_cnt[C_MATCHES] += nmatch; }
            }
            __threadfence_block();
            __syncthreads();
            for (int e = tid; e < no * nh; e += BLOCK) {    // 2. the rest: rows / columns already matched leave the problem
                const int i = e / nh, j = e - i * nh;
                if (A.om[i] || A.hm[j]) A.D[e] = __builtin_nan("");
            }
            __threadfence_block();
            __syncthreads();
            masked_cost(A.D, A.Cm, no, nh, s_red, s_flag);
            if (wv == 0) {
                const int np = wave_lsa(A.Cm, no, nh, (size_t)nh, (size_t)1, W, A.mrows, A.mcols);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {                             // pair events in row order (mot.py:262-300)
                    double sumd = s_cnt[C_SUMD];
                    for (int k = 0; k < np; ++k) {
                        const int i = A.mrows[k], j = A.mcols[k];
                        const double d = A.D[(size_t)i * nh + j];
                        if (!finite_d(d)) continue;
                        const int o = oid[i], h = hid[j];
                        const bool is_switch = A.m[o] >= 0 && A.m[o] != h;                  // max_switch_time = inf
                        if (is_switch && A.hyp_hist[h] < 0) s_cnt[C_ASCEND] += 1;
                        if (A.res_m[h] >= 0 && A.res_m[h] != o) { if (A.last_match[o] < 0) s_cnt[C_MIGRATE] += 1; s_cnt[C_TRANSFER] += 1; }
                        A.hyp_hist[h] = f; A.last_match[o] = f;
                        s_cnt[is_switch ? C_SWITCHES : C_MATCHES] += 1;
                        sumd += d;
                        A.hits[o] += 1; A.frag[o] += A.pend[o]; A.pend[o] = 0; A.prev_ev[o] = 0;
                        A.om[i] = 1; A.hm[j] = 1;
                        A.m[o] = h; A.res_m[h] = o;
                    }
                    s_cnt[C_SUMD] = sumd;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        // 3. misses, 4. false alarms, 5. occurrence state
        int nmiss = 0, nfp = 0;
        for (int i = tid; i < no; i += BLOCK) {
            const int o = oid[i];
            if (!A.om[i]) { ++nmiss; if (A.prev_ev[o] == 0) A.pend[o] += 1; A.prev_ev[o] = 1; }
            A.last_occ[o] = f;
        }
        for (int j = tid; j < nh; j += BLOCK) nfp += A.hm[j] ? 0 : 1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { nmiss += __shfl_xor(nmiss, off); nfp += __shfl_xor(nfp, off); }
        if (lane == 0) { s_flag[wv] = nmiss; s_red[wv] = (double)nfp; }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < NWAVES; ++w) { s_cnt[C_MISSES] += s_flag[w]; s_cnt[C_FP] += s_red[w]; }
            s_cnt[C_FRAMES] += 1;
        }
        __syncthreads();
    }
    // ---- per-object measures (metrics.py:455-506): tracked ratio classes, fragmentations; totals
    {
        double v[6] = {0, 0, 0, 0, 0, 0};                   // objects, unique, mt, pt, ml, frag
        for (int o = tid; o < A.n_gt; o += BLOCK) {
            const int n = A.ocs[o];
            if (n == 0) continue;
            const double ratio = (double)A.hits[o] / (double)n;
            v[0] += n; v[1] += 1; v[2] += ratio >= 0.8 ? 1 : 0; v[3] += (ratio >= 0.2 && ratio < 0.8) ? 1 : 0; v[4] += ratio < 0.2 ? 1 : 0; v[5] += A.frag[o];
        }
        double np_ = 0;
        for (int h = tid; h < A.n_tr; h += BLOCK) np_ += A.hcs[h];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { for (int q = 0; q < 6; ++q) v[q] += __shfl_xor(v[q], off); np_ += __shfl_xor(np_, off); }
        __syncthreads();
        if (lane == 0) { atomicAdd(&s_cnt[C_OBJECTS], v[0]); atomicAdd(&s_cnt[C_UNIQUE], v[1]); atomicAdd(&s_cnt[C_MT], v[2]); atomicAdd(&s_cnt[C_PT], v[3]);
                         atomicAdd(&s_cnt[C_ML], v[4]); atomicAdd(&s_cnt[C_FRAG], v[5]); atomicAdd(&s_cnt[C_PREDS], np_); }       // (integers: any order is exact)
        __syncthreads();
    }
    // ---- id_global_assignment (metrics.py:610-653): objects / hypotheses that occur, in id order
    {
        // compact the ids that occur (ocs / hcs > 0) -- dense ids are already in sorted order
        // (prefix positions by one thread: a few hundred ids)
        int *oi = A.idr, *hi = A.idcol;                      // reused below as LSA outputs after the matrices are built; positions first
        __shared__ int s_no, s_nh;
        if (tid == 0) {
            int c = 0; for (int o = 0; o < A.n_gt; ++o) { oi[o] = A.ocs[o] > 0 ? c++ : -1; } s_no = c;
            c = 0; for (int h = 0; h < A.n_tr; ++h) { hi[h] = A.hcs[h] > 0 ? c++ : -1; } s_nh = c;
        }
        __threadfence_block();
        __syncthreads();
        const int no = s_no, nh = s_nh, n = no + nh;
        const double nan = __builtin_nan("");
        for (size_t e = tid; e < (size_t)n * n; e += BLOCK) {
            const int r = (int)(e / n), c = (int)(e - (size_t)r * n);
            A.fpm[e] = (r >= no && c < nh) ? nan : 0.0;
            A.fnm[e] = (r < no && c >= nh) ? nan : 0.0;
        }
        __threadfence_block();
        __syncthreads();
        for (int o = tid; o < A.n_gt; o += BLOCK) if (oi[o] >= 0) { const int r = oi[o]; const double oc = A.ocs[o]; for (int c = 0; c < nh; ++c) A.fnm[(size_t)r * n + c] = oc; A.fnm[(size_t)r * n + nh + r] = oc; }
        __threadfence_block();
        __syncthreads();
        for (int h = tid; h < A.n_tr; h += BLOCK) if (hi[h] >= 0) { const int c = hi[h]; const double hc = A.hcs[h]; for (int r = 0; r < no; ++r) A.fpm[(size_t)r * n + c] = hc; A.fpm[(size_t)(c + no) * n + c] = hc; }
        __threadfence_block();
        __syncthreads();
        for (size_t e = tid; e < (size_t)A.n_gt * A.n_tr; e += BLOCK) {
            const int ex = A.tps[e];
            if (!ex) continue;
            const int o = (int)(e / A.n_tr), h = (int)(e - (size_t)o * A.n_tr);
            const size_t k = (size_t)oi[o] * n + hi[h];
            A.fpm[k] -= ex; A.fnm[k] -= ex;
        }
        __threadfence_block();
        __syncthreads();
        for (size_t e = tid; e < (size_t)n * n; e += BLOCK) A.D[e] = A.fpm[e] + A.fnm[e];       // (A.D is sized for it by the host)
        __threadfence_block();
        __syncthreads();
        double idfp = 0.0, idfn = 0.0;
        if (n > 0) {
            masked_cost(A.D, A.idc, n, n, s_red, s_flag);
            if (wv == 0) {
                LsaWork Wg = n <= 512 ? W : A.idw;
                const int np = wave_lsa(A.idc, n, n, (size_t)n, (size_t)1, Wg, A.mrows, A.mcols);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < np; k += WAVE) {
                    const size_t e = (size_t)A.mrows[k] * n + A.mcols[k];
                    if (finite_d(A.D[e])) { idfp += A.fpm[e]; idfn += A.fnm[e]; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { idfp += __shfl_xor(idfp, off); idfn += __shfl_xor(idfn, off); }
                if (lane == 0) { s_cnt[C_IDFP] = idfp; s_cnt[C_IDFN] = idfn; }
            }
        }
        __syncthreads();
        if (tid == 0) s_cnt[C_IDTP] = s_cnt[C_OBJECTS] - s_cnt[C_IDFN];
        __syncthreads();
    }
    if (tid < C_N) A.out[tid] = s_cnt[tid];
}

}  // namespace

// HOST buffers. gt_ids / tr_ids: per-frame ids re-labelled 0..n-1 (TrackEval's preprocessing), concatenated; *_ltrb (., 4) float64
// x0 y0 x1 y1; *_off (n_frames + 1) offsets of the frames; alphas19: the thresholds (np.arange(0.05, 0.99, 0.05), passed in so that they
// are numpy's own bits). stats (7, 19): HOTA_TP, HOTA_FN, HOTA_FP, LocA sum, AssA, AssRe, AssPr -- the fields of hota.hota_sequence.
// match-matrix offsets of the frames (exclusive prefix sum of gt boxes x tracker boxes) when the caller's offsets live in device memory
__global__ void __launch_bounds__(BLOCK) eval_moff_kernel(const long long *__restrict__ goff, const long long *__restrict__ toff, int n_frames, long long *__restrict__ moff)
{
    __shared__ long long s_part[BLOCK];
    const int tid = threadIdx.x, per = (n_frames + BLOCK - 1) / BLOCK, lo = min(n_frames, tid * per), hi = min(n_frames, lo + per);
    long long sum = 0;
    for (int f = lo; f < hi; ++f) sum += (goff[f + 1] - goff[f]) * (toff[f + 1] - toff[f]);
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) { long long run = 0; for (int k = 0; k < BLOCK; ++k) { const long long v = s_part[k]; s_part[k] = run; run += v; } moff[n_frames] = run; }
    __syncthreads();
    long long run = s_part[tid];
    for (int f = lo; f < hi; ++f) { moff[f] = run; run += (goff[f + 1] - goff[f]) * (toff[f + 1] - toff[f]); }
}

// the four kernels on arrays that are ALL in device memory (moff == nullptr: computed here); scratch is one allocation, carved
static int hota_run(const int *gid, const int *tid, const double *gb, const double *tb, const long long *goff, const long long *toff, const long long *moff_in,
                    int n_frames, int n_gt, int n_tr, long long ng, long long nt, long long nm_in, const double *alphas19, double *stats, hipStream_t st, const char *who)
{
    const long long nm = nm_in > 0 ? nm_in : 1, npair = (long long)n_gt * n_tr;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_moff = carve(sizeof(long long) * (n_frames + 1));
    const size_t o_sim = carve(sizeof(double) * nm), o_score = carve(sizeof(double) * nm), o_rs = carve(sizeof(double) * ng), o_cs = carve(sizeof(double) * nt);
    const size_t o_posg = carve(sizeof(int) * (size_t)n_frames * n_gt), o_post = carve(sizeof(int) * (size_t)n_frames * n_tr);
    const size_t o_gcnt = carve(sizeof(int) * n_gt), o_tcnt = carve(sizeof(int) * n_tr), o_gas = carve(sizeof(double) * npair);
    const size_t o_mr = carve(sizeof(int) * (ng < nt ? ng : nt) + 64), o_mc = carve(sizeof(int) * (ng < nt ? ng : nt) + 64);
    const size_t o_cnt3 = carve(sizeof(unsigned long long) * 3 * NA + sizeof(double) * NA), o_loc = carve(sizeof(double) * (size_t)n_frames * NA);
    const size_t o_match = carve(sizeof(int) * NA * (size_t)npair), o_err = carve(sizeof(int)), o_out = carve(sizeof(double) * 7 * NA);
    unsigned char *d = nullptr;
    TLK_HIP(hipMalloc((void **)&d, off));
    hipError_t e = hipMemsetAsync(d + o_posg, 0xff, sizeof(int) * (size_t)n_frames * n_gt, st);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_post, 0xff, sizeof(int) * (size_t)n_frames * n_tr, st);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_gcnt, 0, sizeof(int) * n_gt, st);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_tcnt, 0, sizeof(int) * n_tr, st);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_cnt3, 0, sizeof(unsigned long long) * 3 * NA, st);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_match, 0, sizeof(int) * NA * (size_t)npair, st);
    if (e == hipSuccess) e = hipMemsetAsync(d + o_err, 0, sizeof(int), st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_cnt3 + sizeof(unsigned long long) * 3 * NA, alphas19, sizeof(double) * NA, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        if (moff_in) e = hipMemcpyAsync(d + o_moff, moff_in, sizeof(long long) * (n_frames + 1), hipMemcpyDeviceToDevice, st);
        else hipLaunchKernelGGL(eval_moff_kernel, dim3(1), dim3(BLOCK), 0, st, goff, toff, n_frames, (long long *)(d + o_moff));
    }
    if (e != hipSuccess) { hipFree(d); return fail(TLK_EHIP, std::string(who) + ": " + hipGetErrorString(e)); }
    HotaIn in{gid, tid, gb, tb, goff, toff, (const long long *)(d + o_moff), n_frames, n_gt, n_tr};
    hipLaunchKernelGGL(hota_sim_kernel, dim3(n_frames), dim3(BLOCK), 0, st, in, (double *)(d + o_sim), (double *)(d + o_rs), (double *)(d + o_cs), (int *)(d + o_posg),
                       (int *)(d + o_post), (int *)(d + o_gcnt), (int *)(d + o_tcnt));
    hipLaunchKernelGGL(hota_potential_kernel, dim3((unsigned)((npair + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, in, (const double *)(d + o_sim), (const double *)(d + o_rs),
                       (const double *)(d + o_cs), (const int *)(d + o_posg), (const int *)(d + o_post), (const int *)(d + o_gcnt), (const int *)(d + o_tcnt), (double *)(d + o_gas));
    hipLaunchKernelGGL(hota_match_kernel, dim3((n_frames + NWAVES - 1) / NWAVES), dim3(BLOCK), NWAVES * 512 * (sizeof(double) + sizeof(int)), st, in, (const double *)(d + o_sim),
                       (const double *)(d + o_gas), (double *)(d + o_score), (int *)(d + o_mr), (int *)(d + o_mc), (unsigned long long *)(d + o_cnt3), (double *)(d + o_loc),
                       (int *)(d + o_match), (int *)(d + o_err));
    hipLaunchKernelGGL(hota_final_kernel, dim3(NA), dim3(BLOCK), 0, st, in, (const int *)(d + o_match), (const int *)(d + o_gcnt), (const int *)(d + o_tcnt),
                       (const unsigned long long *)(d + o_cnt3), (const double *)(d + o_loc), (double *)(d + o_out));
    e = hipGetLastError();
    int err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&err, d + o_err, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(stats, d + o_out, sizeof(double) * 7 * NA, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string(who) + ": " + hipGetErrorString(e));
    if (err) return fail(err, std::string(who) + ": a frame exceeds the solver's capacity (512 boxes per frame and side)");
    return TLK_OK;
}

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
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_gid = carve(sizeof(int) * ng), o_tid = carve(sizeof(int) * nt), o_gb = carve(sizeof(double) * 4 * ng), o_tb = carve(sizeof(double) * 4 * nt);
    const size_t o_goff = carve(sizeof(long long) * (n_frames + 1)), o_toff = carve(sizeof(long long) * (n_frames + 1)), o_moff = carve(sizeof(long long) * (n_frames + 1));
    unsigned char *d = nullptr;
    TLK_HIP(hipMalloc((void **)&d, off));
    hipError_t e = hipMemcpy(d + o_gid, gt_ids, sizeof(int) * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_tid, tr_ids, sizeof(int) * nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_gb, gt_ltrb, sizeof(double) * 4 * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_tb, tr_ltrb, sizeof(double) * 4 * nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_goff, gt_off, sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_toff, tr_off, sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_moff, moff.data(), sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); return fail(TLK_EHIP, std::string("tlk_hota_sequence_f64: ") + hipGetErrorString(e)); }
    const int rc = hota_run((const int *)(d + o_gid), (const int *)(d + o_tid), (const double *)(d + o_gb), (const double *)(d + o_tb), (const long long *)(d + o_goff),
                            (const long long *)(d + o_toff), (const long long *)(d + o_moff), n_frames, n_gt, n_tr, ng, nt, moff[n_frames], alphas19, stats, (hipStream_t)0,
                            "tlk_hota_sequence_f64");
    hipFree(d);
    return rc;
}

// The same with every array already in device memory -- the tracker side straight from the engine's HBM-resident per-video table
// (tracklab_amd.evaluate.evaluate_device_log), the ground truth uploaded once by the caller. n_gt_boxes / n_tr_boxes / n_match are host scalars
// (totals; n_match >= sum over the frames of gt boxes x tracker boxes -- it sizes the similarity matrices).
extern "C" int tlk_hota_sequence_dev_f64(const int32_t *gt_ids_dev, const double *gt_ltrb_dev, const int64_t *gt_off_dev, const int32_t *tr_ids_dev,
                                         const double *tr_ltrb_dev, const int64_t *tr_off_dev, int n_frames, int n_gt, int n_tr, int64_t n_gt_boxes,
                                         int64_t n_tr_boxes, int64_t n_match, const double *alphas19, double *stats, void *hip_stream)
{
    if (n_frames < 0 || n_gt < 0 || n_tr < 0 || n_gt_boxes < 0 || n_tr_boxes < 0 || n_match < 0 || !gt_off_dev || !tr_off_dev || !alphas19 || !stats)
        return fail(TLK_EINVAL, "tlk_hota_sequence_dev_f64: bad argument");
    for (int k = 0; k < 7 * NA; ++k) stats[k] = 0.0;
    if (n_tr_boxes == 0) { for (int a = 0; a < NA; ++a) stats[NA + a] = (double)n_gt_boxes; return TLK_OK; }
    if (n_gt_boxes == 0) { for (int a = 0; a < NA; ++a) stats[2 * NA + a] = (double)n_tr_boxes; return TLK_OK; }
    if (!gt_ids_dev || !tr_ids_dev || !gt_ltrb_dev || !tr_ltrb_dev) return fail(TLK_EINVAL, "tlk_hota_sequence_dev_f64: null pointer");
    return hota_run((const int *)gt_ids_dev, (const int *)tr_ids_dev, gt_ltrb_dev, tr_ltrb_dev, (const long long *)gt_off_dev, (const long long *)tr_off_dev, nullptr,
                    n_frames, n_gt, n_tr, n_gt_boxes, n_tr_boxes, n_match, alphas19, stats, (hipStream_t)hip_stream, "tlk_hota_sequence_dev_f64");
}

// CLEAR-MOT + ID counts of one sequence. HOST buffers as tlk_hota_sequence_f64, except: ids dense 0..n-1 in the SORTED order of the original
// ids (np.unique's inverse: the global ID assignment orders its matrix by id), boxes (x, y, w, h). counts19: the SUM_FIELDS of
// tracklab_amd/clearmot.py in that order (what its pack() all-reduces); clearmot.finalize() derives MOTA / MOTP / IDF1 ... from them.
// the accumulator kernel on arrays that are ALL in device memory
static int clear_run(const int *gid, const int *tid, const double *gb, const double *tb, const long long *goff, const long long *toff, int n_frames, int n_gt, int n_tr,
                     double max_iou, double *counts19, hipStream_t st, const char *who)
{
    const size_t n = (size_t)n_gt + n_tr, nn = n * n > 512 * 512 ? n * n : 512 * 512, ngt = n_gt ? n_gt : 1, ntr = n_tr ? n_tr : 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_D = carve(sizeof(double) * nn), o_C = carve(sizeof(double) * nn), o_fp = carve(sizeof(double) * (n * n + 1)), o_fn = carve(sizeof(double) * (n * n + 1));
    const size_t o_g = carve(sizeof(int) * 8 * ngt), o_t = carve(sizeof(int) * 3 * ntr), o_tps = carve(sizeof(int) * ngt * ntr);
    const size_t o_om = carve(512), o_hm = carve(512), o_mr = carve(sizeof(int) * (n + 512)), o_mc = carve(sizeof(int) * (n + 512));
    const size_t o_idr = carve(sizeof(int) * (n + 1)), o_idc = carve(sizeof(int) * (n + 1));
    const size_t o_wu = carve(sizeof(double) * 3 * (n + 1)), o_wi = carve(sizeof(int) * 4 * (n + 1)), o_wb = carve(2 * (n + 1));
    const size_t o_out = carve(sizeof(double) * C_N), o_err = carve(sizeof(int));
    unsigned char *d = nullptr;
    TLK_HIP(hipMalloc((void **)&d, off));
    hipError_t e = hipMemsetAsync(d + o_err, 0, sizeof(int), st);
    if (e != hipSuccess) { hipFree(d); return fail(TLK_EHIP, std::string(who) + ": " + hipGetErrorString(e)); }
    ClearDev A;
    A.gid = gid; A.tid = tid; A.gbox = gb; A.tbox = tb; A.goff = goff; A.toff = toff;
    A.T = n_frames; A.n_gt = n_gt; A.n_tr = n_tr; A.max_iou = max_iou;
    A.D = (double *)(d + o_D); A.Cm = (double *)(d + o_C); A.fpm = (double *)(d + o_fp); A.fnm = (double *)(d + o_fn); A.idc = (double *)(d + o_C);
    int *gi = (int *)(d + o_g), *ti = (int *)(d + o_t);
    A.m = gi; A.last_occ = gi + ngt; A.last_match = gi + 2 * ngt; A.ocs = gi + 3 * ngt; A.hits = gi + 4 * ngt; A.prev_ev = gi + 5 * ngt; A.pend = gi + 6 * ngt; A.frag = gi + 7 * ngt;
    A.res_m = ti; A.hyp_hist = ti + ntr; A.hcs = ti + 2 * ntr;
    A.tps = (int *)(d + o_tps); A.om = d + o_om; A.hm = d + o_hm; A.mrows = (int *)(d + o_mr); A.mcols = (int *)(d + o_mc);
    A.idr = (int *)(d + o_idr); A.idcol = (int *)(d + o_idc);
    double *wu = (double *)(d + o_wu); int *wi = (int *)(d + o_wi);
    A.idw.u = wu; A.idw.v = wu + (n + 1); A.idw.spc = wu + 2 * (n + 1);
    A.idw.path = wi; A.idw.row4col = wi + (n + 1); A.idw.remaining = wi + 2 * (n + 1); A.idw.col4row = wi + 3 * (n + 1);
    A.idw.SR = d + o_wb; A.idw.SC = d + o_wb + (n + 1);
    A.out = (double *)(d + o_out); A.err = (int *)(d + o_err);
    hipLaunchKernelGGL(clear_seq_kernel, dim3(1), dim3(BLOCK), 0, st, A);
    e = hipGetLastError();
    int err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&err, d + o_err, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(counts19, d + o_out, sizeof(double) * C_N, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string(who) + ": " + hipGetErrorString(e));
    if (err) return fail(err, std::string(who) + ": a frame exceeds the solver's capacity (512 boxes per frame and side)");
    return TLK_OK;
}

extern "C" int tlk_clear_sequence_f64(const int32_t *gt_ids, const double *gt_ltwh, const int64_t *gt_off, const int32_t *tr_ids, const double *tr_ltwh,
                                      const int64_t *tr_off, int n_frames, int n_gt, int n_tr, double max_iou, double *counts19)
{
    if (n_frames < 0 || n_gt < 0 || n_tr < 0 || !gt_off || !tr_off || !counts19) return fail(TLK_EINVAL, "tlk_clear_sequence_f64: bad argument");
    const long long ng = n_frames ? gt_off[n_frames] : 0, nt = n_frames ? tr_off[n_frames] : 0;
    if ((ng && (!gt_ids || !gt_ltwh)) || (nt && (!tr_ids || !tr_ltwh))) return fail(TLK_EINVAL, "tlk_clear_sequence_f64: null pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_clear_sequence_f64: no HIP device (libtlk has no CPU fallback)");
    for (int f = 0; f < n_frames; ++f) {
        const long long g = gt_off[f + 1] - gt_off[f], t = tr_off[f + 1] - tr_off[f];
        if (g < 0 || t < 0) return fail(TLK_EINVAL, "tlk_clear_sequence_f64: offsets must ascend");
        if (g > 512 || t > 512) return fail(TLK_ECAPACITY, "tlk_clear_sequence_f64: at most 512 boxes per frame and side");
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_gid = carve(sizeof(int) * (ng + 1)), o_tid = carve(sizeof(int) * (nt + 1)), o_gb = carve(sizeof(double) * 4 * (ng + 1)), o_tb = carve(sizeof(double) * 4 * (nt + 1));
    const size_t o_goff = carve(sizeof(long long) * (n_frames + 1)), o_toff = carve(sizeof(long long) * (n_frames + 1));
    unsigned char *d = nullptr;
    TLK_HIP(hipMalloc((void **)&d, off));
    hipError_t e = hipSuccess;
    if (ng) e = hipMemcpy(d + o_gid, gt_ids, sizeof(int) * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess && nt) e = hipMemcpy(d + o_tid, tr_ids, sizeof(int) * nt, hipMemcpyHostToDevice);
    if (e == hipSuccess && ng) e = hipMemcpy(d + o_gb, gt_ltwh, sizeof(double) * 4 * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess && nt) e = hipMemcpy(d + o_tb, tr_ltwh, sizeof(double) * 4 * nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_goff, gt_off, sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + o_toff, tr_off, sizeof(long long) * (n_frames + 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); return fail(TLK_EHIP, std::string("tlk_clear_sequence_f64: ") + hipGetErrorString(e)); }
    const int rc = clear_run((const int *)(d + o_gid), (const int *)(d + o_tid), (const double *)(d + o_gb), (const double *)(d + o_tb), (const long long *)(d + o_goff),
                             (const long long *)(d + o_toff), n_frames, n_gt, n_tr, max_iou, counts19, (hipStream_t)0, "tlk_clear_sequence_f64");
    hipFree(d);
    return rc;
}

// The same with every array already in device memory (see tlk_hota_sequence_dev_f64). Offsets must ascend; ids dense in the sorted order of the originals.
extern "C" int tlk_clear_sequence_dev_f64(const int32_t *gt_ids_dev, const double *gt_ltwh_dev, const int64_t *gt_off_dev, const int32_t *tr_ids_dev,
                                          const double *tr_ltwh_dev, const int64_t *tr_off_dev, int n_frames, int n_gt, int n_tr, double max_iou, double *counts19,
                                          void *hip_stream)
{
    if (n_frames < 0 || n_gt < 0 || n_tr < 0 || !gt_off_dev || !tr_off_dev || !counts19 || !gt_ids_dev || !tr_ids_dev || !gt_ltwh_dev || !tr_ltwh_dev)
        return fail(TLK_EINVAL, "tlk_clear_sequence_dev_f64: bad argument");
    return clear_run((const int *)gt_ids_dev, (const int *)tr_ids_dev, gt_ltwh_dev, tr_ltwh_dev, (const long long *)gt_off_dev, (const long long *)tr_off_dev, n_frames, n_gt,
                     n_tr, max_iou, counts19, (hipStream_t)hip_stream, "tlk_clear_sequence_dev_f64");
}
