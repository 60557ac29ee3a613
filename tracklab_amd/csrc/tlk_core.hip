// tlk_core.hip -- error plumbing + stateless kernels (similarity matrices, batched LSA).
#include "tlk_common.hpp"
#include "tlk_pyset.hpp"

namespace tlk {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace tlk

using namespace tlk;

extern "C" const char *tlk_last_error(void) { return tlk::g_err.c_str(); }
extern "C" int tlk_version(void) { return 100; }
extern "C" int tlk_device_count(int *count)
{
    if (!count) return fail(TLK_EINVAL, "tlk_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(TLK_ENODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return TLK_OK;
}

// ---------------------------------------------------------------------------------------------
// Similarity matrix: one thread per (i, j); a wavefront covers 64 consecutive j of one row so the
// b2 reads and the out writes are coalesced; b1[i] is a wave-uniform (scalar) load.
// Algorithmic bytes: (n + m) * 32 read + n*m*8 written -> HBM/launch bound at tracker sizes.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) iou_matrix_kernel(int variant, const double *__restrict__ b1, int n,
                                                           const double *__restrict__ b2, int m,
                                                           double *__restrict__ out)
{
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= m || i >= n) return;
    double a[4] = {b1[4 * i], b1[4 * i + 1], b1[4 * i + 2], b1[4 * i + 3]};
    double b[4] = {b2[4 * j], b2[4 * j + 1], b2[4 * j + 2], b2[4 * j + 3]};
    out[(size_t)i * m + j] = box_similarity(variant, a, b);
}

// ct_dist rescale (association.py:169-171): d/max, then max' - d. Single block; matrices are tiny.
__global__ void __launch_bounds__(BLOCK) ct_rescale_kernel(double *out, int total)
{
    __shared__ double s_red[NWAVES];
    double mx = -INFINITY;
    bool nan = false;
    for (int k = threadIdx.x; k < total; k += BLOCK) { double v = out[k]; nan |= (v != v); mx = v > mx ? v : mx; }
    double m1 = block_max_nan(nan ? NAN : mx, true, s_red);
    double mx2 = -INFINITY;
    nan = false;
    for (int k = threadIdx.x; k < total; k += BLOCK) { double v = out[k] / m1; out[k] = v; nan |= (v != v); mx2 = v > mx2 ? v : mx2; }
    double m2 = block_max_nan(nan ? NAN : mx2, true, s_red);
    for (int k = threadIdx.x; k < total; k += BLOCK) out[k] = m2 - out[k];
}

extern "C" int tlk_iou_matrix_f64(int variant, const double *b1, int n, const double *b2, int m, double *out,
                                  void *hip_stream)
{
    if (variant < TLK_IOU || variant > TLK_CT) return fail(TLK_EINVAL, "tlk_iou_matrix_f64: bad variant");
    if (n < 0 || m < 0) return fail(TLK_EINVAL, "tlk_iou_matrix_f64: negative size");
    if (n == 0 || m == 0) return TLK_OK;
    if (!b1 || !b2 || !out) return fail(TLK_EINVAL, "tlk_iou_matrix_f64: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    dim3 grid((m + BLOCK - 1) / BLOCK, n);
    hipLaunchKernelGGL(iou_matrix_kernel, grid, dim3(BLOCK), 0, st, variant, b1, n, b2, m, out);
    if (variant == TLK_CT) hipLaunchKernelGGL(ct_rescale_kernel, dim3(1), dim3(BLOCK), 0, st, out, n * m);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// ---------------------------------------------------------------------------------------------
// Batched LSA: one wavefront per problem, 4 problems per 256-thread workgroup, work arrays in LDS,
// cost rows read from HBM/L2 (each row scan is one coalesced 64-lane read).
// ---------------------------------------------------------------------------------------------
static int g_debug_hop_limit = 0;       // tlk_debug_lsa_hop_limit

__global__ void __launch_bounds__(BLOCK) lsa_kernel(const double *__restrict__ cost, int batch, int nr, int nc,
                                                    int *__restrict__ rows, int *__restrict__ cols,
                                                    int *__restrict__ n_pairs, int maxdim, int hop_limit)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int prob = blockIdx.x * NWAVES + w;
    if (prob >= batch) return;                    // whole wave exits together; no block barriers below
    const size_t per_wave = (size_t)maxdim * (3 * sizeof(double) + 4 * sizeof(int) + 2);
    unsigned char *base = smem + (size_t)w * ((per_wave + 15) & ~(size_t)15);
    LsaWork W;
    W.u = (double *)base;
    W.v = W.u + maxdim;
    W.spc = W.v + maxdim;
    W.path = (int *)(W.spc + maxdim);
    W.row4col = W.path + maxdim;
    W.remaining = W.row4col + maxdim;
    W.col4row = W.remaining + maxdim;
    W.SR = (unsigned char *)(W.col4row + maxdim);
    W.SC = W.SR + maxdim;
    W.hop_limit = hop_limit;
    const double *c = cost + (size_t)prob * nr * nc;
    // scipy: "matrix contains invalid numeric entries" on NaN / -inf
    bool bad = false;
    for (int k = lane; k < nr * nc; k += WAVE) { double v = c[k]; bad |= (v != v) || (v == -INFINITY); }
    const int k = nr < nc ? nr : nc;
    int *ro = rows + (size_t)prob * k, *co = cols + (size_t)prob * k;
    if (__ballot(bad)) { if (lane == 0) n_pairs[prob] = -2; return; }
    const int np = wave_lsa(c, nr, nc, (size_t)nc, (size_t)1, W, ro, co);
    if (lane == 0) n_pairs[prob] = np;
}

extern "C" int tlk_lsa_f64(const double *cost, int batch, int nr, int nc, int32_t *rows, int32_t *cols,
                           int32_t *n_pairs, void *hip_stream)
{
    if (batch < 0 || nr < 0 || nc < 0) return fail(TLK_EINVAL, "tlk_lsa_f64: negative size");
    if (batch == 0) return TLK_OK;
    if (!n_pairs) return fail(TLK_EINVAL, "tlk_lsa_f64: null n_pairs");
    hipStream_t st = (hipStream_t)hip_stream;
    if (nr == 0 || nc == 0) { TLK_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t) * batch, st)); return TLK_OK; }
    if (!cost || !rows || !cols) return fail(TLK_EINVAL, "tlk_lsa_f64: null pointer");
    const int maxdim = nr > nc ? nr : nc;
    const size_t per_wave = ((size_t)maxdim * (3 * sizeof(double) + 4 * sizeof(int) + 2) + 15) & ~(size_t)15;
    const size_t smem = per_wave * NWAVES;
    if (smem > 160 * 1024) return fail(TLK_ECAPACITY, "tlk_lsa_f64: problem too large for LDS work arrays");
    TLK_HIP(hipFuncSetAttribute((const void *)lsa_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(lsa_kernel, dim3((batch + NWAVES - 1) / NWAVES), dim3(BLOCK), smem, st, cost, batch, nr, nc,
                       rows, cols, n_pairs, maxdim, g_debug_hop_limit);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

// Test hook (tests/test_gpu_kernels.py): cap the augmenting-path walk of tlk_lsa_f64's solver at `hops` steps (0 = the natural bound, the
// number of rows) so that the bounded-loop exit -- n_pairs = -3, TLK_EINTERNAL in the tracker banks -- can be exercised on purpose.
extern "C" int tlk_debug_lsa_hop_limit(int hops)
{
    if (hops < 0) return fail(TLK_EINVAL, "tlk_debug_lsa_hop_limit: hops >= 0");
    g_debug_hop_limit = hops;
    return TLK_OK;
}

// ---------------------------------------------------------------------------------------------
// lap.lapjv(cost, extend_cost=True, cost_limit=L) as ByteTrack calls it (byte_track/matching.py:37-48): the (nr x nc) cost is
// embedded in an (nr+nc)^2 problem -- every other entry cost_limit/2, the lower-right block 0 -- and solved as a square LSA; a
// row stays unmatched (x = -1) when it is assigned to a padding column, i.e. when no pair cheaper than cost_limit exists for it.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) lapjv_extend_kernel(const double *__restrict__ cost, int batch, int nr, int nc, double half_limit,
                                                             double *__restrict__ ext)
{
    const int n = nr + nc;
    const long long tot = (long long)batch * n * n;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
        const int b = (int)(e / ((long long)n * n));
        const int r = (int)((e - (long long)b * n * n) / n), c = (int)(e - (long long)b * n * n - (long long)r * n);
        double v = half_limit;
        if (r >= nr && c >= nc) v = 0.0;
        else if (r < nr && c < nc) v = cost[((size_t)b * nr + r) * nc + c];
        ext[e] = v;
    }
}
__global__ void __launch_bounds__(BLOCK) lapjv_unpack_kernel(const int *__restrict__ rows, const int *__restrict__ cols, const int *__restrict__ np,
                                                             int batch, int nr, int nc, int *__restrict__ x, int *__restrict__ y)
{
    const int b = blockIdx.x, n = nr + nc;
    for (int i = threadIdx.x; i < nr; i += BLOCK) x[(size_t)b * nr + i] = -1;
    for (int j = threadIdx.x; j < nc; j += BLOCK) y[(size_t)b * nc + j] = -1;
    __syncthreads();
    if (np[b] < 0) return;
    for (int k = threadIdx.x; k < n; k += BLOCK) {
        const int r = rows[(size_t)b * n + k], c = cols[(size_t)b * n + k];
        if (r < nr && c < nc) { x[(size_t)b * nr + r] = c; y[(size_t)b * nc + c] = r; }
    }
}

extern "C" int tlk_lsa_lapjv_limit_f64(const double *cost_dev, int batch, int nr, int nc, double cost_limit, int32_t *x_dev, int32_t *y_dev,
                                       void *hip_stream)
{
    if (batch < 0 || nr < 0 || nc < 0) return fail(TLK_EINVAL, "tlk_lsa_lapjv_limit_f64: negative size");
    if (!(cost_limit < INFINITY) || cost_limit != cost_limit) return fail(TLK_EINVAL, "tlk_lsa_lapjv_limit_f64: cost_limit must be finite");
    if (batch == 0 || (nr == 0 && nc == 0)) return TLK_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    if (nr == 0 || nc == 0) {                      // matching.py:38-39: nothing to match
        if (nr && x_dev) TLK_HIP(hipMemsetAsync(x_dev, 0xff, sizeof(int32_t) * (size_t)batch * nr, st));
        if (nc && y_dev) TLK_HIP(hipMemsetAsync(y_dev, 0xff, sizeof(int32_t) * (size_t)batch * nc, st));
        return TLK_OK;
    }
    if (!cost_dev || !x_dev || !y_dev) return fail(TLK_EINVAL, "tlk_lsa_lapjv_limit_f64: null pointer");
    const int n = nr + nc;
    double *ext = nullptr; int *rows = nullptr, *cols = nullptr, *np = nullptr;
    TLK_HIP(hipMallocAsync((void **)&ext, sizeof(double) * (size_t)batch * n * n, st));
    TLK_HIP(hipMallocAsync((void **)&rows, sizeof(int) * (size_t)batch * n * 2 + sizeof(int) * batch, st));
    cols = rows + (size_t)batch * n; np = cols + (size_t)batch * n;
    const long long tot = (long long)batch * n * n;
    hipLaunchKernelGGL(lapjv_extend_kernel, dim3((unsigned)((tot + BLOCK - 1) / BLOCK < 65535 ? (tot + BLOCK - 1) / BLOCK : 65535)), dim3(BLOCK), 0, st,
                       cost_dev, batch, nr, nc, cost_limit / 2., ext);
    int rc = tlk_lsa_f64(ext, batch, n, n, rows, cols, np, hip_stream);
    if (rc == TLK_OK) {
        hipLaunchKernelGGL(lapjv_unpack_kernel, dim3(batch), dim3(BLOCK), 0, st, (const int *)rows, (const int *)cols, (const int *)np, batch, nr, nc,
                           x_dev, y_dev);
        if (hipGetLastError() != hipSuccess) rc = fail(TLK_EHIP, "tlk_lsa_lapjv_limit_f64: launch failed");
    }
    hipFreeAsync(ext, st); hipFreeAsync(rows, st);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// list(set(a) - set(b)) in CPython 3.10's iteration order (tlk_pyset.hpp): the stateless form of what the StrongSORT-family
// association kernels run after their appearance stage (sort/linear_assignment.py:126-128 of both plugins).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) pyset_difference_kernel(const int *__restrict__ a, int na, const int *__restrict__ b, int nb, int key_cap,
                                                                 int *__restrict__ in_b, int *__restrict__ ws, unsigned cap, int force_table,
                                                                 int *__restrict__ out, int *__restrict__ n_out)
{
    __shared__ int s_scan[NWAVES];
    for (int k = threadIdx.x; k < key_cap; k += BLOCK) in_b[k] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += BLOCK) in_b[b[k]] = 1;
    __syncthreads();
    const int nu = block_compact(na, [&](int r) { return in_b[a[r]] == 0; }, [&](int r, int pos) { out[pos] = a[r]; }, s_scan);
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = nu;
        if (na > 0 && (force_table || !pyset::ascending_is_exact(na, a[na - 1], nb, out, nu)))
            n = pyset::difference_order_serial(a, na, in_b, nb, out, ws, cap, force_table > 1 ? -1 : nu);
        *n_out = n;
    }
}

extern "C" int tlk_pyset_difference_order(const int32_t *a, int na, const int32_t *b, int nb, int32_t *out, int32_t *n_out, int force_table)
{
    if (na < 0 || nb < 0 || (na && !a) || (nb && !b) || !out || !n_out) return fail(TLK_EINVAL, "tlk_pyset_difference_order: bad arguments");
    int key_cap = 1;
    for (int i = 0; i < na; ++i) {
        if (a[i] < 0 || (i && a[i] <= a[i - 1])) return fail(TLK_EINVAL, "tlk_pyset_difference_order: a must be ascending, distinct and non-negative");
        key_cap = a[i] + 1;
    }
    for (int i = 0; i < nb; ++i) {
        bool found = false;
        for (int k = 0; k < na && !found; ++k) found = a[k] == b[i];
        if (!found) return fail(TLK_EINVAL, "tlk_pyset_difference_order: b must be a subset of a");
        for (int k = 0; k < i; ++k) if (b[k] == b[i]) return fail(TLK_EINVAL, "tlk_pyset_difference_order: b must hold distinct keys");
    }
    *n_out = 0;
    if (na == 0) return TLK_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_pyset_difference_order: no HIP device (libtlk has no CPU fallback)");
    const unsigned cap = pyset::table_capacity((unsigned)na);
    int *d = nullptr;
    const size_t ints = (size_t)na * 2 + nb + key_cap + 4 * (size_t)cap + 1;
    TLK_HIP(hipMalloc((void **)&d, sizeof(int) * ints));
    int *d_a = d, *d_b = d_a + na, *d_out = d_b + nb, *d_inb = d_out + na, *d_ws = d_inb + key_cap, *d_n = d_ws + 4 * (size_t)cap;
    hipError_t e = hipMemcpy(d_a, a, sizeof(int) * na, hipMemcpyHostToDevice);
    if (e == hipSuccess && nb) e = hipMemcpy(d_b, b, sizeof(int) * nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pyset_difference_kernel, dim3(1), dim3(BLOCK), 0, 0, (const int *)d_a, na, (const int *)d_b, nb, key_cap, d_inb, d_ws, cap,
                           force_table, d_out, d_n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && *n_out > 0) e = hipMemcpy(out, d_out, sizeof(int) * (size_t)*n_out, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_pyset_difference_order: ") + hipGetErrorString(e));
    return TLK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// non_max_suppression(boxes, max_bbox_overlap, scores) of plugins/track/strong_sort/sort/preprocessing.py:6-73
// (identical in bpbreid_strong_sort/sort/preprocessing.py; dead code in the reference -- the live NMS is the detector's,
// tlk_yolox_decode_nms). One workgroup: (key, index) pairs sorted ascending by a bitonic network in LDS (key = score, or the bottom edge y2
// without scores; equal keys keep ascending index order -- np.argsort's default sort is not stable, the reference's order for ties is
// implementation-defined), then the greedy loop from the top of the order: the pick's overlap (w * h) / area[other] with every box still in the
// list is evaluated by all threads at once. fp64 throughout, operation for operation what the numpy code computes.
namespace {
constexpr int NMS_MAX = 1024;

__global__ void __launch_bounds__(BLOCK) deepsort_nms_kernel(const double *__restrict__ boxes, const double *__restrict__ scores, int n, double thr,
                                                             int *__restrict__ pick, int *__restrict__ n_pick)
{
    __shared__ double s_key[NMS_MAX], s_x2[NMS_MAX], s_y2[NMS_MAX], s_area[NMS_MAX];
    __shared__ int s_idx[NMS_MAX];
    __shared__ unsigned char s_alive[NMS_MAX];
    __shared__ int s_top;
    const int tid = threadIdx.x;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += BLOCK) {
        if (i < n) {
            const double *b = boxes + (size_t)i * 4;
            const double x2 = b[2] + b[0], y2 = b[3] + b[1];
            s_x2[i] = x2; s_y2[i] = y2; s_area[i] = (x2 - b[0] + 1) * (y2 - b[1] + 1);
            s_key[i] = scores ? scores[i] : y2; s_idx[i] = i;
        } else { s_key[i] = INFINITY; s_idx[i] = 0x7fffffff; }                 // padding sorts behind every real entry
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += BLOCK) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const double ka = s_key[i], kb = s_key[p];
                    const int ia = s_idx[i], ib = s_idx[p];
                    const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
                    if (a_gt_b == up) { s_key[i] = kb; s_key[p] = ka; s_idx[i] = ib; s_idx[p] = ia; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += BLOCK) s_alive[i] = 1;
    if (tid == 0) s_top = n - 1;
    __syncthreads();
    int np_ = 0;
    for (;;) {
        const int top = s_top;                                                // position (in the sorted order) of the pick
        if (top < 0) break;
        const int i = s_idx[top];
        if (tid == 0) pick[np_] = i;
        ++np_;
        const double x1i = boxes[(size_t)i * 4], y1i = boxes[(size_t)i * 4 + 1], x2i = s_x2[i], y2i = s_y2[i];
        for (int p = tid; p < top; p += BLOCK) {
            if (!s_alive[p]) continue;
            const int j = s_idx[p];
            const double x1j = boxes[(size_t)j * 4], y1j = boxes[(size_t)j * 4 + 1];
            const double xx1 = x1i > x1j ? x1i : x1j, yy1 = y1i > y1j ? y1i : y1j;
            const double xx2 = x2i < s_x2[j] ? x2i : s_x2[j], yy2 = y2i < s_y2[j] ? y2i : s_y2[j];
            double w = xx2 - xx1 + 1, h = yy2 - yy1 + 1;
            w = w > 0 ? w : 0; h = h > 0 ? h : 0;
            if ((w * h) / s_area[j] > thr) s_alive[p] = 0;
        }
        __syncthreads();
        if (tid < WAVE) {                                                     // next pick: the highest position still alive below `top`
            int nxt = -1;
            for (int hi = top - 1; hi >= 0 && nxt < 0; hi -= WAVE) {
                const int p = hi - tid;
                const unsigned long long m = __ballot(p >= 0 && s_alive[p]);
                if (m) nxt = hi - (__ffsll((long long)m) - 1);
            }
            if (tid == 0) s_top = nxt;
        }
        __syncthreads();
    }
    if (tid == 0) *n_pick = np_;
}
}  // namespace

extern "C" int tlk_deepsort_nms_f64(const double *boxes_xywh_dev, const double *scores_dev, int n, double max_bbox_overlap, int32_t *pick_dev, int32_t *n_pick_dev,
                                    void *hip_stream)
{
    if (n < 0 || !pick_dev || !n_pick_dev || (n > 0 && !boxes_xywh_dev)) return fail(TLK_EINVAL, "tlk_deepsort_nms_f64: bad argument");
    if (n > NMS_MAX) return fail(TLK_ECAPACITY, "tlk_deepsort_nms_f64: at most 1024 boxes");
    if (n == 0) { TLK_HIP(hipMemsetAsync(n_pick_dev, 0, sizeof(int32_t), (hipStream_t)hip_stream)); return TLK_OK; }
    hipLaunchKernelGGL(deepsort_nms_kernel, dim3(1), dim3(BLOCK), 0, (hipStream_t)hip_stream, boxes_xywh_dev, scores_dev, n, max_bbox_overlap, pick_dev, n_pick_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
