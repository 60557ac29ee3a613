// tlk_pose.hip -- pose-estimator pre/post-processing of config 4 (tracklab/wrappers/pose_estimator/rtmlib_api.py:27-33 ->
// rtmlib.RTMPose(image, bboxes)): per box centre/scale (padding 1.25, aspect fixed to the network input), top-down affine
// crop with cv2.warpAffine's fixed-point bilinear (10-bit coordinates, 1/32-pixel positions, 15-bit weights, constant 0
// border), mean/std normalisation; SimCC arg-max decode back to image coordinates. Third-party arithmetic (rtmlib 0.0.13,
// OpenCV imgwarp.cpp), restated: see oracle/src/pose.c for the statement these kernels are bit-exact to.
//   pose_prep_kernel   thread = box: centre, scale, forward matrix by the 6x6 LU solve of cv2.getAffineTransform, its inverse
//   pose_warp_kernel   thread = 8 output pixels of one crop row: 4 taps per pixel gathered from the frame in HBM (an axis-aligned
//                      affine map: consecutive pixels read consecutive source bytes), LDS table (u8 -> normalised value), 16-B stores
//   simcc_decode_kernel wavefront = (box, keypoint): first-maximum arg-max over the x and y logits, score = min of the maxima
#include "tlk_common.hpp"
#include <hip/hip_fp16.h>

using namespace tlk;

namespace {

constexpr int LAYOUT_NCHW = 0, LAYOUT_NHWC = 1;
struct alignas(2) bf16p_t { unsigned short x; };
template <typename T> __device__ __forceinline__ T pcvt(float v);
template <> __device__ __forceinline__ float pcvt<float>(float v) { return v; }
template <> __device__ __forceinline__ __half pcvt<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ bf16p_t pcvt<bf16p_t>(float v)
{
    unsigned int u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    bf16p_t r; r.x = (unsigned short)(u >> 16); return r;
}
template <typename T, int N> struct alignas(sizeof(T) * N) PPack { T v[N]; };

constexpr int META = 10;        // per box: centre (2), scale (2), inverse warp matrix (6)

__device__ void get_affine(const float (&src)[6], const float (&dst)[6], double (&M)[6])      // cv2.getAffineTransform (cv::solve, LU)
{
    double a[36], b[6];
    for (int i = 0; i < 3; ++i) {
        const int j = i * 12, k = i * 12 + 6;
        a[j] = a[k + 3] = src[i * 2]; a[j + 1] = a[k + 4] = src[i * 2 + 1]; a[j + 2] = a[k + 5] = 1;
        a[j + 3] = a[j + 4] = a[j + 5] = 0; a[k] = a[k + 1] = a[k + 2] = 0;
        b[i * 2] = dst[i * 2]; b[i * 2 + 1] = dst[i * 2 + 1];
    }
    const int n = 6;
    for (int i = 0; i < n; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j) if (fabs(a[j * n + i]) > fabs(a[k * n + i])) k = j;
        if (fabs(a[k * n + i]) < 2.220446049250313e-16 * 100) { for (int q = 0; q < 6; ++q) M[q] = 0; return; }
        if (k != i) {
            for (int j = i; j < n; ++j) { const double t = a[i * n + j]; a[i * n + j] = a[k * n + j]; a[k * n + j] = t; }
            const double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        const double d = -1 / a[i * n + i];
        for (int j = i + 1; j < n; ++j) {
            const double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; ++c) a[j * n + c] += alpha * a[i * n + c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * b[k];
        b[i] = s / a[i * n + i];
    }
    for (int q = 0; q < 6; ++q) M[q] = b[q];
}

__global__ void __launch_bounds__(BLOCK) pose_prep_kernel(const double *__restrict__ boxes, int box_stride, const int *__restrict__ counts,
                                                          int B, int max_n, int in_w, int in_h, double padding, double *__restrict__ meta)
{
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= B * max_n) return;
    const int b = slot / max_n, i = slot - b * max_n;
    double *m = meta + (size_t)slot * META;
    if (i >= counts[b]) { for (int q = 0; q < META; ++q) m[q] = 0.0; return; }
    const double *bx = boxes + (size_t)slot * box_stride;
    // bbox_xyxy2cs + _fix_aspect_ratio
    const double cx = (bx[0] + bx[2]) * 0.5, cy = (bx[1] + bx[3]) * 0.5;
    const double w = (bx[2] - bx[0]) * padding, h = (bx[3] - bx[1]) * padding;
    const double ar = (double)in_w / (double)in_h;
    double sw, sh;
    if (w > h * ar) { sw = w; sh = w / ar; } else { sw = h * ar; sh = h; }
    // get_warp_matrix(center, scale, 0, (in_w, in_h)) on float32 point triplets
    float src[6], dst[6];
    const double sdx = 0.0 * 1.0 - (sw * -0.5) * 0.0, sdy = 0.0 * 0.0 + (sw * -0.5) * 1.0;
    src[0] = (float)cx; src[1] = (float)cy; src[2] = (float)(cx + sdx); src[3] = (float)(cy + sdy);
    { const float d0 = src[0] - src[2], d1 = src[1] - src[3]; src[4] = src[2] + (-d1); src[5] = src[3] + d0; }
    dst[0] = (float)(in_w * 0.5); dst[1] = (float)(in_h * 0.5); dst[2] = (float)(in_w * 0.5 + 0.0); dst[3] = (float)(in_h * 0.5 + in_w * -0.5);
    { const float d0 = dst[0] - dst[2], d1 = dst[1] - dst[3]; dst[4] = dst[2] + (-d1); dst[5] = dst[3] + d0; }
    double M[6];
    get_affine(src, dst, M);
    // cv2.warpAffine without WARP_INVERSE_MAP inverts the matrix first
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    m[0] = cx; m[1] = cy; m[2] = sw; m[3] = sh;
    for (int q = 0; q < 6; ++q) m[4 + q] = M[q];
}

__device__ __forceinline__ int cv_round_d(double v) { return (int)rint(v); }
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(BLOCK) pose_warp_kernel(const unsigned char *__restrict__ frames, int B, int H, int W, const int *__restrict__ counts,
                                                          int max_n, int in_w, int in_h, const double *__restrict__ meta,
                                                          double m0, double m1, double m2, double s0, double s1, double s2, T *__restrict__ out, int swap_rb)
{
    __shared__ float s_lut[3][256];
    const int tid = threadIdx.x;
    {   // (u8 - mean) / std in float64, stored float32 (rtmlib keeps float64 until the cast in inference())
        const double mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
        for (int c = 0; c < 3; ++c) s_lut[c][tid] = (float)(((double)tid - mean[c]) / stdv[c]);
    }
    __syncthreads();
    const int groups = in_w / 8;
    const long long gid = (long long)blockIdx.x * BLOCK + tid;
    const long long per_slot = (long long)groups * in_h;
    if (gid >= per_slot * B * max_n) return;
    const int slot = (int)(gid / per_slot);
    const int rem = (int)(gid - (long long)slot * per_slot);
    const int y = rem / groups, x_base = (rem - y * groups) * 8;
    const int b = slot / max_n, i = slot - b * max_n;
    T px[8][3];
    if (i < counts[b]) {
        const double *M = meta + (size_t)slot * META + 4;
        const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, round_delta = AB_SCALE / 32 / 2;
        const int X0 = cv_round_d((M[1] * y + M[2]) * AB_SCALE) + round_delta, Y0 = cv_round_d((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        const unsigned char *img = frames + (size_t)b * H * W * 3;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int x = x_base + k;
            const int X = (X0 + cv_round_d(M[0] * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
            const int Y = (Y0 + cv_round_d(M[3] * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
            const int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
            const int ax = X & 31, ay = Y & 31;
            // BilinearTab_i: (32-ay)(32-ax), (32-ay)ax, ay(32-ax), ay*ax, each x32; the (0,0) entry is (32767, 0, 0, 1)
            int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
            if ((ax | ay) == 0) { w00 = 32767; w11 = 1; }
            if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) px[k][c] = pcvt<T>(s_lut[c][0]);
                continue;
            }
            const bool x0in = sx >= 0 && sx < W, x1in = sx + 1 >= 0 && sx + 1 < W, y0in = sy >= 0 && sy < H, y1in = sy + 1 >= 0 && sy + 1 < H;
            const unsigned char *p00 = img + ((size_t)(y0in ? sy : 0) * W + (x0in ? sx : 0)) * 3;
            const unsigned char *p01 = img + ((size_t)(y0in ? sy : 0) * W + (x1in ? sx + 1 : 0)) * 3;
            const unsigned char *p10 = img + ((size_t)(y1in ? sy + 1 : 0) * W + (x0in ? sx : 0)) * 3;
            const unsigned char *p11 = img + ((size_t)(y1in ? sy + 1 : 0) * W + (x1in ? sx + 1 : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int v00 = (x0in && y0in) ? p00[c] : 0, v01 = (x1in && y0in) ? p01[c] : 0;
                const int v10 = (x0in && y1in) ? p10[c] : 0, v11 = (x1in && y1in) ? p11[c] : 0;
                int r = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
                r = r < 0 ? 0 : (r > 255 ? 255 : r);
                px[k][c] = pcvt<T>(s_lut[c][r]);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) px[k][c] = pcvt<T>(0.f);
    }
    if (swap_rb) {      // output channel c = source channel 2 - c (the look-up table is already per SOURCE channel)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const T t0 = px[k][0]; px[k][0] = px[k][2]; px[k][2] = t0; }
    }
    if (LAYOUT == LAYOUT_NCHW) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            PPack<T, 8> p;
#pragma unroll
            for (int k = 0; k < 8; ++k) p.v[k] = px[k][c];
            *reinterpret_cast<PPack<T, 8> *>(out + (((size_t)slot * 3 + c) * in_h + y) * in_w + x_base) = p;
        }
    } else {
        T *o = out + (((size_t)slot * in_h + y) * in_w + x_base) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            PPack<T, 8> p;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int idx = k * 8 + e; p.v[e] = px[idx / 3][idx % 3]; }
            *reinterpret_cast<PPack<T, 8> *>(o + k * 8) = p;
        }
    }
}

// first maximum of row[0..n): every lane scans ascending indices with a strict >, lanes combine by (value, -index)
__device__ __forceinline__ void wave_argmax(const float *__restrict__ row, int n, int lane, float &best, int &idx)
{
    best = -INFINITY; idx = 0x7fffffff;
    for (int i = lane; i < n; i += WAVE) { const float v = row[i]; if (v > best) { best = v; idx = i; } }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(idx, off);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (idx == 0x7fffffff) idx = 0;      // all -inf / NaN row: np.argmax returns 0 for all-equal rows
}

__global__ void __launch_bounds__(BLOCK) simcc_decode_kernel(const float *__restrict__ sx, const float *__restrict__ sy, int n, int K, int Wx, int Wy,
                                                             float split_ratio, const double *__restrict__ meta, int in_w, int in_h,
                                                             double *__restrict__ kps, float *__restrict__ scores)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.x * NWAVES + w;
    if (job >= n * K) return;
    const int box = job / K;
    float mx, my; int ax, ay;
    wave_argmax(sx + (size_t)job * Wx, Wx, lane, mx, ax);
    wave_argmax(sy + (size_t)job * Wy, Wy, lane, my, ay);
    if (lane == 0) {
        const float val = mx > my ? my : mx;              // get_simcc_maximum: the smaller of the two maxima
        float lx = (float)ax, ly = (float)ay;
        if (val <= 0.f) { lx = -1.f; ly = -1.f; }
        const float kx = lx / split_ratio, ky = ly / split_ratio;
        const double *m = meta + (size_t)box * META;
        kps[(size_t)job * 3] = (double)kx / (double)in_w * m[2] + m[0] - m[2] / 2;
        kps[(size_t)job * 3 + 1] = (double)ky / (double)in_h * m[3] + m[1] - m[3] / 2;
        kps[(size_t)job * 3 + 2] = (double)val;
        scores[job] = val;
    }
}

// keypoints_conf = np.mean(scores, axis=1) in float32: numpy's pairwise sum (8 accumulators, then the tail) / K
__global__ void __launch_bounds__(BLOCK) simcc_conf_kernel(const float *__restrict__ scores, int n, int K, float *__restrict__ conf)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *a = scores + (size_t)i * K;
    float res;
    if (K < 8) { res = 0.f; for (int k = 0; k < K; ++k) res += a[k]; }
    else {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int k = 8;
        for (; k < K - (K % 8); k += 8) for (int j = 0; j < 8; ++j) r[j] += a[k + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; k < K; ++k) res += a[k];
    }
    conf[i] = res / (float)K;
}

template <typename T>
void launch_pose_warp(const unsigned char *frames, int B, int H, int W, const int *counts, int max_n, int in_w, int in_h, const double *meta,
                      const double *mean, const double *stdv, int layout, void *out, hipStream_t st, int swap_rb)
{
    const int sw0 = swap_rb ? 2 : 0, sw2 = swap_rb ? 0 : 2;
    const long long units = (long long)B * max_n * in_h * (in_w / 8);
    const dim3 grid((unsigned)((units + BLOCK - 1) / BLOCK));
    if (layout == LAYOUT_NCHW)
        hipLaunchKernelGGL((pose_warp_kernel<T, LAYOUT_NCHW>), grid, dim3(BLOCK), 0, st, frames, B, H, W, counts, max_n, in_w, in_h, meta,
                           mean[sw0], mean[1], mean[sw2], stdv[sw0], stdv[1], stdv[sw2], (T *)out, swap_rb);
    else
        hipLaunchKernelGGL((pose_warp_kernel<T, LAYOUT_NHWC>), grid, dim3(BLOCK), 0, st, frames, B, H, W, counts, max_n, in_w, in_h, meta,
                           mean[sw0], mean[1], mean[sw2], stdv[sw0], stdv[1], stdv[sw2], (T *)out, swap_rb);
}

}  // namespace

extern "C" int tlk_pose_crop_warp_norm(const uint8_t *frames_dev, int batch, int h, int w, const double *boxes_xyxy_dev, int box_stride,
                                       const int32_t *counts_dev, int max_n, int in_w, int in_h, const double *mean3, const double *std3,
                                       int layout, int dtype, void *out_dev, double *meta_dev, void *hip_stream)
{
    if (batch < 0 || h <= 0 || w <= 0 || max_n < 0 || in_w <= 0 || in_h <= 0 || box_stride < 4) return fail(TLK_EINVAL, "tlk_pose_crop_warp_norm: bad size");
    if (in_w % 8 != 0) return fail(TLK_EINVAL, "tlk_pose_crop_warp_norm: in_w must be a multiple of 8");
    const int swap_rb = (layout & TLK_SWAP_RB) ? 1 : 0;
    layout &= ~TLK_SWAP_RB;
    if (layout < 0 || layout > 1 || dtype < 0 || dtype > 2) return fail(TLK_EINVAL, "tlk_pose_crop_warp_norm: bad layout/dtype");
    if (batch == 0 || max_n == 0) return TLK_OK;
    if (!frames_dev || !boxes_xyxy_dev || !counts_dev || !mean3 || !std3 || !out_dev || !meta_dev) return fail(TLK_EINVAL, "tlk_pose_crop_warp_norm: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    const int slots = batch * max_n;
    hipLaunchKernelGGL(pose_prep_kernel, dim3((slots + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, boxes_xyxy_dev, box_stride, (const int *)counts_dev,
                       batch, max_n, in_w, in_h, 1.25, meta_dev);
    if (dtype == 0) launch_pose_warp<float>(frames_dev, batch, h, w, (const int *)counts_dev, max_n, in_w, in_h, meta_dev, mean3, std3, layout, out_dev, st, swap_rb);
    else if (dtype == 1) launch_pose_warp<__half>(frames_dev, batch, h, w, (const int *)counts_dev, max_n, in_w, in_h, meta_dev, mean3, std3, layout, out_dev, st, swap_rb);
    else launch_pose_warp<bf16p_t>(frames_dev, batch, h, w, (const int *)counts_dev, max_n, in_w, in_h, meta_dev, mean3, std3, layout, out_dev, st, swap_rb);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}

extern "C" int tlk_simcc_decode(const float *simcc_x_dev, const float *simcc_y_dev, int n, int n_keypoints, int wx, int wy, double split_ratio,
                                const double *meta_dev, int in_w, int in_h, double *kps_xyc_dev, float *scores_dev, float *conf_dev,
                                void *hip_stream)
{
    if (n < 0 || n_keypoints <= 0 || wx <= 0 || wy <= 0 || in_w <= 0 || in_h <= 0) return fail(TLK_EINVAL, "tlk_simcc_decode: bad size");
    if (n == 0) return TLK_OK;
    if (!simcc_x_dev || !simcc_y_dev || !meta_dev || !kps_xyc_dev || !scores_dev) return fail(TLK_EINVAL, "tlk_simcc_decode: null pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    const int jobs = n * n_keypoints;
    hipLaunchKernelGGL(simcc_decode_kernel, dim3((jobs + NWAVES - 1) / NWAVES), dim3(BLOCK), 0, st, simcc_x_dev, simcc_y_dev, n, n_keypoints, wx, wy,
                       (float)split_ratio, meta_dev, in_w, in_h, kps_xyc_dev, scores_dev);
    if (conf_dev)
        hipLaunchKernelGGL(simcc_conf_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, (const float *)scores_dev, n, n_keypoints, conf_dev);
    TLK_HIP(hipGetLastError());
    return TLK_OK;
}
