// tlk_ecc.hip -- StrongSORT's camera-motion estimator on the device: `ECC(src, dst)` of plugins/track/strong_sort/sort/track.py:129-211
// (= plugins/track/bpbreid_strong_sort/ecc.py:4-99): BGR2GRAY + resize(0.1) of the frame, then cv2.findTransformECC(previous, current,
// eye(2,3), MOTION_EUCLIDEAN, 100 iterations / eps 1e-5, gaussFiltSize 1), translation rescaled to frame pixels.
//
// OpenCV's algorithm restated (video/src/ecc.cpp): PARITY UNPINNED -- see oracle/src/ecc.c, the CPU restatement this kernel is checked
// against (same arithmetic, same fixed summation order).
//
//   gray_resize_kernel   frame (h,w,3) u8 -> (round(h/10), round(w/10)) grey image                                   (tlk_cv.hpp)
//   ecc_kernel           ONE workgroup of 1024 threads runs the whole Gauss-Newton iteration: the 0.1-scaled images are 192 x 108 px
//                        (83 KB as float), every iteration is three sweeps over them (warp + masked moments; zero-mean + Jacobian +
//                        Hessian / projections; error projection) separated by workgroup reductions of up to 13 doubles, and a
//                        3 x 3 solve every thread repeats for itself. No grid-wide synchronisation, no host round trip: typically
//                        10-30 iterations, a few us each, on the association stream.
#include "tlk_common.hpp"
#include "tlk_cv.hpp"

using namespace tlk;
using namespace tlk::cv;

namespace {

constexpr int ECC_THREADS = 1024, ECC_WAVES = ECC_THREADS / 64;
constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS;

// sum of v[k] over the workgroup in the order oracle/src/ecc.c fixes: balanced pairwise tree inside each wavefront, then the 16
// wavefront sums left to right. Every thread returns with the totals.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *s_red)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int step = 1; step < 64; step <<= 1) x = x + __shfl_xor(x, step, 64);
        v[k] = x;
    }
    __syncthreads();                                                // (the previous reduction's readers are done)
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) s_red[(threadIdx.x >> 6) * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int g = 0; g < ECC_WAVES; ++g) s += s_red[g * K + k];
        v[k] = s;
    }
}

enum { PIX_PLAIN, PIX_GX, PIX_GY };
template <int WHAT>
__device__ __forceinline__ float pix(const unsigned char *__restrict__ img, int h, int w, int y, int x)
{
    if (WHAT == PIX_PLAIN) return (float)img[y * w + x];
    // filter2D(image, [-0.5, 0, 0.5]) / its transpose, BORDER_REFLECT_101, evaluated where it is sampled
    if (WHAT == PIX_GX) return 0.5f * (float)img[y * w + reflect101(x + 1, w)] - 0.5f * (float)img[y * w + reflect101(x - 1, w)];
    return 0.5f * (float)img[reflect101(y + 1, h) * w + x] - 0.5f * (float)img[reflect101(y - 1, h) * w + x];
}
// cv::warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, BORDER_CONSTANT 0) of the float image / its gradients at fixed-point coordinates (X, Y)
__device__ __forceinline__ void warp_linear3(const unsigned char *__restrict__ img, int h, int w, int X, int Y, float &o, float &ox, float &oy)
{
    const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
    o = ox = oy = 0.f;
    if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) return;
    const float fx = (float)(X & (INTER_TAB - 1)) * (1.f / INTER_TAB), fy = (float)(Y & (INTER_TAB - 1)) * (1.f / INTER_TAB);
    const float wx0 = 1.f - fx, wx1 = fx, wy0 = 1.f - fy, wy1 = fy;
    const float wt[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
    const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
    const bool in[4] = {x0 && y0, x1 && y0, x0 && y1, x1 && y1};
    const int ty[4] = {sy, sy, sy + 1, sy + 1}, tx[4] = {sx, sx + 1, sx, sx + 1};
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v = in[k] ? pix<PIX_PLAIN>(img, h, w, ty[k], tx[k]) : 0.f, vx = in[k] ? pix<PIX_GX>(img, h, w, ty[k], tx[k]) : 0.f,
                    vy = in[k] ? pix<PIX_GY>(img, h, w, ty[k], tx[k]) : 0.f;
        if (k == 0) { a = v * wt[0]; b = vx * wt[0]; c = vy * wt[0]; }
        else { a = a + v * wt[k]; b = b + vx * wt[k]; c = c + vy * wt[k]; }
    }
    o = a; ox = b; oy = c;
}

struct EccScratch { float *iw, *gxw, *gyw, *tz, *j0; unsigned char *mask; };

// templ / image: (h, w) uint8 (previous / current 0.1-scaled grey frame). out6: the (2,3) warp as doubles of the float32 values, translation
// divided by `scale` (float32 arithmetic, the reference's numpy float32 / python float under NumPy >= 2). status: iterations run (>= 1), or -1
// where OpenCV throws (NaN correlation, non-positive lambda denominator): the caller skips the camera update. rho_out: final correlation.
__global__ void __launch_bounds__(ECC_THREADS) ecc_kernel(const unsigned char *__restrict__ templ, const unsigned char *__restrict__ image, int h, int w,
                                                          EccScratch S, int max_iter, double eps, float scale, double *__restrict__ out6, int *__restrict__ status,
                                                          double *__restrict__ rho_out)
{
    __shared__ double s_red[ECC_WAVES * 13];
    const int tid = threadIdx.x, npx = h * w;
    float map[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    double rho = -1.0, last_rho = -eps;
    int it = 0, bad = 0;
    for (int i = 1; i <= max_iter && fabs(rho - last_rho) >= eps; ++i) {
        it = i;
        double M[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = (double)map[k];
        // sweep A: warps + masked first and second moments
        double a5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int p = tid; p < npx; p += ECC_THREADS) {
            const int y = p / w, x = p - y * w;
            const int bx = __double2int_rn((M[1] * y + M[2]) * AB_SCALE), by = __double2int_rn((M[4] * y + M[5]) * AB_SCALE);
            const int ad = __double2int_rn(M[0] * x * AB_SCALE), bd = __double2int_rn(M[3] * x * AB_SCALE);
            const int X = (bx + AB_SCALE / INTER_TAB / 2 + ad) >> (AB_BITS - INTER_BITS), Y = (by + AB_SCALE / INTER_TAB / 2 + bd) >> (AB_BITS - INTER_BITS);
            float v, vx, vy;
            warp_linear3(image, h, w, X, Y, v, vx, vy);
            const int nx = (bx + AB_SCALE / 2 + ad) >> AB_BITS, ny = (by + AB_SCALE / 2 + bd) >> AB_BITS;
            const bool m = nx >= 0 && nx < w && ny >= 0 && ny < h;
            S.iw[p] = v; S.gxw[p] = vx; S.gyw[p] = vy; S.mask[p] = m ? 1 : 0;
            const double t = (double)(float)templ[p];
            a5[0] += m ? 1.0 : 0.0;
            a5[1] += m ? (double)v : 0.0;
            a5[2] += m ? (double)v * (double)v : 0.0;
            a5[3] += m ? t : 0.0;
            a5[4] += m ? t * t : 0.0;
        }
        block_sum<5>(a5, s_red);
        const double nz = a5[0], sc = nz != 0.0 ? 1.0 / nz : 0.0;
        const double img_mean = a5[1] * sc, tmp_mean = a5[3] * sc;
        double img_var = a5[2] * sc - img_mean * img_mean, tmp_var = a5[4] * sc - tmp_mean * tmp_mean;
        if (img_var < 0.0) img_var = 0.0;
        if (tmp_var < 0.0) tmp_var = 0.0;
        const double img_std = sqrt(img_var), tmp_std = sqrt(tmp_var);
        const double tmp_norm = sqrt(nz * tmp_std * tmp_std), img_norm = sqrt(nz * img_std * img_std);
        // sweep B: zero-mean, Jacobian, Hessian and projections (every thread re-reads only what it wrote itself)
        const float h0 = map[0], h1 = map[3], im_f = (float)img_mean, tm_f = (float)tmp_mean;
        double a[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) a[k] = 0.0;
        for (int p = tid; p < npx; p += ECC_THREADS) {
            const int y = p / w, x = p - y * w;
            float iwv = S.iw[p], tzv = 0.f;
            if (S.mask[p]) { iwv = iwv - im_f; tzv = (float)templ[p] - tm_f; S.iw[p] = iwv; }
            S.tz[p] = tzv;
            const float Xf = (float)x, Yf = (float)y;
            const float hx_a = Xf * (-h1), hx_b = Yf * (-h0), hy_a = Xf * h0, hy_b = Yf * (-h1);
            const float hatX = hx_a + hx_b, hatY = hy_a + hy_b;
            const float gx = S.gxw[p], gy = S.gyw[p];
            const float ja = gx * hatX, jb = gy * hatY;
            const float j0 = ja + jb;
            S.j0[p] = j0;
            const double J0 = j0, J1 = gx, J2 = gy, I = iwv, T = tzv;
            a[0] += J0 * J0; a[1] += J0 * J1; a[2] += J0 * J2; a[3] += J1 * J1; a[4] += J1 * J2; a[5] += J2 * J2;
            a[6] += J0 * I; a[7] += J1 * I; a[8] += J2 * I; a[9] += J0 * T; a[10] += J1 * T; a[11] += J2 * T; a[12] += T * I;
        }
        block_sum<13>(a, s_red);
        // Hessian (float, symmetric); inverse by the adjugate in double (cv::invert of a 3 x 3 CV_32F matrix), back to float
        const float H[3][3] = {{(float)a[0], (float)a[1], (float)a[2]}, {(float)a[1], (float)a[3], (float)a[4]}, {(float)a[2], (float)a[4], (float)a[5]}};
        float Hi[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        double d = (double)H[0][0] * ((double)H[1][1] * H[2][2] - (double)H[1][2] * H[2][1]) - (double)H[0][1] * ((double)H[1][0] * H[2][2] - (double)H[1][2] * H[2][0])
                 + (double)H[0][2] * ((double)H[1][0] * H[2][1] - (double)H[1][1] * H[2][0]);
        if (d != 0.0) {
            d = 1.0 / d;
            Hi[0][0] = (float)(((double)H[1][1] * H[2][2] - (double)H[1][2] * H[2][1]) * d);
            Hi[0][1] = (float)(((double)H[0][2] * H[2][1] - (double)H[0][1] * H[2][2]) * d);
            Hi[0][2] = (float)(((double)H[0][1] * H[1][2] - (double)H[0][2] * H[1][1]) * d);
            Hi[1][0] = (float)(((double)H[1][2] * H[2][0] - (double)H[1][0] * H[2][2]) * d);
            Hi[1][1] = (float)(((double)H[0][0] * H[2][2] - (double)H[0][2] * H[2][0]) * d);
            Hi[1][2] = (float)(((double)H[0][2] * H[1][0] - (double)H[0][0] * H[1][2]) * d);
            Hi[2][0] = (float)(((double)H[1][0] * H[2][1] - (double)H[1][1] * H[2][0]) * d);
            Hi[2][1] = (float)(((double)H[0][1] * H[2][0] - (double)H[0][0] * H[2][1]) * d);
            Hi[2][2] = (float)(((double)H[0][0] * H[1][1] - (double)H[0][1] * H[1][0]) * d);
        }
        const double correlation = a[12];
        last_rho = rho;
        rho = correlation / (img_norm * tmp_norm);
        if (isnan(rho)) { bad = 1; break; }
        const float ip[3] = {(float)a[6], (float)a[7], (float)a[8]}, tp[3] = {(float)a[9], (float)a[10], (float)a[11]};
        float iph[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) iph[r] = (float)((double)Hi[r][0] * ip[0] + (double)Hi[r][1] * ip[1] + (double)Hi[r][2] * ip[2]);
        const double lambda_n = img_norm * img_norm - ((double)ip[0] * iph[0] + (double)ip[1] * iph[1] + (double)ip[2] * iph[2]);
        const double lambda_d = correlation - ((double)tp[0] * iph[0] + (double)tp[1] * iph[1] + (double)tp[2] * iph[2]);
        if (lambda_d <= 0.0) { rho = -1.0; bad = 1; break; }
        const float lam_f = (float)(lambda_n / lambda_d);
        // sweep C: error image and its projection
        double e3[3] = {0.0, 0.0, 0.0};
        for (int p = tid; p < npx; p += ECC_THREADS) {
            const float lt = lam_f * S.tz[p];
            const double e = (double)(lt - S.iw[p]);
            e3[0] += (double)S.j0[p] * e; e3[1] += (double)S.gxw[p] * e; e3[2] += (double)S.gyw[p] * e;
        }
        block_sum<3>(e3, s_red);
        const float ep[3] = {(float)e3[0], (float)e3[1], (float)e3[2]};
        float dp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) dp[r] = (float)((double)Hi[r][0] * ep[0] + (double)Hi[r][1] * ep[1] + (double)Hi[r][2] * ep[2]);
        // update_warping_matrix_ECC, MOTION_EUCLIDEAN
        double new_theta = (double)dp[0];
        new_theta += asin((double)map[3]);
        map[2] += dp[1]; map[5] += dp[2];
        map[0] = map[4] = (float)cos(new_theta);
        map[3] = (float)sin(new_theta);
        map[1] = -map[3];
    }
    if (tid == 0) {
        if (!bad) { map[2] = map[2] / scale; map[5] = map[5] / scale; }
        for (int k = 0; k < 6; ++k) out6[k] = (double)map[k];
        *status = bad ? -1 : it;
        *rho_out = rho;
    }
}

__global__ void ecc_first_kernel(double *__restrict__ out6, int *__restrict__ status, double *__restrict__ rho_out)
{
    if (threadIdx.x == 0) { out6[0] = 1; out6[1] = 0; out6[2] = 0; out6[3] = 0; out6[4] = 1; out6[5] = 0; *status = 0; *rho_out = 0.0; }
}

}  // namespace

struct tlk_ecc {
    int device, h, w, dh, dw, have_prev, cur;
    unsigned char *gray[2], *frame_stage;
    EccScratch S;
    double *warp, *rho; int *status;
};

static void ecc_free(tlk_ecc *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    void *ptrs[] = {c->gray[0], c->gray[1], c->frame_stage, c->S.iw, c->S.gxw, c->S.gyw, c->S.tz, c->S.j0, c->S.mask, c->warp, c->rho, c->status};
    for (void *p : ptrs) if (p) hipFree(p);
    delete c;
}

extern "C" int tlk_ecc_create(int h, int w, int device, tlk_ecc **out)
{
    if (!out || h < 20 || w < 20) return fail(TLK_EINVAL, "tlk_ecc_create: bad argument (frames of at least 20 x 20 px)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TLK_ENODEVICE, "tlk_ecc_create: no HIP device (libtlk has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TLK_EINVAL, "tlk_ecc_create: bad device index");
    TLK_HIP(hipSetDevice(device));
    tlk_ecc *c = new tlk_ecc();
    memset(c, 0, sizeof(*c));
    c->device = device; c->h = h; c->w = w;
    c->dh = (int)lrint(h * 0.1); c->dw = (int)lrint(w * 0.1);           // cv::resize(dsize = 0, fx = fy = 0.1): saturate_cast<int>(size * 0.1)
    const size_t npx = (size_t)c->dh * c->dw;
#define ECC_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) { ecc_free(c); return fail(TLK_EHIP, std::string("tlk_ecc_create: hipMalloc: ") + hipGetErrorString(e_)); } } while (0)
    ECC_ALLOC(c->gray[0], npx); ECC_ALLOC(c->gray[1], npx); ECC_ALLOC(c->frame_stage, (size_t)h * w * 3);
    ECC_ALLOC(c->S.iw, sizeof(float) * npx); ECC_ALLOC(c->S.gxw, sizeof(float) * npx); ECC_ALLOC(c->S.gyw, sizeof(float) * npx);
    ECC_ALLOC(c->S.tz, sizeof(float) * npx); ECC_ALLOC(c->S.j0, sizeof(float) * npx); ECC_ALLOC(c->S.mask, npx);
    ECC_ALLOC(c->warp, sizeof(double) * 6); ECC_ALLOC(c->rho, sizeof(double)); ECC_ALLOC(c->status, sizeof(int));
#undef ECC_ALLOC
    TLK_HIP(hipDeviceSynchronize());
    *out = c;
    return TLK_OK;
}

extern "C" int tlk_ecc_destroy(tlk_ecc *c) { ecc_free(c); return TLK_OK; }
extern "C" int tlk_ecc_reset(tlk_ecc *c) { if (!c) return fail(TLK_EINVAL, "tlk_ecc_reset: null handle"); c->have_prev = 0; c->cur = 0; return TLK_OK; }

extern "C" int tlk_ecc_apply_dev(tlk_ecc *c, const uint8_t *frame_dev, double *warp6_dev, int *status_dev, void *hip_stream)
{
    if (!c || !frame_dev || !warp6_dev || !status_dev) return fail(TLK_EINVAL, "tlk_ecc_apply_dev: null pointer");
    TLK_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int b = c->cur, pb = 1 - b, npx = c->dh * c->dw;
    hipLaunchKernelGGL(gray_resize_kernel, dim3((unsigned)((npx + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, frame_dev, c->h, c->w, c->gray[b], c->dh, c->dw);
    if (c->have_prev)
        hipLaunchKernelGGL(ecc_kernel, dim3(1), dim3(ECC_THREADS), 0, st, (const unsigned char *)c->gray[pb], (const unsigned char *)c->gray[b], c->dh, c->dw, c->S, 100, 1e-5,
                           (float)0.1, warp6_dev, status_dev, c->rho);
    else
        hipLaunchKernelGGL(ecc_first_kernel, dim3(1), dim3(64), 0, st, warp6_dev, status_dev, c->rho);
    TLK_HIP(hipGetLastError());
    c->have_prev = 1; c->cur = pb;
    return TLK_OK;
}

extern "C" int tlk_ecc_apply(tlk_ecc *c, const uint8_t *frame_host, double *warp6_host, int *status, double *rho)
{
    if (!c || !frame_host || !warp6_host || !status) return fail(TLK_EINVAL, "tlk_ecc_apply: null pointer");
    TLK_HIP(hipSetDevice(c->device));
    TLK_HIP(hipMemcpy(c->frame_stage, frame_host, (size_t)c->h * c->w * 3, hipMemcpyHostToDevice));
    const int rc = tlk_ecc_apply_dev(c, c->frame_stage, c->warp, c->status, nullptr);
    if (rc != TLK_OK) return rc;
    TLK_HIP(hipMemcpy(warp6_host, c->warp, sizeof(double) * 6, hipMemcpyDeviceToHost));
    TLK_HIP(hipMemcpy(status, c->status, sizeof(int), hipMemcpyDeviceToHost));
    if (rho) TLK_HIP(hipMemcpy(rho, c->rho, sizeof(double), hipMemcpyDeviceToHost));
    return TLK_OK;
}

// test entry: cv2.findTransformECC alone on two (h, w) uint8 host images (the 0.1-scaled grey frames), translation NOT rescaled
extern "C" int tlk_ecc_find_transform(const uint8_t *templ_host, const uint8_t *image_host, int h, int w, int max_iter, double eps, double *warp6_host, int *status, double *rho, int device)
{
    if (!templ_host || !image_host || !warp6_host || !status || h < 2 || w < 2) return fail(TLK_EINVAL, "tlk_ecc_find_transform: bad argument");
    tlk_ecc *c = nullptr;
    const int rc = tlk_ecc_create(h * 10, w * 10, device, &c);
    if (rc != TLK_OK) return rc;
    if (c->dh != h || c->dw != w) { ecc_free(c); return fail(TLK_EINVAL, "tlk_ecc_find_transform: size"); }
    hipError_t e = hipMemcpy(c->gray[0], templ_host, (size_t)h * w, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->gray[1], image_host, (size_t)h * w, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ecc_kernel, dim3(1), dim3(ECC_THREADS), 0, (hipStream_t)0, (const unsigned char *)c->gray[0], (const unsigned char *)c->gray[1], h, w, c->S, max_iter, eps,
                           1.0f, c->warp, c->status, c->rho);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(warp6_host, c->warp, sizeof(double) * 6, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(status, c->status, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && rho) e = hipMemcpy(rho, c->rho, sizeof(double), hipMemcpyDeviceToHost);
    ecc_free(c);
    if (e != hipSuccess) return fail(TLK_EHIP, std::string("tlk_ecc_find_transform: ") + hipGetErrorString(e));
    return TLK_OK;
}
