/* conv.c -- CPU checker of libtlk's fp32 convolution (tracklab_amd/csrc/tlk_conv.hip).  TEST INFRASTRUCTURE: only tests/, smoke() and
 * bench.py's cpu_baseline may call it.
 *
 * What it restates: the convolution + folded-BatchNorm bias + residual add + activation of the reference's fp32 backbones, which run
 * inside third-party runtimes that are not vendored (ONNXRuntime behind tracklab/wrappers/bbox_detector/rtmlib_api.py:21 and
 * wrappers/pose_estimator/rtmlib_api.py:21, torchreid behind wrappers/reid/kpreid_api.py:147-182).  A convolution has no reference
 * summation order (cuDNN / ONNXRuntime / MIOpen each pick their own), so parity with the reference is a TOLERANCE statement
 * (tests compare with torch's own fp32 / fp64 convolution, rtol 2e-5 of the absolute-value convolution); what this file pins
 * bit-for-bit is the kernel's own contract: one fmaf chain per output element over k = (kh, kw, ci) in the order 0,4,1,5,2,6,3,7 within
 * every group of 8 (groups ascending, K rounded up to a multiple of 8 with zero terms, zero terms for taps outside the image), then
 * + bias, + residual, activation -- v_mfma_f32_32x32x2_f32 is exactly that chain on gfx950.  SiLU uses expf of this libm, the device
 * uses its own exp: that one activation is compared to 2 ulp-ish tolerance in the tests, the others bit-exactly. */
#include <math.h>
#include <stddef.h>

#include "orc.h"

void orc_conv2d_nhwc_f32(const float *x, const float *w, const float *bias, const float *res, float *y, int n, int h, int wd, int cin, int cout,
                         int kh, int kw, int stride, int pad, int act)
{
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (wd + 2 * pad - kw) / stride + 1;
    const int K = kh * kw * cin, K8 = (K + 7) / 8 * 8;
    static const int order[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int in = 0; in < n; ++in)
        for (int oh = 0; oh < ho; ++oh)
            for (int ow = 0; ow < wo; ++ow) {
                float *yo = y + (((size_t)in * ho + oh) * wo + ow) * cout;
                const float *ro = res ? res + (((size_t)in * ho + oh) * wo + ow) * cout : NULL;
                for (int co = 0; co < cout; ++co) {
                    float acc = 0.f;
                    for (int g = 0; g < K8; g += 8)
                        for (int t = 0; t < 8; ++t) {
                            const int k = g + order[t];
                            float a = 0.f, b = 0.f;
                            if (k < K) {
                                const int tap = k / cin, ci = k - tap * cin, ih = oh * stride + tap / kw - pad, iw = ow * stride + tap % kw - pad;
                                if (ih >= 0 && ih < h && iw >= 0 && iw < wd) a = x[(((size_t)in * h + ih) * wd + iw) * cin + ci];
                                b = w[(size_t)co * K + k];
                            }
                            acc = fmaf(a, b, acc);
                        }
                    const int res_post = act & 0x100, a = act & 0xff;      /* 0x100 = TLK_ACT_RES_AFTER: the residual joins after the activation */
                    float v = acc + (bias ? bias[co] : 0.f);
                    if (ro && !res_post) v += ro[co];
                    if (a == 1) v = v > 0.f ? v : 0.f;
                    else if (a == 2) v = v / (1.f + expf(-v));
                    if (ro && res_post) v += ro[co];
                    yo[co] = v;
                }
            }
}

/* The depthwise halves of RTMPose's CSPNeXt blocks (mmdet DepthwiseSeparableConvModule, run by the reference inside ONNXRuntime behind
 * tracklab/wrappers/pose_estimator/rtmlib_api.py:21-36): y[n,oy,ox,c] = act( sum_{ky,kx} x[n, oy+ky-k/2, ox+kx-k/2, c] * w[ky,kx,c] + bias[c] ).
 * Summation contract of tlk_dwconv2d_nhwc (tlk_dwconv.hip): one fmaf chain per output element, ky ascending then kx ascending; a row outside
 * the image contributes nothing, a column outside the image enters as a zero term.  Parity with the reference: tolerance against torch's
 * own depthwise convolution in the tests (no reference summation order exists). */
void orc_dwconv2d_nhwc_f32(const float *x, const float *w, const float *bias, float *y, int n, int h, int wd, int c, int k, int act)
{
    const int pad = k / 2;
    for (int in = 0; in < n; ++in)
        for (int oy = 0; oy < h; ++oy)
            for (int ox = 0; ox < wd; ++ox)
                for (int ch = 0; ch < c; ++ch) {
                    float acc = 0.f;
                    for (int ky = 0; ky < k; ++ky) {
                        const int iy = oy + ky - pad;
                        if (iy < 0 || iy >= h) continue;
                        for (int kx = 0; kx < k; ++kx) {
                            const int ix = ox + kx - pad;
                            const float a = (ix >= 0 && ix < wd) ? x[(((size_t)in * h + iy) * wd + ix) * c + ch] : 0.f;
                            acc = fmaf(a, w[((size_t)ky * k + kx) * c + ch], acc);
                        }
                    }
                    float v = acc + (bias ? bias[ch] : 0.f);
                    if (act == 1) v = v > 0.f ? v : 0.f;
                    else if (act == 2) v = v / (1.f + expf(-v));
                    y[(((size_t)in * h + oy) * wd + ox) * c + ch] = v;
                }
}

/* The pooling half of an SPPBottleneck (YOLOX CSPDarknet / RTMPose CSPNeXt, inside ONNXRuntime in the reference: wrappers/bbox_detector/
 * rtmlib_api.py:21, wrappers/pose_estimator/rtmlib_api.py:21): y (n,h,w,4c) = [x | max 5x5 | max 9x9 | max 13x13], stride 1, windows clipped at
 * the border (-inf padding).  Checker of tlk_spp_maxpool_nhwc; max is exact, so this equals torch's max_pool2d + cat bit for bit. */
void orc_spp_maxpool_nhwc_f32(const float *x, float *y, int n, int h, int wd, int c)
{
    static const int rad[3] = {2, 4, 6};
    for (int in = 0; in < n; ++in)
        for (int oy = 0; oy < h; ++oy)
            for (int ox = 0; ox < wd; ++ox) {
                float *yo = y + (((size_t)in * h + oy) * wd + ox) * 4 * c;
                for (int ch = 0; ch < c; ++ch) {
                    yo[ch] = x[(((size_t)in * h + oy) * wd + ox) * c + ch];
                    for (int r = 0; r < 3; ++r) {
                        float m = -INFINITY;
                        for (int iy = oy - rad[r]; iy <= oy + rad[r]; ++iy)
                            for (int ix = ox - rad[r]; ix <= ox + rad[r]; ++ix)
                                if (iy >= 0 && iy < h && ix >= 0 && ix < wd) {
                                    const float v = x[(((size_t)in * h + iy) * wd + ix) * c + ch];
                                    if (v > m) m = v;
                                }
                        yo[(r + 1) * c + ch] = m;
                    }
                }
            }
}
