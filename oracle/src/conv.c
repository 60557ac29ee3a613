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
                    float v = acc + (bias ? bias[co] : 0.f);
                    if (ro) v += ro[co];
                    if (act == 1) v = v > 0.f ? v : 0.f;
                    else if (act == 2) v = v / (1.f + expf(-v));
                    yo[co] = v;
                }
            }
}
