/* heads.c -- ORACLE (test infrastructure, never linked into the product): plain-C restatement of the r06 prediction-head kernels of
 * tracklab_amd/csrc/tlk_heads.hip, loop for loop in the arithmetic contract include/tlk.h states for them.
 *
 * What they stand for in the reference (third-party graphs, absent from /root/reference -- parity with THOSE is unpinned, as for every backbone
 * piece; what is pinned is kernel == this file, and this file against a torch fp32 statement of the same head in tests/test_gpu_heads.py):
 *   orc_yolox_head      the outputs of YOLOX's decoupled head as rtmlib's ONNX model emits them (tracklab/wrappers/bbox_detector/rtmlib_api.py:21,30):
 *                       reg (4) | sigmoid(obj) | sigmoid(cls) per anchor, levels concatenated along the anchors;
 *   orc_reid_part_head  BPBReID / KPR's part-based head (tracklab/wrappers/reid/kpreid_api.py:147-182: embeddings (N, K, D), visibility_scores (N, K)):
 *                       pixel-wise part classifier, softmax over the parts, attention-weighted average, visibility from the strongest attention. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "orc.h"

static float dot_chain(const float *x, const float *w, int c)
{
    float acc = 0.f;
    for (int i = 0; i < c; ++i) acc = fmaf(x[i], w[i], acc);
    return acc;
}

static float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

/* one level: cls / reg (batch, hw, c) fp32 NHWC dense, w (5 + ncls, c), b (5 + ncls); out rows [a_off, a_off + hw) of (batch, A, 5 + ncls) */
void orc_yolox_head_level(const float *cls, const float *reg, const float *w, const float *b, int batch, int hw, int c, int ncls, int a_off, int A, float *out)
{
    const int no = 5 + ncls;
    for (int n = 0; n < batch; ++n)
        for (int p = 0; p < hw; ++p) {
            const float *xr = reg + ((size_t)n * hw + p) * c, *xc = cls + ((size_t)n * hw + p) * c;
            float *o = out + ((size_t)n * A + a_off + p) * no;
            for (int k = 0; k < 4; ++k) o[k] = dot_chain(xr, w + (size_t)k * c, c) + b[k];
            o[4] = sigmoidf_(dot_chain(xr, w + (size_t)4 * c, c) + b[4]);
            for (int k = 5; k < no; ++k) o[k] = sigmoidf_(dot_chain(xc, w + (size_t)k * c, c) + b[k]);
        }
}

/* feat (rows_dense, hw, d) fp32; counts / slot_base may be NULL; emb (rows, k, d), vis (rows, k); returns 1 when a live embedding is not finite */
int orc_reid_part_head(const float *feat, int hw, int d, int k, const float *w, const float *b, const int32_t *counts, const int32_t *slot_base,
                       int rows, int max_dets, float vis_thr, float *emb, unsigned char *vis)
{
    int bad = 0;
    float att[8], lg[8];
    for (int r = 0; r < rows; ++r) {
        float *e = emb + (size_t)r * k * d;
        unsigned char *v = vis + (size_t)r * k;
        long long src = r;
        memset(e, 0, sizeof(float) * (size_t)k * d);
        memset(v, 0, (size_t)k);
        if (counts) {
            const int f = r / max_dets, j = r % max_dets;
            if (j >= counts[f]) continue;
            if (slot_base) src = slot_base[f] + j;
        }
        const float *fm = feat + (size_t)src * hw * d;
        float den[8] = {0}, mx[8] = {0};
        for (int p = 0; p < hw; ++p) {
            const float *x = fm + (size_t)p * d;
            float m, s = 0.f;
            for (int q = 0; q < k; ++q) lg[q] = dot_chain(x, w + (size_t)q * d, d) + b[q];
            m = lg[0];
            for (int q = 1; q < k; ++q) m = fmaxf(m, lg[q]);
            for (int q = 0; q < k; ++q) { att[q] = expf(lg[q] - m); s += att[q]; }
            for (int q = 0; q < k; ++q) {
                att[q] = att[q] / s;
                den[q] += att[q];
                mx[q] = (att[q] != att[q] || att[q] > mx[q]) ? att[q] : mx[q];
                for (int i = 0; i < d; ++i) e[(size_t)q * d + i] = fmaf(att[q], x[i], e[(size_t)q * d + i]);
            }
        }
        for (int q = 0; q < k; ++q) {
            const float dn = den[q] < 1e-6f ? 1e-6f : den[q];
            v[q] = (q == 0 || mx[q] > vis_thr) ? 1 : 0;
            for (int i = 0; i < d; ++i) {
                const float val = e[(size_t)q * d + i] / dn;
                e[(size_t)q * d + i] = val;
                if (!(fabsf(val) <= 3.4028234664e38f)) bad = 1;
            }
        }
    }
    return bad;
}
