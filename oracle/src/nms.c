/*
 * oracle/src/nms.c -- CPU ORACLE (test infrastructure, NOT product code).
 * `non_max_suppression(boxes, max_bbox_overlap, scores=None)` of plugins/track/strong_sort/sort/preprocessing.py:6-73 (identical copy:
 * plugins/track/bpbreid_strong_sort/sort/preprocessing.py). SURVEY 8a row G2. The function is DEAD CODE in the reference (no call site,
 * and `boxes.astype(np.float)` fails on NumPy >= 1.24); the golden vectors come from the function itself run with `np.float = float`
 * restored (tests/golden/make_golden.py gen_nms).
 *   boxes (n, 4) float64 (x, y, w, h); corners x2 = w + x, y2 = h + y; area = (x2 - x1 + 1) (y2 - y1 + 1)
 *   order = argsort(scores), or argsort(y2) without scores; pick the last of the order, drop every earlier one whose
 *   max(0, xx2 - xx1 + 1) * max(0, yy2 - yy1 + 1) / area[other] exceeds max_bbox_overlap; repeat
 * np.argsort's default kind is not stable: equal keys come out in an implementation-defined order. Here (and in the HIP kernel) equal keys
 * keep ascending index order; the goldens use distinct keys.
 */
#include "orc.h"
#include <stdlib.h>

typedef struct { double k; int i; } nms_key;
static int nms_cmp(const void *a, const void *b)
{
    const nms_key *p = a, *q = b;
    if (p->k < q->k) return -1;
    if (p->k > q->k) return 1;
    return p->i - q->i;
}

int orc_deepsort_nms(const double *boxes, const double *scores, int n, double max_overlap, int32_t *pick)
{
    if (n <= 0) return 0;
    double *x2 = malloc(sizeof(double) * n), *y2 = malloc(sizeof(double) * n), *area = malloc(sizeof(double) * n);
    nms_key *key = malloc(sizeof(nms_key) * n);
    int *idxs = malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) {
        const double *b = boxes + (size_t)i * 4;
        x2[i] = b[2] + b[0]; y2[i] = b[3] + b[1];
        area[i] = (x2[i] - b[0] + 1) * (y2[i] - b[1] + 1);
        key[i].k = scores ? scores[i] : y2[i]; key[i].i = i;
    }
    qsort(key, (size_t)n, sizeof(nms_key), nms_cmp);
    for (int i = 0; i < n; ++i) idxs[i] = key[i].i;
    int m = n, np_ = 0;
    while (m > 0) {
        const int last = m - 1, i = idxs[last];
        pick[np_++] = i;
        int k = 0;
        for (int p = 0; p < last; ++p) {
            const int j = idxs[p];
            const double *bi = boxes + (size_t)i * 4, *bj = boxes + (size_t)j * 4;
            const double xx1 = bi[0] > bj[0] ? bi[0] : bj[0], yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
            const double xx2 = x2[i] < x2[j] ? x2[i] : x2[j], yy2 = y2[i] < y2[j] ? y2[i] : y2[j];
            double w = xx2 - xx1 + 1, h = yy2 - yy1 + 1;
            w = w > 0 ? w : 0; h = h > 0 ? h : 0;
            const double overlap = (w * h) / area[j];
            if (!(overlap > max_overlap)) idxs[k++] = j;
        }
        m = k;
    }
    free(x2); free(y2); free(area); free(key); free(idxs);
    return np_;
}
