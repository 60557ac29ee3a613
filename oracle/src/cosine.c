/* CPU ORACLE (test infrastructure) -- see orc.h.
 * Cosine gallery distance of plain StrongSORT: plugins/track/strong_sort/sort/nn_matching.py:30-50
 * (_cosine_distance), :73-91 (_nn_cosine_distance), :144-161 (NearestNeighborDistanceMetric.distance). */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

/* gallery (Gtot, D) f32 rows grouped per track by CSR offsets (T+1); dets (N, D) f32 -> out (T, N) f64:
 * min over the track's gallery rows of 1 - (g/|g|).(d/|d|), float32 arithmetic like numpy on float32 inputs. */
void orc_cosine_gallery_min_f32(const float *gallery, const int32_t *offsets, int T, const float *dets, int N, int D, double *out)
{
    int G = offsets[T];
    float *gn = malloc(sizeof(float) * (size_t)(G > 0 ? G : 1) * D), *dn = malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * D);
    for (int pass = 0; pass < 2; ++pass) {
        const float *src = pass ? dets : gallery; float *dst = pass ? dn : gn; int R = pass ? N : G;
        for (int r = 0; r < R; ++r) {
            float ss = 0;
            for (int d = 0; d < D; ++d) ss += src[(size_t)r * D + d] * src[(size_t)r * D + d];
            float nrm = sqrtf(ss);
            for (int d = 0; d < D; ++d) dst[(size_t)r * D + d] = src[(size_t)r * D + d] / nrm;
        }
    }
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < N; ++n) {
            float best = INFINITY;
            for (int g = offsets[t]; g < offsets[t + 1]; ++g) {
                float dot = 0;
                for (int d = 0; d < D; ++d) dot += gn[(size_t)g * D + d] * dn[(size_t)n * D + d];
                float v = 1.f - dot;
                if (v < best) best = v;
            }
            out[(size_t)t * N + n] = (double)best;
        }
    free(gn); free(dn);
}
