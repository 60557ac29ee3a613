/* CPU ORACLE (test infrastructure) -- see orc.h. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double dmax(double a, double b) { return a > b ? a : b; }   /* np.maximum on non-NaN */
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* oc_sort/association.py:5-21 (iou), :24-55 (giou), :58-95 (diou), :97-147 (ciou), :150-171 (ct_dist) */
void orc_iou_matrix(int variant, const double *b1, int n, int s1, const double *b2, int m, int s2, double *out)
{
    for (int i = 0; i < n; ++i) {
        const double *a = b1 + (size_t)i * s1;
        for (int j = 0; j < m; ++j) {
            const double *b = b2 + (size_t)j * s2;
            double *o = out + (size_t)i * m + j;
            if (variant == ORC_CT) {
                double cx1 = (a[0] + a[2]) / 2.0, cy1 = (a[1] + a[3]) / 2.0;
                double cx2 = (b[0] + b[2]) / 2.0, cy2 = (b[1] + b[3]) / 2.0;
                double dx = cx1 - cx2, dy = cy1 - cy2;
                *o = sqrt(dx * dx + dy * dy);
                continue;
            }
            double xx1 = dmax(a[0], b[0]), yy1 = dmax(a[1], b[1]);
            double xx2 = dmin(a[2], b[2]), yy2 = dmin(a[3], b[3]);
            double w = dmax(0., xx2 - xx1), h = dmax(0., yy2 - yy1);
            double wh = w * h;
            double iou = wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
            if (variant == ORC_IOU) { *o = iou; continue; }
            double xxc1 = dmin(a[0], b[0]), yyc1 = dmin(a[1], b[1]);
            double xxc2 = dmax(a[2], b[2]), yyc2 = dmax(a[3], b[3]);
            if (variant == ORC_GIOU) {
                double wc = xxc2 - xxc1, hc = yyc2 - yyc1;
                double area_enclose = wc * hc;
                double giou = iou - (area_enclose - wh) / area_enclose;
                *o = (giou + 1.) / 2.0;
                continue;
            }
            double cx1 = (a[0] + a[2]) / 2.0, cy1 = (a[1] + a[3]) / 2.0;
            double cx2 = (b[0] + b[2]) / 2.0, cy2 = (b[1] + b[3]) / 2.0;
            double inner = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
            double outer = (xxc2 - xxc1) * (xxc2 - xxc1) + (yyc2 - yyc1) * (yyc2 - yyc1);
            if (variant == ORC_DIOU) { *o = ((iou - inner / outer) + 1) / 2.0; continue; }
            /* ciou */
            double w1 = a[2] - a[0], h1 = a[3] - a[1], w2 = b[2] - b[0], h2 = b[3] - b[1];
            h2 = h2 + 1.; h1 = h1 + 1.;
            double arct = atan(w2 / h2) - atan(w1 / h1);
            double v = (4 / (M_PI * M_PI)) * (arct * arct);
            double S = 1 - iou;
            double alpha = v / (S + v);
            *o = ((iou - inner / outer - alpha * v) + 1) / 2.0;
        }
    }
    if (variant == ORC_CT && n > 0 && m > 0) {           /* association.py:169-171 */
        size_t tot = (size_t)n * m;
        double mx = out[0];
        for (size_t k = 1; k < tot; ++k) mx = (out[k] > mx || isnan(out[k])) ? out[k] : mx;
        for (size_t k = 0; k < tot; ++k) out[k] = out[k] / mx;
        double mx2 = out[0];
        for (size_t k = 1; k < tot; ++k) mx2 = (out[k] > mx2 || isnan(out[k])) ? out[k] : mx2;
        for (size_t k = 0; k < tot; ++k) out[k] = mx2 - out[k];
    }
}

/* Rectangular LSAP, shortest augmenting path with the exact scan / tie-break order of
 * scipy 1.15 (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp): remaining[] filled in
 * reverse, ties prefer an unassigned column (later entry wins), tall matrices transposed,
 * output sorted by row. */
static int64_t aug_path(int nc, const double *cost, size_t rs, size_t cs, const double *u, const double *v,
                        int64_t *path, const int64_t *row4col, double *spc, int64_t i,
                        char *SR, char *SC, int64_t *remaining, int nr, double *p_minval)
{
    double minval = 0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    memset(SR, 0, (size_t)nr);
    memset(SC, 0, (size_t)nc);
    for (int j = 0; j < nc; ++j) spc[j] = INFINITY;
    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; ++it) {
            int64_t j = remaining[it];
            double r = minval + cost[(size_t)i * rs + (size_t)j * cs] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
        }
        minval = lowest;
        if (minval == INFINITY) return -1;
        int64_t j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_minval = minval;
    return sink;
}

int orc_lsa(const double *cost, int nr0, int nc0, int64_t *rows, int64_t *cols)
{
    if (nr0 == 0 || nc0 == 0) return 0;
    int transpose = nc0 < nr0;
    int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    size_t rs = transpose ? 1 : (size_t)nc0, cs = transpose ? (size_t)nc0 : 1;
    for (size_t k = 0; k < (size_t)nr0 * nc0; ++k)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) return -2;
    double *u = calloc(nr, sizeof(double)), *v = calloc(nc, sizeof(double)), *spc = malloc(sizeof(double) * nc);
    int64_t *path = malloc(sizeof(int64_t) * nc), *col4row = malloc(sizeof(int64_t) * nr);
    int64_t *row4col = malloc(sizeof(int64_t) * nc), *remaining = malloc(sizeof(int64_t) * nc);
    char *SR = malloc(nr), *SC = malloc(nc);
    for (int j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    int ret = nr;
    for (int64_t cur = 0; cur < nr; ++cur) {
        double minval;
        int64_t sink = aug_path(nc, cost, rs, cs, u, v, path, row4col, spc, cur, SR, SC, remaining, nr, &minval);
        if (sink < 0) { ret = -1; break; }
        u[cur] += minval;
        for (int i = 0; i < nr; ++i) if (SR[i] && i != cur) u[i] += minval - spc[col4row[i]];
        for (int j = 0; j < nc; ++j) if (SC[j]) v[j] -= minval - spc[j];
        int64_t j = sink;
        for (;;) {
            int64_t i = path[j];
            row4col[j] = i;
            int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    if (ret >= 0) {
        if (transpose) {
            /* a[i] = col4row[v], b[i] = v over v in argsort(col4row): i.e. pairs sorted by original row */
            int k = 0;
            for (int64_t r = 0; r < nc; ++r)          /* original row r == transposed column r */
                if (row4col[r] != -1) { rows[k] = r; cols[k] = row4col[r]; ++k; }
        } else {
            for (int i = 0; i < nr; ++i) { rows[i] = i; cols[i] = col4row[i]; }
        }
    }
    free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining); free(SR); free(SC);
    return ret;
}

/* lap.lapjv(cost, extend_cost=True, cost_limit=L) (byte_track/matching.py:37-48; third-party `lap`, restated from its documented
 * embedding): square (nr+nc) problem with L/2 padding and a zero lower-right block. x (nr) / y (nc): partner or -1. */
int orc_lapjv_limit(const double *cost, int nr, int nc, double cost_limit, int32_t *x, int32_t *y)
{
    for (int i = 0; i < nr; ++i) x[i] = -1;
    for (int j = 0; j < nc; ++j) y[j] = -1;
    if (nr == 0 || nc == 0) return 0;
    const int n = nr + nc;
    double *ext = malloc(sizeof(double) * (size_t)n * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c)
            ext[(size_t)r * n + c] = (r >= nr && c >= nc) ? 0.0 : ((r < nr && c < nc) ? cost[(size_t)r * nc + c] : cost_limit / 2.);
    int64_t *rows = malloc(sizeof(int64_t) * n), *cols = malloc(sizeof(int64_t) * n);
    const int np = orc_lsa(ext, n, n, rows, cols);
    int matched = 0;
    for (int k = 0; k < np; ++k)
        if (rows[k] < nr && cols[k] < nc) { x[rows[k]] = (int32_t)cols[k]; y[cols[k]] = (int32_t)rows[k]; ++matched; }
    free(ext); free(rows); free(cols);
    return np < 0 ? np : matched;
}
