/*
 * oracle/src/pyset.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Iteration order of `list(set(a) - set(b))` for small non-negative ints as CPython 3.10 produces it (Objects/setobject.c: open
 * addressing, 9 linear probes, perturb shift 5, growth x4, set_difference's copy-and-discard / rebuild split). The reference's
 * matching cascades build their unmatched-track lists that way (plugins/track/strong_sort/sort/linear_assignment.py:126,
 * bpbreid_strong_sort/sort/linear_assignment.py), and the order of that list is the ROW order of the IoU stage that follows, hence
 * the order in which LSA-assigned-but-rejected pairs return to the unmatched lists and tracks born in the same frame get their ids.
 * int keys hash to themselves; -1 (the only int whose hash is remapped) never occurs.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

#define PS_EMPTY (-1)
#define PS_DUMMY (-2)
#define LINEAR_PROBES 9
#define PERTURB_SHIFT 5

typedef struct { int *t; size_t mask; size_t fill, used; } pyset;

static void ps_init(pyset *s) { s->mask = 7; s->t = malloc(sizeof(int) * 8); for (int i = 0; i < 8; ++i) s->t[i] = PS_EMPTY; s->fill = s->used = 0; }
static void ps_free(pyset *s) { free(s->t); s->t = NULL; }

static void ps_insert_clean(int *t, size_t mask, int key)
{
    size_t perturb = (size_t)key, i = (size_t)key & mask;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
        do { if (t[e] == PS_EMPTY) { t[e] = key; return; } e++; } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
static void ps_resize(pyset *s, size_t minused)
{
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    int *nt = malloc(sizeof(int) * newsize);
    for (size_t i = 0; i < newsize; ++i) nt[i] = PS_EMPTY;
    for (size_t i = 0; i <= s->mask; ++i) if (s->t[i] >= 0) ps_insert_clean(nt, newsize - 1, s->t[i]);
    free(s->t);
    s->t = nt; s->mask = newsize - 1; s->fill = s->used;
}
static int ps_contains(const pyset *s, int key)
{
    size_t perturb = (size_t)key, i = (size_t)key & s->mask;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= s->mask) ? LINEAR_PROBES : 0;
        do { if (s->t[e] == PS_EMPTY) return 0; if (s->t[e] == key) return 1; e++; } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & s->mask;
    }
}
static void ps_add(pyset *s, int key)
{
    size_t perturb = (size_t)key, i = (size_t)key & s->mask;
    long freeslot = -1;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= s->mask) ? LINEAR_PROBES : 0;
        do {
            if (s->t[e] == PS_EMPTY) {
                if (freeslot >= 0) { s->t[freeslot] = key; s->used++; return; }
                s->t[e] = key; s->fill++; s->used++;
                if (s->fill * 5 < s->mask * 3) return;
                ps_resize(s, s->used > 50000 ? s->used * 2 : s->used * 4);
                return;
            }
            if (s->t[e] == key) return;
            if (s->t[e] == PS_DUMMY) freeslot = (long)e;
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & s->mask;
    }
}
static void ps_discard(pyset *s, int key)
{
    size_t perturb = (size_t)key, i = (size_t)key & s->mask;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= s->mask) ? LINEAR_PROBES : 0;
        do { if (s->t[e] == PS_EMPTY) return; if (s->t[e] == key) { s->t[e] = PS_DUMMY; s->used--; return; } e++; } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & s->mask;
    }
}
/* set_copy: make_new_set -> set_merge into an empty set */
static void ps_copy(pyset *dst, const pyset *src)
{
    ps_init(dst);
    if (src->used == 0) return;
    if ((dst->fill + src->used) * 5 >= dst->mask * 3) ps_resize(dst, (dst->used + src->used) * 2);
    if (dst->mask == src->mask && src->fill == src->used) { memcpy(dst->t, src->t, sizeof(int) * (src->mask + 1)); dst->fill = src->fill; dst->used = src->used; return; }
    dst->fill = dst->used = src->used;
    for (size_t i = 0; i <= src->mask; ++i) if (src->t[i] >= 0) ps_insert_clean(dst->t, dst->mask, src->t[i]);
}

/* out = list(set(a) - set(b)) with a, b given in the order Python would have iterated the source lists / generators; returns len */
int orc_pyset_difference_order(const int *a, int na, const int *b, int nb, int *out)
{
    pyset A, B, R;
    ps_init(&A); ps_init(&B);
    for (int i = 0; i < na; ++i) ps_add(&A, a[i]);
    for (int i = 0; i < nb; ++i) ps_add(&B, b[i]);
    int n = 0;
    if ((A.used >> 2) > B.used) {                          /* set_copy_and_difference */
        ps_copy(&R, &A);
        for (size_t i = 0; i <= B.mask; ++i) if (B.t[i] >= 0) ps_discard(&R, B.t[i]);
    } else {
        ps_init(&R);
        for (size_t i = 0; i <= A.mask; ++i) if (A.t[i] >= 0 && !ps_contains(&B, A.t[i])) ps_add(&R, A.t[i]);
    }
    for (size_t i = 0; i <= R.mask; ++i) if (R.t[i] >= 0) out[n++] = R.t[i];
    ps_free(&A); ps_free(&B); ps_free(&R);
    return n;
}

/* 1 (default): CPython 3.10's own iteration order -- what the reference does and what the HIP association kernels reproduce
 * (tracklab_amd/csrc/tlk_pyset.hpp); 0: ascending, kept as a switch for the tests that show where the two differ (once a track
 * index exceeds the table size of a small result set the order changes the rows of the IoU stage and with them the ids of
 * tracks born in one frame). */
static int g_python_set_order = 1;
void orc_set_python_set_order(int on) { g_python_set_order = on; }
int orc_get_python_set_order(void) { return g_python_set_order; }
