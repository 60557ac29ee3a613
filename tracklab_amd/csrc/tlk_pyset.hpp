// tlk_pyset.hpp -- iteration order of CPython 3.10's `list(set(a) - set(b))` for small non-negative ints, on the device.
//
// The StrongSORT-family matching cascades build their unmatched-track list that way
// (plugins/track/strong_sort/sort/linear_assignment.py:126-127, plugins/track/bpbreid_strong_sort/sort/linear_assignment.py:127-128:
// `unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))`). The list is the ROW order of the IoU / OKS stage
// that follows, so LSA-assigned-but-rejected pairs return to the unmatched lists in that order and tracks born in one frame get
// their ids in that order. CPython's order is ascending only while every key is below the table size of the result set; the
// association kernels therefore reproduce Objects/setobject.c: open addressing with 9 linear probes, perturb shift 5, growth to
// the first power of two above 4 x used once fill * 5 >= mask * 3, and set_difference's two forms (copy-and-discard when
// len(a) / 4 > len(b), rebuild otherwise). int keys hash to themselves (-1, the one remapped hash, never occurs).
//
// One lane runs the table emulation (a few dependent LDS accesses per key), and only when a key can wrap: while every key of a
// table is below its size each key sits in slot `key`, i.e. the table iterates ascending and the emulation is skipped.
#pragma once
#include "tlk_common.hpp"

namespace tlk {
namespace pyset {

constexpr int EMPTY = -1, DUMMY = -2;
constexpr int LINEAR_PROBES = 9, PERTURB_SHIFT = 5;

// table size after n set_add_key calls on an empty set (no dummies: fill == used): 8, 32 from the 5th key, 128 from the 19th, ...
__host__ __device__ inline unsigned size_after_adds(unsigned n)
{
    unsigned mask = 7;
    for (;;) {
        const unsigned f = (mask * 3 + 4) / 5;            // first fill with fill * 5 >= mask * 3
        if (n < f) return mask + 1;
        unsigned ns = 8;
        while (ns <= f * 4) ns <<= 1;
        mask = ns - 1;
    }
}
// table size of set_copy's target (make_new_set + set_merge into an empty set)
__host__ __device__ inline unsigned size_of_copy(unsigned n)
{
    if (n * 5 < 7 * 3) return 8;
    unsigned ns = 8;
    while (ns <= n * 2) ns <<= 1;
    return ns;
}
// ints of work space one emulation needs for at most max_keys keys: two tables (a, result), each with a resize target
__host__ __device__ inline unsigned table_capacity(unsigned max_keys)
{
    const unsigned a = size_after_adds(max_keys), c = size_of_copy(max_keys);
    return a > c ? a : c;
}

struct Set { int *t, *alt; unsigned mask, fill, used; };

__device__ inline void set_init(Set &s, int *b0, int *b1)
{
    s.t = b0; s.alt = b1; s.mask = 7; s.fill = s.used = 0;
    for (int i = 0; i < 8; ++i) b0[i] = EMPTY;
}
__device__ inline void insert_clean(int *t, unsigned mask, int key)        // set_insert_clean
{
    unsigned perturb = (unsigned)key, i = (unsigned)key & mask;
    for (;;) {
        unsigned e = i;
        int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
        do { if (t[e] == EMPTY) { t[e] = key; return; } e++; } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
__device__ inline void set_resize(Set &s, unsigned minused)                // set_table_resize
{
    unsigned ns = 8;
    while (ns <= minused) ns <<= 1;
    int *nt = s.alt;
    for (unsigned i = 0; i < ns; ++i) nt[i] = EMPTY;
    for (unsigned i = 0; i <= s.mask; ++i) if (s.t[i] >= 0) insert_clean(nt, ns - 1, s.t[i]);
    s.alt = s.t; s.t = nt; s.mask = ns - 1; s.fill = s.used;
}
__device__ inline void set_add(Set &s, int key)                            // set_add_entry
{
    unsigned perturb = (unsigned)key, i = (unsigned)key & s.mask;
    int freeslot = -1;
    for (;;) {
        unsigned e = i;
        int probes = (i + LINEAR_PROBES <= s.mask) ? LINEAR_PROBES : 0;
        do {
            const int v = s.t[e];
            if (v == EMPTY) {
                if (freeslot >= 0) { s.t[freeslot] = key; s.used++; return; }
                s.t[e] = key; s.fill++; s.used++;
                if (s.fill * 5 < s.mask * 3) return;
                set_resize(s, s.used * 4);
                return;
            }
            if (v == key) return;
            if (v == DUMMY) freeslot = (int)e;
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & s.mask;
    }
}
__device__ inline void set_discard(Set &s, int key)                        // set_discard_entry
{
    unsigned perturb = (unsigned)key, i = (unsigned)key & s.mask;
    for (;;) {
        unsigned e = i;
        int probes = (i + LINEAR_PROBES <= s.mask) ? LINEAR_PROBES : 0;
        do {
            const int v = s.t[e];
            if (v == EMPTY) return;
            if (v == key) { s.t[e] = DUMMY; s.used--; return; }
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & s.mask;
    }
}

// ONE thread: out[] = list(set(a) - set(b)) in CPython's order, a[0..na) distinct keys in insertion order, in_b[key] != 0 <=> key in b,
// nb = len(set(b)), b a subset of a. ws = 4 * cap ints (cap >= table_capacity(na)). Returns len(out).
// n_asc >= 0: out[0..n_asc) already holds the difference in ascending order (saves the membership sweep over a).
__device__ inline int difference_order_serial(const int *a, int na, const int *in_b, int nb, int *out, int *ws, unsigned cap, int n_asc = -1)
{
    Set A, R;
    // while every key of a is below the size its table ends up with, key k sits in slot k: the table iterates in a's own (ascending)
    // order and need not be built (the normal case -- what wraps is the small RESULT set); `ord(i)` = i-th key in a's iteration order
    const bool a_plain = na == 0 || (unsigned)a[na - 1] < size_after_adds((unsigned)na);
    unsigned a_mask = size_after_adds((unsigned)na) - 1, a_used = (unsigned)na;
    if (!a_plain) {
        set_init(A, ws, ws + cap);
        for (int i = 0; i < na; ++i) set_add(A, a[i]);
        a_mask = A.mask; a_used = A.used;
    }
    const unsigned a_slots = a_plain ? (unsigned)na : a_mask + 1;
    auto a_at = [&](unsigned i) { return a_plain ? a[i] : A.t[i]; };        // < 0: empty slot
    int n = 0;
    if ((a_used >> 2) > (unsigned)nb) {                                     // set_copy_and_difference
        set_init(R, ws + 2 * cap, ws + 3 * cap);
        if (a_used != 0) {
            if ((R.fill + a_used) * 5 >= R.mask * 3) {                      // set_merge's up-front resize of the (empty) target
                unsigned ns = 8;
                while (ns <= a_used * 2) ns <<= 1;
                for (unsigned i = 0; i < ns; ++i) R.t[i] = EMPTY;
                R.mask = ns - 1;
            }
            if (R.mask == a_mask && !a_plain) { for (unsigned i = 0; i <= a_mask; ++i) R.t[i] = A.t[i]; }
            else { for (unsigned i = 0; i < a_slots; ++i) { const int k = a_at(i); if (k >= 0) insert_clean(R.t, R.mask, k); } }    // (a plain table copies to the same slots either way)
            R.fill = R.used = a_used;
        }
        for (unsigned i = 0; i <= R.mask; ++i) if (R.t[i] >= 0 && !in_b[R.t[i]]) out[n++] = R.t[i];     // discards leave dummies in place
    } else {                                                                // set_difference: add every key of a that is not in b
        set_init(R, ws + 2 * cap, ws + 3 * cap);
        if (a_plain && n_asc >= 0) { for (int i = 0; i < n_asc; ++i) set_add(R, out[i]); }
        else for (unsigned i = 0; i < a_slots; ++i) { const int k = a_at(i); if (k >= 0 && !in_b[k]) set_add(R, k); }
        for (unsigned i = 0; i <= R.mask; ++i) if (R.t[i] >= 0) out[n++] = R.t[i];
    }
    return n;
}

// can the ascending list be used as is? asc[0..nu) = the difference ascending, a ascending with largest key amax
__device__ inline bool ascending_is_exact(int na, int amax, int nb, const int *asc, int nu)
{
    const unsigned sa = size_after_adds((unsigned)na);
    if ((unsigned)amax >= sa) return false;                                // a itself wraps: its iteration order is not ascending
    if (((unsigned)na >> 2) > (unsigned)nb) return (unsigned)amax < size_of_copy((unsigned)na);
    return nu == 0 || (unsigned)asc[nu - 1] < size_after_adds((unsigned)nu);
}

}  // namespace pyset
}  // namespace tlk
