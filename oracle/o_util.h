/* TEST INFRASTRUCTURE ONLY (oracle).  Small C helpers for the CPU restatement:
 * growable arrays, strings, a port of libstdc++'s std::sort (system library,
 * GCC 11 <bits/stl_algo.h>/<bits/stl_heap.h>: the reference relies on its exact
 * permutation for ties at /root/reference/StrainCall/PartialOrderGraph.cpp:466,
 * NonparametricClustering.cpp:647,675 and StrainCall.cpp:1027), and
 * std::mt19937 + generate_canonical<double,53> (libstdc++ <bits/random.tcc>).
 */
#ifndef O_UTIL_H
#define O_UTIL_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld; /* DoubleL, /root/reference/StrainCall/PartialOrderGraph.hpp:229 */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(2); }
    return p;
}
static char *xstrdup(const char *s) {
    size_t n = strlen(s);
    char *p = (char *)xmalloc(n + 1);
    memcpy(p, s, n + 1);
    return p;
}
static char *xstrndup(const char *s, size_t n) {
    char *p = (char *)xmalloc(n + 1);
    memcpy(p, s, n);
    p[n] = 0;
    return p;
}
/* r = a + b (new string) */
static char *str_cat(const char *a, const char *b) {
    size_t la = strlen(a), lb = strlen(b);
    char *p = (char *)xmalloc(la + lb + 1);
    memcpy(p, a, la);
    memcpy(p + la, b, lb + 1);
    return p;
}
static char *str_char(char c) {
    char *p = (char *)xmalloc(2);
    p[0] = c; p[1] = 0;
    return p;
}

/* generic growable vector of fixed-size items */
#define VEC(T) struct { T *v; int n, cap; }
#define vec_init(a) do { (a).v = NULL; (a).n = 0; (a).cap = 0; } while (0)
#define vec_push(a, x) do { \
        if ((a).n == (a).cap) { (a).cap = (a).cap ? (a).cap * 2 : 8; \
            (a).v = xrealloc((a).v, sizeof(*(a).v) * (size_t)(a).cap); } \
        (a).v[(a).n++] = (x); } while (0)
#define vec_free(a) do { free((a).v); (a).v = NULL; (a).n = (a).cap = 0; } while (0)
#define vec_clear(a) do { (a).n = 0; } while (0)

typedef VEC(int) IntVec;
typedef VEC(void *) PtrVec;

/* ---------------------------------------------------------------------------
 * std::sort on a permutation.  `idx[0..n)` is permuted exactly as libstdc++
 * would permute a vector whose elements are compared by `less(ctx, a, b)`
 * (a, b are the VALUES stored in idx, i.e. the identities of the elements).
 * ------------------------------------------------------------------------- */
typedef int (*less_fn)(void *ctx, int a, int b);
typedef struct { int *a; less_fn less; void *ctx; } SortCtx;

static void ss_unguarded_linear_insert(SortCtx *s, int last) {
    int val = s->a[last];
    int next = last - 1;
    while (s->less(s->ctx, val, s->a[next])) {
        s->a[last] = s->a[next];
        last = next;
        --next;
    }
    s->a[last] = val;
}
static void ss_insertion_sort(SortCtx *s, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (s->less(s->ctx, s->a[i], s->a[first])) {
            int val = s->a[i];
            memmove(&s->a[first + 1], &s->a[first], sizeof(int) * (size_t)(i - first));
            s->a[first] = val;
        } else
            ss_unguarded_linear_insert(s, i);
    }
}
static void ss_push_heap(SortCtx *s, int first, int hole, int top, int value) {
    int parent = (hole - 1) / 2;
    while (hole > top && s->less(s->ctx, s->a[first + parent], value)) {
        s->a[first + hole] = s->a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    s->a[first + hole] = value;
}
static void ss_adjust_heap(SortCtx *s, int first, int hole, int len, int value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (s->less(s->ctx, s->a[first + child], s->a[first + child - 1])) child--;
        s->a[first + hole] = s->a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        s->a[first + hole] = s->a[first + child - 1];
        hole = child - 1;
    }
    ss_push_heap(s, first, hole, top, value);
}
static void ss_pop_heap(SortCtx *s, int first, int last, int result) {
    int value = s->a[result];
    s->a[result] = s->a[first];
    ss_adjust_heap(s, first, 0, last - first, value);
}
static void ss_make_heap(SortCtx *s, int first, int last) {
    if (last - first < 2) return;
    int len = last - first, parent = (len - 2) / 2;
    for (;;) {
        int value = s->a[first + parent];
        ss_adjust_heap(s, first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}
static void ss_partial_sort_all(SortCtx *s, int first, int last) {
    /* __partial_sort(first, last, last): heap_select then sort_heap */
    ss_make_heap(s, first, last);
    int l = last;
    while (l - first > 1) { --l; ss_pop_heap(s, first, l, l); }
}
static void ss_swap(SortCtx *s, int i, int j) { int t = s->a[i]; s->a[i] = s->a[j]; s->a[j] = t; }
static void ss_move_median_to_first(SortCtx *s, int result, int a, int b, int c) {
#define L(x, y) s->less(s->ctx, s->a[x], s->a[y])
    if (L(a, b)) {
        if (L(b, c)) ss_swap(s, result, b);
        else if (L(a, c)) ss_swap(s, result, c);
        else ss_swap(s, result, a);
    } else if (L(a, c)) ss_swap(s, result, a);
    else if (L(b, c)) ss_swap(s, result, c);
    else ss_swap(s, result, b);
#undef L
}
static int ss_unguarded_partition(SortCtx *s, int first, int last, int pivot) {
    for (;;) {
        while (s->less(s->ctx, s->a[first], s->a[pivot])) ++first;
        --last;
        while (s->less(s->ctx, s->a[pivot], s->a[last])) --last;
        if (!(first < last)) return first;
        ss_swap(s, first, last);
        ++first;
    }
}
static void ss_introsort_loop(SortCtx *s, int first, int last, int depth) {
    while (last - first > 16) {
        if (depth == 0) { ss_partial_sort_all(s, first, last); return; }
        --depth;
        int mid = first + (last - first) / 2;
        ss_move_median_to_first(s, first, first + 1, mid, last - 1);
        int cut = ss_unguarded_partition(s, first + 1, last, first);
        ss_introsort_loop(s, cut, last, depth);
        last = cut;
    }
}
static void std_sort_perm(int *idx, int n, less_fn less, void *ctx) {
    if (n <= 0) return;
    SortCtx s = { idx, less, ctx };
    int lg = 0;
    for (unsigned t = (unsigned)n; t > 1; t >>= 1) lg++;
    ss_introsort_loop(&s, 0, n, lg * 2);
    if (n > 16) {
        ss_insertion_sort(&s, 0, 16);
        for (int i = 16; i != n; ++i) ss_unguarded_linear_insert(&s, i);
    } else
        ss_insertion_sort(&s, 0, n);
}

/* ---------------------------------------------------------------------------
 * std::mt19937 and generate_canonical<double,53>
 * ------------------------------------------------------------------------- */
typedef struct { uint32_t x[624]; int p; } MT;
static void mt_seed(MT *m, uint32_t seed) {
    m->x[0] = seed;
    for (int i = 1; i < 624; i++)
        m->x[i] = 1812433253u * (m->x[i - 1] ^ (m->x[i - 1] >> 30)) + (uint32_t)i;
    m->p = 624;
}
static void mt_twist(MT *m) {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu;
    for (int k = 0; k < 624 - 397; ++k) {
        uint32_t y = (m->x[k] & UP) | (m->x[k + 1] & LO);
        m->x[k] = m->x[k + 397] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    }
    for (int k = 624 - 397; k < 623; ++k) {
        uint32_t y = (m->x[k] & UP) | (m->x[k + 1] & LO);
        m->x[k] = m->x[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    }
    uint32_t y = (m->x[623] & UP) | (m->x[0] & LO);
    m->x[623] = m->x[396] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    m->p = 0;
}
static uint32_t mt_next(MT *m) {
    if (m->p >= 624) mt_twist(m);
    uint32_t z = m->x[m->p++];
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    return z;
}
/* generate_canonical<double,53>(mt19937): two draws, sum = x0 + x1*2^32 in
 * double (rounded to nearest), divided by 2^64; clamp below 1. */
static double mt_canonical(MT *m) {
    double sum = 0.0, tmp = 1.0;
    for (int k = 2; k != 0; --k) {
        sum += (double)mt_next(m) * tmp;
        tmp *= 4294967296.0;
    }
    double r = sum / tmp;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}

#endif
