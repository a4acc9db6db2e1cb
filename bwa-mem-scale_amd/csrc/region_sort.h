// region_sort.h — ksort.h's introsort (ks_introsort: median-of-3 quicksort, 16-element cut-off, combsort fallback,
// final insertion sort; /root/reference/src/ksort.h) over 24-byte sort records, operation by operation: the sort is
// unstable, so only the same sequence of comparisons and swaps leaves tied records in the reference's order.
// Used for the two sorts of mem_sort_dedup_patch (bwamem.cpp:176-180), by dedup.hip and pair.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bwams {
namespace {

struct SortRec { int64_t k; int32_t s, q, idx; int32_t pad_; };      // ars2: k = re; ars: k = rb, s = score, q = qb

struct LtEnd   { __device__ __forceinline__ bool operator()(const SortRec &a, const SortRec &b) const { return a.k < b.k; } };
struct LtScore { __device__ __forceinline__ bool operator()(const SortRec &a, const SortRec &b) const {
    return a.s > b.s || (a.s == b.s && (a.k < b.k || (a.k == b.k && a.q < b.q))); } };

// mem_mark_primary_se's two orders (alnreg_hlt / alnreg_hlt2, bwamem.cpp:182-186): k = hash (unsigned), s = score, q = is_alt
struct LtHash  { __device__ __forceinline__ bool operator()(const SortRec &a, const SortRec &b) const {
    return a.s > b.s || (a.s == b.s && (a.q < b.q || (a.q == b.q && (uint64_t)a.k < (uint64_t)b.k))); } };
struct LtHash2 { __device__ __forceinline__ bool operator()(const SortRec &a, const SortRec &b) const {
    return a.q < b.q || (a.q == b.q && (a.s > b.s || (a.s == b.s && (uint64_t)a.k < (uint64_t)b.k))); } };
// pair64_lt (utils.cpp:45) with x = k and y = (s, q), both halves non-negative
struct LtXY    { __device__ __forceinline__ bool operator()(const SortRec &a, const SortRec &b) const {
    return a.k < b.k || (a.k == b.k && (a.s < b.s || (a.s == b.s && a.q < b.q))); } };

// smem_lt_2 (bwamem.cpp:73): MEMs of the ERT walk by (start, end); k = start, s = end
struct LtStartEnd { __device__ __forceinline__ bool operator()(const SortRec &a, const SortRec &b) const {
    return a.k == b.k ? a.s < b.s : a.k < b.k; } };

template <class LT> __device__ __forceinline__ void r_insertsort(SortRec *a, int s, int t, LT lt) {
    for (int i = s + 1; i < t; ++i)
        for (int j = i; j > s && lt(a[j], a[j - 1]); --j) { const SortRec x = a[j]; a[j] = a[j - 1]; a[j - 1] = x; }
}
template <class LT> __device__ __forceinline__ void r_combsort(SortRec *a, int n, LT lt) {
    const double shrink = 1.2473309501039786540366528676643;
    bool do_swap;
    unsigned long long gap = (unsigned long long)n;
    do {
        if (gap > 2) {
            gap = (unsigned long long)((double)gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = false;
        for (long long i = 0; i < (long long)n - (long long)gap; ++i) {
            const long long j = i + (long long)gap;
            if (lt(a[j], a[i])) { const SortRec x = a[i]; a[i] = a[j]; a[j] = x; do_swap = true; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) r_insertsort(a, 0, n, lt);
}
template <class LT> __device__ __forceinline__ void r_introsort(SortRec *a, int n, LT lt) {
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { const SortRec x = a[0]; a[0] = a[1]; a[1] = x; } return; }
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d);
    int stk_l[40], stk_r[40], stk_d[40], top = 0;
    int s = 0, t = n - 1;
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { r_combsort(a + s, t - s + 1, lt); t = s; continue; }
            int i = s, j = t, k = i + ((j - i) >> 1) + 1;
            if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
            else k = lt(a[j], a[i]) ? i : j;
            const SortRec rp = a[k];
            if (k != t) { a[k] = a[t]; a[t] = rp; }
            for (;;) {
                do ++i; while (lt(a[i], rp));
                do --j; while (i <= j && lt(rp, a[j]));
                if (j <= i) break;
                const SortRec x = a[i]; a[i] = a[j]; a[j] = x;
            }
            { const SortRec x = a[i]; a[i] = a[t]; a[t] = x; }
            if (i - s > t - i) {
                if (i - s > 16) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) { r_insertsort(a, 0, n, lt); return; }
            --top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
        }
    }
}

// The sorts are called through this non-inlined wrapper: the pointer stays generic (LDS or HBM, flat accesses).
// Instantiated directly on a __shared__ array the inlined introsort spun forever on gfx950 (ROCm 7.2) for a
// six-record input that the same code sorts correctly through a generic pointer; see profiles/r01_notes.md.
__device__ __noinline__ void sort_records(SortRec *a, int n, int by_score) {
    switch (by_score) {
    case 0: r_introsort(a, n, LtEnd()); break;
    case 1: r_introsort(a, n, LtScore()); break;
    case 2: r_introsort(a, n, LtHash()); break;
    case 3: r_introsort(a, n, LtHash2()); break;
    case 4: r_introsort(a, n, LtXY()); break;
    default: r_introsort(a, n, LtStartEnd()); break;
    }
}

// The same sorts for a one-wavefront block with the records in LDS: every lane counts the records that sort before
// its own (a rank sort: O(n^2 / 64) LDS reads, no dependent chain), which is the unique sorted order — and hence
// ksort.h's — whenever no two records compare equal.  A lane-wide ballot checks that; if some do, lane 0 runs the
// operation-exact introsort on the untouched input instead.  All 64 lanes call this; a and tmp hold n records each.
template <class LT> __device__ __forceinline__ bool wave_rank_pass(const SortRec *a, SortRec *tmp, int n, int lane, LT lt) {
    bool tie = false;
    for (int ib = 0; ib < n; ib += 64) {
        const int i = ib + lane;
        int eq = 0;
        if (i < n) {
            const SortRec x = a[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const SortRec y = a[j];
                const bool l = lt(y, x);
                rank += l ? 1 : 0;
                eq += (!l && !lt(x, y)) ? 1 : 0;
            }
            tmp[rank] = x;                                   // collisions only with equal records (then tmp is not used)
        }
        tie = tie || (__ballot(eq > 1) != 0);
    }
    return tie;
}
__device__ __forceinline__ void wave_sort_records(SortRec *a, SortRec *tmp, int n, int by_score, int lane) {
    if (n < 2) return;
    const bool tie = by_score ? wave_rank_pass(a, tmp, n, lane, LtScore()) : wave_rank_pass(a, tmp, n, lane, LtEnd());
    __syncthreads();
    if (!tie) {
        for (int i = lane; i < n; i += 64) a[i] = tmp[i];
    } else if (lane == 0) sort_records(a, n, by_score);
    __syncthreads();
}

}  // namespace
}  // namespace bwams
