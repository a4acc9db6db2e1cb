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
template <class LT> __device__ __forceinline__ void r_introsort(SortRec *a, int n, LT lt, int depth0 = 0) {
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { const SortRec x = a[0]; a[0] = a[1]; a[1] = x; } return; }
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d);
    int stk_l[40], stk_r[40], stk_d[40], top = 0;
    int s = 0, t = n - 1;
    d <<= 1;
    if (depth0 > 0) d = depth0;               // tests: reach the comb-sort fallback on any input
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
__device__ __noinline__ void sort_records(SortRec *a, int n, int by_score, int depth0 = 0) {
    switch (by_score) {
    case 0: r_introsort(a, n, LtEnd(), depth0); break;
    case 1: r_introsort(a, n, LtScore(), depth0); break;
    case 2: r_introsort(a, n, LtHash()); break;
    case 3: r_introsort(a, n, LtHash2()); break;
    case 4: r_introsort(a, n, LtXY()); break;
    default: r_introsort(a, n, LtStartEnd()); break;
    }
}


// ks_combsort operation by operation with the whole wavefront (all 64 lanes call this; a in LDS, tmp = n records of LDS
// scratch).  A pass with gap g compares (i, i + g) for i = 0 .. n - g - 1 in order; position i + g may have been written by
// the comparison at i - g, never by any other, so the pass is g independent chains (the residue classes of i mod g), each
// walked in order by one lane: the same compare-and-swap sequence per chain, hence the same array after the pass.  Gaps shrink
// to 2 (two lanes), where the loop repeats until a pass swaps nothing; ks_combsort then finishes with its insertion sort — a
// stable sort of what the passes left, i.e. the rank sort below.  This is introsort's depth-limit fallback in the wave tiers:
// no lane sorts alone on LDS while 63 wait at a barrier (the configuration that once hung, profiles/r01_notes.md 20).
template <class LT> __device__ void wave_combsort(SortRec *a, int n, SortRec *tmp, int lane, LT lt) {
    const double shrink = 1.2473309501039786540366528676643;
    unsigned long long gap = (unsigned long long)n;
    bool do_swap;
    do {
        if (gap > 2) {
            gap = (unsigned long long)((double)gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        const int g = (int)gap;
        bool sw = false;
        for (int r = lane; r < g; r += 64)
            for (int i = r; i + g < n; i += g)
                if (lt(a[i + g], a[i])) { const SortRec x = a[i]; a[i] = a[i + g]; a[i + g] = x; sw = true; }
        __syncthreads();
        do_swap = __ballot(sw) != 0;
    } while (do_swap || gap > 2);
    if (gap != 1) {
        for (int x0 = 0; x0 < n; x0 += 64) {
            const int x = x0 + lane;
            if (x < n) {
                const SortRec v = a[x];
                int pos = 0;
                for (int y = 0; y < n; ++y) {
                    const SortRec w = a[y];
                    pos += (lt(w, v) || (!lt(v, w) && y < x)) ? 1 : 0;
                }
                tmp[pos] = v;
            }
        }
        __syncthreads();
        for (int x = lane; x < n; x += 64) a[x] = tmp[x];
        __syncthreads();
    }
}

// The same sorts for a one-wavefront block with the records in LDS: every lane counts the records that sort before
// its own (a rank sort: O(n^2 / 64) LDS reads, no dependent chain), which is the unique sorted order — and hence
// ksort.h's — whenever no two records compare equal.  A lane-wide ballot checks that; if some do, the wave runs the
// operation-exact introsort (wave_introsort) on the untouched input instead.  All 64 lanes call this; a and tmp hold n
// records each.  (Round 1 sent lane 0 alone into the sequential introsort here while 63 lanes waited at the barrier:
// the configuration in which an inlined LDS instantiation once hung, profiles/r01_notes.md 20.  No lane-0-only sort call
// on LDS is left in the wave tiers: the depth-limit fallback is wave_combsort above.)
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
// The same contract (tmp = the records in sorted order, returns whether two of them compare equal) for more than a few dozen records: a
// bitonic network over the next power of two P >= n (tmp holds P records; the pads, pad_ = 1, sort behind everything).  n log^2 n / 128
// compare-exchanges per lane instead of n^2 / 64 comparisons: a read in a satellite array reaches de-duplication with 500 .. 2000
// regions, and two rank sorts of those were a millisecond of a wavefront that has a CU to itself.
template <class LT> __device__ __forceinline__ bool wave_bitonic_pass(const SortRec *a, SortRec *tmp, int n, int P, int lane, LT lt) {
    for (int i = lane; i < P; i += 64) {
        SortRec x;
        if (i < n) x = a[i]; else { x.k = 0; x.s = 0; x.q = 0; x.idx = 0; x.pad_ = 1; }
        tmp[i] = x;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int e = lane; e < P; e += 64) {
                const int x = e ^ j;
                if (x > e) {
                    const SortRec A = tmp[e], B = tmp[x];
                    const bool b_lt_a = !B.pad_ && (A.pad_ || lt(B, A));
                    if (b_lt_a == ((e & k) == 0)) { tmp[e] = B; tmp[x] = A; }
                }
            }
            __syncthreads();
        }
    bool tie = false;
    for (int ib = 1; ib < n; ib += 64) {
        const int i = ib + lane;
        tie = tie || (__ballot(i < n && !lt(tmp[i - 1], tmp[i])) != 0);
    }
    return tie;
}
// ks_introsort operation by operation, but with the whole wavefront on every step (all 64 lanes call this; a in LDS,
// tmp = n records of LDS scratch, stk = 120 ints of LDS).  The Hoare partition of a range [s, t] around the pivot rp
// (moved to a[t]) is determined by two lists: the "up stoppers" (x in s+1..t, ascending, with !lt(a[x], rp)) and the
// "down stoppers" (x in t-1..s+1, descending, with !lt(rp, a[x])); the scalar loop swaps the k-th up stopper with the
// k-th down stopper while the former lies below the latter, m swaps in all, and the pivot lands on
// min(up[m], down[m-1]).  The lists are built with ballots, the swaps are independent.  The closing insertion sort over
// the whole array is a stable sort of what the partitions left: a rank sort (ksort's median of three never examines
// a[s], so the element may lie far from its place: a windowed clean-up would be wrong).  The comb-sort fallback of the
// depth limit is wave_combsort (whose scratch is the part of tmp the stopper lists do not need at that moment: they are dead).
// depth0 > 0 replaces the 2 ceil(log2 n) depth budget (tests reach the fallback with it).
template <class LT> __device__ void wave_introsort(SortRec *a, int n, SortRec *tmp, int *stk, int lane, LT lt, int by_score, int depth0 = 0) {
    if (n < 2) return;
    if (n == 2) {
        if (lane == 0 && lt(a[1], a[0])) { const SortRec x = a[0]; a[0] = a[1]; a[1] = x; }
        __syncthreads();
        return;
    }
    uint16_t *ls = reinterpret_cast<uint16_t *>(tmp), *rs = ls + n;          // 4 n bytes of the 24 n
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d);
    int top = 0, s = 0, t = n - 1;
    d <<= 1;
    if (depth0 > 0) d = depth0;
    (void)by_score;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (;;) {
        if (s < t) {
            if (--d == 0) {
                __syncthreads();
                wave_combsort(a + s, t - s + 1, tmp, lane, lt);
                t = s;
                continue;
            }
            int i = s, j = t, k = i + ((j - i) >> 1) + 1;
            {
                const SortRec ak = a[k], ai = a[i], aj = a[j];
                if (lt(ak, ai)) { if (lt(ak, aj)) k = j; }
                else k = lt(aj, ai) ? i : j;
            }
            const SortRec rp = a[k];
            __syncthreads();
            if (lane == 0 && k != t) { a[k] = a[t]; a[t] = rp; }
            __syncthreads();
            int NL = 0, NR = 0;
            for (int x0 = s + 1; x0 <= t; x0 += 64) {
                const int x = x0 + lane;
                const bool f = x <= t && !lt(a[x], rp);
                const unsigned long long m = __ballot(f);
                if (f) ls[NL + __popcll(m & below)] = (uint16_t)x;
                NL += __popcll(m);
            }
            for (int x0 = t - 1; x0 >= s + 1; x0 -= 64) {
                const int x = x0 - lane;
                const bool f = x >= s + 1 && !lt(rp, a[x]);
                const unsigned long long m = __ballot(f);
                if (f) rs[NR + __popcll(m & below)] = (uint16_t)x;
                NR += __popcll(m);
            }
            __syncthreads();
            const int np = NL < NR ? NL : NR;
            int m_sw = 0;
            for (int k0 = 0; k0 < np; k0 += 64) {
                const int kk = k0 + lane;
                m_sw += __popcll(__ballot(kk < np && ls[kk] < rs[kk]));
            }
            for (int k0 = 0; k0 < m_sw; k0 += 64) {
                const int kk = k0 + lane;
                if (kk < m_sw) { const int p = ls[kk], q = rs[kk]; const SortRec x = a[p]; a[p] = a[q]; a[q] = x; }
            }
            int i_f = ls[m_sw];                      // up[m] exists: t itself is an up stopper
            if (m_sw >= 1 && (int)rs[m_sw - 1] < i_f) i_f = rs[m_sw - 1];
            __syncthreads();
            if (lane == 0) { const SortRec x = a[i_f]; a[i_f] = a[t]; a[t] = x; }
            __syncthreads();
            i = i_f;
            if (i - s > t - i) {
                if (i - s > 16) { stk[3 * top] = s; stk[3 * top + 1] = i - 1; stk[3 * top + 2] = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { stk[3 * top] = i + 1; stk[3 * top + 1] = t; stk[3 * top + 2] = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
            __syncthreads();                         // the stack entries were written by every lane (same values)
        } else {
            if (top == 0) break;
            --top; s = stk[3 * top]; t = stk[3 * top + 1]; d = stk[3 * top + 2];
        }
    }
    __syncthreads();
    for (int x0 = 0; x0 < n; x0 += 64) {
        const int x = x0 + lane;
        if (x < n) {
            const SortRec v = a[x];
            int pos = 0;
            for (int y = 0; y < n; ++y) {
                const SortRec w = a[y];
                pos += (lt(w, v) || (!lt(v, w) && y < x)) ? 1 : 0;
            }
            tmp[pos] = v;                            // the stopper lists are dead by now
        }
    }
    __syncthreads();
    for (int x = lane; x < n; x += 64) a[x] = tmp[x];
    __syncthreads();
}

// force_exact: take the operation-exact path even when no two keys are equal (tests); depth0: see wave_introsort
// cap: records tmp holds (a power of two at or above n lets the bitonic pass run)
__device__ __forceinline__ void wave_sort_records(SortRec *a, SortRec *tmp, int n, int by_score, int lane, bool force_exact = false, int depth0 = 0,
                                                  int cap = 0) {
    __shared__ int l_sort_stk[120];
    if (n < 2) return;
    bool tie = true;
    int P = 128;
    while (P < n) P <<= 1;
    if (!force_exact) {
        if (n > 96 && P <= cap) tie = by_score ? wave_bitonic_pass(a, tmp, n, P, lane, LtScore()) : wave_bitonic_pass(a, tmp, n, P, lane, LtEnd());
        else tie = by_score ? wave_rank_pass(a, tmp, n, lane, LtScore()) : wave_rank_pass(a, tmp, n, lane, LtEnd());
    }
    __syncthreads();
    if (!tie) {
        for (int i = lane; i < n; i += 64) a[i] = tmp[i];
        __syncthreads();
    } else if (by_score) {
        wave_introsort(a, n, tmp, l_sort_stk, lane, LtScore(), 1, depth0);
    } else {
        wave_introsort(a, n, tmp, l_sort_stk, lane, LtEnd(), 0, depth0);
    }
}

}  // namespace
}  // namespace bwams
