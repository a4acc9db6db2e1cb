// chain.hip — seed chaining and chain filtering on the device.
//
// Replaces, for a whole chunk of reads, the reference's
//   mem_chain_seeds   /root/reference/src/bwamem.cpp:789-959  (+ test_and_merge :379-421)
//   mem_chain_flt     bwamem.cpp:528-646  (+ mem_chain_weight :451-470)
// consuming the SMEMs and SA coordinates that the seeding stage left in HBM.  mem_flt_chained_seeds
// (bwamem.cpp:491-526) acts only on reads of ~1100 bases and more: this file flags such reads, the
// re-scoring itself is seed_sw.hip.
//
// The work is sequential per read (each seed is tested against the chain found by an ordered
// lookup; then the read's chains are sorted and filtered pairwise) and reads are independent.
//   chain_kernel        one lane per read with few seeds: chaining, chain weights; for reads with
//                       few chains also the sort and the filter.
//   chain_wave_kernel   one wave per read with many seeds, heaviest reads first: the same code run
//                       in lockstep by 64 lanes (uniform loads = one request; lane 0 stores), the
//                       chain records and an ordered array of chain positions in LDS; a read in
//                       which a chain position repeats goes to chain_redo_kernel (B-tree, HBM).
//   chain_heavy_kernel  one wave per read with many chains: the sort runs on lane 0 over an LDS
//                       copy, the quadratic pairwise filter runs 64 kept chains at a time.
// All state lives in HBM scratch indexed by the read's slice of the SA-coordinate array: a seed
// IS an SA hit, so every per-seed array shares the index space of sa_coord, and a chain is named
// by its first seed.  The latency of dependent loads is what a lane pays for, so the hot records
// are laid out to be fetched whole: a B-tree node carries its keys' positions (160 B, ten 16-byte
// loads in flight at once, searched in registers), and a chain record carries everything
// test_and_merge and mem_chain_weight read (64 B; the weight is maintained incrementally).
//
// Two generic pieces decide tie cases and are therefore kept behaviour-identical to klib:
//   * the ordered map is a B-tree of order t = 5 (what kb_init(chn, 512 + 8) gives for the
//     48-byte mem_chain_t, kbtree.h:64) with kbtree.h's search, split and insert rules, so that
//     equal positions resolve to the same chain and the in-order traversal is the same; while
//     all positions of a read are distinct the tree's shape cannot matter, and the wave tiers
//     use a plain sorted array until one repeats;
//   * chains are sorted by weight with ksort.h's introsort (median-of-3, 16-element cut-off,
//     final insertion sort, comb-sort depth fallback), which is not stable.
#include "common.h"
#include "chain_kernels.h"
#include "wave_ops.h"

namespace bwams {
namespace {

constexpr int KB_T = 5;
constexpr int KB_MAXK = 2 * KB_T - 1;
constexpr int kLaneSeeds = 32;       // reads with more seeds than this are chained by a whole wave (chain_wave_kernel)
constexpr int kLightChains = 16;     // reads with more chains than this go to chain_heavy_kernel
struct alignas(16) Node {            // 160 B = ten 16-byte loads
    int16_t n, internal;
    int32_t ptr[KB_MAXK + 1];        // node ids relative to the read's node region.  32 bits: a read has no cap on its chains in the reference
    //                                  (kb_putp grows the tree, src/bwamem.cpp:830); 16-bit ids ended a read of > 131 k chains with an error
    int64_t pos[KB_MAXK];            // reference position of each key's chain (the sort key)
    int32_t key[KB_MAXK];            // chain ids (creation order within the read)
};
static_assert(sizeof(Node) == 160, "node layout");

struct alignas(16) ChainRec {        // 48 B, one per chain in creation order
    int64_t last_rbeg, endr;         // last seed's rbeg; running `end` of the reference-side weight
    uint16_t first_qbeg, last_qbeg, last_len, endq;  // query coordinates (reads hold at most 65534 bases: bwams_seed_upload)
    int32_t rid, n;                  // reference sequence, seeds
    int32_t first_idx, last_idx;     // first / last seed (index relative to the read's first SA hit)
    int32_t wq, wr;                  // query-/reference-side weights so far
};
static_assert(sizeof(ChainRec) == 48, "chain record layout");

// LDS bytes a wave needs to chain a read of at most K seeds: K chain records + the ordered array of
// (position, chain id), 12 B per chain
__host__ __device__ constexpr size_t lds_bytes(int K) { return (size_t)K * (sizeof(ChainRec) + 12) + 64; }
constexpr int kClassS = 128, kClassM1 = 256, kClassM = 512, kClassL1 = 850, kClassL = 1700;      // seeds per read: 7.7, 15, 31, 51, 102 KB of LDS
// two classes in between (round 3): a read holds its wavefront for milliseconds, so what a class costs is reads / (waves a CU's LDS admits):
// (512, 665] at 40 KB = four per CU instead of three, (850, 1275] at 76.6 KB = two instead of one
constexpr int kClassM2 = 665, kClassL2 = 1275;
static_assert(lds_bytes(kClassL) <= 160 * 1024, "class L must fit one CU's LDS");
// class XL: reads beyond class L keep only the ordered array (12 B per chain) in LDS, the chain records in HBM
constexpr int kClassXL = 4096;       // 49 KB
// class XL2 (round 4): on a repeat-rich genome reads reach 10^4 seeds (a dozen SMEMs with max_occ hits each); beyond class XL they went
// through the B-tree in HBM one seed at a time: 40 ms for a handful of reads, the stage's long pole on the grch38_like genome
constexpr int kClassXL2 = 13000;     // 156 KB: a CU's LDS
__host__ __device__ constexpr size_t lds_bytes_xl(int K) { return (size_t)K * 12 + 64; }

// bns_pos2rid / bns_intv2rid (bntseq.cpp:397-421) with a one-entry cache of the last sequence found
struct RidCache { int64_t lo, hi; int rid; };

__device__ __forceinline__ int pos2rid(const DevBns &b, int64_t pos_f, RidCache &rc) {
    if (pos_f >= b.l_pac) return -1;
    if (pos_f >= rc.lo && pos_f < rc.hi) return rc.rid;
    int left = 0, mid = 0, right = b.n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= b.contigs[mid].offset) {
            if (mid == b.n_seqs - 1) break;
            if (pos_f < b.contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    const int64_t off = b.contigs[mid].offset;
    if (pos_f >= off) { rc.lo = off; rc.hi = mid == b.n_seqs - 1 ? b.l_pac : b.contigs[mid + 1].offset; rc.rid = mid; }
    return mid;
}
__device__ __forceinline__ int64_t depos(const DevBns &b, int64_t pos) {
    return pos >= b.l_pac ? (b.l_pac << 1) - 1 - pos : pos;
}
__device__ __forceinline__ int intv2rid(const DevBns &b, int64_t rb, int64_t re, RidCache &rc) {
    if (rb < b.l_pac && re > b.l_pac) return -2;
    const int rid_b = pos2rid(b, depos(b, rb), rc);
    const int rid_e = rb < re ? pos2rid(b, depos(b, re - 1), rc) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

// The same for a wavefront working on one read: lane l keeps the offset of sequence l (or, with more than 64 sequences, of
// sequence 64 l) in a register, and a position's sequence is a ballot (plus one coalesced load for the second level)
// instead of a binary search of dependent global loads — seeds of a read in a repeat family jump between sequences, the
// one-entry cache misses, and those loads were most of the per-seed latency of the wave tiers.
struct WaveBns { int64_t off; };
__device__ __forceinline__ WaveBns wave_bns_load(const DevBns &b, int lane) {
    WaveBns w;
    const int stride = b.n_seqs <= 64 ? 1 : 64;
    const int64_t i = (int64_t)lane * stride;
    w.off = i < b.n_seqs ? b.contigs[i].offset : INT64_MAX;
    return w;
}
__device__ __forceinline__ int pos2rid_w(const DevBns &b, const WaveBns &w, int64_t pos_f, int lane) {
    if (pos_f >= b.l_pac) return -1;
    const int c = __popcll(__ballot(w.off <= pos_f)) - 1;          // offset[0] = 0 <= pos_f: c >= 0
    if (b.n_seqs <= 64) return c;
    const int i = c * 64 + lane;
    const int64_t o = i < b.n_seqs ? b.contigs[i].offset : INT64_MAX;
    return c * 64 + __popcll(__ballot(o <= pos_f)) - 1;
}
__device__ __forceinline__ int intv2rid_w(const DevBns &b, const WaveBns &w, int64_t rb, int64_t re, int lane) {
    if (rb < b.l_pac && re > b.l_pac) return -2;
    const int rid_b = pos2rid_w(b, w, depos(b, rb), lane);
    const int rid_e = rb < re ? pos2rid_w(b, w, depos(b, re - 1), lane) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

// per-read view of the scratch
struct ReadCtx {
    Node *nodes;
    ChainRec *crec;
    int32_t n_nodes, cap_nodes, root;
    int32_t n_keys;
    bool overflow;
    bool wr;                 // this lane performs the stores (lane-per-read: always; wave-per-read: lane 0)
};

__device__ __forceinline__ int32_t new_node(ReadCtx &c) {
    if (c.n_nodes >= c.cap_nodes) { c.overflow = true; return 0; }
    if (c.wr) { Node *x = &c.nodes[c.n_nodes]; x->n = 0; x->internal = 0; }
    return c.n_nodes++;
}

// Register copy of a node.  Filled field by field (independent loads, all in flight at once) rather
// than by a struct copy: a byte-wise copy of the mixed 16/32/64-bit layout is not promoted to
// registers by the compiler and would bounce through scratch memory.
struct NodeR {
    int32_t n, internal;
    int64_t pos[KB_MAXK];
    int32_t key[KB_MAXK];
    int32_t ptr[KB_MAXK + 1];
};
__device__ __forceinline__ void load_node(const Node *p, NodeR &x, bool want_ptr) {
    x.n = p->n; x.internal = p->internal;
#pragma unroll
    for (int t = 0; t < KB_MAXK; ++t) { x.pos[t] = p->pos[t]; x.key[t] = p->key[t]; }
#pragma unroll
    for (int t = 0; t <= KB_MAXK; ++t) x.ptr[t] = want_ptr ? (int32_t)p->ptr[t] : 0;
}

// __kb_getp_aux (kbtree.h:124-139) on a register copy of the node: keys are sorted, so the lower
// bound is the number of keys below k, and "found" means some key equals k
__device__ __forceinline__ int node_search(const NodeR &x, int64_t k, bool &eq) {
    int cnt = 0;
    bool e = false;
#pragma unroll
    for (int t = 0; t < KB_MAXK; ++t)
        if (t < x.n) { cnt += x.pos[t] < k ? 1 : 0; e |= x.pos[t] == k; }
    eq = e;
    return e ? cnt : cnt - 1;
}
__device__ __forceinline__ int32_t sel_key(const NodeR &x, int i) {
    int32_t v = x.key[0];
#pragma unroll
    for (int t = 1; t < KB_MAXK; ++t) v = i == t ? x.key[t] : v;
    return v;
}
__device__ __forceinline__ int64_t sel_pos(const NodeR &x, int i) {
    int64_t v = x.pos[0];
#pragma unroll
    for (int t = 1; t < KB_MAXK; ++t) v = i == t ? x.pos[t] : v;
    return v;
}
__device__ __forceinline__ int32_t sel_ptr(const NodeR &x, int i) {
    int32_t v = x.ptr[0];
#pragma unroll
    for (int t = 1; t <= KB_MAXK; ++t) v = i == t ? x.ptr[t] : v;
    return v;
}

// kb_intervalp (kbtree.h:159-176): the chain with the closest position <= k, or -1
__device__ __forceinline__ int32_t kbt_lower(const ReadCtx &c, int64_t k, int64_t &lower_pos) {
    int32_t lower = -1, xi = c.root;
    for (;;) {
        NodeR x;
        load_node(&c.nodes[xi], x, true);
        bool eq;
        const int i = node_search(x, k, eq);
        if (i >= 0) { lower = sel_key(x, i); lower_pos = sel_pos(x, i); }
        if (i >= 0 && eq) return lower;
        if (!x.internal) return lower;
        xi = sel_ptr(x, i + 1);
    }
}

// __kb_split (kbtree.h:183-199), in memory (one call per ~5 insertions)
__device__ __forceinline__ void kbt_split(ReadCtx &c, int32_t xi, int i, int32_t yi) {
    const int32_t zi = new_node(c);
    if (c.overflow || !c.wr) return;
    Node *x = &c.nodes[xi], *y = &c.nodes[yi], *z = &c.nodes[zi];
    z->internal = y->internal;
    z->n = KB_T - 1;
    for (int t = 0; t < KB_T - 1; ++t) { z->key[t] = y->key[KB_T + t]; z->pos[t] = y->pos[KB_T + t]; }
    if (y->internal) for (int t = 0; t < KB_T; ++t) z->ptr[t] = y->ptr[KB_T + t];
    y->n = KB_T - 1;
    const int xn = x->n;
    for (int t = xn; t > i; --t) x->ptr[t + 1] = x->ptr[t];
    x->ptr[i + 1] = zi;
    for (int t = xn - 1; t >= i; --t) { x->key[t + 1] = x->key[t]; x->pos[t + 1] = x->pos[t]; }
    x->key[i] = y->key[KB_T - 1];
    x->pos[i] = y->pos[KB_T - 1];
    x->n = (int16_t)(xn + 1);
}

// __kb_putp_aux / kb_putp (kbtree.h:200-233)
__device__ __forceinline__ void kbt_put(ReadCtx &c, int32_t id, int64_t k) {
    ++c.n_keys;
    int32_t xi = c.root;
    if (c.nodes[xi].n == KB_MAXK) {
        const int32_t s = new_node(c);
        if (c.overflow) return;
        if (c.wr) { c.nodes[s].internal = 1; c.nodes[s].n = 0; c.nodes[s].ptr[0] = xi; }
        c.root = s;
        kbt_split(c, s, 0, xi);
        if (c.overflow) return;
        xi = s;
    }
    for (;;) {
        Node *px = &c.nodes[xi];
        NodeR x;
        load_node(px, x, true);
        bool eq;
        if (!x.internal) {
            const int i = node_search(x, k, eq);          // insert after slot i
            if (c.wr) {
#pragma unroll
                for (int t = KB_MAXK - 1; t >= 1; --t)
                    if (t > i + 1 && t <= x.n) { px->key[t] = x.key[t - 1]; px->pos[t] = x.pos[t - 1]; }
#pragma unroll
                for (int t = 0; t < KB_MAXK; ++t)
                    if (t == i + 1) { px->key[t] = id; px->pos[t] = k; }
                px->n = (int16_t)(x.n + 1);
            }
            return;
        }
        int i = node_search(x, k, eq) + 1;
        int32_t ci = sel_ptr(x, i);
        if (c.nodes[ci].n == KB_MAXK) {
            const int64_t median = c.nodes[ci].pos[KB_T - 1];
            kbt_split(c, xi, i, ci);
            if (c.overflow) return;
            if (k > median) ci = c.nodes[xi].ptr[i + 1];      // chain_cmp(*k, x->key[i]) > 0: ++i
        }
        xi = ci;
    }
}

struct WPath { int32_t leaf; int16_t slot, n; bool ok; };     // what a lookup remembers for the insertion that may follow

// in-order traversal (__kb_traverse, kbtree.h:345-368) with an explicit stack; si = children already
// descended (height <= 16 covers 5^16 keys)
__device__ __forceinline__ int32_t kbt_traverse(const ReadCtx &c, int32_t *out) {
    int32_t sx[16];
    int32_t si[16];
    int sp = 0, n = 0;
    sx[0] = c.root; si[0] = 0;
    while (sp >= 0) {
        const Node *x = &c.nodes[sx[sp]];
        if (!x->internal) {
            for (int t = 0; t < x->n; ++t) { if (c.wr) out[n] = x->key[t]; ++n; }
            --sp;
            continue;
        }
        const int i = si[sp];
        if (i > 0 && i - 1 < x->n) { if (c.wr) out[n] = x->key[i - 1]; ++n; }
        if (i <= x->n && sp < 15) {
            si[sp] = i + 1;
            ++sp;
            sx[sp] = x->ptr[i]; si[sp] = 0;
        } else --sp;
    }
    return n;
}

// ---- ksort.h introsort over {w, id} pairs, descending w --------------------------------------
__device__ __forceinline__ bool flt_lt(uint2 a, uint2 b) { return a.x > b.x; }
__device__ __forceinline__ void swp(uint2 *a, int i, int j) { const uint2 t = a[i]; a[i] = a[j]; a[j] = t; }

__device__ __forceinline__ void flt_insertsort(uint2 *a, int s, int t) {
    for (int i = s + 1; i < t; ++i)
        for (int j = i; j > s && flt_lt(a[j], a[j - 1]); --j) swp(a, j, j - 1);
}
__device__ __forceinline__ void flt_combsort(uint2 *a, int n) {
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
            if (flt_lt(a[j], a[i])) { swp(a, (int)i, (int)j); do_swap = true; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) flt_insertsort(a, 0, n);
}
__device__ __forceinline__ void flt_introsort(uint2 *a, int n) {
    if (n < 1) return;
    if (n == 2) { if (flt_lt(a[1], a[0])) swp(a, 0, 1); return; }
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d);
    int stk_l[40], stk_r[40], stk_d[40], top = 0;
    int s = 0, t = n - 1;
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { flt_combsort(a + s, t - s + 1); t = s; continue; }
            int i = s, j = t, k = i + ((j - i) >> 1) + 1;
            if (flt_lt(a[k], a[i])) { if (flt_lt(a[k], a[j])) k = j; }
            else k = flt_lt(a[j], a[i]) ? i : j;
            const uint2 rp = a[k];
            if (k != t) swp(a, k, t);
            for (;;) {
                do ++i; while (flt_lt(a[i], rp));
                do --j; while (i <= j && flt_lt(rp, a[j]));
                if (j <= i) break;
                swp(a, i, j);
            }
            swp(a, i, t);
            if (i - s > t - i) {
                if (i - s > 16) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) { flt_insertsort(a, 0, n); return; }
            --top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
        }
    }
}

// ---- the same introsort, run by a whole wavefront over an LDS array -----------------------------------------
// ksort.h's introsort is not stable, so ties (equal weights are common) come out in an order that only the same
// sequence of operations reproduces — but that sequence need not be EXECUTED sequentially:
//   * Hoare's partition is a function of two flag vectors over the untouched range — "stops the upward scan"
//     (!lt(a[x], pivot)) and "stops the downward scan" (!lt(pivot, a[x])): its k-th swap exchanges the k-th upward stopper
//     with the k-th downward stopper for as long as the former lies left of the latter (positions already swapped are never
//     scanned again), and the pivot lands on the next upward stopper or on the last swapped downward position, whichever
//     comes first.  The stoppers are listed with ballots, the swaps are independent;
//   * the final insertion sort over the whole array is a STABLE sort, whose result is unique: a rank sort gives it.
// The control flow between partitions (median of three, explicit stack, depth budget, comb-sort fallback on lane 0) is
// ksort's own.  tmp: 2 * n uint16 (the stopper lists), stk: 3 * 40 ints, both LDS.  All 64 lanes call this.
__device__ void wave_flt_introsort(uint2 *a, int n, uint16_t *tmp, int *stk, int lane) {
    if (n < 2) return;
    if (n == 2) {
        if (lane == 0 && flt_lt(a[1], a[0])) swp(a, 0, 1);
        __syncthreads();
        return;
    }
    uint16_t *ls = tmp, *rs = tmp + n;
    int d;
    for (d = 2; (1ul << d) < (unsigned long)n; ++d);
    int top = 0, s = 0, t = n - 1;
    d <<= 1;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (;;) {
        if (s < t) {
            if (--d == 0) {
                if (lane == 0) flt_combsort(a + s, t - s + 1);
                __syncthreads();
                t = s;
                continue;
            }
            int i = s, j = t, k = i + ((j - i) >> 1) + 1;
            {
                const uint2 ak = a[k], ai = a[i], aj = a[j];
                if (flt_lt(ak, ai)) { if (flt_lt(ak, aj)) k = j; }
                else k = flt_lt(aj, ai) ? i : j;
            }
            const uint2 rp = a[k];
            __syncthreads();
            if (lane == 0 && k != t) { a[k] = a[t]; a[t] = rp; }
            __syncthreads();
            int NL = 0, NR = 0;
            for (int x0 = s + 1; x0 <= t; x0 += 64) {
                const int x = x0 + lane;
                const bool f = x <= t && !flt_lt(a[x], rp);
                const unsigned long long m = __ballot(f);
                if (f) ls[NL + __popcll(m & below)] = (uint16_t)x;
                NL += __popcll(m);
            }
            for (int x0 = t - 1; x0 >= s + 1; x0 -= 64) {
                const int x = x0 - lane;
                const bool f = x >= s + 1 && !flt_lt(rp, a[x]);
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
                if (kk < m_sw) swp(a, ls[kk], rs[kk]);
            }
            int i_f = ls[m_sw];
            if (m_sw >= 1 && (int)rs[m_sw - 1] < i_f) i_f = rs[m_sw - 1];
            __syncthreads();
            if (lane == 0) swp(a, i_f, t);
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
    // the closing insertion sort = THE stable sort of what the partitions left.  (Not a local clean-up: ksort's median of
    // three never examines a[s], which may lie beyond the pivot and then travels a long way in the insertion sort.)  Stable =
    // unique: up to 64 chains a rank sort (every lane counts, from uniform LDS reads, the elements that sort before its own);
    // beyond, a sorting network over the 64-bit keys (weight descending, position ascending — no two equal), n log^2 n / 128
    // compare-exchanges per lane instead of n^2 / 64 comparisons.  The network is the bitonic sorter in its standard form (every
    // comparator leaves the smaller key at the lower index), so the pads of a power of two are virtual: a comparator whose upper
    // end lies at or beyond n does nothing.  tmp: 16 * n bytes (keys, then the output copy).
    __syncthreads();
    uint2 *out = reinterpret_cast<uint2 *>(tmp);          // n * 8 bytes: the stopper lists are dead by now
    if (n > 64) {
        unsigned long long *key = reinterpret_cast<unsigned long long *>(tmp);
        out = reinterpret_cast<uint2 *>(key + n);
        for (int x = lane; x < n; x += 64) key[x] = ((unsigned long long)(0x7fffffffu - a[x].x) << 32) | (unsigned long long)(unsigned)x;
        __syncthreads();
        int P = 128;
        while (P < n) P <<= 1;
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool flip = j == (k >> 1);
                for (int c0 = 0; c0 < (P >> 1); c0 += 64) {
                    const int c = c0 + lane;
                    const int i = ((c & ~(j - 1)) << 1) | (c & (j - 1));        // j is a power of two
                    const int l = flip ? (i ^ (k - 1)) : i + j;
                    if (l < n) {
                        const unsigned long long x = key[i], y = key[l];
                        if (y < x) { key[i] = y; key[l] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for (int x = lane; x < n; x += 64) out[x] = a[(uint32_t)key[x]];
        __syncthreads();
    } else {
        const int x = lane;
        const uint2 v = x < n ? a[x] : make_uint2(0u, 0u);
        int pos = 0;
        for (int y = 0; y < n; ++y) {
            const uint32_t wy = a[y].x;
            pos += (wy > v.x || (wy == v.x && y < x)) ? 1 : 0;
        }
        if (x < n) out[pos] = v;
        __syncthreads();
    }
    for (int x = lane; x < n; x += 64) a[x] = out[x];
    __syncthreads();
}

// ---- the pairwise filter of mem_chain_flt, sequential form --------------------------------------
// fl[0..n_chn) sorted; rec[i] = {chn_beg, chn_end, w | is_alt << 31, first}.  Leaves kept[] set.
__device__ void filter_seq(const bwams_mem_opt_t &opt, int n_chn, uint4 *rec, int32_t *kept, int32_t *sel) {
    int n_sel = 0;
    kept[0] = 3;
    sel[n_sel++] = 0;
    for (int i = 1; i < n_chn; ++i) {
        bool large_ovlp = false;
        const uint4 ri = rec[i];
        const int bi = (int)ri.x, ei = (int)ri.y, wi = (int)(ri.z & 0x7fffffffu);
        const bool alt_i = (ri.z >> 31) != 0;
        int k;
        for (k = 0; k < n_sel; ++k) {
            const int j = sel[k];
            const uint4 rj = rec[j];
            const int bj = (int)rj.x, ej = (int)rj.y;
            const int b_max = bj > bi ? bj : bi;
            const int e_min = ej < ei ? ej : ei;
            const bool alt_j = (rj.z >> 31) != 0;
            if (e_min > b_max && (!alt_j || alt_i)) {
                const int li = ei - bi, lj = ej - bj;
                const int min_l = li < lj ? li : lj;
                if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
                    large_ovlp = true;
                    if ((int)rj.w < 0) rec[j].w = (uint32_t)i;
                    const int wj = (int)(rj.z & 0x7fffffffu);
                    if ((float)wi < (float)wj * opt.drop_ratio && wj - wi >= (opt.min_seed_len << 1)) break;
                }
            }
        }
        if (k == n_sel) {
            sel[n_sel++] = i;
            kept[i] = large_ovlp ? 2 : 3;
        }
    }
    for (int i = 0; i < n_sel; ++i) {
        const int f = (int)rec[sel[i]].w;
        if (f >= 0) kept[f] = 1;
    }
}

// max_chain_extend, compaction, per-read totals (bwamem.cpp:618-640); sequential
__device__ void finish_read(const ChainArgs &A, int64_t r, int64_t base, int n_chn, int L) {
    uint2 *fl = A.flt + base;
    uint4 *rec = A.f_rec + base;
    int32_t *kept = A.f_kept + base, *first = A.f_first + base;
    const ChainRec *crec = reinterpret_cast<const ChainRec *>(A.crec) + base;
    int i, k;
    for (i = k = 0; i < n_chn; ++i) {
        if (kept[i] == 0 || kept[i] == 3) continue;
        if (++k >= A.opt.max_chain_extend) break;
    }
    for (; i < n_chn; ++i)
        if (kept[i] < 3) kept[i] = 0;
    int n_seeds = 0;
    for (i = k = 0; i < n_chn; ++i) {
        if (kept[i] == 0) continue;
        const uint2 f = fl[i];
        const uint32_t alt = rec[i].z & 0x80000000u;
        fl[k] = make_uint2(f.x | ((unsigned)kept[i] << 29) | alt, f.y);
        first[k] = (int32_t)rec[i].w;
        n_seeds += crec[f.y].n;
        ++k;
    }
    A.n_kept[r] = k;
    A.n_kept_seeds[r] = n_seeds;
    // mem_flt_chained_seeds re-scores seeds only for long reads (min_l <= 0.05 * l_query)
    if (k) {
        const double min_l = A.opt.min_chain_weight ? (double)(1.1f * (float)A.opt.min_chain_weight) : (double)5.5f * log((double)L);
        if (!(min_l > (double)(0.05f * (float)L))) atomicAdd(&A.ctr->chain_longread, 1ull);
    }
}

__device__ __forceinline__ uint4 make_rec(const ChainArgs &A, const ChainRec &c, uint32_t w) {
    const uint32_t alt = A.bns.contigs[c.rid].is_alt != 0 ? 0x80000000u : 0u;
    return make_uint4((uint32_t)c.first_qbeg, (uint32_t)(c.last_qbeg + c.last_len), w | alt, 0xffffffffu);
}

// ---- the chaining kernel: one lane per read --------------------------------------------------
// lane per read: the read's slice of the sorted SMEM array and its seed count (the sort key that
// groups reads of similar cost into the same wave)
// a read's slice of the (rid, m, n)-sorted SMEM array from the array itself: lane per SMEM, a boundary where the read id changes (reads
// without SMEMs keep the zeroed [0, 0)).  A lane per read bisected the array twice: 46 dependent loads per read, 0.76 ms per million.
__global__ void chain_slice_kernel(ChainArgs A) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_smem) return;
    const int64_t r = (int64_t)A.smem[i].rid;
    if (i == 0 || (int64_t)A.smem[i - 1].rid != r) A.slice[2 * r] = i;
    if (i + 1 == A.n_smem || (int64_t)A.smem[i + 1].rid != r) A.slice[2 * r + 1] = i + 1;
}
__global__ void chain_count_kernel(ChainArgs A, uint32_t *keys, uint32_t *vals) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < A.nseq;
    int64_t cnt = 0;
    if (live) {
        const int64_t beg = A.slice[2 * r], end = A.slice[2 * r + 1];
        cnt = beg < end ? A.sa_off[end] - A.sa_off[beg] : 0;
        keys[r] = (uint32_t)(cnt < 0xffffffffll ? cnt : 0xffffffffll);
        vals[r] = (uint32_t)r;
    }
    // class boundaries in the descending order: [0, c[0]) > L, [c[0], c[1]) > L1, [c[1], c[2]) > M, [c[2], c[3]) > M1,
    // [c[3], c[4]) > S, [c[4], c[5]) > lane tier; [0, c[6]) beyond XL, [c[6], c[0]) XL; [c[0], c[7]) class L, [c[7], c[1]) class L2;
    // [c[1], c[8]) class L1, [c[8], c[2]) class M2; [0, c[9]) beyond XL2 (HBM), [c[9], c[6]) XL2.  One atomic per wave and class.
    const int thr[10] = {kClassL, kClassL1, kClassM, kClassM1, kClassS, kLaneSeeds, kClassXL, kClassL2, kClassM2, kClassXL2};
    if (!__ballot(live && cnt > kLaneSeeds)) return;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        const unsigned long long m = __ballot(live && cnt > thr[c]);
        if (m && (threadIdx.x & 63) == 0) atomicAdd(&A.ctr->chain_class[c], (unsigned long long)__popcll(m));
    }
}

// Chain one read.  `nl` lanes run this function in lockstep on the same read: 1 (a lane per read) or
// 64 (a wave per read: every load is wave-uniform, i.e. one request; lane 0 performs the stores).
// nodes / crec_w: where the B-tree and the chain records live while chaining — the read's HBM scratch,
// or LDS (then the records are copied out to crec_g at the end; the tree is not needed afterwards).
// LDS = true: nodes / crec_w point into LDS and every access through them must stay a ds_* instruction
// (a flat access would wait on vmcnt, i.e. on the round trip of every global store issued before it),
// so the pointers are never mixed with HBM pointers in this instantiation.
// ---- an ordered array instead of the B-tree (wave tiers) ---------------------------------------------------------
// While every chain position of a read is distinct, kbtree answers kb_intervalp with THE greatest key <= k and
// traverses in key order whatever its shape: a sorted array in LDS gives the same answers with two ballots per
// lookup (which 64-key chunk, which key in it) and a lane-parallel shift per insertion, in 12 B per chain instead of
// the nodes' ~45.  Equal positions are where the tree's shape starts to matter (which of the equal keys a lookup
// meets, where a duplicate lands): inserting a position that is already present makes the attempt give up, and the
// read is chained again with the B-tree (chain_redo_kernel).
__device__ __forceinline__ int sarr_lower(const int64_t *key, int n, int64_t k, int lane, bool &eq) {
    int base = 0;
    if (n > 4096) {                                                                 // a level above: 4096-key stretches (n <= 2^18)
        const int nsc = (n + 4095) >> 12;
        const unsigned long long ms = __ballot(lane < nsc && key[lane << 12] <= k);
        if (!ms) { eq = false; return -1; }
        base = (__popcll(ms) - 1) << 12;
        key += base;
        n = n - base < 4096 ? n - base : 4096;
    }
    const int nch = (n + 63) >> 6;
    const unsigned long long mc = __ballot(lane < nch && key[lane << 6] <= k);      // chunks whose first key is <= k
    if (!mc) { eq = false; return -1; }
    const int c = __popcll(mc) - 1;
    const int i = (c << 6) + lane;
    const int64_t my = i < n ? key[i] : 0;
    const unsigned long long mk = __ballot(i < n && my <= k);
    const int idx = base + (c << 6) + __popcll(mk) - 1;
    eq = __ballot(i < n && my == k) != 0;
    return idx;
}
__device__ __forceinline__ void sarr_insert(int64_t *key, int32_t *cid, int n, int at, int64_t k, int32_t id, int lane) {
    for (int base = ((n - at) >> 6) << 6; base >= 0; base -= 64) {                  // move [at, n) up by one, top chunk first
        const int i = at + base + lane;
        const bool mv = i < n;
        const int64_t kk = mv ? key[i] : 0;
        const int32_t cc = mv ? cid[i] : 0;
        if (mv) { key[i + 1] = kk; cid[i + 1] = cc; }
    }
    if (lane == 0) { key[at] = k; cid[at] = id; }
}

// ---- 64 seeds per pass (the wave tiers' ordered-array mode) ------------------------------------------------------------------------
// mem_chain_seeds takes a read's seeds one after the other: look the seed's position up, test_and_merge with the chain found, else a
// new chain.  One seed per trip of a wavefront is a chain of dependent LDS round trips (two for the lookup, the array entry, the chain
// record, the insertion's shifts: ~1700 cycles per seed, and a read in a repeat family has thousands).  But the hits of ONE SMEM rarely
// touch each other: they lie at different loci, each merges into the chain of its own locus or starts one.  So a pass takes 64 hits of
// an SMEM, a lane each — binary search, chain record, test_and_merge against the state BEFORE the pass — and then settles, in seed
// order, which of those decisions the sequential loop would have reached too:
//   * a seed whose looked-up chain an earlier seed of the pass has extended: not settled (its test must see the new last seed);
//   * a seed with a chain started by an earlier seed of the pass at a position between its looked-up one and its own: that chain is
//     its true predecessor.  It holds one seed of the same SMEM (same query span), so test_and_merge reduces to "same sequence, same
//     strand side, at most w further on": if that holds the seed would extend a chain born in this pass — not settled; if not, the
//     seed starts a chain of its own, whatever its looked-up chain said;
//   * otherwise the decision stands.
// The pass commits the seeds before the first unsettled one — extensions scatter to distinct chain records, the new chains take ids
// in seed order and their positions are merged into the ordered array in one sweep — and repeats with the rest.  Decisions and their
// order are the sequential loop's, so chains, ids and seed lists are identical.  A pass that settles a single seed costs more than
// the one-seed step, so after such a pass the next seeds are taken one at a time (tandem repeats: every hit extends the previous one's chain).
struct SeedCtx {
    bool crec_hbm;
    const int64_t *pos;
    int32_t *s_next;
    int2 *s_ql;
    ChainRec *crec;
    int64_t *s_key;
    int32_t *s_cid;
};
// one seed, the whole wave in lockstep: returns false when the read needs the B-tree
__device__ __forceinline__ bool chain_seed_one(const ChainArgs &A, const SeedCtx &S, int &n_keys, int32_t g, int64_t rbeg, int rid, int qbeg,
                                               int slen, int lane) {
    const bool wr = lane == 0;
    const int64_t l_pac = A.bns.l_pac;
    bool to_add = true, eq = false;
    int idx = -1;
    if (n_keys) {
        idx = sarr_lower(S.s_key, n_keys, rbeg, lane, eq);
        if (idx >= 0) {
            const int32_t lower = S.s_cid[idx];
            const int64_t fr = S.s_key[idx];
            ChainRec ch = S.crec[lower];
            const int64_t lr = ch.last_rbeg;
            const int64_t qend = ch.last_qbeg + ch.last_len, rend = lr + ch.last_len;
            if (rid != ch.rid) to_add = true;
            else if (qbeg >= ch.first_qbeg && qbeg + slen <= qend && rbeg >= fr && rbeg + slen <= rend) to_add = false;
            else if ((lr < l_pac || fr < l_pac) && rbeg >= l_pac) to_add = true;
            else {
                const int64_t x = qbeg - ch.last_qbeg, y = rbeg - lr;
                if (y >= 0 && x - y <= A.opt.w && y - x <= A.opt.w && x - ch.last_len < A.opt.max_chain_gap && y - ch.last_len < A.opt.max_chain_gap) {
                    if (wr) { S.s_ql[g] = make_int2(qbeg, slen); S.s_next[g] = -1; S.s_next[ch.last_idx] = g; }
                    if (qbeg >= ch.endq) ch.wq += slen; else if (qbeg + slen > ch.endq) ch.wq += qbeg + slen - ch.endq;
                    ch.endq = (uint16_t)((int)ch.endq > qbeg + slen ? (int)ch.endq : qbeg + slen);
                    if (rbeg >= ch.endr) ch.wr += slen; else if (rbeg + slen > ch.endr) ch.wr += (int)(rbeg + slen - ch.endr);
                    ch.endr = ch.endr > rbeg + slen ? ch.endr : rbeg + slen;
                    ch.last_rbeg = rbeg; ch.last_qbeg = (uint16_t)qbeg; ch.last_len = (uint16_t)slen; ch.last_idx = g; ch.n += 1;
                    if (wr) S.crec[lower] = ch;
                    to_add = false;
                }
            }
        }
    }
    if (to_add) {
        if (n_keys && eq) return false;                                           // a second chain at this position: the B-tree decides
        if (wr) {
            S.s_ql[g] = make_int2(qbeg, slen); S.s_next[g] = -1;
            ChainRec ch;
            ch.last_rbeg = rbeg; ch.endr = rbeg + slen;
            ch.first_qbeg = (uint16_t)qbeg; ch.last_qbeg = (uint16_t)qbeg; ch.last_len = (uint16_t)slen; ch.endq = (uint16_t)(qbeg + slen);
            ch.rid = rid; ch.n = 1; ch.first_idx = g; ch.last_idx = g; ch.wq = slen; ch.wr = slen;
            S.crec[n_keys] = ch;
        }
        sarr_insert(S.s_key, S.s_cid, n_keys, n_keys ? idx + 1 : 0, rbeg, n_keys, lane);
        ++n_keys;
    }
    return true;
}
__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
    return (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) |
                     (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)v, l));
}
// the read's seeds [0, cnt), 64 per pass across its SMEMs sm[beg .. end); all 64 lanes call this.  Returns false when the read needs the B-tree.
__device__ bool chain_seeds_batch(const ChainArgs &A, const SeedCtx &S, int &n_keys, const bwams_smem_t *sm, int64_t beg, int64_t end, int64_t base,
                                  int32_t cnt, RidCache &rc, const WaveBns &wb, int lane) {
    enum { NONE = 0, NOOP = 1, MERGE = 2, NEW = 3, UNSETTLED = 4 };
    const int64_t l_pac = A.bns.l_pac;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int w = A.opt.w, max_gap = A.opt.max_chain_gap;
    int64_t i_lo = beg;                                                           // the SMEM of the batch's first seed
#ifdef BWAMS_CHAINDBG
    unsigned long long tq[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#define TQ(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tq[k] += t_ - tq0; tq0 = t_; }
#else
#define TQ(k)
#endif
    // up to 64 SMEMs: lane l keeps SMEM l's first hit and query span (every dependent global load costs a microsecond here)
    const int nsm = (int)(end - beg);
    const bool sm_regs = nsm <= 64;
    int sm_start = INT32_MAX, sm_q = 0, sm_l = 0;
    if (sm_regs && lane < nsm) {
        sm_start = (int)(A.sa_off[beg + lane] - base);
        sm_q = (int)sm[beg + lane].m; sm_l = (int)sm[beg + lane].n + 1 - sm_q;
    }
    for (int32_t gb = 0; gb < cnt; gb += 64) {
        const int32_t g = gb + lane;
        bool pend = g < cnt;
        const int64_t rbeg = pend ? S.pos[g] : 0;
        // the seed's SMEM: the last one whose first hit is <= g
        int qbeg = 0, slen = 0;
        if (sm_regs) {                               // the SMEMs' first hits and spans sit in lanes: count, then fetch from the owner lane
            int k = 0;
            for (int i = 0; i < nsm; ++i) k += __builtin_amdgcn_readlane(sm_start, i) <= g ? 1 : 0;
            const int src = k > 0 ? k - 1 : 0;
            qbeg = __shfl(sm_q, src); slen = __shfl(sm_l, src);
        } else {
            int64_t lo = i_lo, hi = end - 1;                                      // answer in [lo, hi]
            if (pend) {
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) >> 1;
                    if (A.sa_off[mid] - base <= g) lo = mid; else hi = mid - 1;
                }
                qbeg = (int)sm[lo].m; slen = (int)sm[lo].n + 1 - qbeg;
            }
            i_lo = readlane64(lo, 0);
        }
        int rid = -1;
        if (A.bns.n_seqs <= 64) {                    // the sequences' offsets sit in lanes (wb): count the ones at or below both ends
            const int64_t rb = rbeg, re = rbeg + slen;   // (a uniform loop: every lane takes part in the readlanes; bns_intv2rid, bntseq.cpp:407-421)
            const int64_t pb = depos(A.bns, rb), pe = rb < re ? depos(A.bns, re - 1) : pb;
            int cb = 0, ce = 0;
            for (int i = 0; i < A.bns.n_seqs; ++i) {
                const int64_t o = readlane64(wb.off, i);
                cb += o <= pb ? 1 : 0; ce += o <= pe ? 1 : 0;
            }
            rid = (rb < l_pac && re > l_pac) ? -2 : cb == ce ? cb - 1 : -1;
        } else if (pend) rid = intv2rid(A.bns, rbeg, rbeg + slen, rc);
        pend = pend && rid >= 0;
        int one_by_one = 0;
        unsigned long long mp;
        TQ(0)
        while ((mp = __ballot(pend)) != 0ull) {
            if (one_by_one) {                                                     // the first pending seed, on the whole wave
                const int l0 = __builtin_ctzll(mp);
                if (!chain_seed_one(A, S, n_keys, gb + l0, readlane64(rbeg, l0), __builtin_amdgcn_readlane(rid, l0), __builtin_amdgcn_readlane(qbeg, l0),
                                    __builtin_amdgcn_readlane(slen, l0), lane)) return false;
                if (lane == l0) pend = false;
#ifdef BWAMS_CHAINDBG
                if (lane == 0) atomicAdd(&A.ctr->dbg[62], 1ull);
#endif
                --one_by_one;
                TQ(5)
                continue;
            }
            const int n = n_keys;
            // lookup: the greatest key <= rbeg, a binary search per lane
            int idx = -1;
            if (n) {
                int step = 1;
                while ((step << 1) <= n) step <<= 1;
                for (; step > 0; step >>= 1) {
                    const int t = idx + step;
                    if (pend && t < n && S.s_key[t] <= rbeg) idx = t;
                }
            }
            TQ(1)
            int64_t lo_key = INT64_MIN;
            int32_t lower = -1;
            int act = NONE;
            ChainRec ch;
            ch.last_rbeg = 0; ch.endr = 0; ch.first_qbeg = ch.last_qbeg = ch.last_len = ch.endq = 0; ch.rid = ch.n = ch.first_idx = ch.last_idx = ch.wq = ch.wr = 0;
            if (pend) {
                act = NEW;
                if (idx >= 0) {
                    lo_key = S.s_key[idx];
                    lower = S.s_cid[idx];
                    ch = S.crec[lower];
                    const int64_t lr = ch.last_rbeg;
                    const int64_t qend = ch.last_qbeg + ch.last_len, rend = lr + ch.last_len;
                    if (rid != ch.rid) act = NEW;
                    else if (qbeg >= ch.first_qbeg && qbeg + slen <= qend && rbeg >= lo_key && rbeg + slen <= rend) act = NOOP;
                    else if ((lr < l_pac || lo_key < l_pac) && rbeg >= l_pac) act = NEW;
                    else {
                        const int64_t x = qbeg - ch.last_qbeg, y = rbeg - lr;
                        if (y >= 0 && x - y <= w && y - x <= w && x - ch.last_len < max_gap && y - ch.last_len < max_gap) act = MERGE;
                    }
                }
            }
            const bool eq = pend && idx >= 0 && lo_key == rbeg;
            TQ(2)
            // settle in seed order
            int64_t best = lo_key;                   // the greatest position <= rbeg among the looked-up chain and the chains earlier seeds of this pass start
            int best_rid = -1, best_q = 0, best_l = 0;
            bool displaced = false, touched = false;
            int first_open = 64;
            // Two seeds of a pass can only touch each other when they look up the same place of the array (a chain started by the earlier
            // one lies between the later one's looked-up position and its own; or both look up the same chain): only those lanes go
            // through the ordered walk below, the others' decisions stand.  (The walk over all 64 lanes was most of a pass: 29 k cycles.)
            // (who shares: every pending lane counts itself into the top byte of its place's chain id — ids stay below 2^24 —, reads the
            // count back and takes itself out again: three LDS operations instead of a walk over the lanes)
            bool shared = false;
            {
                unsigned *mark = reinterpret_cast<unsigned *>(S.s_cid);
                const bool in_arr = pend && idx >= 0;
                if (in_arr) atomicAdd(&mark[idx], 1u << 24);
                if (in_arr) shared = (mark[idx] >> 24) > 1u;
                if (in_arr) atomicSub(&mark[idx], 1u << 24);
                const bool under = pend && idx < 0;
                const int n_under = __popcll(__ballot(under));
                if (under && n_under > 1) shared = true;
            }
            unsigned long long rem = __ballot(pend && shared);
            while (rem) {
                const int i = __builtin_ctzll(rem);
                rem &= rem - 1;
                int fin = act;
                if (displaced) {                     // test_and_merge against a chain of one seed {best_q, best_l, best}
                    const int64_t x = qbeg - best_q, y = rbeg - best;
                    if (rid != best_rid) fin = NEW;
                    else if (qbeg >= best_q && qbeg + slen <= best_q + best_l && rbeg + slen <= best + best_l) fin = NOOP;
                    else if (best < l_pac && rbeg >= l_pac) fin = NEW;
                    else fin = (x - y <= w && y - x <= w && x - best_l < max_gap && y - best_l < max_gap) ? UNSETTLED : NEW;
                } else if (touched) fin = UNSETTLED;
                const int a_i = __builtin_amdgcn_readlane(fin, i);
                if (a_i == UNSETTLED) { first_open = i; break; }
                if (lane == i) act = fin;
                if (a_i == NEW) {
                    const int64_t rb_i = readlane64(rbeg, i);
                    const int rid_i = __builtin_amdgcn_readlane(rid, i), q_i = __builtin_amdgcn_readlane(qbeg, i), l_i = __builtin_amdgcn_readlane(slen, i);
                    if (lane > i && rb_i <= rbeg && rb_i > best) { best = rb_i; best_rid = rid_i; best_q = q_i; best_l = l_i; displaced = true; }
                } else if (a_i == MERGE) {
                    const int low_i = __builtin_amdgcn_readlane(lower, i);
                    if (lane > i && lower == low_i) touched = true;
                }
            }
            const bool com = pend && lane < first_open;
            TQ(3)
            const unsigned long long m_new = __ballot(com && act == NEW);
            // a second chain at a position (in the array, or started earlier in this pass): the B-tree decides
            if (__ballot(com && act == NEW && (eq || (displaced && best == rbeg)))) return false;
            if (com && act == MERGE) {
                S.s_ql[g] = make_int2(qbeg, slen); S.s_next[g] = -1; S.s_next[ch.last_idx] = g;
                if (qbeg >= ch.endq) ch.wq += slen; else if (qbeg + slen > ch.endq) ch.wq += qbeg + slen - ch.endq;
                ch.endq = (uint16_t)((int)ch.endq > qbeg + slen ? (int)ch.endq : qbeg + slen);
                if (rbeg >= ch.endr) ch.wr += slen; else if (rbeg + slen > ch.endr) ch.wr += (int)(rbeg + slen - ch.endr);
                ch.endr = ch.endr > rbeg + slen ? ch.endr : rbeg + slen;
                ch.last_rbeg = rbeg; ch.last_qbeg = (uint16_t)qbeg; ch.last_len = (uint16_t)slen; ch.last_idx = g; ch.n += 1;
                S.crec[lower] = ch;
            }
            if (m_new) {
                const int m = __popcll(m_new);
                const bool mine = com && act == NEW;
                // the old entries move up by the number of new positions below them, top chunk first (a chunk is read whole before it is written)
                // (an entry's shift = the new positions below its chunk — one ballot — plus those inside the chunk below it: the few lanes
                // whose predecessor lies in this chunk; a loop over all new lanes per chunk cost 200 k cycles per pass at 8000 chains)
                int min_idx = mine ? idx : n;
                for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(min_idx, d); min_idx = o < min_idx ? o : min_idx; }
                for (int cb = n > 0 ? ((n - 1) >> 6) << 6 : -64; cb >= 0 && cb + 63 > min_idx; cb -= 64) {
                    const int p = cb + lane;
                    const bool mv = p < n;
                    const int64_t kk = mv ? S.s_key[p] : 0;
                    const int32_t cc = mv ? S.s_cid[p] : 0;
                    int sh = __popcll(__ballot(mine && idx < cb));
                    unsigned long long mm = __ballot(mine && idx >= cb && idx < cb + 64);
                    while (mm) { const int i = __builtin_ctzll(mm); mm &= mm - 1; sh += __builtin_amdgcn_readlane(idx, i) < p ? 1 : 0; }
                    if (mv && sh) { S.s_key[p + sh] = kk; S.s_cid[p + sh] = cc; }
                }
                int rank = 0;                        // new positions below mine: lower places of the array, or the same place and a lower position
                { unsigned long long mm = m_new; while (mm) { const int i = __builtin_ctzll(mm); mm &= mm - 1; rank += __builtin_amdgcn_readlane(idx, i) < idx ? 1 : 0; } }
                { unsigned long long mm = m_new & __ballot(shared); while (mm) { const int i = __builtin_ctzll(mm); mm &= mm - 1; rank += (__builtin_amdgcn_readlane(idx, i) == idx && readlane64(rbeg, i) < rbeg) ? 1 : 0; } }
                if (mine) {
                    const int32_t cid = n + __popcll(m_new & below);              // chains are numbered in creation order
                    S.s_ql[g] = make_int2(qbeg, slen); S.s_next[g] = -1;
                    ChainRec nc;
                    nc.last_rbeg = rbeg; nc.endr = rbeg + slen;
                    nc.first_qbeg = (uint16_t)qbeg; nc.last_qbeg = (uint16_t)qbeg; nc.last_len = (uint16_t)slen; nc.endq = (uint16_t)(qbeg + slen);
                    nc.rid = rid; nc.n = 1; nc.first_idx = g; nc.last_idx = g; nc.wq = slen; nc.wr = slen;
                    S.crec[cid] = nc;
                    S.s_key[idx + 1 + rank] = rbeg; S.s_cid[idx + 1 + rank] = cid;
                }
                n_keys = n + m;
            }
            if (com) pend = false;
#ifdef BWAMS_CHAINDBG
            if (lane == 0) { atomicAdd(&A.ctr->dbg[61], 1ull); atomicAdd(&A.ctr->dbg[63], (unsigned long long)__popcll(__ballot(com))); atomicAdd(&A.ctr->dbg[64], (unsigned long long)__popcll(m_new)); }
#endif
            if (__popcll(__ballot(com)) <= 1) one_by_one = 8;
            if (S.crec_hbm) __threadfence_block();                            // the chain records of class XL are read back through L2
            TQ(4)
        }
    }
#ifdef BWAMS_CHAINDBG
    if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&A.ctr->dbg[65 + k], tq[k]);
#endif
    return true;
}

__device__ void heavy_read(const ChainArgs &A, int64_t r, int64_t base, int n_chn, unsigned char *lds, int cap, int lane);
__host__ __device__ constexpr size_t heavy_lds_bytes(int cap);
// CONT = 0: kbtree (exact for any input); CONT = 1: sorted array, returns false when the read needs the B-tree
template <bool LDS, int CONT = 0, bool CREC_HBM = false>
__device__ __forceinline__ bool chain_read(const ChainArgs &A, int64_t r, int lane, int nl, Node *nodes, int32_t cap_nodes,
                                           ChainRec *crec_w, int32_t cap_chains) {
    const bool wr = lane == 0;
#ifdef BWAMS_CHAINDBG
    const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
    if (wr) {
        A.n_kept[r] = 0;
        A.n_kept_seeds[r] = 0;
        A.read_base[r] = 0;
        A.n_chn[r] = 0;
    }
    const bwams_smem_t *sm = A.smem;
    const int64_t beg = A.slice[2 * r], end = A.slice[2 * r + 1];
    const int L = (int)(A.cum[r + 1] - A.cum[r]);
    if (beg == end || L < A.opt.min_seed_len) return true;

    // frac_rep: query span covered by over-frequent SMEMs
    int b = 0, e = 0, l_rep = 0;
    if (nl == 64 && end - beg <= 64) {           // one coalesced load, then a scalar walk: a load per SMEM waited a round trip each
        const bool has = beg + lane < end;
        const int my_b = has ? (int)sm[beg + lane].m : 0, my_e = has ? (int)sm[beg + lane].n + 1 : 0;
        unsigned long long rep = __ballot(has && sm[beg + lane].s > (int64_t)A.opt.max_occ);
        while (rep) {
            const int i = __builtin_ctzll(rep);
            rep &= rep - 1;
            const int sb = __builtin_amdgcn_readlane(my_b, i), se = __builtin_amdgcn_readlane(my_e, i);
            if (sb > e) { l_rep += e - b; b = sb; e = se; }
            else e = e > se ? e : se;
        }
    } else
    for (int64_t i = beg; i < end; ++i) {
        const int sb = (int)sm[i].m, se = (int)sm[i].n + 1;
        if (sm[i].s <= (int64_t)A.opt.max_occ) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    if (wr) A.frac_rep[r] = (float)l_rep / (float)L;

    const int64_t base = A.sa_off[beg];
    const int32_t cnt = (int32_t)(A.sa_off[end] - base);
    if (wr) A.read_base[r] = base;
    if (cnt == 0) return true;
    const int64_t *pos = A.sa_coord + base;
    int32_t *s_next = A.s_next + base;
    int2 *s_ql = A.s_ql + base;
    ChainRec *crec_g = reinterpret_cast<ChainRec *>(A.crec) + base;
    ChainRec *crec;
    ReadCtx c;
    if constexpr (LDS) {
        crec = CREC_HBM ? crec_g : crec_w;
        c.nodes = nodes; c.cap_nodes = cap_nodes;
    } else {
        crec = crec_g;
        const int64_t nbase = (base >> 1) + 2 * r;
        c.nodes = reinterpret_cast<Node *>(A.nodes) + nbase;
        c.cap_nodes = (int32_t)((((base + cnt) >> 1) + 2 * (r + 1)) - nbase);
    }
    c.crec = crec;
    c.n_nodes = 0; c.n_keys = 0; c.overflow = false; c.wr = wr;
    int64_t *s_key = nullptr;            // CONT = 1: the ordered array lives where the nodes would
    int32_t *s_cid = nullptr;
    if constexpr (CONT == 1) {
        s_key = reinterpret_cast<int64_t *>(nodes);
        s_cid = reinterpret_cast<int32_t *>(s_key + cap_chains);
    } else c.root = new_node(c);
    RidCache rc;
    rc.lo = 0; rc.hi = -1; rc.rid = 0;
    WaveBns wb;
    wb.off = 0;
    if (nl == 64 && A.bns.n_seqs <= 4096) wb = wave_bns_load(A.bns, lane);
    if (LDS && cnt > cap_chains) { if (wr) atomicAdd(&A.ctr->chain_overflow, 1ull); return true; }

    const int64_t l_pac = A.bns.l_pac;
#ifdef BWAMS_CHAINDBG
    const unsigned long long tp1 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (CONT == 1) {
        if (nl == 64 && A.seed_batch) {          // 64 seeds per pass
            SeedCtx S;
            S.pos = pos; S.s_next = s_next; S.s_ql = s_ql; S.crec = crec; S.s_key = s_key; S.s_cid = s_cid; S.crec_hbm = CREC_HBM;
            int nk = 0;
            if (!chain_seeds_batch(A, S, nk, sm, beg, end, base, cnt, rc, wb, lane)) return false;
            c.n_keys = nk;
        }
    }
    if (!(CONT == 1 && nl == 64 && A.seed_batch))
    for (int64_t i = beg; i < end; ++i) {
        const int qbeg = (int)sm[i].m, slen = (int)sm[i].n + 1 - (int)sm[i].m;
        const int32_t g0 = (int32_t)(A.sa_off[i] - base), g1 = (int32_t)(A.sa_off[i + 1] - base);
        int64_t pos64 = 0;                       // wave mode: the positions of 64 seeds, one per lane
        for (int32_t g = g0; g < g1; ++g) {
            int64_t rbeg;
            if (nl == 64) {
                // one load per 64 seeds: a load per seed would wait (vmcnt) for the round trip of the
                // stores the previous seed issued
                const int j = (g - g0) & 63;
                if (j == 0) pos64 = g + lane < g1 ? pos[g + lane] : 0;
                rbeg = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(pos64 >> 32), j) << 32) |
                                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)pos64, j));
            } else rbeg = pos[g];
            const int rid = nl == 64 && A.bns.n_seqs <= 4096 ? intv2rid_w(A.bns, wb, rbeg, rbeg + slen, lane)
                                                               : intv2rid(A.bns, rbeg, rbeg + slen, rc);
            if (rid < 0) continue;
            bool to_add = true;
            WPath path;
            path.ok = false;
            if (c.n_keys) {
                int64_t fr = 0;
                int32_t lower;
                if constexpr (CONT == 1) {
                    bool eq;
                    const int idx = sarr_lower(s_key, c.n_keys, rbeg, lane, eq);
                    lower = -1;
                    if (idx >= 0) { lower = s_cid[idx]; fr = s_key[idx]; }
                    path.slot = (int16_t)0; path.leaf = idx; path.ok = !eq;          // leaf: where the key would go - 1; ok: not present yet
                } else lower = kbt_lower(c, rbeg, fr);
                if (lower >= 0) {                                        // test_and_merge
                    ChainRec ch = crec[lower];
                    const int64_t lr = ch.last_rbeg;
                    const int64_t qend = ch.last_qbeg + ch.last_len, rend = lr + ch.last_len;
                    if (rid != ch.rid) to_add = true;
                    else if (qbeg >= ch.first_qbeg && qbeg + slen <= qend && rbeg >= fr && rbeg + slen <= rend) to_add = false;   // contained
                    else if ((lr < l_pac || fr < l_pac) && rbeg >= l_pac) to_add = true;
                    else {
                        const int64_t x = qbeg - ch.last_qbeg, y = rbeg - lr;
                        if (y >= 0 && x - y <= A.opt.w && y - x <= A.opt.w && x - ch.last_len < A.opt.max_chain_gap &&
                            y - ch.last_len < A.opt.max_chain_gap) {
                            if (wr) {
                                s_ql[g] = make_int2(qbeg, slen);
                                s_next[g] = -1;
                                s_next[ch.last_idx] = g;
                            }
                            // mem_chain_weight's two running sums, one seed further
                            if (qbeg >= ch.endq) ch.wq += slen; else if (qbeg + slen > ch.endq) ch.wq += qbeg + slen - ch.endq;
                            ch.endq = (uint16_t)((int)ch.endq > qbeg + slen ? (int)ch.endq : qbeg + slen);
                            if (rbeg >= ch.endr) ch.wr += slen; else if (rbeg + slen > ch.endr) ch.wr += (int)(rbeg + slen - ch.endr);
                            ch.endr = ch.endr > rbeg + slen ? ch.endr : rbeg + slen;
                            ch.last_rbeg = rbeg; ch.last_qbeg = (uint16_t)qbeg; ch.last_len = (uint16_t)slen; ch.last_idx = g; ch.n += 1;
                            if (wr) crec[lower] = ch;
                            to_add = false;
                        }
                    }
                }
            }
            if (to_add) {
                if (wr) { s_ql[g] = make_int2(qbeg, slen); s_next[g] = -1; }
                ChainRec ch;
                ch.last_rbeg = rbeg; ch.endr = rbeg + slen;
                ch.first_qbeg = (uint16_t)qbeg; ch.last_qbeg = (uint16_t)qbeg; ch.last_len = (uint16_t)slen; ch.endq = (uint16_t)(qbeg + slen);
                ch.rid = rid; ch.n = 1; ch.first_idx = g; ch.last_idx = g; ch.wq = slen; ch.wr = slen;
                const int32_t cid = c.n_keys;                            // chains are numbered in creation order
                if (wr) crec[cid] = ch;
                if constexpr (CONT == 1) {
                    if (c.n_keys && !path.ok) return false;                          // a second chain at this position: the B-tree decides
                    sarr_insert(s_key, s_cid, c.n_keys, c.n_keys ? path.leaf + 1 : 0, rbeg, cid, lane);
                    ++c.n_keys;
                } else kbt_put(c, cid, rbeg);
                if (c.overflow) { if (wr) atomicAdd(&A.ctr->chain_overflow, 1ull); return true; }
            }
        }
    }
    if (c.n_keys == 0) return true;
#ifdef BWAMS_CHAINDBG
    const unsigned long long tp2 = __builtin_amdgcn_s_memtime();
#endif

    // chains in B-tree order, their weights, the weight floor
    int32_t *ord = A.f_first + base;
    int32_t n_trav;
    if constexpr (CONT == 1) {
        n_trav = c.n_keys;
        for (int32_t t = lane; t < n_trav; t += nl) ord[t] = s_cid[t];
        __threadfence_block();                                                       // ord is read back by every lane below
    } else n_trav = kbt_traverse(c, ord);
    uint2 *fl = A.flt + base;
    int n_chn = 0;
    if (nl == 64) {                             // 64 chains per trip: a chain per trip waited for ord[t] and crec[id] one after the other
        uint2 first = make_uint2(0u, 0u);
        for (int32_t t0 = 0; t0 < n_trav; t0 += 64) {
            const int32_t t = t0 + lane;
            const bool ok = t < n_trav;
            int32_t id = 0;
            if (ok) { if constexpr (CONT == 1) id = s_cid[t]; else id = ord[t]; }
            const ChainRec *pc = &crec[id];
            int w = pc->wr < pc->wq ? pc->wr : pc->wq;
            w = w < (1 << 30) ? w : (1 << 30) - 1;
            if (t0 == 0) first = make_uint2((unsigned)__builtin_amdgcn_readfirstlane(w), (unsigned)__builtin_amdgcn_readfirstlane(id));
            const bool keep = ok && w >= A.opt.min_chain_weight;
            const unsigned long long m = __ballot(keep);
            if (keep) fl[n_chn + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2((unsigned)w, (unsigned)id);
            n_chn += __popcll(m);
        }
        if (n_chn == 0 && wr) fl[0] = first;    // a_[0] stays in place when everything is dropped
        __threadfence_block();                  // fl is read back by lane 0 (light reads) below
    } else
    for (int32_t t = 0; t < n_trav; ++t) {
        const int32_t id = ord[t];
        const ChainRec ch = crec[id];
        int w = ch.wr < ch.wq ? ch.wr : ch.wq;
        w = w < (1 << 30) ? w : (1 << 30) - 1;
        if (t == 0 && wr) fl[0] = make_uint2((unsigned)w, (unsigned)id);      // a_[0] stays in place when everything is dropped
        if (w < A.opt.min_chain_weight) continue;
        if (wr) fl[n_chn] = make_uint2((unsigned)w, (unsigned)id);
        ++n_chn;
    }
    if (n_chn == 0) n_chn = 1;                  // the reference keeps a_[0] in that case (bwamem.cpp:549-572)
    if constexpr (LDS && !CREC_HBM)             // chain records out of LDS, for the filter and the emit stages
        for (int32_t k = lane; k < c.n_keys; k += nl) crec_g[k] = crec_w[k];
    if constexpr (LDS) {
        // the read's LDS is free now (chain records copied out, the ordered array dead): its sort and filter run here, on the same wave,
        // when they fit — no second kernel, no lane-0 loops through HBM for the reads with a dozen chains
        if (nl == 64) {
#ifdef BWAMS_CHAINDBG
            if (wr) {
                const unsigned long long tp3 = __builtin_amdgcn_s_memtime();
                atomicAdd(&A.ctr->dbg[56], tp1 - tp0); atomicAdd(&A.ctr->dbg[57], tp2 - tp1); atomicAdd(&A.ctr->dbg[58], tp3 - tp2);
                atomicAdd(&A.ctr->dbg[59], 1ull); atomicAdd(&A.ctr->dbg[60], (unsigned long long)cnt);
            }
#endif
            unsigned char *region = CREC_HBM ? reinterpret_cast<unsigned char *>(nodes) : reinterpret_cast<unsigned char *>(crec_w);
            const size_t bytes = (size_t)cap_chains * (CREC_HBM ? 12 : sizeof(ChainRec) + 12);
            const int cap_f = (int)((bytes - 64) / 41);
            if (cap_f >= 64 && n_chn <= cap_f) {
                if (wr) A.n_chn[r] = n_chn;
                __threadfence_block();          // crec_g, fl: written above by other lanes
                heavy_read(A, r, base, n_chn, region, cap_f, lane);
                return true;
            }
        }
    }
    if (!wr) return true;
    A.n_chn[r] = n_chn;
    if (n_chn > kLightChains) {                 // sort + filter by a whole wave (chain_heavy_kernel)
        const unsigned long long slot = atomicAdd(&A.ctr->n_heavy, 1ull);
        A.heavy[slot] = (int32_t)r;
        return true;
    }
    flt_introsort(fl, n_chn);
    uint4 *rec = A.f_rec + base;
    int32_t *kept = A.f_kept + base;
    for (int i = 0; i < n_chn; ++i) {
        rec[i] = make_rec(A, crec[fl[i].y], fl[i].x);
        kept[i] = 0;
    }
    filter_seq(A.opt, n_chn, rec, kept, A.f_sel + base);
    if constexpr (LDS) __threadfence_block();   // crec_g was written by the other lanes just above
    finish_read(A, r, base, n_chn, L);
    return true;
}

// reads with few seeds: one lane per read, state in HBM scratch
__global__ __launch_bounds__(64) void chain_kernel(ChainArgs A, const uint32_t *__restrict__ n_seeds) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    if (n_seeds[r] > (uint32_t)kLaneSeeds) return;          // a wave kernel's
    chain_read<false>(A, r, 0, 1, nullptr, 0, nullptr, 0);
}

// reads with many seeds: one wave (= one block) per read, B-tree and chain records in LDS.  The reads
// order[lo .. hi) (sorted by descending seed count, so a size class is a range) are handed out by ticket.
// K = 0: no LDS (reads too large for a CU's LDS): state in HBM scratch.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void chain_wave_kernel(ChainArgs A, const unsigned long long *lo_p, const unsigned long long *hi_p,
                                                        unsigned long long *ticket, int K) {
    extern __shared__ __align__(16) unsigned char l_mem[];
    const int lane = threadIdx.x;
    const int64_t lo = lo_p ? (int64_t)*lo_p : 0, hi = (int64_t)*hi_p;
    ChainRec *l_crec = reinterpret_cast<ChainRec *>(l_mem);
    Node *l_nodes = reinterpret_cast<Node *>(l_mem + (size_t)(K > 0 ? K : 0) * sizeof(ChainRec));
    for (;;) {
        const unsigned long long t = wave_ticket(ticket, 1ull);
        if (lo + (int64_t)t >= hi) break;
        const int64_t r = (int64_t)A.order[lo + (int64_t)t];
#ifdef BWAMS_CHAINDBG
        const unsigned long long tk0 = wall_clock64();
#endif
        if (K < 0) {                 // class XL: ordered array of -K chains in LDS, chain records in HBM
            if (!chain_read<true, 1, true>(A, r, lane, 64, reinterpret_cast<Node *>(l_mem), 0, nullptr, -K) && lane == 0)
                A.redo[atomicAdd(&A.ctr->chain_redo, 1ull)] = (int32_t)r;
        } else if (K) {
            if (!chain_read<true, 1>(A, r, lane, 64, l_nodes, 0, l_crec, K) && lane == 0)
                A.redo[atomicAdd(&A.ctr->chain_redo, 1ull)] = (int32_t)r;
        } else chain_read<false>(A, r, lane, 64, nullptr, 0, nullptr, 0);
        __syncthreads();
#ifdef BWAMS_CHAINDBG
        if (lane == 0) {
            const int c = K < 0 ? 0 : K == kClassL ? 1 : K == kClassL2 ? 2 : K == kClassL1 ? 3 : K == kClassM2 ? 4 : K == kClassM ? 5 : K == kClassM1 ? 6 : 7;
            const unsigned long long dt = wall_clock64() - tk0;
            atomicAdd(&A.ctr->dbg[32 + 3 * c], 1ull); atomicAdd(&A.ctr->dbg[33 + 3 * c], dt); atomicMax(&A.ctr->dbg[34 + 3 * c], dt);
        }
#endif
    }
}

// reads whose chain positions repeat: again, with the B-tree (state in HBM scratch, wave per read)
__global__ __launch_bounds__(64) void chain_redo_kernel(ChainArgs A) {
    const int lane = threadIdx.x;
    const int64_t n = (int64_t)A.ctr->chain_redo;
    for (;;) {
        const int64_t t = (int64_t)wave_ticket(&A.ctr->chain_redo_ticket, 1ull);
        if (t >= n) break;
        chain_read<false>(A, A.redo[t], lane, 64, nullptr, 0, nullptr, 0);
    }
}

// ---- reads with many chains: one wave per read ---------------------------------------------------
// Size classes (chains per read): each class is its own launch with the LDS its largest read needs — 41 B per chain —
// so that the many reads with a few dozen chains run 12 waves to a CU instead of sharing the footprint of the rare read
// with thousands (reads in repeat families reach max_occ hits per SMEM: on a GRCh38-size index a read can carry > 2000
// chains, and the quadratic filter of such a read on one lane through HBM took 40 ms).  Every launch walks the whole
// list of many-chain reads with its own ticket counter and skips the other classes' reads.
constexpr int kHeavyCap[6] = {64, 128, 256, 512, 960, 3840};         // 2.7 / 5 / 10.5 / 21 / 39 / 157 KB of LDS
__host__ __device__ constexpr size_t heavy_lds_bytes(int cap) { return (size_t)cap * 41 + 64; }
static_assert(heavy_lds_bytes(kHeavyCap[5]) <= 160 * 1024, "the largest class must fit one CU's LDS");

// one read's sort, pairwise filter and totals by a whole wavefront.  lds: heavy_lds_bytes(cap) bytes, n_chn <= cap; fl[0 .. n_chn) in HBM
// holds the {weight, chain id} pairs in B-tree order.  All 64 lanes call this.
__device__ void heavy_read(const ChainArgs &A, int64_t r, int64_t base, int n_chn, unsigned char *lds, int cap, int lane) {
    uint4 *l_rec = reinterpret_cast<uint4 *>(lds);                       // by sorted position: {beg, end, w | alt, first}
    uint4 *l_sel = l_rec + cap;                // the kept ("selected") chains, in selection order: {beg, end, w | alt, position}
    uint2 *l_fl = reinterpret_cast<uint2 *>(l_sel + cap);
    uint8_t *l_kept = reinterpret_cast<uint8_t *>(l_fl + cap);
    const int L = (int)(A.cum[r + 1] - A.cum[r]);
    uint2 *fl = A.flt + base;
    const ChainRec *crec = reinterpret_cast<const ChainRec *>(A.crec) + base;
    const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        const int bk = n_chn <= 32 ? 0 : n_chn <= 64 ? 1 : n_chn <= 128 ? 2 : n_chn <= 256 ? 3 : n_chn <= 512 ? 4 : n_chn <= 960 ? 5 : 6;
        atomicAdd(&A.ctr->dbg[bk], 1ull);
    }
    {
        __syncthreads();
        for (int i = lane; i < n_chn; i += 64) l_fl[i] = fl[i];
        __syncthreads();
        // ksort's introsort, operation for operation, by the whole wave; l_sel is free until the filter: it lends the sort
        // its stopper lists / output copy (8 B per chain) and, behind them, the partition stack
        wave_flt_introsort(l_fl, n_chn, reinterpret_cast<uint16_t *>(l_sel), reinterpret_cast<int *>(reinterpret_cast<char *>(l_sel) + (size_t)n_chn * 8), lane);
        const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
        for (int i = lane; i < n_chn; i += 64) {
            const uint2 f = l_fl[i];
            l_rec[i] = make_rec(A, crec[f.y], f.x);
            l_kept[i] = 0;
        }
        __syncthreads();
        // pairwise filter (mem_chain_flt, bwamem.cpp:575-617).  The sequential loop takes chain i through the chains selected so far,
        // in selection order, up to the first one that drops it; every selected chain it meets on the way with a large overlap
        // records i as its first shadowed chain if it has none yet.  Here SIXTY-FOUR candidates go through the selection at once, a
        // lane each: first through the chains selected before the batch (one broadcast LDS read per selected chain for all 64
        // lanes), then through the batch's own survivors in order (candidate l' is selected iff it is still alive when every
        // earlier candidate has been resolved; its record reaches the later lanes through v_readlane).  `first` is the SMALLEST
        // candidate that meets the chain: ds_min on the 0xffffffff-initialised field, issued by the lowest such lane — no round trip.
        // One candidate at a time against 64 selected chains per trip cost ~1800 cycles of latency per chain, whatever the selection's size.
        int n_sel = 1;
        if (lane == 0) { const uint4 r0 = l_rec[0]; l_sel[0] = make_uint4(r0.x, r0.y, r0.z, 0u); l_kept[0] = 3; }
        __syncthreads();
        const float mask_level = A.opt.mask_level, drop_ratio = A.opt.drop_ratio;
        const int max_gap = A.opt.max_chain_gap, min_dw = A.opt.min_seed_len << 1;
        for (int ib = 1; ib < n_chn; ib += 64) {
            const int i = ib + lane;
            const bool valid = i < n_chn;
            const uint4 ri = valid ? l_rec[i] : make_uint4(0, 0, 0, 0);
            const int bi = (int)ri.x, ei = (int)ri.y, wi = (int)(ri.z & 0x7fffffffu), li = ei - bi;
            const bool alt_i = (ri.z >> 31) != 0;
            const float fwi = (float)wi;
            bool alive = valid, large = false;
            const int n_sel0 = n_sel;
            for (int k0 = 0; k0 < n_sel0; k0 += 4) {
                uint4 rj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rj[u] = l_sel[k0 + u < n_sel0 ? k0 + u : n_sel0 - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u >= n_sel0) break;
                    const int bj = (int)rj[u].x, ej = (int)rj[u].y;
                    const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
                    const bool alt_j = (rj[u].z >> 31) != 0;
                    const int lj = ej - bj, min_l = li < lj ? li : lj;
                    const bool lg = alive && e_min > b_max && (!alt_j || alt_i) && (float)(e_min - b_max) >= (float)min_l * mask_level && min_l < max_gap;
                    const unsigned long long m_lg = __ballot(lg);
                    if (m_lg) {
                        if (lane == __builtin_ctzll(m_lg)) atomicMin(&l_rec[rj[u].w].w, (uint32_t)i);
                        if (lg) {
                            large = true;
                            const int wj = (int)(rj[u].z & 0x7fffffffu);
                            if (fwi < (float)wj * drop_ratio && wj - wi >= min_dw) alive = false;
                        }
                    }
                }
                if (!__ballot(alive)) break;
            }
            // the batch's own survivors, in order
            unsigned long long todo = __ballot(alive);
            while (todo) {
                const int lp = __builtin_ctzll(todo);                                    // alive, and every earlier candidate resolved: selected
                const int bj = __builtin_amdgcn_readlane(bi, lp), ej = __builtin_amdgcn_readlane(ei, lp);
                const uint32_t zj = (uint32_t)__builtin_amdgcn_readlane((int)ri.z, lp);
                if (lane == lp) { l_sel[n_sel] = make_uint4(ri.x, ri.y, ri.z, (uint32_t)i); l_kept[i] = large ? 2 : 3; }
                ++n_sel;
                const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
                const bool alt_j = (zj >> 31) != 0;
                const int lj = ej - bj, min_l = li < lj ? li : lj;
                const bool lg = alive && lane > lp && e_min > b_max && (!alt_j || alt_i) && (float)(e_min - b_max) >= (float)min_l * mask_level && min_l < max_gap;
                const unsigned long long m_lg = __ballot(lg);
                if (m_lg) {
                    if (lane == __builtin_ctzll(m_lg)) l_rec[ib + lp].w = (uint32_t)i;     // just selected: nobody has met it before
                    if (lg) {
                        large = true;
                        const int wj = (int)(zj & 0x7fffffffu);
                        if (fwi < (float)wj * drop_ratio && wj - wi >= min_dw) alive = false;
                    }
                }
                todo = __ballot(alive) & ~((2ull << lp) - 1ull);
            }
            __syncthreads();
        }
        for (int k = lane; k < n_sel; k += 64) {
            const int f = (int)l_rec[l_sel[k].w].w;
            if (f >= 0) l_kept[f] = 1;
        }
        __syncthreads();
        if (lane == 0) {
            const unsigned long long t_2 = __builtin_amdgcn_s_memtime();
            atomicAdd(&A.ctr->dbg[8], t_1 - t_0);
            atomicAdd(&A.ctr->dbg[9], t_2 - t_1);
            atomicMax(&A.ctr->dbg[12], t_1 - t_0);
            atomicMax(&A.ctr->dbg[13], t_2 - t_1);
            atomicMax(&A.ctr->dbg[14], (unsigned long long)n_chn);
            atomicMax(&A.ctr->dbg[15], (unsigned long long)n_sel);
            atomicAdd(&A.ctr->dbg[10], (unsigned long long)n_sel);
            atomicAdd(&A.ctr->dbg[11], (unsigned long long)n_chn);
        }
        // max_chain_extend (bwamem.cpp:618-625) can only bite when it is smaller than the chain count
        if (A.opt.max_chain_extend <= n_chn) {
            if (lane == 0) {
                int i, k;
                for (i = k = 0; i < n_chn; ++i) {
                    if (l_kept[i] == 0 || l_kept[i] == 3) continue;
                    if (++k >= A.opt.max_chain_extend) break;
                }
                for (; i < n_chn; ++i)
                    if (l_kept[i] < 3) l_kept[i] = 0;
            }
            __syncthreads();
        }
        // compaction of the kept chains and the read's totals, 64 chains at a time
        int32_t *first = A.f_first + base;
        int n_out = 0, n_seeds = 0;
        for (int ib = 0; ib < n_chn; ib += 64) {
            const int i = ib + lane;
            const int kp = i < n_chn ? (int)l_kept[i] : 0;
            const unsigned long long m = __ballot(kp != 0);
            if (kp) {
                const int o = n_out + __popcll(m & ((1ull << lane) - 1ull));
                const uint2 f = l_fl[i];
                const uint4 rc4 = l_rec[i];
                fl[o] = make_uint2(f.x | ((unsigned)kp << 29) | (rc4.z & 0x80000000u), f.y);
                first[o] = (int32_t)rc4.w;
                n_seeds += crec[f.y].n;
            }
            n_out += __popcll(m);
        }
        for (int d = 32; d >= 1; d >>= 1) n_seeds += __shfl_xor(n_seeds, d);
        if (lane == 0) {
            A.n_kept[r] = n_out;
            A.n_kept_seeds[r] = n_seeds;
            if (n_out) {
                const double min_l = A.opt.min_chain_weight ? (double)(1.1f * (float)A.opt.min_chain_weight) : (double)5.5f * log((double)L);
                if (!(min_l > (double)(0.05f * (float)L))) atomicAdd(&A.ctr->chain_longread, 1ull);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void chain_heavy_kernel(ChainArgs A, const unsigned long long *n_heavy_p, int cap_lo, int cap,
                                                         unsigned long long *ticket) {
    extern __shared__ __align__(16) unsigned char l_heavy[];
    const int kLdsChains = cap;
    const int lane = threadIdx.x;
    const int64_t n_heavy = (int64_t)*n_heavy_p;
    for (;;) {
        const int64_t hi = (int64_t)wave_ticket(ticket, 1ull);
        if (hi >= n_heavy) break;
        const int64_t r = A.heavy[hi];
        const int64_t base = A.read_base[r];
        const int n_chn = A.n_chn[r];
        if (n_chn <= cap_lo || (n_chn > cap && cap != kHeavyCap[5])) continue;          // another class's read
        const int L = (int)(A.cum[r + 1] - A.cum[r]);
        uint2 *fl = A.flt + base;
        uint4 *rec = A.f_rec + base;
        int32_t *kept = A.f_kept + base;
        const ChainRec *crec = reinterpret_cast<const ChainRec *>(A.crec) + base;
        if (n_chn > kLdsChains) {              // beyond the LDS budget: the sequential form, on lane 0
            const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                atomicAdd(&A.ctr->dbg[6], 1ull);
                flt_introsort(fl, n_chn);
                for (int i = 0; i < n_chn; ++i) { rec[i] = make_rec(A, crec[fl[i].y], fl[i].x); kept[i] = 0; }
                filter_seq(A.opt, n_chn, rec, kept, A.f_sel + base);
                finish_read(A, r, base, n_chn, L);
                atomicAdd(&A.ctr->dbg[7], __builtin_amdgcn_s_memtime() - t_0);
            }
            continue;
        }
        heavy_read(A, r, base, n_chn, l_heavy, cap, lane);
    }
}

// flat chain and seed records.  Sixteen lanes per read, a lane per kept chain (sixteen chains a trip): the chains' seed offsets are a
// prefix sum of their seed counts inside the group, and every lane walks its own chain's seed list — a lane per read took the chains
// one after the other, three dependent loads each before the first seed (1.8 ms per million reads; 4.4 on the grch38_like genome)
__global__ __launch_bounds__(256) void chain_emit_kernel(ChainArgs A, const int64_t *__restrict__ chain_off, const int64_t *__restrict__ seed_off,
                                                         bwams_chain_t *chains, bwams_chain_seed_t *seeds) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const bool live = r < A.nseq;                  // (whole 16-lane groups: the shuffles below stay inside a group)
    const int nk = live ? A.n_kept[r] : 0;
    if (nk == 0) return;
    const int64_t base = A.read_base[r];
    const uint2 *fl = A.flt + base;
    const int32_t *first = A.f_first + base;
    const int32_t *s_next = A.s_next + base;
    const int2 *s_ql = A.s_ql + base;
    const int64_t *pos = A.sa_coord + base;
    const ChainRec *crec = reinterpret_cast<const ChainRec *>(A.crec) + base;
    int64_t so0 = seed_off[r];
    const int64_t co = chain_off[r];
    const float frac = A.frac_rep[r];
    for (int j0 = 0; j0 < nk; j0 += 16) {
        const int j = j0 + sub;
        const bool has = j < nk;
        uint2 f = make_uint2(0u, 0u);
        ChainRec ch;
        ch.n = 0; ch.first_idx = -1; ch.rid = 0;
        if (has) { f = fl[j]; ch = crec[f.y]; }
        const int n = has ? ch.n : 0;
        int incl = n;                             // inclusive prefix of the seed counts over the group's sixteen lanes
        for (int d = 1; d < 16; d <<= 1) {
            const int o = __shfl_up(incl, d, 16);
            if (sub >= d) incl += o;
        }
        const int total = __shfl(incl, 15, 16);
        if (has) {
            int64_t so = so0 + (incl - n);
            bwams_chain_t c;
            c.seqid = (int32_t)r; c.cseed = 0;
            c.n = n;
            int m = 1;
            while (m < n) m <<= 1;               // SEEDS_PER_CHAIN = 1, doubled on demand (bwamem.cpp:398-412)
            c.m = m;
            c.first = first[j];
            c.rid = ch.rid;
            c.w_kept_alt = f.x;
            c.frac_rep = frac;
            c.pos = pos[ch.first_idx];
            c.seed_off = so;
            chains[co + j] = c;
            for (int32_t g = ch.first_idx; g >= 0; g = s_next[g]) {
                bwams_chain_seed_t sd;
                sd.rbeg = pos[g];
                sd.qbeg = s_ql[g].x; sd.len = s_ql[g].y; sd.score = sd.len;
                sd.done = 0; sd.pad0_[0] = sd.pad0_[1] = sd.pad0_[2] = 0;
                sd.aln = 0; sd.pad1_ = 0;
                seeds[so++] = sd;
            }
        }
        so0 += total;
    }
}

}  // namespace

size_t chain_node_bytes(int64_t n_sa, int64_t nseq) { return (size_t)((n_sa >> 1) + 2 * nseq + 4) * sizeof(Node); }
size_t chain_rec_bytes(int64_t n_sa) { return (size_t)(n_sa > 0 ? n_sa : 1) * sizeof(ChainRec); }

void launch_chain_count(const ChainArgs &A, uint32_t *keys, uint32_t *vals, hipStream_t st) {
    if (A.nseq <= 0) return;
    (void)hipMemsetAsync(A.slice, 0, (size_t)A.nseq * 16, st);
    if (A.n_smem > 0) chain_slice_kernel<<<(unsigned)((A.n_smem + 255) / 256), 256, 0, st>>>(A);
    chain_count_kernel<<<(unsigned)((A.nseq + 255) / 256), 256, 0, st>>>(A, keys, vals);
}
// The tiers are independent of each other: they run concurrently on the auxiliary streams (forked from
// and joined back into the batch's stream), the filter of the many-chain reads after all of them.
int launch_chain(const ChainArgs &A, const uint32_t *n_seeds, int cu_count, hipStream_t st, hipStream_t *aux,
                 hipEvent_t fork, hipEvent_t *join) {
    if (A.nseq <= 0) return 0;
    unsigned long long *cls = A.ctr->chain_class, *tk = A.ctr->chain_ticket;
    // the opt-in for more than 64 KB of dynamic LDS belongs to the CURRENT device: set per launch (a batch on a second GPU of the
    // process needs it too), and checked
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(chain_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes_xl(kClassXL2)) != hipSuccess) return -1;
    if (hipEventRecord(fork, st) != hipSuccess) return -1;
    for (int i = 0; i < 7; ++i)
        if (hipStreamWaitEvent(aux[i], fork, 0) != hipSuccess) return -1;
    // heaviest first: reads beyond the LDS budget (HBM state) and classes L, L1, then M, M1, S, then the lane tier
    // (the reads beyond a CU's LDS keep their state in HBM: every step is a dependent L2 / HBM access, so they get many
    // waves — no LDS limits them; at GRCh38 size, with max_occ hits per repeat SMEM, they were the stage's long pole on two)
    // (class XL, round 2: the five reads per million beyond class L — 2915 seeds the longest — took 19.5 ms in the HBM tier and
    // were the stage's long pole; with the ordered array in LDS only their chain records are in HBM)
    // class XL2 wants a whole CU's LDS per read: on the batch's own stream, which does not wait for the fork event, so that its blocks
    // are placed before the other classes' have filled every CU (behind them they found no CU free until the others drained: 42 ms on
    // the grch38_like genome for reads of ~2 ms each); blocks without a read leave at once
    chain_wave_kernel<<<(unsigned)cu_count, 64, lds_bytes_xl(kClassXL2), st>>>(A, cls + 9, cls + 6, tk + 9, -kClassXL2);
    chain_wave_kernel<<<(unsigned)(cu_count * 8), 64, 0, aux[0]>>>(A, nullptr, cls + 9, tk + 0, 0);
    chain_wave_kernel<<<(unsigned)cu_count, 64, lds_bytes_xl(kClassXL), aux[0]>>>(A, cls + 6, cls + 0, tk + 6, -kClassXL);
    chain_wave_kernel<<<(unsigned)cu_count, 64, lds_bytes(kClassL), aux[1]>>>(A, cls + 0, cls + 7, tk + 1, kClassL);
    chain_wave_kernel<<<(unsigned)(cu_count * 2), 64, lds_bytes(kClassL2), aux[5]>>>(A, cls + 7, cls + 1, tk + 7, kClassL2);
    chain_wave_kernel<<<(unsigned)(cu_count * 3), 64, lds_bytes(kClassL1), aux[2]>>>(A, cls + 1, cls + 8, tk + 2, kClassL1);
    chain_wave_kernel<<<(unsigned)(cu_count * 4), 64, lds_bytes(kClassM2), aux[6]>>>(A, cls + 8, cls + 2, tk + 8, kClassM2);
    chain_wave_kernel<<<(unsigned)(cu_count * 8), 64, lds_bytes(kClassM1), aux[2]>>>(A, cls + 3, cls + 4, tk + 4, kClassM1);
    chain_wave_kernel<<<(unsigned)(cu_count * 12), 64, lds_bytes(kClassS), aux[3]>>>(A, cls + 4, cls + 5, tk + 5, kClassS);
    // (the lane tier on the batch's own stream behind class XL2 instead: 19.1 -> 20.3 ms on the uniform genome, 50 -> 54 on grch38_like)
    chain_kernel<<<(unsigned)((A.nseq + 63) / 64), 64, 0, aux[1]>>>(A, n_seeds);
    // class M behind the lane tier (5 ms) rather than behind class L or S (kernel trace at GRCh38 size: L 8.9-10.3 ms + M 4.3-6.7 was
    // the stage's longest stream; S 7.8, L1 7.7 + M1 2.9, XL 0.9 + 7.2)
    // (round 4: on the stream of class L, the shortest; aux[4] shares a hardware queue with class S's stream, 16 ms on a repeat-rich genome)
    chain_wave_kernel<<<(unsigned)(cu_count * 5), 64, lds_bytes(kClassM), aux[5]>>>(A, cls + 2, cls + 3, tk + 3, kClassM);
    for (int i = 0; i < 7; ++i) {
        if (hipEventRecord(join[i], aux[i]) != hipSuccess) return -1;
        if (hipStreamWaitEvent(st, join[i], 0) != hipSuccess) return -1;
    }
    chain_redo_kernel<<<(unsigned)(cu_count * 2), 64, 0, st>>>(A);
    // the filter of the many-chain reads: three size classes, concurrently, the class of the longest reads first
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(chain_heavy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)heavy_lds_bytes(kHeavyCap[5])) != hipSuccess) return -1;
    if (hipEventRecord(fork, st) != hipSuccess) return -1;
    for (int i = 0; i < 5; ++i)
        if (hipStreamWaitEvent(aux[i], fork, 0) != hipSuccess) return -1;
    unsigned long long *htk = A.ctr->heavy_tickets;
    // The class of the longest reads (a whole CU's LDS per read, a few dozen reads per million) goes on the batch's own stream: it
    // starts without waiting for the fork event, i.e. before the other classes' blocks have filled every CU's LDS — behind them
    // (kernel trace) its blocks found no CU with 157 KB free until the other classes drained: 16.8 ms for 26 reads of ~4 ms.
    // Its blocks without a read leave at once.
    // (round 4: five classes instead of three — a wave of this kernel runs on LDS latency, so what a class's footprint leaves of a CU's
    // occupancy is its speed: the 257..512-chain reads ran four to a CU under the 960 class's 39 KB)
    chain_heavy_kernel<<<(unsigned)cu_count, 64, heavy_lds_bytes(kHeavyCap[5]), st>>>(A, &A.ctr->n_heavy, kHeavyCap[4], kHeavyCap[5], htk + 5);
    chain_heavy_kernel<<<(unsigned)(cu_count * 4), 64, heavy_lds_bytes(kHeavyCap[4]), aux[1]>>>(A, &A.ctr->n_heavy, kHeavyCap[3], kHeavyCap[4], htk + 4);
    chain_heavy_kernel<<<(unsigned)(cu_count * 7), 64, heavy_lds_bytes(kHeavyCap[3]), aux[2]>>>(A, &A.ctr->n_heavy, kHeavyCap[2], kHeavyCap[3], htk + 3);
    chain_heavy_kernel<<<(unsigned)(cu_count * 12), 64, heavy_lds_bytes(kHeavyCap[2]), aux[0]>>>(A, &A.ctr->n_heavy, kHeavyCap[1], kHeavyCap[2], htk + 2);
    chain_heavy_kernel<<<(unsigned)(cu_count * 16), 64, heavy_lds_bytes(kHeavyCap[1]), aux[3]>>>(A, &A.ctr->n_heavy, kHeavyCap[0], kHeavyCap[1], htk + 1);
    chain_heavy_kernel<<<(unsigned)(cu_count * 24), 64, heavy_lds_bytes(kHeavyCap[0]), aux[4]>>>(A, &A.ctr->n_heavy, 0, kHeavyCap[0], htk + 0);
    for (int i = 0; i < 5; ++i) {
        if (hipEventRecord(join[i], aux[i]) != hipSuccess) return -1;
        if (hipStreamWaitEvent(st, join[i], 0) != hipSuccess) return -1;
    }
    return 0;
}
void launch_chain_emit(const ChainArgs &A, const int64_t *chain_off, const int64_t *seed_off, bwams_chain_t *chains,
                       bwams_chain_seed_t *seeds, hipStream_t st) {
    if (A.nseq <= 0) return;
    chain_emit_kernel<<<(unsigned)((A.nseq * 16 + 255) / 256), 256, 0, st>>>(A, chain_off, seed_off, chains, seeds);
}

}  // namespace bwams
