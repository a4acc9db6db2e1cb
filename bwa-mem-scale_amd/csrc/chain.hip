// chain.hip — seed chaining and chain filtering on the device.
//
// Replaces, for a whole chunk of reads, the reference's
//   mem_chain_seeds   /root/reference/src/bwamem.cpp:789-959  (+ test_and_merge :379-421)
//   mem_chain_flt     bwamem.cpp:528-646  (+ mem_chain_weight :451-470)
// and the short-read early-out of mem_flt_chained_seeds (bwamem.cpp:491-526), consuming the
// SMEMs and SA coordinates that the seeding stage left in HBM.
//
// The work is inherently sequential per read (each seed is tested against the chain found by
// an ordered lookup, then the chains of the read are sorted and filtered pairwise), and reads
// are independent: one lane per read, all state in HBM scratch indexed by the read's slice of
// the SA-coordinate array.  A seed IS an SA hit, so every per-seed array shares the index space
// of sa_coord; a chain is named by its first seed.
//
// Two generic pieces decide tie cases and are therefore kept behaviour-identical to klib:
//   * the ordered map is a B-tree of order t = 5 (what kb_init(chn, 512 + 8) gives for the
//     48-byte mem_chain_t, kbtree.h:64) with kbtree.h's search, split and insert rules, so that
//     equal positions resolve to the same chain and the in-order traversal is the same;
//   * chains are sorted by weight with ksort.h's introsort (median-of-3, 16-element cut-off,
//     final insertion sort, comb-sort depth fallback), which is not stable.
#include "common.h"
#include "chain_kernels.h"

namespace bwams {
namespace {

constexpr int KB_T = 5;
constexpr int KB_MAXK = 2 * KB_T - 1;

struct Node {                  // 80 B
    int32_t n, internal;
    int32_t key[KB_MAXK];      // chain ids (relative seed index of the chain's first seed)
    int32_t ptr[KB_MAXK + 1];  // node ids relative to the read's node region
};

__device__ __forceinline__ int pos2rid(const DevBns &b, int64_t pos_f) {
    int left = 0, mid = 0, right = b.n_seqs;
    if (pos_f >= b.l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= b.contigs[mid].offset) {
            if (mid == b.n_seqs - 1) break;
            if (pos_f < b.contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
__device__ __forceinline__ int64_t depos(const DevBns &b, int64_t pos) {
    return pos >= b.l_pac ? (b.l_pac << 1) - 1 - pos : pos;
}
__device__ __forceinline__ int intv2rid(const DevBns &b, int64_t rb, int64_t re) {
    if (rb < b.l_pac && re > b.l_pac) return -2;
    const int rid_b = pos2rid(b, depos(b, rb));
    const int rid_e = rb < re ? pos2rid(b, depos(b, re - 1)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

__device__ __forceinline__ int kb_cmp(int64_t a, int64_t b) { return (b < a) - (a < b); }

// per-read view of the scratch
struct ReadCtx {
    const int64_t *pos;     // sa_coord + base: pos[id] is the key of chain id
    Node *nodes;
    int32_t n_nodes, cap_nodes, root;
    int32_t n_keys;
    bool overflow;
};

__device__ __forceinline__ int32_t new_node(ReadCtx &c) {
    if (c.n_nodes >= c.cap_nodes) { c.overflow = true; return 0; }
    Node *x = &c.nodes[c.n_nodes];
    x->n = 0; x->internal = 0;
    return c.n_nodes++;
}

__device__ int getp_aux(const ReadCtx &c, const Node *x, int64_t k, int *r) {
    int begin = 0, end = x->n;
    if (x->n == 0) return -1;
    while (begin < end) {
        const int mid = (begin + end) >> 1;
        if (kb_cmp(c.pos[x->key[mid]], k) < 0) begin = mid + 1;
        else end = mid;
    }
    if (begin == x->n) { *r = 1; return x->n - 1; }
    if ((*r = kb_cmp(k, c.pos[x->key[begin]])) < 0) --begin;
    return begin;
}

__device__ int32_t kbt_lower(const ReadCtx &c, int64_t k) {
    int r = 0;
    int32_t lower = -1, xi = c.root;
    for (;;) {
        const Node *x = &c.nodes[xi];
        const int i = getp_aux(c, x, k, &r);
        if (i >= 0 && r == 0) return x->key[i];
        if (i >= 0) lower = x->key[i];
        if (!x->internal) return lower;
        xi = x->ptr[i + 1];
    }
}

__device__ void kbt_split(ReadCtx &c, int32_t xi, int i, int32_t yi) {
    const int32_t zi = new_node(c);
    if (c.overflow) return;
    Node *x = &c.nodes[xi], *y = &c.nodes[yi], *z = &c.nodes[zi];
    z->internal = y->internal;
    z->n = KB_T - 1;
    for (int t = 0; t < KB_T - 1; ++t) z->key[t] = y->key[KB_T + t];
    if (y->internal) for (int t = 0; t < KB_T; ++t) z->ptr[t] = y->ptr[KB_T + t];
    y->n = KB_T - 1;
    for (int t = x->n; t > i; --t) x->ptr[t + 1] = x->ptr[t];
    x->ptr[i + 1] = zi;
    for (int t = x->n - 1; t >= i; --t) x->key[t + 1] = x->key[t];
    x->key[i] = y->key[KB_T - 1];
    ++x->n;
}

__device__ void kbt_put(ReadCtx &c, int32_t id) {
    const int64_t k = c.pos[id];
    ++c.n_keys;
    int32_t xi = c.root;
    if (c.nodes[xi].n == KB_MAXK) {
        const int32_t s = new_node(c);
        if (c.overflow) return;
        c.nodes[s].internal = 1; c.nodes[s].n = 0; c.nodes[s].ptr[0] = xi;
        c.root = s;
        kbt_split(c, s, 0, xi);
        if (c.overflow) return;
        xi = s;
    }
    for (;;) {
        Node *x = &c.nodes[xi];
        int r;
        if (!x->internal) {
            const int i = getp_aux(c, x, k, &r);
            for (int t = x->n - 1; t > i; --t) x->key[t + 1] = x->key[t];
            x->key[i + 1] = id;
            ++x->n;
            return;
        }
        int i = getp_aux(c, x, k, &r) + 1;
        if (c.nodes[x->ptr[i]].n == KB_MAXK) {
            kbt_split(c, xi, i, x->ptr[i]);
            if (c.overflow) return;
            if (kb_cmp(k, c.pos[x->key[i]]) > 0) ++i;
        }
        xi = x->ptr[i];
    }
}

// in-order traversal (__kb_traverse, kbtree.h:345-368) with an explicit stack; si = children already
// descended (height <= 16 covers 5^16 keys)
__device__ int32_t kbt_traverse(const ReadCtx &c, int32_t *out) {
    int32_t sx[16];
    int32_t si[16];
    int sp = 0, n = 0;
    sx[0] = c.root; si[0] = 0;
    while (sp >= 0) {
        const Node *x = &c.nodes[sx[sp]];
        if (!x->internal) {
            for (int t = 0; t < x->n; ++t) out[n++] = x->key[t];
            --sp;
            continue;
        }
        const int i = si[sp];
        if (i > 0 && i - 1 < x->n) out[n++] = x->key[i - 1];
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

__device__ void flt_insertsort(uint2 *a, int s, int t) {
    for (int i = s + 1; i < t; ++i)
        for (int j = i; j > s && flt_lt(a[j], a[j - 1]); --j) swp(a, j, j - 1);
}
__device__ void flt_combsort(uint2 *a, int n) {
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
__device__ void flt_introsort(uint2 *a, int n) {
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

// ---- the chaining kernel: one lane per read --------------------------------------------------
__global__ __launch_bounds__(64) void chain_kernel(ChainArgs A) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    A.n_kept[r] = 0;
    A.n_kept_seeds[r] = 0;
    A.read_base[r] = 0;
    const bwams_smem_t *sm = A.smem;
    // slice of this read in the (rid, m, n)-sorted SMEM array
    int64_t lo = 0, hi = A.n_smem;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sm[mid].rid < r) lo = mid + 1; else hi = mid; }
    const int64_t beg = lo;
    hi = A.n_smem;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sm[mid].rid <= r) lo = mid + 1; else hi = mid; }
    const int64_t end = lo;
    const int L = (int)(A.cum[r + 1] - A.cum[r]);
    if (beg == end || L < A.opt.min_seed_len) return;

    // frac_rep: query span covered by over-frequent SMEMs
    int b = 0, e = 0, l_rep = 0;
    for (int64_t i = beg; i < end; ++i) {
        const int sb = (int)sm[i].m, se = (int)sm[i].n + 1;
        if (sm[i].s <= (int64_t)A.opt.max_occ) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    A.frac_rep[r] = (float)l_rep / (float)L;

    const int64_t base = A.sa_off[beg];
    const int32_t cnt = (int32_t)(A.sa_off[end] - base);
    A.read_base[r] = base;
    if (cnt == 0) return;
    int32_t *s_next = A.s_next + base;
    int2 *s_ql = A.s_ql + base;
    int32_t *c_last = A.c_last + base, *c_n = A.c_n + base, *c_rid = A.c_rid + base;

    ReadCtx c;
    c.pos = A.sa_coord + base;
    const int64_t nbase = (base >> 1) + 2 * r;
    c.nodes = reinterpret_cast<Node *>(A.nodes) + nbase;
    c.cap_nodes = (int32_t)((((base + cnt) >> 1) + 2 * (r + 1)) - nbase);
    c.n_nodes = 0; c.n_keys = 0; c.overflow = false;
    c.root = new_node(c);

    const int64_t l_pac = A.bns.l_pac;
    for (int64_t i = beg; i < end; ++i) {
        const int qbeg = (int)sm[i].m, slen = (int)sm[i].n + 1 - (int)sm[i].m;
        const int32_t g0 = (int32_t)(A.sa_off[i] - base), g1 = (int32_t)(A.sa_off[i + 1] - base);
        for (int32_t g = g0; g < g1; ++g) {
            const int64_t rbeg = c.pos[g];
            const int rid = intv2rid(A.bns, rbeg, rbeg + slen);
            if (rid < 0) continue;
            bool to_add = true;
            if (c.n_keys) {
                const int32_t lower = kbt_lower(c, rbeg);
                if (lower >= 0) {                                        // test_and_merge
                    const int32_t li = c_last[lower];
                    const int2 lq = s_ql[li], fq = s_ql[lower];
                    const int64_t lr = c.pos[li], fr = c.pos[lower];
                    const int64_t qend = lq.x + lq.y, rend = lr + lq.y;
                    if (rid != c_rid[lower]) to_add = true;
                    else if (qbeg >= fq.x && qbeg + slen <= qend && rbeg >= fr && rbeg + slen <= rend) to_add = false;   // contained
                    else if ((lr < l_pac || fr < l_pac) && rbeg >= l_pac) to_add = true;
                    else {
                        const int64_t x = qbeg - lq.x, y = rbeg - lr;
                        if (y >= 0 && x - y <= A.opt.w && y - x <= A.opt.w && x - lq.y < A.opt.max_chain_gap &&
                            y - lq.y < A.opt.max_chain_gap) {
                            s_ql[g] = make_int2(qbeg, slen);
                            s_next[g] = -1;
                            s_next[li] = g;
                            c_last[lower] = g;
                            c_n[lower] += 1;
                            to_add = false;
                        }
                    }
                }
            }
            if (to_add) {
                s_ql[g] = make_int2(qbeg, slen);
                s_next[g] = -1;
                c_last[g] = g; c_n[g] = 1; c_rid[g] = rid;
                kbt_put(c, g);
                if (c.overflow) { atomicAdd(&A.ctr->chain_overflow, 1ull); return; }
            }
        }
    }
    if (c.n_keys == 0) return;

    // chains in B-tree order, their weights, the weight floor
    int32_t *ord = A.f_first + base;            // reused below as `first`
    const int32_t n_trav = kbt_traverse(c, ord);
    uint2 *fl = A.flt + base;
    int n_chn = 0;
    for (int32_t t = 0; t < n_trav; ++t) {
        const int32_t id = ord[t];
        int64_t endq = 0, endr = 0;
        int wq = 0, wr = 0;
        for (int32_t g = id; g >= 0; g = s_next[g]) {
            const int2 q = s_ql[g];
            const int64_t rb = c.pos[g];
            if (q.x >= endq) wq += q.y; else if (q.x + q.y > endq) wq += (int)(q.x + q.y - endq);
            endq = endq > q.x + q.y ? endq : q.x + q.y;
            if (rb >= endr) wr += q.y; else if (rb + q.y > endr) wr += (int)(rb + q.y - endr);
            endr = endr > rb + q.y ? endr : rb + q.y;
        }
        int w = wr < wq ? wr : wq;
        w = w < (1 << 30) ? w : (1 << 30) - 1;
        if (t == 0) fl[0] = make_uint2((unsigned)w, (unsigned)id);      // a_[0] stays in place when everything is dropped
        if (w < A.opt.min_chain_weight) continue;
        fl[n_chn++] = make_uint2((unsigned)w, (unsigned)id);
    }
    if (n_chn == 0) n_chn = 1;                  // the reference keeps a_[0] in that case (bwamem.cpp:549-572)
    flt_introsort(fl, n_chn);

    // pairwise filter
    int32_t *first = A.f_first + base;
    int32_t *kept = A.f_kept + base;
    int32_t *sel = A.f_sel + base;
    int2 *be = A.f_be + base;
    for (int i = 0; i < n_chn; ++i) {
        const int32_t id = (int32_t)fl[i].y;
        const int2 lq = s_ql[c_last[id]];
        be[i] = make_int2(s_ql[id].x, lq.x + lq.y);
        first[i] = -1; kept[i] = 0;
    }
    int n_sel = 0;
    kept[0] = 3;
    sel[n_sel++] = 0;
    for (int i = 1; i < n_chn; ++i) {
        bool large_ovlp = false;
        const int2 bi = be[i];
        const int wi = (int)fl[i].x;
        const bool alt_i = A.bns.contigs[c_rid[fl[i].y]].is_alt != 0;
        int k;
        for (k = 0; k < n_sel; ++k) {
            const int j = sel[k];
            const int2 bj = be[j];
            const int b_max = bj.x > bi.x ? bj.x : bi.x;
            const int e_min = bj.y < bi.y ? bj.y : bi.y;
            const bool alt_j = A.bns.contigs[c_rid[fl[j].y]].is_alt != 0;
            if (e_min > b_max && (!alt_j || alt_i)) {
                const int li = bi.y - bi.x, lj = bj.y - bj.x;
                const int min_l = li < lj ? li : lj;
                if ((float)(e_min - b_max) >= (float)min_l * A.opt.mask_level && min_l < A.opt.max_chain_gap) {
                    large_ovlp = true;
                    if (first[j] < 0) first[j] = i;
                    const int wj = (int)fl[j].x;
                    if ((float)wi < (float)wj * A.opt.drop_ratio && wj - wi >= (A.opt.min_seed_len << 1)) break;
                }
            }
        }
        if (k == n_sel) {
            sel[n_sel++] = i;
            kept[i] = large_ovlp ? 2 : 3;
        }
    }
    for (int i = 0; i < n_sel; ++i) {
        const int f = first[sel[i]];
        if (f >= 0) kept[f] = 1;
    }
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
        const bool alt = A.bns.contigs[c_rid[f.y]].is_alt != 0;
        fl[k] = make_uint2(f.x | ((unsigned)kept[i] << 29) | (alt ? 0x80000000u : 0u), f.y);
        first[k] = first[i];
        n_seeds += c_n[f.y];
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

// lane per read: flat chain and seed records
__global__ void chain_emit_kernel(ChainArgs A, const int64_t *__restrict__ chain_off, const int64_t *__restrict__ seed_off,
                                  bwams_chain_t *chains, bwams_chain_seed_t *seeds) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    const int nk = A.n_kept[r];
    if (nk == 0) return;
    const int64_t base = A.read_base[r];
    const uint2 *fl = A.flt + base;
    const int32_t *first = A.f_first + base;
    const int32_t *s_next = A.s_next + base;
    const int2 *s_ql = A.s_ql + base;
    const int64_t *pos = A.sa_coord + base;
    int64_t so = seed_off[r];
    const float frac = A.frac_rep[r];
    for (int j = 0; j < nk; ++j) {
        const int32_t id = (int32_t)fl[j].y;
        const int n = A.c_n[base + id];
        bwams_chain_t c;
        c.seqid = (int32_t)r; c.cseed = 0;
        c.n = n;
        int m = 1;
        while (m < n) m <<= 1;                   // SEEDS_PER_CHAIN = 1, doubled on demand (bwamem.cpp:398-412)
        c.m = m;
        c.first = first[j];
        c.rid = A.c_rid[base + id];
        c.w_kept_alt = fl[j].x;
        c.frac_rep = frac;
        c.pos = pos[id];
        c.seed_off = so;
        chains[chain_off[r] + j] = c;
        for (int32_t g = id; g >= 0; g = s_next[g]) {
            bwams_chain_seed_t s;
            s.rbeg = pos[g];
            s.qbeg = s_ql[g].x; s.len = s_ql[g].y; s.score = s.len;
            s.done = 0; s.pad0_[0] = s.pad0_[1] = s.pad0_[2] = 0;
            s.aln = 0; s.pad1_ = 0;
            seeds[so++] = s;
        }
    }
}

}  // namespace

size_t chain_node_bytes(int64_t n_sa, int64_t nseq) { return (size_t)((n_sa >> 1) + 2 * nseq + 4) * sizeof(Node); }

void launch_chain(const ChainArgs &A, hipStream_t st) {
    if (A.nseq <= 0) return;
    chain_kernel<<<(unsigned)((A.nseq + 63) / 64), 64, 0, st>>>(A);
}
void launch_chain_emit(const ChainArgs &A, const int64_t *chain_off, const int64_t *seed_off, bwams_chain_t *chains,
                       bwams_chain_seed_t *seeds, hipStream_t st) {
    if (A.nseq <= 0) return;
    chain_emit_kernel<<<(unsigned)((A.nseq + 63) / 64), 64, 0, st>>>(A, chain_off, seed_off, chains, seeds);
}

}  // namespace bwams
