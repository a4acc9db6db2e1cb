/*
 * chain_oracle.c — CPU restatement of seed chaining, chain filtering and the
 * chain-to-alignment driver (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * Follows, in /root/reference/src:
 *   mem_chain_seeds                 bwamem.cpp:789-959   (orc_chain_seeds)
 *   test_and_merge                  bwamem.cpp:379-421   (test_and_merge)
 *   kbtree (kb_intervalp/kb_putp/__kb_traverse)  kbtree.h:124-236,345-368  (kbt_*)
 *   bns_pos2rid / bns_intv2rid      bntseq.cpp:397-421   (pos2rid, intv2rid)
 *   mem_chain_weight                bwamem.cpp:451-470   (chain_weight)
 *   mem_chain_flt                   bwamem.cpp:528-646   (orc_chain_flt)
 *   ks_introsort / ks_combsort      ksort.h              (flt_introsort, flt_combsort)
 *   mem_flt_chained_seeds, mem_seed_sw   bwamem.cpp:425-449, :491-526   (flt_chained_seeds, seed_sw)
 *   bns_fetch_seq                   bntseq.cpp:545-574   (the contig clip inside seed_sw)
 *   cal_max_gap                     bwamem.cpp:94-104    (cal_max_gap)
 *   mem_chain2aln_across_reads_V2   bwamem.cpp:2773-3760 (orc_chain2aln)
 *   bns_fetch_seq_v2                bntseq.cpp:484-520   (the contig clip of rmax)
 *
 * PINNING: bwamem.cpp itself cannot be compiled here (it includes the un-vendored
 * safestringlib), so the driver logic above is PARITY UNPINNED.  The two generic klib
 * pieces whose exact behaviour decides tie cases — the B-tree (which chain is "lower",
 * where an equal key goes, the traversal order) and the unstable introsort of chains by
 * weight — ARE pinned: tests/test_oracle_chain.py compares kbt_* and flt_introsort with the
 * reference's own kbtree.h and ksort.h, compiled from where they lie into
 * oracle/_ref/libref_chain.so (ref_harness_chain.cpp).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bwams_oracle.h"

#define H0_ (-99)            /* macro.h:56 */
#define MAX_BAND_TRY 2       /* bwamem.cpp:79 */

/* ------------------------------------------------------------------ contigs */
static int pos2rid(const orc_bns_t *b, int64_t pos_f)
{
    int left = 0, mid = 0, right = b->n_seqs;
    if (pos_f >= b->l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= b->contigs[mid].offset) {
            if (mid == b->n_seqs - 1) break;
            if (pos_f < b->contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
static int64_t depos(const orc_bns_t *b, int64_t pos, int *is_rev)
{
    return (*is_rev = (pos >= b->l_pac)) ? (b->l_pac << 1) - 1 - pos : pos;
}
static int intv2rid(const orc_bns_t *b, int64_t rb, int64_t re)
{
    int is_rev, rid_b, rid_e;
    if (rb < b->l_pac && re > b->l_pac) return -2;
    rid_b = pos2rid(b, depos(b, rb, &is_rev));
    rid_e = rb < re ? pos2rid(b, depos(b, re - 1, &is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

/* ------------------------------------------------------------------ B-tree */
/* t = ((520 - 4 - 8) / (8 + 48) + 1) >> 1 = 5 for kb_init(chn, KB_DEFAULT_SIZE + 8) with the
 * 48-byte mem_chain_t key (kbtree.h:64, bwamem.cpp:830). Keys are chain ids; order = chain pos. */
#define KB_T 5
#define KB_MAXK (2 * KB_T - 1)
typedef struct { int32_t n, internal; int32_t key[KB_MAXK]; int32_t ptr[KB_MAXK + 1]; } kbnode_t;
typedef struct orc_kbt {
    kbnode_t *nodes; int32_t n_nodes, cap, root; int64_t n_keys;
    const int64_t *pos;      /* pos[id] */
} orc_kbt_t;

static int kb_cmp(int64_t a, int64_t b) { return (b < a) - (a < b); }   /* chain_cmp, bwamem.cpp:63 */

static int32_t kbt_new_node(orc_kbt_t *b)
{
    if (b->n_nodes == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 16;
        b->nodes = (kbnode_t *)realloc(b->nodes, (size_t)b->cap * sizeof(kbnode_t));
    }
    memset(&b->nodes[b->n_nodes], 0, sizeof(kbnode_t));
    return b->n_nodes++;
}
void orc_kbt_init(orc_kbt_t *b, const int64_t *pos)
{
    memset(b, 0, sizeof *b);
    b->pos = pos;
    b->root = kbt_new_node(b);
}
void orc_kbt_free(orc_kbt_t *b) { free(b->nodes); b->nodes = 0; }

/* __kb_getp_aux (kbtree.h:124-139) */
static int kbt_getp_aux(const orc_kbt_t *b, const kbnode_t *x, int64_t k, int *r)
{
    int tr, *rr = r ? r : &tr, begin = 0, end = x->n;
    if (x->n == 0) return -1;
    while (begin < end) {
        int mid = (begin + end) >> 1;
        if (kb_cmp(b->pos[x->key[mid]], k) < 0) begin = mid + 1;
        else end = mid;
    }
    if (begin == x->n) { *rr = 1; return x->n - 1; }
    if ((*rr = kb_cmp(k, b->pos[x->key[begin]])) < 0) --begin;
    return begin;
}
/* kb_intervalp (kbtree.h:159-176): id of the closest key <= k, or -1 */
int32_t orc_kbt_lower(const orc_kbt_t *b, int64_t k)
{
    int i, r = 0;
    int32_t lower = -1, x = b->root;
    while (x >= 0) {
        const kbnode_t *nd = &b->nodes[x];
        i = kbt_getp_aux(b, nd, k, &r);
        if (i >= 0 && r == 0) return nd->key[i];
        if (i >= 0) lower = nd->key[i];
        if (!nd->internal) return lower;
        x = nd->ptr[i + 1];
    }
    return lower;
}
/* __kb_split (kbtree.h:183-199): y = child i of x is full */
static void kbt_split(orc_kbt_t *b, int32_t xi, int i, int32_t yi)
{
    const int32_t zi = kbt_new_node(b);
    kbnode_t *x = &b->nodes[xi], *y = &b->nodes[yi], *z = &b->nodes[zi];
    z->internal = y->internal;
    z->n = KB_T - 1;
    memcpy(z->key, y->key + KB_T, sizeof(int32_t) * (KB_T - 1));
    if (y->internal) memcpy(z->ptr, y->ptr + KB_T, sizeof(int32_t) * KB_T);
    y->n = KB_T - 1;
    memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(int32_t) * (size_t)(x->n - i));
    x->ptr[i + 1] = zi;
    memmove(x->key + i + 1, x->key + i, sizeof(int32_t) * (size_t)(x->n - i));
    x->key[i] = y->key[KB_T - 1];
    ++x->n;
}
/* __kb_putp_aux / kb_putp (kbtree.h:200-233) */
void orc_kbt_put(orc_kbt_t *b, int32_t id)
{
    const int64_t k = b->pos[id];
    ++b->n_keys;
    int32_t xi = b->root;
    if (b->nodes[xi].n == KB_MAXK) {
        const int32_t s = kbt_new_node(b);
        b->nodes[s].internal = 1; b->nodes[s].n = 0; b->nodes[s].ptr[0] = xi;
        b->root = s;
        kbt_split(b, s, 0, xi);
        xi = s;
    }
    for (;;) {
        kbnode_t *x = &b->nodes[xi];
        int i;
        if (!x->internal) {
            i = kbt_getp_aux(b, x, k, 0);
            if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (size_t)(x->n - i - 1) * sizeof(int32_t));
            x->key[i + 1] = id;
            ++x->n;
            return;
        }
        i = kbt_getp_aux(b, x, k, 0) + 1;
        if (b->nodes[x->ptr[i]].n == KB_MAXK) {
            kbt_split(b, xi, i, x->ptr[i]);
            x = &b->nodes[xi];                       /* the node array may have moved */
            if (kb_cmp(k, b->pos[x->key[i]]) > 0) ++i;
        }
        xi = x->ptr[i];
    }
}
/* __kb_traverse (kbtree.h:345-368): in-order */
static int64_t kbt_walk(const orc_kbt_t *b, int32_t xi, int32_t *out, int64_t n)
{
    const kbnode_t *x = &b->nodes[xi];
    for (int i = 0; i <= x->n; ++i) {
        if (x->internal) n = kbt_walk(b, x->ptr[i], out, n);
        if (i < x->n) out[n++] = x->key[i];
    }
    return n;
}
int64_t orc_kbt_traverse(const orc_kbt_t *b, int32_t *out) { return kbt_walk(b, b->root, out, 0); }

/* Test hook: run a script of (lookup, optional insert) operations; pos[i] is the key of id i.
 * lower[i] receives the id found by the lookup made before id i is (optionally) inserted. */
int64_t orc_kbt_script(int64_t n, const int64_t *pos, const uint8_t *do_put, int32_t *lower, int32_t *order)
{
    orc_kbt_t b;
    orc_kbt_init(&b, pos);
    for (int64_t i = 0; i < n; ++i) {
        lower[i] = b.n_keys ? orc_kbt_lower(&b, pos[i]) : -1;
        if (do_put[i]) orc_kbt_put(&b, (int32_t)i);
    }
    const int64_t m = orc_kbt_traverse(&b, order);
    orc_kbt_free(&b);
    return m;
}

/* ------------------------------------------------------------------ sort */
typedef struct { uint32_t w; int32_t id; } flt_t;
#define flt_lt(a, b) ((a).w > (b).w)        /* bwamem.cpp:89 */

static void flt_insertsort(flt_t *s, flt_t *t)       /* __ks_insertsort, ksort.h */
{
    flt_t *i, *j, tmp;
    for (i = s + 1; i < t; ++i)
        for (j = i; j > s && flt_lt(*j, *(j - 1)); --j) { tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
static void flt_combsort(size_t n, flt_t *a)
{
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    flt_t tmp, *i, *j;
    do {
        if (gap > 2) {
            gap = (size_t)(gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (i = a; i < a + n - gap; ++i) {
            j = i + gap;
            if (flt_lt(*j, *i)) { tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) flt_insertsort(a, a + n);
}
static void flt_introsort(size_t n, flt_t *a)
{
    struct { flt_t *left, *right; int depth; } stack[8 * 64 + 2], *top = stack;
    int d;
    flt_t rp, tmp, *s, *t, *i, *j, *k;
    if (n < 1) return;
    if (n == 2) { if (flt_lt(a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }
    for (d = 2; 1ul << d < n; ++d);
    s = a; t = a + (n - 1); d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { flt_combsort((size_t)(t - s) + 1, s); t = s; continue; }
            i = s; j = t; k = i + ((j - i) >> 1) + 1;
            if (flt_lt(*k, *i)) { if (flt_lt(*k, *j)) k = j; }
            else k = flt_lt(*j, *i) ? i : j;
            rp = *k;
            if (k != t) { tmp = *k; *k = *t; *t = tmp; }
            for (;;) {
                do ++i; while (flt_lt(*i, rp));
                do --j; while (i <= j && flt_lt(rp, *j));
                if (j <= i) break;
                tmp = *i; *i = *j; *j = tmp;
            }
            tmp = *i; *i = *t; *t = tmp;
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { flt_insertsort(a, a + n); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}
/* Test hook: sort ids 0..n-1 by weight descending exactly as ks_introsort(mem_flt) would. */
void orc_flt_sort(int64_t n, const uint32_t *w, int32_t *order)
{
    flt_t *a = (flt_t *)malloc((size_t)(n ? n : 1) * sizeof(flt_t));
    for (int64_t i = 0; i < n; ++i) { a[i].w = w[i]; a[i].id = (int32_t)i; }
    flt_introsort((size_t)n, a);
    for (int64_t i = 0; i < n; ++i) order[i] = a[i].id;
    free(a);
}

/* ------------------------------------------------------------------ chaining */
typedef struct {            /* a chain under construction */
    int32_t n, m, rid, is_alt;
    int64_t pos;
    bwams_chain_seed_t *seeds;
} wchain_t;

static int test_and_merge(const bwams_mem_opt_t *opt, int64_t l_pac, wchain_t *c, const bwams_chain_seed_t *p, int seed_rid)
{
    int64_t qend, rend, x, y;
    const bwams_chain_seed_t *last = &c->seeds[c->n - 1];
    qend = last->qbeg + last->len;
    rend = last->rbeg + last->len;
    if (seed_rid != c->rid) return 0;
    if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
        return 1;
    if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
    x = p->qbeg - last->qbeg;
    y = p->rbeg - last->rbeg;
    if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
        if (c->n == c->m) {
            c->m <<= 1;
            c->seeds = (bwams_chain_seed_t *)realloc(c->seeds, (size_t)c->m * sizeof(bwams_chain_seed_t));
        }
        c->seeds[c->n++] = *p;
        return 1;
    }
    return 0;
}

static uint32_t chain_weight(int n, const bwams_chain_seed_t *seeds)
{
    int64_t end;
    int j, w = 0, tmp;
    for (j = 0, end = 0; j < n; ++j) {
        const bwams_chain_seed_t *s = &seeds[j];
        if (s->qbeg >= end) w += s->len;
        else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
        end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
    }
    tmp = w; w = 0;
    for (j = 0, end = 0; j < n; ++j) {
        const bwams_chain_seed_t *s = &seeds[j];
        if (s->rbeg >= end) w += s->len;
        else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
        end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
    }
    w = w < tmp ? w : tmp;
    return (uint32_t)(w < 1 << 30 ? w : (1 << 30) - 1);
}

/* mem_chain_flt for one read's chains (the reference calls it per read, bwamem.cpp:1354-1361).
 * a[] is reordered and compacted in place; returns the number kept. */
int orc_chain_flt(const bwams_mem_opt_t *opt, int n_chn, bwams_chain_t *a, const bwams_chain_seed_t *seeds)
{
    int i, k;
    if (n_chn == 0) return 0;
    for (i = k = 0; i < n_chn; ++i) {
        bwams_chain_t *c = &a[i];
        c->first = -1;
        const uint32_t w = chain_weight(c->n, seeds + c->seed_off);
        c->w_kept_alt = (c->w_kept_alt & 0x80000000u) | w;          /* kept = 0 */
        if ((int)w < opt->min_chain_weight) continue;
        a[k++] = *c;
    }
    /* when every chain is below min_chain_weight the reference still forms the range [0, 1)
     * (bwamem.cpp:549-572, `pr.second = i` with i = 1) over the untouched a_[0], which is then
     * kept: restated as is */
    n_chn = k ? k : 1;

    flt_t *srt = (flt_t *)malloc((size_t)n_chn * sizeof(flt_t));
    bwams_chain_t *tmp = (bwams_chain_t *)malloc((size_t)n_chn * sizeof(bwams_chain_t));
    for (i = 0; i < n_chn; ++i) { srt[i].w = BWAMS_CHAIN_W(a[i]); srt[i].id = i; }
    flt_introsort((size_t)n_chn, srt);
    for (i = 0; i < n_chn; ++i) tmp[i] = a[srt[i].id];
    memcpy(a, tmp, (size_t)n_chn * sizeof(bwams_chain_t));
    free(tmp); free(srt);

#define CHN_BEG(ch) (seeds[(ch).seed_off].qbeg)
#define CHN_END(ch) (seeds[(ch).seed_off + (ch).n - 1].qbeg + seeds[(ch).seed_off + (ch).n - 1].len)
#define SET_KEPT(ch, v) ((ch).w_kept_alt = ((ch).w_kept_alt & ~(3u << 29)) | ((uint32_t)(v) << 29))
    int *chains = (int *)malloc((size_t)n_chn * sizeof(int)), n_chains = 0;
    SET_KEPT(a[0], 3);
    chains[n_chains++] = 0;
    for (i = 1; i < n_chn; ++i) {
        int large_ovlp = 0;
        for (k = 0; k < n_chains; ++k) {
            const int j = chains[k];
            const int b_max = CHN_BEG(a[j]) > CHN_BEG(a[i]) ? CHN_BEG(a[j]) : CHN_BEG(a[i]);
            const int e_min = CHN_END(a[j]) < CHN_END(a[i]) ? CHN_END(a[j]) : CHN_END(a[i]);
            if (e_min > b_max && (!BWAMS_CHAIN_IS_ALT(a[j]) || BWAMS_CHAIN_IS_ALT(a[i]))) {
                const int li = CHN_END(a[i]) - CHN_BEG(a[i]);
                const int lj = CHN_END(a[j]) - CHN_BEG(a[j]);
                const int min_l = li < lj ? li : lj;
                if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
                    large_ovlp = 1;
                    if (a[j].first < 0) a[j].first = i;
                    if ((int)BWAMS_CHAIN_W(a[i]) < (int)BWAMS_CHAIN_W(a[j]) * opt->drop_ratio &&
                        (int)BWAMS_CHAIN_W(a[j]) - (int)BWAMS_CHAIN_W(a[i]) >= opt->min_seed_len << 1)
                        break;
                }
            }
        }
        if (k == n_chains) {
            chains[n_chains++] = i;
            SET_KEPT(a[i], large_ovlp ? 2 : 3);
        }
    }
    for (i = 0; i < n_chains; ++i) {
        bwams_chain_t *c = &a[chains[i]];
        if (c->first >= 0) SET_KEPT(a[c->first], 1);
    }
    free(chains);
    for (i = k = 0; i < n_chn; ++i) {
        const unsigned kept = BWAMS_CHAIN_KEPT(a[i]);
        if (kept == 0 || kept == 3) continue;
        if (++k >= opt->max_chain_extend) break;
    }
    for (; i < n_chn; ++i)
        if (BWAMS_CHAIN_KEPT(a[i]) < 3) SET_KEPT(a[i], 0);
    for (i = k = 0; i < n_chn; ++i)
        if (BWAMS_CHAIN_KEPT(a[i]) != 0) a[k++] = a[i];
    return k;
}

/* mem_seed_sw (bwamem.cpp:425-449): local SW score of a short seed in a +-50 window, or -1 */
#define MEM_SHORT_EXT 50
#define MEM_SHORT_LEN 200
static int seed_sw(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, int l_query,
                   const uint8_t *query, const bwams_chain_seed_t *s)
{
    const int64_t l_pac = bns->l_pac;
    int qb, qe;
    int64_t rb, re, mid;
    if (s->len >= MEM_SHORT_LEN) return -1;
    qb = s->qbeg; qe = s->qbeg + s->len;
    rb = s->rbeg; re = s->rbeg + s->len;
    mid = (rb + re) >> 1;
    qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
    qe += MEM_SHORT_EXT; qe = qe < l_query ? qe : l_query;
    rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
    re += MEM_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
    if (rb < l_pac && l_pac < re) {
        if (mid < l_pac) re = l_pac;
        else rb = l_pac;
    }
    if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return -1;
    {   /* bns_fetch_seq: clip to the reference sequence holding mid */
        int is_rev;
        const int rid = pos2rid(bns, depos(bns, mid, &is_rev));
        int64_t far_beg = bns->contigs[rid].offset, far_end = far_beg + bns->contigs[rid].len;
        if (is_rev) { const int64_t t0 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t0; }
        rb = rb > far_beg ? rb : far_beg;
        re = re < far_end ? re : far_end;
    }
    bwams_sw_opt_t sw;
    sw.o_del = opt->o_del; sw.e_del = opt->e_del; sw.o_ins = opt->o_ins; sw.e_ins = opt->e_ins;
    sw.zdrop = opt->zdrop; sw.end_bonus = 0;
    memcpy(sw.mat, opt->mat, 25);
    int out[7];
    orc_ksw_align2(&sw, qe - qb, query + qb, (int)(re - rb), ref_string + rb, 0x80000 /* KSW_XSTART */, out);
    return out[0];
}

/* mem_flt_chained_seeds (bwamem.cpp:491-526) for one read's kept chains: drops the seeds whose local
 * score is below min_HSP_score and compacts each chain's seed list in place (c->n shrinks). */
static void flt_chained_seeds(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, int l_query,
                              const uint8_t *query, int n_chn, bwams_chain_t *a, bwams_chain_seed_t *seeds)
{
    const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l_query);
    const int min_HSP_score = (int)(opt->a * min_l + .499);
    if (min_l > 0.05f * l_query) return;
    for (int i = 0; i < n_chn; ++i) {
        bwams_chain_t *c = &a[i];
        bwams_chain_seed_t *cs = seeds + c->seed_off;
        int j, k;
        for (j = k = 0; j < c->n; ++j) {
            bwams_chain_seed_t *s = &cs[j];
            s->score = seed_sw(opt, bns, ref_string, l_query, query, s);
            if (s->score < 0 || s->score >= min_HSP_score) {
                s->score = s->score < 0 ? s->len * opt->a : s->score;
                cs[k++] = *s;
            }
        }
        c->n = k;
    }
}

/* mem_chain_seeds for a batch (one work item), followed — when do_flt — by the per-read
 * mem_chain_flt and mem_flt_chained_seeds (bwamem.cpp:1354-1372; the latter needs ref_string and enc_qdb
 * and only acts on reads of ~1100 bases and more).
 * Output: flat chains grouped by read (chain_off[nseq+1]); the seeds of chain c are
 * seeds[c.seed_off .. +c.n).  Returns the number of chains, -1 on overflow, -2 when a read is
 * long enough for mem_flt_chained_seeds to re-score seeds (not restated). */
int64_t orc_chain_seeds(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const bwams_smem_t *smem, int64_t num_smem,
                        const int64_t *sa_coord, const int64_t *sa_off, const int64_t *cum_len, int32_t nseq, int do_flt,
                        bwams_chain_t *chains, int64_t chain_cap, bwams_chain_seed_t *seeds, int64_t seed_cap,
                        int64_t *chain_off, int64_t *n_seeds_out, const uint8_t *ref_string, const uint8_t *enc_qdb)
{
    const int64_t l_pac = bns->l_pac;
    int64_t smem_ptr = 0, pos = 0, n_chains = 0, n_seeds = 0;
    int l;
    for (l = 0; l <= nseq; ++l) chain_off[l] = 0;          /* chain_off[l+1] = chains of read l, prefix-summed at the end */
    for (l = 0; l < nseq && pos < num_smem - 1; ++l) {
        const int l_seq = (int)(cum_len[l + 1] - cum_len[l]);
        if ((int64_t)smem[smem_ptr].rid > l) continue;
        if (l_seq < opt->min_seed_len) continue;

        int b, e, l_rep;
        b = e = l_rep = 0;
        pos = smem_ptr - 1;
        do {
            pos++;
            const bwams_smem_t *p = &smem[pos];
            const int sb = (int)p->m, se = (int)p->n + 1;
            if (p->s <= opt->max_occ) continue;
            if (sb > e) l_rep += e - b, b = sb, e = se;
            else e = e > se ? e : se;
        } while (pos < num_smem - 1 && smem[pos].rid == smem[pos + 1].rid);
        l_rep += e - b;

        /* working chains of this read: id = creation order */
        const int64_t max_c = sa_off[pos + 1] - sa_off[smem_ptr] + 1;
        wchain_t *wc = (wchain_t *)calloc((size_t)max_c, sizeof(wchain_t));
        int64_t *wpos = (int64_t *)malloc((size_t)max_c * sizeof(int64_t));
        int32_t n_wc = 0;
        orc_kbt_t tree;
        orc_kbt_init(&tree, wpos);

        for (int64_t i = smem_ptr; i <= pos; ++i) {
            const bwams_smem_t *p = &smem[i];
            const int32_t slen = (int32_t)p->n + 1 - (int32_t)p->m;
            const int64_t step = p->s > opt->max_occ ? p->s / opt->max_occ : 1;
            int64_t k, mypos = sa_off[i];
            int32_t count;
            for (k = count = 0; k < p->s && count < opt->max_occ; k += step, ++count) {
                bwams_chain_seed_t s;
                memset(&s, 0, sizeof s);
                s.rbeg = sa_coord[mypos++];
                s.qbeg = (int32_t)p->m;
                s.score = s.len = slen;
                const int rid = intv2rid(bns, s.rbeg, s.rbeg + s.len);
                if (rid < 0) continue;
                int to_add = 0;
                if (tree.n_keys) {
                    const int32_t lower = orc_kbt_lower(&tree, s.rbeg);
                    if (lower < 0 || !test_and_merge(opt, l_pac, &wc[lower], &s, rid)) to_add = 1;
                } else to_add = 1;
                if (to_add) {
                    wchain_t *c = &wc[n_wc];
                    c->n = 1; c->m = 4;
                    c->seeds = (bwams_chain_seed_t *)malloc((size_t)c->m * sizeof(bwams_chain_seed_t));
                    c->seeds[0] = s;
                    c->rid = rid;
                    c->is_alt = !!bns->contigs[rid].is_alt;
                    c->pos = wpos[n_wc] = s.rbeg;
                    orc_kbt_put(&tree, n_wc++);
                }
            }
        }
        smem_ptr = pos + 1;

        /* traversal order -> flat output */
        int32_t *order = (int32_t *)malloc((size_t)(n_wc ? n_wc : 1) * sizeof(int32_t));
        const int64_t n_trav = orc_kbt_traverse(&tree, order);
        int64_t need_seeds = 0;
        for (int64_t i = 0; i < n_trav; ++i) need_seeds += wc[order[i]].n;
        int overflow = n_chains + n_trav > chain_cap || n_seeds + need_seeds > seed_cap;
        const int64_t first_chain = n_chains;
        if (!overflow) {
            for (int64_t i = 0; i < n_trav; ++i) {
                const wchain_t *c = &wc[order[i]];
                bwams_chain_t *o = &chains[n_chains++];
                memset(o, 0, sizeof *o);
                o->seqid = l; o->n = c->n; o->rid = c->rid;
                for (o->m = 1; o->m < o->n; o->m <<= 1) {}      /* SEEDS_PER_CHAIN = 1, doubled on demand (bwamem.cpp:398-412) */
                o->w_kept_alt = (uint32_t)c->is_alt << 31;
                o->frac_rep = (float)l_rep / l_seq;
                o->pos = c->pos;
                o->seed_off = n_seeds;
                memcpy(seeds + n_seeds, c->seeds, (size_t)c->n * sizeof(bwams_chain_seed_t));
                n_seeds += c->n;
            }
        }
        for (int32_t i = 0; i < n_wc; ++i) free(wc[i].seeds);
        free(order); free(wc); free(wpos);
        orc_kbt_free(&tree);
        if (overflow) return -1;

        if (do_flt) {
            const int kept = orc_chain_flt(opt, (int)(n_chains - first_chain), chains + first_chain, seeds);
            n_chains = first_chain + kept;
            const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l_seq);
            if (!(min_l > 0.05f * l_seq) && kept && do_flt != 2) {          /* do_flt == 2: stop after mem_chain_flt (test hook) */
                if (!ref_string || !enc_qdb) return -2;
                flt_chained_seeds(opt, bns, ref_string, l_seq, enc_qdb + cum_len[l], kept, chains + first_chain, seeds);
            }
        }
        chain_off[l + 1] = n_chains - first_chain;
    }
    for (l = 0; l < nseq; ++l) chain_off[l + 1] += chain_off[l];
    if (n_seeds_out) *n_seeds_out = n_seeds;
    return n_chains;
}

/* ------------------------------------------------------------------ chain -> alignment regions */
/* ---- ERT mode: the tail of mem_kernel1_core_ert (bwamem.cpp:1193-1203) for a chunk whose MEMs and hits the
 * reference's ERT walk (get_seeds / reseed / last, ertseeding.cpp) produced: ks_introsort(mem_smem_sort_lt),
 * mem_chain_new (:961-1050), mem_chain_flt, mem_flt_chained_seeds.  mems of read l: [mem_off[l], mem_off[l+1]);
 * its hit array starts at hits + hit_off[l] (mem.hitbeg is relative to it). ---- */
typedef int (*orc_lt_fn)(const void *ctx, int a, int b);
void orc_idx_introsort(size_t n, int *a, orc_lt_fn lt, const void *ctx);
static int ert_mem_lt(const void *ctx, int a, int b)              /* smem_lt_2, bwamem.cpp:73 */
{
    const bwams_ert_mem_t *m = (const bwams_ert_mem_t *)ctx;
    return m[a].start == m[b].start ? m[a].end < m[b].end : m[a].start < m[b].start;
}
int64_t orc_chain_new_ert(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const bwams_ert_mem_t *mems, const int64_t *mem_off,
                          const uint64_t *hits, const int64_t *hit_off, const int64_t *cum_len, int32_t nseq, int do_flt,
                          bwams_chain_t *chains, int64_t chain_cap, bwams_chain_seed_t *seeds, int64_t seed_cap,
                          int64_t *chain_off, int64_t *n_seeds_out, const uint8_t *ref_string, const uint8_t *enc_qdb)
{
    const int64_t l_pac = bns->l_pac;
    int64_t n_chains = 0, n_seeds = 0;
    chain_off[0] = 0;
    for (int l = 0; l < nseq; ++l) {
        const int len = (int)(cum_len[l + 1] - cum_len[l]);
        const int nm = (int)(mem_off[l + 1] - mem_off[l]);
        const int64_t first_chain = n_chains;
        chain_off[l + 1] = n_chains;
        if (nm == 0) continue;                                     /* (an EMF-resolved read has no MEMs either) */
        const bwams_ert_mem_t *mm = mems + mem_off[l];
        const uint64_t *hh = hits + hit_off[l];
        int *ord = (int *)malloc((size_t)nm * sizeof(int));
        for (int i = 0; i < nm; ++i) ord[i] = i;
        orc_idx_introsort((size_t)nm, ord, ert_mem_lt, mm);        /* bwamem.cpp:1193 */
        if (len < opt->min_seed_len) { free(ord); continue; }      /* mem_chain_new returns at once (:978) */
        int i, b = 0, e = 0, l_rep = 0;
        int64_t max_c = 1;
        for (i = 0; i < nm; ++i) {                                 /* frac_rep (:980-987) */
            const bwams_ert_mem_t *p = &mm[ord[i]];
            const int sb = p->start, se = p->end;
            max_c += p->hitcount < opt->max_occ ? p->hitcount : opt->max_occ;
            if (p->hitcount <= opt->max_occ) continue;
            if (sb > e) l_rep += e - b, b = sb, e = se;
            else e = e > se ? e : se;
        }
        l_rep += e - b;
        wchain_t *wc = (wchain_t *)calloc((size_t)max_c, sizeof(wchain_t));
        int64_t *wpos = (int64_t *)malloc((size_t)max_c * sizeof(int64_t));
        int32_t n_wc = 0;
        orc_kbt_t tree;
        orc_kbt_init(&tree, wpos);
        for (i = 0; i < nm; ++i) {
            const bwams_ert_mem_t *p = &mm[ord[i]];
            const int slen = p->end - p->start;
            const int step = p->hitcount > opt->max_occ ? p->hitcount / opt->max_occ : 1;
            int64_t k;
            int count;
            for (k = count = 0; k < p->hitcount && count < opt->max_occ; k += step, ++count) {
                bwams_chain_seed_t s;
                memset(&s, 0, sizeof s);
                if (p->forward || p->fetch_leaves) s.rbeg = (int64_t)hh[p->hitbeg + k];
                else s.rbeg = (l_pac << 1) - ((int64_t)hh[p->hitbeg + k] + slen - p->end_correction);
                s.qbeg = p->start;
                s.len = p->end - p->start;
                s.score = s.len;
                const int rid = intv2rid(bns, s.rbeg, s.rbeg + s.len);
                if (rid < 0) continue;
                int to_add = 0;
                if (tree.n_keys) {
                    const int32_t lower = orc_kbt_lower(&tree, s.rbeg);
                    if (lower < 0 || !test_and_merge(opt, l_pac, &wc[lower], &s, rid)) to_add = 1;
                } else to_add = 1;
                if (to_add) {
                    wchain_t *c = &wc[n_wc];
                    c->n = 1; c->m = 4;
                    c->seeds = (bwams_chain_seed_t *)malloc((size_t)c->m * sizeof(bwams_chain_seed_t));
                    c->seeds[0] = s;
                    c->rid = rid;
                    c->is_alt = !!bns->contigs[rid].is_alt;
                    c->pos = wpos[n_wc] = s.rbeg;
                    orc_kbt_put(&tree, n_wc++);
                }
            }
        }
        free(ord);
        int32_t *order = (int32_t *)malloc((size_t)(n_wc ? n_wc : 1) * sizeof(int32_t));
        const int64_t n_trav = n_wc ? orc_kbt_traverse(&tree, order) : 0;
        int64_t need_seeds = 0;
        for (int64_t t = 0; t < n_trav; ++t) need_seeds += wc[order[t]].n;
        const int overflow = n_chains + n_trav > chain_cap || n_seeds + need_seeds > seed_cap;
        if (!overflow)
            for (int64_t t = 0; t < n_trav; ++t) {
                const wchain_t *c = &wc[order[t]];
                bwams_chain_t *o = &chains[n_chains++];
                memset(o, 0, sizeof *o);
                o->seqid = l; o->n = c->n; o->rid = c->rid;
                for (o->m = 1; o->m < o->n; o->m <<= 1) {}
                o->w_kept_alt = (uint32_t)c->is_alt << 31;
                o->frac_rep = (float)l_rep / len;
                o->pos = c->pos;
                o->seed_off = n_seeds;
                memcpy(seeds + n_seeds, c->seeds, (size_t)c->n * sizeof(bwams_chain_seed_t));
                n_seeds += c->n;
            }
        for (int32_t t = 0; t < n_wc; ++t) free(wc[t].seeds);
        free(order); free(wc); free(wpos);
        orc_kbt_free(&tree);
        if (overflow) return -1;
        if (do_flt && n_chains > first_chain) {
            const int kept = orc_chain_flt(opt, (int)(n_chains - first_chain), chains + first_chain, seeds);
            n_chains = first_chain + kept;
            const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(len);
            if (!(min_l > 0.05f * len) && kept) {
                if (!ref_string || !enc_qdb) return -2;
                flt_chained_seeds(opt, bns, ref_string, len, enc_qdb + cum_len[l], kept, chains + first_chain, seeds);
            }
        }
        chain_off[l + 1] = n_chains;
    }
    if (n_seeds_out) *n_seeds_out = n_seeds;
    return n_chains;
}

static int cal_max_gap(const bwams_mem_opt_t *opt, int qlen)
{
    int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
    int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < opt->w << 1 ? l : opt->w << 1;
}

static void seedcov(bwams_alnreg_t *a, const bwams_chain_t *c, const bwams_chain_seed_t *seeds)
{
    int i;
    for (i = 0, a->seedcov = 0; i < c->n; ++i) {
        const bwams_chain_seed_t *t = &seeds[c->seed_off + i];
        if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
            a->seedcov += t->len;
    }
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* One side of the extension: the band-retry loop of bwamem.cpp:3225-3390 (left) and
 * :3440-3630 (right).  The reference runs it three times per side (scalar / 16-bit / 8-bit
 * task classes); the classes produce identical outputs, so one loop restates all three. */
static void run_side(const bwams_mem_opt_t *opt, int right, bwams_seqpair_t *pairs, int64_t np, const uint8_t *refbuf,
                     const uint8_t *qerbuf, const int64_t *reg_of_pair, bwams_alnreg_t *regs, const bwams_chain_t *chains,
                     const bwams_chain_seed_t *seeds, const int64_t *cum_len)
{
    bwams_sw_opt_t sw;
    sw.o_del = opt->o_del; sw.e_del = opt->e_del; sw.o_ins = opt->o_ins; sw.e_ins = opt->e_ins;
    sw.zdrop = opt->zdrop; sw.end_bonus = right ? opt->pen_clip3 : opt->pen_clip5;
    memcpy(sw.mat, opt->mat, 25);
    const int pen_clip = sw.end_bonus;
    int64_t *pend = (int64_t *)malloc((size_t)(np ? np : 1) * sizeof(int64_t)), n_pend = np;
    for (int64_t i = 0; i < np; ++i) pend[i] = i;
    for (int t = 0; t < MAX_BAND_TRY; ++t) {
        const int32_t w = opt->w << t;
        int64_t num = 0;
        for (int64_t l = 0; l < n_pend; ++l) {
            bwams_seqpair_t *sp = &pairs[pend[l]];
            orc_bsw_pairs(&sw, sp, refbuf, qerbuf, 1, w, 0);
            bwams_alnreg_t *a = &regs[reg_of_pair[pend[l]]];
            const int prev = a->score;
            a->score = sp->score;
            if (a->score == prev || sp->max_off < (w >> 1) + (w >> 2) || t + 1 == MAX_BAND_TRY) {
                if (!right) {
                    if (sp->gscore <= 0 || sp->gscore <= a->score - pen_clip) {
                        a->qb -= sp->qle; a->rb -= sp->tle;
                        a->truesc = a->score;
                    } else {
                        a->qb = 0; a->rb -= sp->gtle;
                        a->truesc = sp->gscore;
                    }
                } else {
                    if (sp->gscore <= 0 || sp->gscore <= a->score - pen_clip) {
                        a->qe += sp->qle; a->re += sp->tle;
                        a->truesc += a->score - sp->h0;
                    } else {
                        a->qe = (int32_t)(cum_len[sp->seqid + 1] - cum_len[sp->seqid]); a->re += sp->gtle;
                        a->truesc += sp->gscore - sp->h0;
                    }
                }
                a->w = a->w > w ? a->w : w;
                if (a->rb != H0_ && a->qb != H0_ && a->qe != H0_ && a->re != H0_) seedcov(a, &chains[a->chain], seeds);
            } else pend[num++] = pend[l];
        }
        n_pend = num;
    }
    free(pend);
}

/* mem_chain2aln_across_reads_V2 for one work item.  seeds[].aln is written (the region index
 * of each seed within its read).  regs: one per seed, grouped by read (reg_off[nseq+1]), in the
 * order the reference appends them (chain by chain, seeds by descending score then index).
 * Optional dumps of the task lists as built (before extension): left/right pairs and buffers.
 * Returns the number of regions or -1 on overflow. */
int64_t orc_chain2aln(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const uint8_t *enc_qdb,
                      const int64_t *cum_len, int32_t nseq, const bwams_chain_t *chains, const int64_t *chain_off,
                      bwams_chain_seed_t *seeds, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off,
                      orc_task_dump_t *dump)
{
    const int64_t l_pac = bns->l_pac;
    int64_t n_regs = 0, n_seeds_tot = 0;
    for (int64_t c = 0; c < chain_off[nseq]; ++c) n_seeds_tot += chains[c].n;
    if (n_seeds_tot > reg_cap) return -1;

    /* task storage */
    bwams_seqpair_t *pl = (bwams_seqpair_t *)calloc((size_t)n_seeds_tot + 1, sizeof(bwams_seqpair_t));
    bwams_seqpair_t *pr = (bwams_seqpair_t *)calloc((size_t)n_seeds_tot + 1, sizeof(bwams_seqpair_t));
    int64_t *regl = (int64_t *)malloc(((size_t)n_seeds_tot + 1) * sizeof(int64_t));
    int64_t *regr = (int64_t *)malloc(((size_t)n_seeds_tot + 1) * sizeof(int64_t));
    size_t cap_lq = 1 << 16, cap_lr = 1 << 16, cap_rq = 1 << 16, cap_rr = 1 << 16;
    uint8_t *lq = (uint8_t *)malloc(cap_lq), *lr = (uint8_t *)malloc(cap_lr), *rq = (uint8_t *)malloc(cap_rq), *rr = (uint8_t *)malloc(cap_rr);
    int64_t nl = 0, nr = 0, olq = 0, olr = 0, orq = 0, orr = 0;
    /* per read: the seed order lists of its chains (srtg), for the purge pass */
    uint32_t *srtg = (uint32_t *)malloc(((size_t)n_seeds_tot + 1) * sizeof(uint32_t));
    int64_t spos = 0;
#define GROW(buf, cap, need) do { if ((size_t)(need) > cap) { while ((size_t)(need) > cap) cap *= 2; buf = (uint8_t *)realloc(buf, cap); } } while (0)

    for (int l = 0; l < nseq; ++l) {
        reg_off[l] = n_regs;
        const uint8_t *query = enc_qdb + cum_len[l];
        const int l_query = (int)(cum_len[l + 1] - cum_len[l]);
        for (int64_t j = chain_off[l]; j < chain_off[l + 1]; ++j) {
            const bwams_chain_t *c = &chains[j];
            bwams_chain_seed_t *cs = seeds + c->seed_off;
            if (c->n == 0) continue;
            int64_t rmax[2], tmp;
            rmax[0] = l_pac << 1; rmax[1] = 0;
            for (int i = 0; i < c->n; ++i) {
                const bwams_chain_seed_t *t = &cs[i];
                const int64_t b = t->rbeg - (t->qbeg + cal_max_gap(opt, t->qbeg));
                const int64_t e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(opt, l_query - t->qbeg - t->len));
                rmax[0] = rmax[0] < b ? rmax[0] : b;
                rmax[1] = rmax[1] > e ? rmax[1] : e;
            }
            rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
            rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
            if (rmax[0] < l_pac && l_pac < rmax[1]) {
                if (cs[0].rbeg < l_pac) rmax[1] = l_pac;
                else rmax[0] = l_pac;
            }
            {   /* bns_fetch_seq_v2: clip to the contig of the first seed */
                int is_rev;
                const int rid = pos2rid(bns, depos(bns, cs[0].rbeg, &is_rev));
                int64_t far_beg = bns->contigs[rid].offset, far_end = far_beg + bns->contigs[rid].len;
                if (is_rev) { const int64_t t0 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t0; }
                rmax[0] = rmax[0] > far_beg ? rmax[0] : far_beg;
                rmax[1] = rmax[1] < far_end ? rmax[1] : far_end;
            }
            const uint8_t *rseq = ref_string + rmax[0];

            uint64_t *srt = (uint64_t *)malloc((size_t)c->n * sizeof(uint64_t));
            for (int i = 0; i < c->n; ++i) srt[i] = (uint64_t)(uint32_t)cs[i].score << 32 | (uint32_t)i;
            qsort(srt, (size_t)c->n, sizeof(uint64_t), cmp_u64);      /* keys are distinct: any sort gives ks_introsort_64's result */
            for (int i = 0; i < c->n; ++i) srtg[spos++] = (uint32_t)srt[i];

            for (int k = c->n - 1; k >= 0; --k) {
                bwams_chain_seed_t *s = &cs[(uint32_t)srt[k]];
                bwams_alnreg_t *a = &regs[n_regs];
                memset(a, 0, sizeof *a);
                s->aln = (int32_t)(n_regs - reg_off[l]);
                a->w = opt->w;
                a->score = a->truesc = -1;
                a->rid = c->rid;
                a->frac_rep = c->frac_rep;
                a->seedlen0 = s->len;
                a->chain = j;
                a->rb = a->re = H0_; a->qb = a->qe = H0_;
                if (s->qbeg) {
                    bwams_seqpair_t sp;
                    memset(&sp, 0, sizeof sp);
                    sp.h0 = s->len * opt->a;
                    sp.seqid = l; sp.regid = s->aln; sp.id = (int32_t)nl;
                    sp.idq = (int32_t)olq; sp.idr = (int32_t)olr;
                    GROW(lq, cap_lq, olq + s->qbeg);
                    for (int i = 0; i < s->qbeg; ++i) lq[olq + i] = query[s->qbeg - 1 - i];
                    olq += s->qbeg;
                    tmp = s->rbeg - rmax[0];
                    GROW(lr, cap_lr, olr + tmp);
                    for (int64_t i = 0; i < tmp; ++i) lr[olr + i] = rseq[tmp - 1 - i];
                    olr += tmp;
                    sp.len2 = s->qbeg; sp.len1 = (int32_t)tmp;
                    regl[nl] = n_regs;
                    pl[nl++] = sp;
                    a->qb = s->qbeg; a->rb = s->rbeg;
                } else {
                    a->score = a->truesc = s->len * opt->a; a->qb = 0; a->rb = s->rbeg;
                }
                if (s->qbeg + s->len != l_query) {
                    const int64_t qe = s->qbeg + s->len;
                    const int64_t re = s->rbeg + s->len - rmax[0];
                    bwams_seqpair_t sp;
                    memset(&sp, 0, sizeof sp);
                    sp.h0 = H0_;
                    sp.seqid = l; sp.regid = s->aln; sp.id = (int32_t)nr;
                    sp.len2 = (int32_t)(l_query - qe);
                    sp.len1 = (int32_t)(rmax[1] - rmax[0] - re);
                    sp.idq = (int32_t)orq; sp.idr = (int32_t)orr;
                    GROW(rq, cap_rq, orq + sp.len2);
                    GROW(rr, cap_rr, orr + sp.len1);
                    for (int i = 0; i < sp.len2; ++i) rq[orq + i] = query[qe + i];
                    for (int i = 0; i < sp.len1; ++i) rr[orr + i] = rseq[re + i];
                    orq += sp.len2; orr += sp.len1;
                    regr[nr] = n_regs;
                    pr[nr++] = sp;
                    a->qe = (int32_t)qe; a->re = rmax[0] + re;
                } else {
                    a->qe = l_query; a->re = s->rbeg + s->len;
                    if (a->rb != H0_ && a->qb != H0_) seedcov(a, c, seeds);
                }
                ++n_regs;
            }
            free(srt);
        }
    }
    reg_off[nseq] = n_regs;

    if (dump) {     /* the task lists as built */
        dump->n_left = nl; dump->n_right = nr;
        dump->left = pl; dump->right = pr;
        dump->left_ref = lr; dump->left_qer = lq; dump->right_ref = rr; dump->right_qer = rq;
        dump->left_ref_bytes = olr; dump->left_qer_bytes = olq; dump->right_ref_bytes = orr; dump->right_qer_bytes = orq;
        if (dump->build_only) { free(regl); free(regr); free(srtg); return n_regs; }
    }

    run_side(opt, 0, pl, nl, lr, lq, regl, regs, chains, seeds, cum_len);
    for (int64_t i = 0; i < nr; ++i) pr[i].h0 = regs[regr[i]].score;
    run_side(opt, 1, pr, nr, rr, rq, regr, regs, chains, seeds, cum_len);

    /* discard seeds (and their regions) already covered by an earlier region (bwamem.cpp:3648-3755) */
    spos = 0;
    for (int l = 0; l < nseq; ++l) {
        const int l_query = (int)(cum_len[l + 1] - cum_len[l]);
        bwams_alnreg_t *av = regs + reg_off[l];
        const int64_t av_n = reg_off[l + 1] - reg_off[l];
        int lim = 0;
        for (int64_t j = chain_off[l]; j < chain_off[l + 1]; ++j) {
            const bwams_chain_t *c = &chains[j];
            const bwams_chain_seed_t *cs = seeds + c->seed_off;
            uint32_t *srt2 = srtg + spos;
            spos += c->n;
            for (int k = c->n - 1; k >= 0; --k) {
                const bwams_chain_seed_t *s = &cs[srt2[k]];
                int64_t i;
                int v = 0;
                for (i = 0; i < av_n && v < lim; ++i) {
                    const bwams_alnreg_t *p = &av[i];
                    if (p->qb == -1 && p->qe == -1) continue;
                    int64_t rd;
                    int qd, w, max_gap;
                    if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) { v++; continue; }
                    if (s->len - p->seedlen0 > .1 * l_query) { v++; continue; }
                    qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
                    max_gap = cal_max_gap(opt, (int)(qd < rd ? qd : rd));
                    w = max_gap < p->w ? max_gap : p->w;
                    if (qd - rd < w && rd - qd < w) break;
                    qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
                    max_gap = cal_max_gap(opt, (int)(qd < rd ? qd : rd));
                    w = max_gap < p->w ? max_gap : p->w;
                    if (qd - rd < w && rd - qd < w) break;
                    v++;
                }
                if (v < lim) {
                    for (v = k + 1; v < c->n; ++v) {
                        if (srt2[v] == UINT32_MAX) continue;
                        const bwams_chain_seed_t *t = &cs[srt2[v]];
                        if (t->len < s->len * .95) continue;
                        if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
                        if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
                    }
                    if (v == c->n) {
                        av[s->aln].qb = av[s->aln].qe = -1;
                        srt2[k] = UINT32_MAX;
                        continue;
                    }
                }
                lim++;
            }
        }
    }

    if (!dump) { free(pl); free(pr); free(lq); free(lr); free(rq); free(rr); }
    free(regl); free(regr); free(srtg);
    return n_regs;
}

void orc_task_dump_free(orc_task_dump_t *d)
{
    free(d->left); free(d->right); free(d->left_ref); free(d->left_qer); free(d->right_ref); free(d->right_qer);
    memset(d, 0, sizeof *d);
}

/* exported for the pin against bns_depos (bntseq.h:88-91) */
int64_t orc_depos(int64_t l_pac, int64_t pos, int *is_rev) { orc_bns_t b; b.l_pac = l_pac; b.n_seqs = 0; b.contigs = 0; return depos(&b, pos, is_rev); }

