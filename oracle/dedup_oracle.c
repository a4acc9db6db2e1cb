/*
 * dedup_oracle.c — CPU restatement of the tail of mem_kernel2_core: dropping purged regions,
 * mem_sort_dedup_patch with mem_patch_reg, the ALT mark (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * Follows, in /root/reference/src:
 *   mem_kernel2_core (tail)     bwamem.cpp:1446-1481   (orc_regs_finish)
 *   mem_sort_dedup_patch        bwamem.cpp:314-375     (sort_dedup_patch)
 *   mem_patch_reg               bwamem.cpp:199-250     (patch_reg)
 *   bwa_gen_cigar2 (score only) bwa.cpp:380-428        (gen_score)
 *   ksw_global2 (score only)    ksw.cpp:558-649        (orc_ksw_global2_score)
 *   ks_introsort(mem_ars2 / mem_ars)   ksort.h, bwamem.cpp:176-180   (idx_introsort with ars2_lt / ars_lt)
 *
 * PINNING: ksw_global2 is pinned against the reference's own ksw.cpp object (oracle/_ref/libref_sw_*.so) and
 * the two sorts against the reference's ksort.h (oracle/_ref/libref_chain.so) in tests/test_oracle_dedup.py;
 * mem_sort_dedup_patch / mem_patch_reg themselves live in bwamem.cpp (not buildable here): PARITY UNPINNED.
 */
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

#define MINUS_INF (-0x40000000)
#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

typedef struct { int32_t h, e; } eh_t;

int orc_ksw_global2_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del,
                          int e_del, int o_ins, int e_ins, int w)
{
    const int m = 5, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, k, score;
    int8_t *qp = (int8_t *)malloc((size_t)(qlen > 0 ? qlen : 1) * m);
    eh_t *eh = (eh_t *)calloc((size_t)qlen + 1, 8);
    for (k = i = 0; k < m; ++k) {
        const int8_t *p = &mat[k * m];
        for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
    }
    eh[0].h = 0; eh[0].e = MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) eh[j].h = -(o_ins + e_ins * j), eh[j].e = MINUS_INF;
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
    for (i = 0; i < tlen; ++i) {
        int32_t f = MINUS_INF, h1, beg, end, t;
        const int8_t *q = &qp[target[i] * qlen];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j];
            int32_t h, mm = p->h, e = p->e;
            p->h = h1;
            mm += q[j];
            h = mm >= e ? mm : e;
            h = h >= f ? h : f;
            h1 = h;
            t = mm - oe_del;
            e -= e_del;
            e = e > t ? e : t;
            p->e = e;
            t = mm - oe_ins;
            f -= e_ins;
            f = f > t ? f : t;
        }
        eh[end].h = h1; eh[end].e = MINUS_INF;
    }
    score = eh[qlen].h;
    free(eh); free(qp);
    return score;
}

/* bwa_gen_cigar2 with n_cigar = NM = 0: the score of the global alignment of query[0, l_query) against
 * ref_string[rb, re); both are reversed on the reverse strand (bwa.cpp:394-399) */
int64_t orc_dedup_dp_calls = 0, orc_dedup_dp_cells = 0;      /* test / diagnostic counters */
static int gen_score(const bwams_mem_opt_t *opt, int w_, int64_t l_pac, const uint8_t *ref_string, int l_query,
                     const uint8_t *query, int64_t rb, int64_t re, int *score)
{
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
    const int64_t rlen = re - rb;
    uint8_t *q = (uint8_t *)malloc((size_t)l_query), *r = (uint8_t *)malloc((size_t)rlen);
    if (rb >= l_pac) {
        for (int i = 0; i < l_query; ++i) q[i] = query[l_query - 1 - i];
        for (int64_t i = 0; i < rlen; ++i) r[i] = ref_string[rb + rlen - 1 - i];
    } else {
        memcpy(q, query, (size_t)l_query);
        memcpy(r, ref_string + rb, (size_t)rlen);
    }
    if (l_query == re - rb && w_ == 0) {
        *score = 0;
        for (int i = 0; i < l_query; ++i) *score += opt->mat[r[i] * 5 + q[i]];
    } else {
        int w, max_gap, max_ins, max_del, min_w;
        max_ins = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_ins) / opt->e_ins + 1.);
        max_del = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_del) / opt->e_del + 1.);
        max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = (max_gap + abs((int)(rlen - l_query)) + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = abs((int)(rlen - l_query)) + 3;
        w = w > min_w ? w : min_w;
        orc_dedup_dp_calls++; orc_dedup_dp_cells += (int64_t)rlen * (2 * w + 1 < l_query ? 2 * w + 1 : l_query);
        *score = orc_ksw_global2_score(l_query, q, (int)rlen, r, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w);
    }
    free(q); free(r);
    return 1;
}

static int patch_reg(const bwams_mem_opt_t *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query,
                     const bwams_alnreg_t *a, const bwams_alnreg_t *b, int *_w)
{
    int w, score = 0, q_s, r_s;
    double r;
    if (ref_string == 0 || query == 0) return 0;             /* bwamem.cpp:206: mate rescue passes bns = pac = query = 0 */
    if (a->rb < l_pac && b->rb >= l_pac) return 0;
    if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
    w = (int)((a->re - b->rb) - (a->qe - b->qb));
    w = w > 0 ? w : -w;
    r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
    r = r > 0. ? r : -r;
    if (a->re < b->rb || a->qe < b->qb) {
        if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
    } else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
    w += a->w + b->w;
    w = w < opt->w << 2 ? w : opt->w << 2;
    gen_score(opt, w, l_pac, ref_string, b->qe - a->qb, query + a->qb, a->rb, b->re, &score);
    q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
    r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
    if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
    *_w = w;
    return score;
}

/* ks_introsort over an index array; lt(ctx, i, j) compares elements i and j of the caller's array */
typedef int (*lt_fn)(const void *ctx, int a, int b);
static void idx_insertsort(int *s, int *t, lt_fn lt, const void *ctx)
{
    int *i, *j, tmp;
    for (i = s + 1; i < t; ++i)
        for (j = i; j > s && lt(ctx, *j, *(j - 1)); --j) { tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
static void idx_combsort(size_t n, int *a, lt_fn lt, const void *ctx)
{
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap, tmp, *i, *j;
    size_t gap = n;
    do {
        if (gap > 2) {
            gap = (size_t)(gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (i = a; i < a + n - gap; ++i) {
            j = i + gap;
            if (lt(ctx, *j, *i)) { tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) idx_insertsort(a, a + n, lt, ctx);
}
void orc_idx_introsort(size_t n, int *a, lt_fn lt, const void *ctx)
{
    struct { int *left, *right; int depth; } stack[8 * 64 + 2], *top = stack;
    int d, rp, tmp, *s, *t, *i, *j, *k;
    if (n < 1) return;
    if (n == 2) { if (lt(ctx, a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }
    for (d = 2; 1ul << d < n; ++d);
    s = a; t = a + (n - 1); d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { idx_combsort((size_t)(t - s) + 1, s, lt, ctx); t = s; continue; }
            i = s; j = t; k = i + ((j - i) >> 1) + 1;
            if (lt(ctx, *k, *i)) { if (lt(ctx, *k, *j)) k = j; }
            else k = lt(ctx, *j, *i) ? i : j;
            rp = *k;
            if (k != t) { tmp = *k; *k = *t; *t = tmp; }
            for (;;) {
                do ++i; while (lt(ctx, *i, rp));
                do --j; while (i <= j && lt(ctx, rp, *j));
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
            if (top == stack) { idx_insertsort(a, a + n, lt, ctx); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}
static int ars2_lt(const void *ctx, int a, int b)
{
    const bwams_alnreg_t *r = (const bwams_alnreg_t *)ctx;
    return r[a].re < r[b].re;
}
static int ars_lt(const void *ctx, int a, int b)
{
    const bwams_alnreg_t *r = (const bwams_alnreg_t *)ctx;
    return r[a].score > r[b].score || (r[a].score == r[b].score && (r[a].rb < r[b].rb || (r[a].rb == r[b].rb && r[a].qb < r[b].qb)));
}
/* test hooks for the two sorts */
typedef struct { const int64_t *k0, *k1, *k2; } sortkeys_t;
static int hook_ars2_lt(const void *ctx, int a, int b) { const sortkeys_t *k = (const sortkeys_t *)ctx; return k->k0[a] < k->k0[b]; }
static int hook_ars_lt(const void *ctx, int a, int b)
{
    const sortkeys_t *k = (const sortkeys_t *)ctx;
    return k->k0[a] > k->k0[b] || (k->k0[a] == k->k0[b] && (k->k1[a] < k->k1[b] || (k->k1[a] == k->k1[b] && k->k2[a] < k->k2[b])));
}
void orc_ars_sort(int64_t n, int which, const int64_t *k0, const int64_t *k1, const int64_t *k2, int32_t *order)
{
    sortkeys_t k = {k0, k1, k2};
    for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    orc_idx_introsort((size_t)n, order, which ? hook_ars_lt : hook_ars2_lt, &k);
}

static void permute(bwams_alnreg_t *a, int n, const int *ord)
{
    bwams_alnreg_t *tmp = (bwams_alnreg_t *)malloc((size_t)n * sizeof *tmp);
    for (int i = 0; i < n; ++i) tmp[i] = a[ord[i]];
    memcpy(a, tmp, (size_t)n * sizeof *tmp);
    free(tmp);
}
#define N_COMP(r) ((r).n_comp_is_alt & 0x3fffffff)
#define SET_N_COMP(r, v) ((r).n_comp_is_alt = ((r).n_comp_is_alt & ~0x3fffffff) | ((v) & 0x3fffffff))

static int sort_dedup_patch(const bwams_mem_opt_t *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n,
                            bwams_alnreg_t *a)
{
    int m, i, j;
    if (n <= 1) return n;
    int *ord = (int *)malloc((size_t)n * sizeof(int));
    for (i = 0; i < n; ++i) ord[i] = i;
    orc_idx_introsort((size_t)n, ord, ars2_lt, a);
    permute(a, n, ord);
    for (i = 0; i < n; ++i) SET_N_COMP(a[i], 1);
    for (i = 1; i < n; ++i) {
        bwams_alnreg_t *p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
            bwams_alnreg_t *q = &a[j];
            int64_t or_, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;
            or_ = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            } else if (q->rb < p->rb && (score = patch_reg(opt, l_pac, ref_string, query, q, p, &w)) > 0) {
                SET_N_COMP(*p, N_COMP(*p) + N_COMP(*q) + 1);
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    n = m;
    for (i = 0; i < n; ++i) ord[i] = i;
    orc_idx_introsort((size_t)n, ord, ars_lt, a);
    permute(a, n, ord);
    for (i = 1; i < n; ++i)
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    free(ord);
    return m;
}

/* mem_dedup_patch (bwamem.cpp:262-312): the pairwise pass of mem_sort_dedup_patch on the list as it stands (the
 * caller keeps it sorted by end), no sorting, no identical-hit pass */
int orc_dedup_patch(const bwams_mem_opt_t *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n,
                    bwams_alnreg_t *a)
{
    int m, i, j;
    if (n <= 1) return n;
    for (i = 0; i < n; ++i) SET_N_COMP(a[i], 1);
    for (i = 1; i < n; ++i) {
        bwams_alnreg_t *p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
            bwams_alnreg_t *q = &a[j];
            int64_t or_, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;
            or_ = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            } else if (q->rb < p->rb && (score = patch_reg(opt, l_pac, ref_string, query, q, p, &w)) > 0) {
                SET_N_COMP(*p, N_COMP(*p) + N_COMP(*q) + 1);
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    return m;
}
/* sort_alnreg_re / sort_alnreg_score (bwamem.cpp:188-194) */
void orc_sort_alnreg(int n, bwams_alnreg_t *a, int by_score)
{
    if (n < 2) return;
    int *ord = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) ord[i] = i;
    orc_idx_introsort((size_t)n, ord, by_score ? ars_lt : ars2_lt, a);
    permute(a, n, ord);
    free(ord);
}

/* mem_sort_dedup_patch for callers outside this file (pair_oracle.c: query = ref_string = NULL, no patching) */
int orc_sort_dedup_patch(const bwams_mem_opt_t *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n,
                         bwams_alnreg_t *a)
{
    return sort_dedup_patch(opt, l_pac, ref_string, query, n, a);
}

/* The tail of mem_kernel2_core for a work item: regs (grouped by read, reg_off[nseq+1]) are compacted in
 * place per read; out_off[nseq+1] receives the new grouping.  Returns the number of regions left. */
int64_t orc_regs_finish(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const uint8_t *enc_qdb,
                        const int64_t *cum_len, int32_t nseq, bwams_alnreg_t *regs, const int64_t *reg_off, int64_t *out_off)
{
    int64_t n_out = 0;
    for (int l = 0; l < nseq; ++l) {
        bwams_alnreg_t *a = regs + reg_off[l];
        const int n = (int)(reg_off[l + 1] - reg_off[l]);
        int i, m;
        out_off[l] = n_out;
        for (i = 0, m = 0; i < n; ++i)                       /* bwamem.cpp:1446-1456 */
            if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
        m = sort_dedup_patch(opt, bns->l_pac, ref_string, enc_qdb + cum_len[l], m, a);
        for (i = 0; i < m; ++i)                              /* :1470-1481 */
            if (a[i].rid >= 0 && bns->contigs[a[i].rid].is_alt) a[i].n_comp_is_alt = (a[i].n_comp_is_alt & 0x3fffffff) | (1 << 30);
        memmove(regs + n_out, a, (size_t)m * sizeof *a);
        n_out += m;
    }
    out_off[nseq] = n_out;
    return n_out;
}

/* ---- mem_pestat (bwamem_pair.cpp:57-156): insert-size statistics of a chunk of read pairs ---- */
#include <math.h>
#define MIN_RATIO     0.8
#define MIN_DIR_CNT   10
#define MIN_DIR_RATIO 0.05
#define OUTLIER_BOUND 2.0
#define MAPPING_BOUND 3.0
#define MAX_STDDEV    4.0

static int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
    int64_t p2;
    int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
static int cal_sub(const bwams_mem_opt_t *opt, int n, const bwams_alnreg_t *a)
{
    int j;
    for (j = 1; j < n; ++j) {
        int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
        int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
        if (e_min > b_max) {
            int min_l = a[j].qe - a[j].qb < a[0].qe - a[0].qb ? a[j].qe - a[j].qb : a[0].qe - a[0].qb;
            if (e_min - b_max >= min_l * opt->mask_level) break;
        }
    }
    return j < n ? a[j].score : opt->min_seed_len * opt->a;
}
static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
/* the (orientation << 60 | insert size) key of every qualifying pair, in pair order; returns the count */
int64_t orc_pestat_keys(const bwams_mem_opt_t *opt, int64_t l_pac, int n, const bwams_alnreg_t *regs, const int64_t *reg_off,
                        uint64_t *keys)
{
    int64_t m = 0;
    for (int i = 0; i < n >> 1; ++i) {
        const bwams_alnreg_t *r0 = regs + reg_off[i << 1], *r1 = regs + reg_off[i << 1 | 1];
        const int n0 = (int)(reg_off[(i << 1) + 1] - reg_off[i << 1]), n1 = (int)(reg_off[(i << 1 | 1) + 1] - reg_off[i << 1 | 1]);
        int64_t is;
        if (n0 == 0 || n1 == 0) continue;
        if (cal_sub(opt, n0, r0) > MIN_RATIO * r0[0].score) continue;
        if (cal_sub(opt, n1, r1) > MIN_RATIO * r1[0].score) continue;
        if (r0[0].rid != r1[0].rid) continue;
        const int dir = infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
        if (is && is <= opt->max_ins) keys[m++] = (uint64_t)dir << 60 | (uint64_t)is;
    }
    return m;
}

void orc_pestat(const bwams_mem_opt_t *opt, int64_t l_pac, int n, const bwams_alnreg_t *regs, const int64_t *reg_off,
                bwams_pestat_t pes[4])
{
    int i, d, max;
    uint64_t *isz[4];
    size_t cnt[4] = {0, 0, 0, 0};
    memset(pes, 0, 4 * sizeof(bwams_pestat_t));
    for (d = 0; d < 4; ++d) isz[d] = (uint64_t *)malloc(((size_t)(n >> 1) + 1) * 8);
    for (i = 0; i < n >> 1; ++i) {
        const bwams_alnreg_t *r0 = regs + reg_off[i << 1], *r1 = regs + reg_off[i << 1 | 1];
        const int n0 = (int)(reg_off[(i << 1) + 1] - reg_off[i << 1]), n1 = (int)(reg_off[(i << 1 | 1) + 1] - reg_off[i << 1 | 1]);
        int64_t is;
        if (n0 == 0 || n1 == 0) continue;
        if (cal_sub(opt, n0, r0) > MIN_RATIO * r0[0].score) continue;
        if (cal_sub(opt, n1, r1) > MIN_RATIO * r1[0].score) continue;
        if (r0[0].rid != r1[0].rid) continue;
        const int dir = infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
        if (is && is <= opt->max_ins) isz[dir][cnt[dir]++] = (uint64_t)is;
    }
    for (d = 0; d < 4; ++d) {
        bwams_pestat_t *r = &pes[d];
        uint64_t *q = isz[d];
        const size_t qn = cnt[d];
        int p25, p50, p75, x;
        size_t k;
        if (qn < MIN_DIR_CNT) { r->failed = 1; continue; }
        qsort(q, qn, 8, cmp_u64);                      /* ks_introsort_64: the sorted values are what matters */
        p25 = (int)q[(int)(.25 * qn + .499)];
        p50 = (int)q[(int)(.50 * qn + .499)];
        p75 = (int)q[(int)(.75 * qn + .499)];
        (void)p50;
        r->low = (int)(p25 - OUTLIER_BOUND * (p75 - p25) + .499);
        if (r->low < 1) r->low = 1;
        r->high = (int)(p75 + OUTLIER_BOUND * (p75 - p25) + .499);
        for (k = 0, x = 0, r->avg = 0; k < qn; ++k)
            if (q[k] >= (uint64_t)r->low && q[k] <= (uint64_t)r->high) r->avg += q[k], ++x;
        r->avg /= x;
        for (k = 0, r->std = 0; k < qn; ++k)
            if (q[k] >= (uint64_t)r->low && q[k] <= (uint64_t)r->high) r->std += (q[k] - r->avg) * (q[k] - r->avg);
        r->std = sqrt(r->std / x);
        r->low = (int)(p25 - MAPPING_BOUND * (p75 - p25) + .499);
        r->high = (int)(p75 + MAPPING_BOUND * (p75 - p25) + .499);
        if (r->low > r->avg - MAX_STDDEV * r->std) r->low = (int)(r->avg - MAX_STDDEV * r->std + .499);
        if (r->high < r->avg + MAX_STDDEV * r->std) r->high = (int)(r->avg + MAX_STDDEV * r->std + .499);
        if (r->low < 1) r->low = 1;
    }
    for (d = 0, max = 0; d < 4; ++d) max = max > (int)cnt[d] ? max : (int)cnt[d];
    for (d = 0; d < 4; ++d)
        if (pes[d].failed == 0 && cnt[d] < max * MIN_DIR_RATIO) pes[d].failed = 1;
    for (d = 0; d < 4; ++d) free(isz[d]);
}
