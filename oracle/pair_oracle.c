/*
 * pair_oracle.c — CPU restatement of the paired-end tail of worker_sam up to the pairing decision
 * (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * Follows, in /root/reference/src:
 *   mem_sam_pe_batch_pre / _post (rescue part)   bwamem_pair.cpp:838-870, :981-1042   (orc_pair_pe, anchors b[i])
 *   mem_matesw_batch_pre / mem_matesw_batch_post bwamem_pair.cpp:1193-1355, :1497-1601 (matesw: the non-ERT form,
 *       whose body is mem_matesw_orig :283-364; the alignment of a rescue window is the same ksw_align2 call
 *       whether it was batched by _pre or made on the spot by _post's index == -1 branch)
 *   bns_fetch_seq                                bntseq.cpp:545-573                   (fetch_window)
 *   mem_mark_primary_se(_core)                   bwamem.cpp:1905-1980                 (mark_primary_se)
 *   mem_pair                                     bwamem_pair.cpp:366-427              (pair)
 *   hash_64                                      utils.h:117-128
 *
 * PINNING: ksw_align2 is pinned against the reference's ksw.cpp object, the (re) and (score, rb, qb) introsorts
 * against the reference's ksort.h (tests/test_oracle_dedup.py).  The sorts of mem_mark_primary_se and mem_pair
 * compare keys that cannot tie (hash_64 is a bijection; pair64_t.y holds a unique index), so any sorting
 * algorithm gives the reference's order.  The driver logic lives in bwamem_pair.cpp / bwamem.cpp, which are
 * not buildable here: PARITY UNPINNED.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

#ifndef M_SQRT1_2
#define M_SQRT1_2 0.70710678118654752440
#endif
#define KSW_XBYTE  0x10000
#define KSW_XSTOP  0x20000
#define KSW_XSUBO  0x40000
#define KSW_XSTART 0x80000

int orc_sort_dedup_patch(const bwams_mem_opt_t *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n,
                         bwams_alnreg_t *a);

#define IS_ALT(r) (((r).n_comp_is_alt >> 30) & 3)

static uint64_t hash_64(uint64_t key)
{
    key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
    key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
    return key;
}
static int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)       /* bwamem_pair.cpp:57-65 */
{
    int64_t p2;
    int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
static int pos2rid(const orc_bns_t *bns, int64_t pos_f)                           /* bntseq.cpp:397-413 */
{
    int left, mid, right;
    if (pos_f >= bns->l_pac) return -1;
    left = 0; mid = 0; right = bns->n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= bns->contigs[mid].offset) {
            if (mid == bns->n_seqs - 1) break;
            if (pos_f < bns->contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
/* bns_fetch_seq without the copy: clip [*beg, *end) to the sequence and strand that hold mid */
static int fetch_window(const orc_bns_t *bns, int64_t *beg, int64_t mid, int64_t *end)
{
    int64_t far_beg, far_end;
    if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
    const int is_rev = mid >= bns->l_pac;
    const int rid = pos2rid(bns, is_rev ? (bns->l_pac << 1) - 1 - mid : mid);
    far_beg = bns->contigs[rid].offset;
    far_end = far_beg + bns->contigs[rid].len;
    if (is_rev) { int64_t t = far_beg; far_beg = (bns->l_pac << 1) - far_end; far_end = (bns->l_pac << 1) - t; }
    *beg = *beg > far_beg ? *beg : far_beg;
    *end = *end < far_end ? *end : far_end;
    return rid;
}

int orc_dedup_patch(const bwams_mem_opt_t *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n,
                    bwams_alnreg_t *a);
void orc_sort_alnreg(int n, bwams_alnreg_t *a, int by_score);

int64_t orc_pair_sw_calls = 0;       /* diagnostic: ksw_align2 calls made by the rescue */

/* mem_matesw_orig: rescue the mate ms of anchor a into the mate's region list ma (n regions, room for 4 more) */
/* ert != 0: mem_matesw_batch_post_ert (bwamem_pair.cpp:1357-1495) — the list is kept sorted by END position: the
 * rescued region goes in front of the first region ending later; if one ends exactly where it does the list is
 * re-sorted by score, the region inserted by score, and everything sorted by end again; then mem_dedup_patch (no
 * sorting) instead of mem_sort_dedup_patch. */
static int matesw(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const bwams_pestat_t pes[4],
                  const bwams_alnreg_t *a, int l_ms, const uint8_t *ms, bwams_alnreg_t *ma, int *ma_n, int ert)
{
    const int64_t l_pac = bns->l_pac;
    int i, r, skip[4], n = 0, rid = -1;
    bwams_sw_opt_t sw;
    memset(&sw, 0, sizeof sw);
    sw.o_del = opt->o_del; sw.e_del = opt->e_del; sw.o_ins = opt->o_ins; sw.e_ins = opt->e_ins;
    memcpy(sw.mat, opt->mat, 25);
    for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
    for (i = 0; i < *ma_n; ++i) {
        int64_t dist;
        r = infer_dir(l_pac, a->rb, ma[i].rb, &dist);
        if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
    }
    if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
    uint8_t *rev = (uint8_t *)malloc((size_t)(l_ms > 0 ? l_ms : 1));
    for (r = 0; r < 4; ++r) {
        int is_rev, is_larger;
        const uint8_t *seq;
        int64_t rb, re;
        if (skip[r]) continue;
        is_rev = (r >> 1 != (r & 1));
        is_larger = !(r >> 1);
        if (is_rev) {
            for (i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
            seq = rev;
        } else seq = ms;
        if (!is_rev) {
            rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
            re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
        } else {
            rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
            re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
        }
        if (rb < 0) rb = 0;
        if (re > l_pac << 1) re = l_pac << 1;
        if (rb < re) rid = fetch_window(bns, &rb, (rb + re) >> 1, &re);
        if (a->rid == rid && re - rb >= opt->min_seed_len) {
            int out[7];
            bwams_alnreg_t b;
            int tmp, xtra = KSW_XSUBO | KSW_XSTART | (l_ms * opt->a < 250 ? KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
            orc_ksw_align2(&sw, l_ms, seq, (int)(re - rb), ref_string + rb, xtra, out);
            ++orc_pair_sw_calls;
            const int score = out[0], te = out[1], qe = out[2], score2 = out[3], tb = out[5], qb = out[6];
            memset(&b, 0, sizeof b);
            if (score >= opt->min_seed_len && qb >= 0) {
                b.rid = a->rid;
                b.n_comp_is_alt = (int32_t)((uint32_t)IS_ALT(*a) << 30);
                b.qb = is_rev ? l_ms - (qe + 1) : qb;
                b.qe = is_rev ? l_ms - qb : qe + 1;
                b.rb = is_rev ? (l_pac << 1) - (rb + te + 1) : rb + tb;
                b.re = is_rev ? (l_pac << 1) - (rb + tb) : rb + te + 1;
                b.score = score;
                b.csub = score2;
                b.secondary = -1;
                b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
                ++*ma_n;
                if (!ert) {
                    for (i = 0; i < *ma_n - 1; ++i)
                        if (ma[i].score < b.score) break;
                    tmp = i;
                    for (i = *ma_n - 1; i > tmp; --i) ma[i] = ma[i - 1];
                    ma[i] = b;
                } else {
                    int resort = 0;
                    for (i = 0; i < *ma_n - 1; ++i) {
                        if (ma[i].re == b.re) { resort = 1; break; }
                        if (ma[i].re > b.re) break;
                    }
                    if (resort) {
                        orc_sort_alnreg(*ma_n - 1, ma, 1);
                        for (i = 0; i < *ma_n - 1; ++i)
                            if (ma[i].score < b.score) break;
                        tmp = i;
                        for (i = *ma_n - 1; i > tmp; --i) ma[i] = ma[i - 1];
                        ma[i] = b;
                        orc_sort_alnreg(*ma_n, ma, 0);
                    } else {
                        tmp = i;
                        for (i = *ma_n - 1; i > tmp; --i) ma[i] = ma[i - 1];
                        ma[i] = b;
                    }
                }
            }
            ++n;
        }
        if (n) *ma_n = ert ? orc_dedup_patch(opt, l_pac, 0, 0, *ma_n, ma) : orc_sort_dedup_patch(opt, l_pac, 0, 0, *ma_n, ma);
    }
    free(rev);
    return n;
}

/* sort of (score desc, is_alt asc, hash asc) or (is_alt asc, score desc, hash asc): keys never tie */
static int cmp_hash(const void *x, const void *y)
{
    const bwams_alnreg_t *a = (const bwams_alnreg_t *)x, *b = (const bwams_alnreg_t *)y;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    if (IS_ALT(*a) != IS_ALT(*b)) return IS_ALT(*a) < IS_ALT(*b) ? -1 : 1;
    return a->hash < b->hash ? -1 : a->hash > b->hash;
}
static int cmp_hash2(const void *x, const void *y)
{
    const bwams_alnreg_t *a = (const bwams_alnreg_t *)x, *b = (const bwams_alnreg_t *)y;
    if (IS_ALT(*a) != IS_ALT(*b)) return IS_ALT(*a) < IS_ALT(*b) ? -1 : 1;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    return a->hash < b->hash ? -1 : a->hash > b->hash;
}
static void mark_primary_core(const bwams_mem_opt_t *opt, int n, bwams_alnreg_t *a, int *z)
{
    int i, k, tmp, zn = 0;
    tmp = opt->a + opt->b;
    tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
    tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
    z[zn++] = 0;
    for (i = 1; i < n; ++i) {
        for (k = 0; k < zn; ++k) {
            int j = z[k];
            int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
            int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
            if (e_min > b_max) {
                int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
                if (e_min - b_max >= min_l * opt->mask_level) {
                    if (a[j].sub == 0) a[j].sub = a[i].score;
                    if (a[j].score - a[i].score <= tmp && (IS_ALT(a[j]) || !IS_ALT(a[i]))) ++a[j].sub_n;
                    break;
                }
            }
        }
        if (k == zn) z[zn++] = i;
        else a[i].secondary = z[k];
    }
}
int orc_mark_primary_se(const bwams_mem_opt_t *opt, int n, bwams_alnreg_t *a, int64_t id)
{
    int i, n_pri;
    if (n == 0) return 0;
    int *z = (int *)malloc((size_t)n * sizeof(int));
    for (i = n_pri = 0; i < n; ++i) {
        a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64((uint64_t)(id + i));
        if (!IS_ALT(a[i])) ++n_pri;
    }
    qsort(a, (size_t)n, sizeof *a, cmp_hash);
    mark_primary_core(opt, n, a, z);
    for (i = 0; i < n; ++i) {
        bwams_alnreg_t *p = &a[i];
        p->secondary_all = i;
        if (!IS_ALT(*p) && p->secondary >= 0 && IS_ALT(a[p->secondary])) p->alt_sc = a[p->secondary].score;
    }
    if (n_pri >= 0 && n_pri < n) {
        if (n_pri > 0) qsort(a, (size_t)n, sizeof *a, cmp_hash2);
        for (i = 0; i < n; ++i) z[a[i].secondary_all] = i;
        for (i = 0; i < n; ++i) {
            if (a[i].secondary >= 0) {
                a[i].secondary_all = z[a[i].secondary];
                if (IS_ALT(a[i])) a[i].secondary = 0x7fffffff;
            } else a[i].secondary_all = -1;
        }
        if (n_pri > 0) {
            for (i = 0; i < n_pri; ++i) a[i].sub = 0, a[i].secondary = -1;
            mark_primary_core(opt, n_pri, a, z);
        }
    } else {
        for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
    }
    free(z);
    return n_pri;
}

typedef struct { uint64_t x, y; } pair64_t;
static int cmp_128(const void *a, const void *b)
{
    const pair64_t *p = (const pair64_t *)a, *q = (const pair64_t *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return p->y < q->y ? -1 : p->y > q->y;
}
static int pair(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const bwams_pestat_t pes[4], bwams_alnreg_t *a[2],
                int id, int *sub, int *n_sub, int z[2], const int n_pri[2])
{
    const int64_t l_pac = bns->l_pac;
    int r, i, k, y[4], ret;
    size_t vn = 0, un = 0, um = 16;
    pair64_t *v = (pair64_t *)malloc((size_t)(n_pri[0] + n_pri[1] + 1) * sizeof *v);
    pair64_t *u = (pair64_t *)malloc(um * sizeof *u);
    for (r = 0; r < 2; ++r)
        for (i = 0; i < n_pri[r]; ++i) {
            pair64_t key;
            const bwams_alnreg_t *e = &a[r][i];
            key.x = e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb;
            key.x = (uint64_t)e->rid << 32 | (key.x - bns->contigs[e->rid].offset);
            key.y = (uint64_t)e->score << 32 | i << 2 | (e->rb >= l_pac) << 1 | r;
            v[vn++] = key;
        }
    qsort(v, vn, sizeof *v, cmp_128);
    y[0] = y[1] = y[2] = y[3] = -1;
    for (i = 0; i < (int)vn; ++i) {
        for (r = 0; r < 2; ++r) {
            int dir = r << 1 | (v[i].y >> 1 & 1), which;
            if (pes[dir].failed) continue;
            which = r << 1 | ((v[i].y & 1) ^ 1);
            if (y[which] < 0) continue;
            for (k = y[which]; k >= 0; --k) {
                int64_t dist;
                int q;
                double ns;
                if ((v[k].y & 3) != (uint64_t)which) continue;
                dist = (int64_t)v[i].x - v[k].x;
                if (dist > pes[dir].high) break;
                if (dist < pes[dir].low) continue;
                ns = (dist - pes[dir].avg) / pes[dir].std;
                q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a + .499);
                if (q < 0) q = 0;
                if (un == um) { um <<= 1; u = (pair64_t *)realloc(u, um * sizeof *u); }
                u[un].y = (uint64_t)k << 32 | i;
                u[un].x = (uint64_t)q << 32 | (hash_64(u[un].y ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffU);
                ++un;
            }
        }
        y[v[i].y & 3] = i;
    }
    if (un) {
        int tmp = opt->a + opt->b;
        tmp = tmp > opt->o_del + opt->e_del ? tmp : opt->o_del + opt->e_del;
        tmp = tmp > opt->o_ins + opt->e_ins ? tmp : opt->o_ins + opt->e_ins;
        qsort(u, un, sizeof *u, cmp_128);
        i = (int)(u[un - 1].y >> 32); k = (int)(u[un - 1].y << 32 >> 32);
        z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
        z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
        ret = (int)(u[un - 1].x >> 32);
        *sub = un > 1 ? (int)(u[un - 2].x >> 32) : 0;
        for (i = (int)un - 2, *n_sub = 0; i >= 0; --i)
            if (*sub - (int)(u[i].x >> 32) <= tmp) ++*n_sub;
    } else ret = 0, *sub = 0, *n_sub = 0;
    free(u); free(v);
    return ret;
}

/* mem_reorder_primary5 (bwamem.cpp:2009-2031), `mem -5`: among the primary, non-ALT regions scoring >= T the one with the
 * smallest query start (first of equals) changes places with a[0]; secondary / secondary_all indices to either follow. */
void orc_reorder_primary5(int T, int n, bwams_alnreg_t *a)
{
    int k, n_pri = 0, left_st = INT32_MAX, left_k = -1;
    bwams_alnreg_t t;
    for (k = 0; k < n; ++k)
        if (a[k].secondary < 0 && !(((uint32_t)a[k].n_comp_is_alt >> 30) & 3) && a[k].score >= T) ++n_pri;
    if (n_pri <= 1) return;
    for (k = 0; k < n; ++k) {
        const bwams_alnreg_t *p = &a[k];
        if (p->secondary >= 0 || (((uint32_t)p->n_comp_is_alt >> 30) & 3) || p->score < T) continue;
        if (p->qb < left_st) left_st = p->qb, left_k = k;
    }
    if (left_k == 0) return;
    t = a[0], a[0] = a[left_k], a[left_k] = t;
    for (k = 1; k < n; ++k) {
        bwams_alnreg_t *p = &a[k];
        if (p->secondary == 0) p->secondary = left_k;
        else if (p->secondary == left_k) p->secondary = 0;
        if (p->secondary_all == 0) p->secondary_all = left_k;
        else if (p->secondary_all == left_k) p->secondary_all = 0;
    }
}

/* For every pair p (reads 2p, 2p+1) of a chunk: mate rescue, mem_mark_primary_se of both ends, mem_pair.
 * flags: 1 = MEM_F_NO_RESCUE, 2 = useErt, 4 = MEM_F_NOPAIRING (bwamem_pair.cpp:1066); primary5_T >= 0 = MEM_F_PRIMARY5 with
 * that opt->T (:1060-1063).
 * regs / reg_off: the final regions per read (orc_regs_finish); out (capacity out_cap) / out_off receive the
 * regions per read afterwards.  id_base = n_processed >> 1 of the chunk.  Returns the region count, -1 on overflow. */
int64_t orc_pair_pe(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const uint8_t *enc_qdb,
                    const int64_t *cum_len, int32_t n_pairs, const bwams_alnreg_t *regs, const int64_t *reg_off,
                    const bwams_pestat_t pes[4], int64_t id_base, int flags, int primary5_T, bwams_alnreg_t *out, int64_t out_cap,
                    int64_t *out_off, bwams_pair_t *pairs)
{
    const int no_rescue = flags & 1, use_ert = (flags >> 1) & 1, no_pairing = (flags >> 2) & 1;
    int64_t n_out = 0;
    for (int p = 0; p < n_pairs; ++p) {
        int n[2], nb[2], i, j;
        bwams_alnreg_t *a[2], *b[2];
        for (i = 0; i < 2; ++i) {
            const int64_t r = 2 * (int64_t)p + i;
            n[i] = (int)(reg_off[r + 1] - reg_off[r]);
            b[i] = (bwams_alnreg_t *)malloc((size_t)(n[i] + 1) * sizeof(bwams_alnreg_t));
            nb[i] = 0;
            for (j = 0; j < n[i]; ++j)
                if (regs[reg_off[r] + j].score >= regs[reg_off[r]].score - opt->pen_unpaired) b[i][nb[i]++] = regs[reg_off[r] + j];
        }
        for (i = 0; i < 2; ++i) {
            const int64_t r = 2 * (int64_t)p + i;
            const int na = nb[!i] < opt->max_matesw ? nb[!i] : opt->max_matesw;
            a[i] = (bwams_alnreg_t *)malloc((size_t)(n[i] + 4 * na + 1) * sizeof(bwams_alnreg_t));
            memcpy(a[i], regs + reg_off[r], (size_t)n[i] * sizeof(bwams_alnreg_t));
        }
        bwams_pair_t *pr = &pairs[p];
        memset(pr, 0, sizeof *pr);
        pr->z[0] = pr->z[1] = -1;
        if (!no_rescue) {
            for (i = 0; i < 2; ++i)
                for (j = 0; j < n[i]; ++j) a[i][j].flg = 0;           /* bwamem_pair.cpp:1011-1014 */
            for (i = 0; i < 2; ++i) {
                int val = 0;
                if (use_ert) orc_sort_alnreg(n[!i], a[!i], 0);                   /* bwamem_pair.cpp:1017-1018 */
                for (j = 0; j < nb[i] && j < opt->max_matesw; ++j) {
                    const int64_t m = 2 * (int64_t)p + !i;
                    val = matesw(opt, bns, ref_string, pes, &b[i][j], (int)(cum_len[m + 1] - cum_len[m]),
                                 enc_qdb + cum_len[m], a[!i], &n[!i], use_ert);
                    pr->n_matesw += val;
                }
                if (use_ert) {                                                    /* :1033-1041: the LAST anchor's return value decides */
                    if (val) n[!i] = orc_sort_dedup_patch(opt, bns->l_pac, 0, 0, n[!i], a[!i]);
                    else orc_sort_alnreg(n[!i], a[!i], 1);
                }
            }
        }
        const int64_t id = id_base + p;
        pr->n_pri[0] = orc_mark_primary_se(opt, n[0], a[0], id << 1 | 0);
        pr->n_pri[1] = orc_mark_primary_se(opt, n[1], a[1], id << 1 | 1);
        if (primary5_T >= 0) {
            orc_reorder_primary5(primary5_T, n[0], a[0]);
            orc_reorder_primary5(primary5_T, n[1], a[1]);
        }
        if (!no_pairing && pr->n_pri[0] && pr->n_pri[1]) {
            int sub = 0, n_sub = 0, z[2] = {-1, -1};
            pr->score = pair(opt, bns, pes, a, (int)id, &sub, &n_sub, z, pr->n_pri);
            pr->sub = sub; pr->n_sub = n_sub; pr->z[0] = z[0]; pr->z[1] = z[1];
        }
        for (i = 0; i < 2; ++i) {
            out_off[2 * (int64_t)p + i] = n_out;
            if (n_out + n[i] > out_cap) return -1;
            memcpy(out + n_out, a[i], (size_t)n[i] * sizeof(bwams_alnreg_t));
            n_out += n[i];
            free(a[i]); free(b[i]);
        }
    }
    out_off[2 * (int64_t)n_pairs] = n_out;
    return n_out;
}

/* exported for the pins of tests/test_oracle_pair.py against the reference headers (utils.h:117-128, bntseq.h:88-91) */
uint64_t orc_hash_64(uint64_t key) { return hash_64(key); }

