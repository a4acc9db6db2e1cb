/*
 * aln_oracle.c — CPU restatement of the SAM-side alignment step (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 *   ksw_global2 with traceback          /root/reference/src/ksw.cpp:558-668      (orc_ksw_global2_cigar)
 *   bwa_gen_cigar2 (CIGAR, NM, MD)      /root/reference/src/bwa.cpp:380-467      (gen_cigar2)
 *   infer_bw, mem_approx_mapq_se        /root/reference/src/bwamem.cpp:2640-2648, :1983-2008
 *   mem_reg2aln                         /root/reference/src/bwamem.cpp:2533-2628 (orc_reg2aln)
 *
 * PINNING: ksw_global2 with its CIGAR is pinned against the reference's own ksw.cpp object (oracle/_ref/libref_sw_*.so,
 * tests/test_oracle_aln.py).  bwa_gen_cigar2 / mem_reg2aln live in bwa.cpp / bwamem.cpp (safestringlib: not buildable
 * here): PARITY UNPINNED, checked through properties (the CIGAR consumes exactly the query and reference spans, NM and MD
 * recomputed independently from the two sequences and the CIGAR).
 *
 * The reference reads the reference bases from the 2-bit .pac through bns_get_seq; for coordinates on the reverse strand
 * (rb >= l_pac) that yields text[rb, re) of the fw || rc text, i.e. the .0123 array this restatement reads directly.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

#define MINUS_INF (-0x40000000)
#define MEM_MAPQ_COEF 30.0

typedef struct { int32_t h, e; } eh_t;

/* CIGAR ops pushed back to front by the traceback, merged like push_cigar; returns the new count */
static int push_cigar(int n, uint32_t *cigar, int op, int len)
{
    if (n == 0 || op != (int)(cigar[n - 1] & 0xf)) cigar[n++] = (uint32_t)len << 4 | (uint32_t)op;
    else cigar[n - 1] += (uint32_t)len << 4;
    return n;
}

/* cigar must hold qlen + tlen + 2 entries */
int orc_ksw_global2_cigar(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del,
                          int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t *cigar)
{
    const int m = 5, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, k, score;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    uint8_t *z = (uint8_t *)malloc((size_t)(n_col > 0 ? n_col : 1) * (size_t)(tlen > 0 ? tlen : 1));
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
        uint8_t *zi = &z[(long)i * n_col];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j];
            int32_t h, mm = p->h, e = p->e;
            uint8_t d;
            p->h = h1;
            mm += q[j];
            d = mm >= e ? 0 : 1;
            h = mm >= e ? mm : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = mm - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            p->e = e;
            t = mm - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            zi[j - beg] = d;
        }
        eh[end].h = h1; eh[end].e = MINUS_INF;
    }
    score = eh[qlen].h;
    {
        int n = 0, which = 0;
        uint32_t tmp;
        i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
        while (i >= 0 && k >= 0) {
            which = z[(long)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) n = push_cigar(n, cigar, 0, 1), --i, --k;
            else if (which == 1) n = push_cigar(n, cigar, 2, 1), --i;
            else n = push_cigar(n, cigar, 1, 1), --k;
        }
        if (i >= 0) n = push_cigar(n, cigar, 2, i + 1);
        if (k >= 0) n = push_cigar(n, cigar, 1, k + 1);
        for (i = 0; i < n >> 1; ++i) tmp = cigar[i], cigar[i] = cigar[n - 1 - i], cigar[n - 1 - i] = tmp;
        *n_cigar_ = n;
    }
    free(eh); free(qp); free(z);
    return score;
}

static int put_num(char *s, int l, int x)          /* kputw */
{
    char buf[16];
    int n = 0;
    if (x == 0) buf[n++] = '0';
    while (x > 0) { buf[n++] = (char)('0' + x % 10); x /= 10; }
    while (n > 0) s[l++] = buf[--n];
    return l;
}

/* bwa_gen_cigar2: returns 0 when the reference returns a null CIGAR.  cigar: qlen + rlen + 2 entries, md: 3 * rlen + 16 bytes
 * (NUL-terminated, *md_len counts the NUL) */
static int gen_cigar2(const bwams_mem_opt_t *opt, int w_, int64_t l_pac, const uint8_t *ref_string, int l_query,
                      const uint8_t *query_in, int64_t rb, int64_t re, int *score, int *n_cigar, uint32_t *cigar, int *NM,
                      char *md, int *md_len)
{
    const int8_t *mat = opt->mat;
    int i;
    *n_cigar = 0; *NM = -1; *md_len = 0;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
    if (rb < 0 || re > 2 * l_pac) return 0;                                  /* bns_get_seq out of range */
    const int64_t rlen = re - rb;
    uint8_t *query = (uint8_t *)malloc((size_t)l_query), *rseq = (uint8_t *)malloc((size_t)rlen);
    memcpy(query, query_in, (size_t)l_query);
    memcpy(rseq, ref_string + rb, (size_t)rlen);
    if (rb >= l_pac) {                               /* reverse both: indels go to the leftmost position */
        uint8_t tmp;
        for (i = 0; i < l_query >> 1; ++i) tmp = query[i], query[i] = query[l_query - 1 - i], query[l_query - 1 - i] = tmp;
        for (i = 0; i < rlen >> 1; ++i) tmp = rseq[i], rseq[i] = rseq[rlen - 1 - i], rseq[rlen - 1 - i] = tmp;
    }
    if (l_query == re - rb && w_ == 0) {
        cigar[0] = (uint32_t)l_query << 4 | 0;
        *n_cigar = 1;
        for (i = 0, *score = 0; i < l_query; ++i) *score += mat[rseq[i] * 5 + query[i]];
    } else {
        int w, max_gap, max_ins, max_del, min_w;
        max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - opt->o_ins) / opt->e_ins + 1.);
        max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - opt->o_del) / opt->e_del + 1.);
        max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = (max_gap + abs((int)(rlen - l_query)) + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = abs((int)(rlen - l_query)) + 3;
        w = w > min_w ? w : min_w;
        *score = orc_ksw_global2_cigar(l_query, query, (int)rlen, rseq, mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w,
                                       n_cigar, cigar);
    }
    {   /* NM and MD */
        int k, x, y, u, n_mm = 0, n_gap = 0, l = 0;
        const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
        for (k = 0, x = y = u = 0; k < *n_cigar; ++k) {
            const int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
            if (op == 0) {
                for (i = 0; i < len; ++i) {
                    if (query[x + i] != rseq[y + i]) {
                        l = put_num(md, l, u);
                        md[l++] = int2base[rseq[y + i]];
                        ++n_mm; u = 0;
                    } else ++u;
                }
                x += len; y += len;
            } else if (op == 2) {
                if (k > 0 && k < *n_cigar - 1) {
                    l = put_num(md, l, u); md[l++] = '^';
                    for (i = 0; i < len; ++i) md[l++] = int2base[rseq[y + i]];
                    u = 0; n_gap += len;
                }
                y += len;
            } else if (op == 1) x += len, n_gap += len;
        }
        l = put_num(md, l, u); md[l++] = 0;
        *NM = n_mm + n_gap;
        *md_len = l;
    }
    free(query); free(rseq);
    return 1;
}

static int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
    int w;
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    if (w < abs(l1 - l2)) w = abs(l1 - l2);
    return w;
}

int orc_approx_mapq_se(const bwams_mem_opt_t *opt, const bwams_alnreg_t *a)
{
    int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
    double identity;
    const int mapQ_coef_len = opt->mapq_coef_len;
    const double mapQ_coef_fac = mapQ_coef_len > 0 ? log((double)mapQ_coef_len) : 0.;
    sub = a->csub > sub ? a->csub : sub;
    if (sub >= a->score) return 0;
    l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
    identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
    if (a->score == 0) {
        mapq = 0;
    } else if (mapQ_coef_len > 0) {
        double tmp;
        tmp = l < mapQ_coef_len ? 1. : mapQ_coef_fac / log(l);
        tmp *= identity * identity;
        mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
    } else {
        mapq = (int)(MEM_MAPQ_COEF * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - a->frac_rep) + .499);
    return mapq;
}

static int pos2rid(const orc_bns_t *b, int64_t pos_f)      /* bns_pos2rid (bntseq.cpp:397-411) */
{
    int left, mid, right;
    if (pos_f >= b->l_pac) return -1;
    left = 0; mid = 0; right = b->n_seqs;
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

/* mem_reg2aln for one region.  cigar: l_query + (re - rb) + 4 entries; md: 3 * (re - rb) + 16 bytes.  Returns 0 for the
 * unmapped record (ar with rb < 0 or re < 0), 1 otherwise. */
int orc_reg2aln(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, int l_query, const uint8_t *query,
                const bwams_alnreg_t *ar, bwams_aln_t *a, uint32_t *cigar, char *md)
{
    int i, w2, tmp, qb, qe, NM = -1, score = 0, is_rev, last_sc = -(1 << 30), n_cigar = 0, l_MD = 0;
    int64_t pos, rb, re;
    memset(a, 0, sizeof *a);
    if (ar == 0 || ar->rb < 0 || ar->re < 0) {
        a->rid = -1; a->pos = -1; a->flag |= 0x4;
        return 0;
    }
    qb = ar->qb; qe = ar->qe; rb = ar->rb; re = ar->re;
    a->mapq = ar->secondary < 0 ? orc_approx_mapq_se(opt, ar) : 0;
    if (ar->secondary >= 0) a->flag |= 0x100;
    tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
    w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
    i = 0;
    do {
        w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
        gen_cigar2(opt, w2, bns->l_pac, ref_string, qe - qb, query + qb, rb, re, &score, &n_cigar, cigar, &NM, md, &l_MD);
        if (score == last_sc || w2 == opt->w << 2) break;
        last_sc = score;
        w2 <<= 1;
    } while (++i < 3 && score < ar->truesc - opt->a);
    a->NM = NM;
    {
        const int64_t p0 = rb < bns->l_pac ? rb : re - 1;
        is_rev = p0 >= bns->l_pac;
        pos = is_rev ? (bns->l_pac << 1) - 1 - p0 : p0;
    }
    a->is_rev = is_rev;
    if (n_cigar > 0) {                               /* squeeze out leading or trailing deletions */
        if ((cigar[0] & 0xf) == 2) {
            pos += cigar[0] >> 4;
            --n_cigar;
            memmove(cigar, cigar + 1, (size_t)n_cigar * 4);
        } else if ((cigar[n_cigar - 1] & 0xf) == 2) {
            --n_cigar;
        }
    }
    if (qb != 0 || qe != l_query) {                  /* add clipping */
        const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
        if (clip5) {
            memmove(cigar + 1, cigar, (size_t)n_cigar * 4);
            cigar[0] = (uint32_t)clip5 << 4 | 3;
            ++n_cigar;
        }
        if (clip3) cigar[n_cigar++] = (uint32_t)clip3 << 4 | 3;
    }
    a->n_cigar = n_cigar;
    a->md_len = l_MD;
    a->rid = pos2rid(bns, pos);
    a->pos = pos - (a->rid >= 0 ? bns->contigs[a->rid].offset : 0);
    a->score = ar->score; a->sub = ar->sub > ar->csub ? ar->sub : ar->csub;
    a->is_alt = ((uint32_t)ar->n_comp_is_alt >> 30) & 1; a->alt_sc = ar->alt_sc;      /* is_alt:2 of the region, :1 of the record */
    return 1;
}
