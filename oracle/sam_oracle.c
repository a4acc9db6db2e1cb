/*
 * sam_oracle.c — CPU restatement of the single-end SAM text step (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 *   mem_reg2sam       /root/reference/src/bwamem.cpp:2091-2150      (which regions become records, supplementary flag, mapq cap)
 *   mem_gen_alt       /root/reference/src/bwamem_extra.cpp:123-187  (XA strings; get_pri_idx)
 *   mem_aln2sam       /root/reference/src/bwamem.cpp:2380-2531      (the record, m = NULL: single-end; V17 build)
 *   kputw / kputl     /root/reference/src/kstring.h:92-141
 * as worker_sam's single-end branch calls them (bwamem.cpp:1836-1844), after mem_mark_primary_se.  mem_reorder_primary5
 * (MEM_F_PRIMARY5) and MEM_F_REF_HDR are not restated.
 *
 * PARITY UNPINNED: bwamem.cpp / bwamem_extra.cpp include safestringlib (not buildable here) and the reference ships no
 * SAM fixtures.  Checked through properties (tests/test_oracle_sam.py): every line has the eleven mandatory fields, FLAG /
 * POS / CIGAR / NM / MD equal the mem_reg2aln record, SEQ and QUAL are the read (reverse-complemented / reversed on the
 * reverse strand, trimmed by the hard clips), the CIGAR's query length equals the SEQ length, XA / SA entries name regions
 * of the read.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "bwams_oracle.h"

typedef struct { char *s; int64_t l, cap; int over; } sbuf_t;

static void sputc(sbuf_t *b, int c) { if (b->l < b->cap) b->s[b->l] = (char)c; else b->over = 1; ++b->l; }
static void sputsn(sbuf_t *b, const char *p, int64_t n) { for (int64_t i = 0; i < n; ++i) sputc(b, p[i]); }
static void sputs(sbuf_t *b, const char *p) { sputsn(b, p, (int64_t)strlen(p)); }
static void sputl(sbuf_t *b, long c)                     /* kputl; kputw is the same over int */
{
    char buf[32];
    long l = 0, x;
    if (c == 0) { sputc(b, '0'); return; }
    for (x = c < 0 ? -c : c; x > 0; x /= 10) buf[l++] = (char)(x % 10 + '0');
    if (c < 0) buf[l++] = '-';
    for (x = l - 1; x >= 0; --x) sputc(b, buf[x]);
}

typedef struct {                 /* mem_aln_t with its arrays */
    bwams_aln_t a;
    uint32_t *cigar;
    char *md;
    int xa_of;                   /* region whose XA string this record carries, -1: none */
} aln_t;

static int get_pri_idx(double XA_drop_ratio, const bwams_alnreg_t *a, int i)
{
    int k = a[i].secondary_all;
    if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
    return -1;
}

static int reg_is_alt(const bwams_alnreg_t *p) { return (int)(((uint32_t)p->n_comp_is_alt >> 30) & 3) != 0; }

static void run_reg2aln(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, int l_seq, const uint8_t *seq,
                        const bwams_alnreg_t *ar, aln_t *t)
{
    const int64_t span = ar && ar->re > ar->rb ? ar->re - ar->rb : 0;
    t->cigar = (uint32_t *)calloc((size_t)(l_seq + span + 8), 4);
    t->md = (char *)calloc((size_t)(3 * span + 32), 1);
    t->xa_of = -1;
    orc_reg2aln(opt, bns, ref_string, l_seq, seq, ar, &t->a, t->cigar, t->md);
}

/* add_cigar (bwamem.cpp:2380-2391) */
static void add_cigar(const bwams_sam_opt_t *so, const aln_t *p, sbuf_t *str, int which)
{
    if (p->a.n_cigar) {
        for (int i = 0; i < p->a.n_cigar; ++i) {
            int c = (int)(p->cigar[i] & 0xf);
            if (!(so->flag & BWAMS_MEM_F_SOFTCLIP) && !p->a.is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
            sputl(str, (long)(p->cigar[i] >> 4)); sputc(str, "MIDSH"[c]);
        }
    } else sputc(str, '*');
}

/* the XA string of primary region r (mem_gen_alt builds one per region; NULL when nothing was appended) */
static int gen_alt_for(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                       const int32_t *ctg_off, const uint8_t *ref_string, const bwams_alnreg_t *a, int n, int l_seq,
                       const uint8_t *seq, int r, sbuf_t *out)
{
    int cnt = 0, has_alt = 0, any = 0;
    for (int i = 0; i < n; ++i)
        if (get_pri_idx(so->XA_drop_ratio, a, i) == r) { ++cnt; if (reg_is_alt(&a[i])) has_alt = 1; }
    if (cnt == 0) return 0;
    if (cnt > so->max_XA_hits_alt || (!has_alt && cnt > so->max_XA_hits)) return 0;
    for (int i = 0; i < n; ++i) {
        aln_t t;
        if (get_pri_idx(so->XA_drop_ratio, a, i) != r) continue;
        run_reg2aln(opt, bns, ref_string, l_seq, seq, &a[i], &t);
        sputs(out, ctg_names + ctg_off[t.a.rid]);
        sputc(out, ','); sputc(out, "+-"[t.a.is_rev]); sputl(out, (long)(t.a.pos + 1));
        sputc(out, ',');
        for (int k = 0; k < t.a.n_cigar; ++k) { sputl(out, (long)(t.cigar[k] >> 4)); sputc(out, "MIDSHN"[t.cigar[k] & 0xf]); }
        sputc(out, ','); sputl(out, t.a.NM);
        sputc(out, ';');
        free(t.cigar); free(t.md);
        any = 1;
    }
    return any;
}

/* mem_aln2sam with m = NULL */
static void aln2sam(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                    const int32_t *ctg_off, const uint8_t *ref_string, sbuf_t *str, int l_seq, const uint8_t *seq, const char *qual,
                    const char *name, const char *comment, int n, const aln_t *list, int which, const bwams_alnreg_t *regs, int n_regs)
{
    aln_t ptmp = list[which], *p = &ptmp;
    int i;
    p->a.flag |= p->a.rid < 0 ? 0x4 : 0;
    p->a.flag |= p->a.is_rev ? 0x10 : 0;
    sputs(str, name); sputc(str, '\t');
    sputl(str, (p->a.flag & 0xffff) | (p->a.flag & 0x10000 ? 0x100 : 0)); sputc(str, '\t');
    if (p->a.rid >= 0) {
        sputs(str, ctg_names + ctg_off[p->a.rid]); sputc(str, '\t');
        sputl(str, (long)(p->a.pos + 1)); sputc(str, '\t');
        sputl(str, p->a.mapq); sputc(str, '\t');
        add_cigar(so, p, str, which);
    } else sputsn(str, "*\t0\t0\t*", 7);
    sputc(str, '\t');
    sputsn(str, "*\t0\t0", 5);
    sputc(str, '\t');
    if (p->a.flag & 0x100) {
        sputsn(str, "*\t*", 3);
    } else if (!p->a.is_rev) {
        int qb = 0, qe = l_seq;
        if (p->a.n_cigar && which && !(so->flag & BWAMS_MEM_F_SOFTCLIP) && !p->a.is_alt) {
            if ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3) qb += (int)(p->cigar[0] >> 4);
            if ((p->cigar[p->a.n_cigar - 1] & 0xf) == 4 || (p->cigar[p->a.n_cigar - 1] & 0xf) == 3) qe -= (int)(p->cigar[p->a.n_cigar - 1] >> 4);
        }
        for (i = qb; i < qe; ++i) sputc(str, "ACGTN"[seq[i]]);
        sputc(str, '\t');
        if (qual) { for (i = qb; i < qe; ++i) sputc(str, qual[i]); }
        else sputc(str, '*');
    } else {
        int qb = 0, qe = l_seq;
        if (p->a.n_cigar && which && !(so->flag & BWAMS_MEM_F_SOFTCLIP) && !p->a.is_alt) {
            if ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3) qe -= (int)(p->cigar[0] >> 4);
            if ((p->cigar[p->a.n_cigar - 1] & 0xf) == 4 || (p->cigar[p->a.n_cigar - 1] & 0xf) == 3) qb += (int)(p->cigar[p->a.n_cigar - 1] >> 4);
        }
        for (i = qe - 1; i >= qb; --i) sputc(str, "TGCAN"[seq[i]]);
        sputc(str, '\t');
        if (qual) { for (i = qe - 1; i >= qb; --i) sputc(str, qual[i]); }
        else sputc(str, '*');
    }
    if (p->a.n_cigar) {
        sputsn(str, "\tNM:i:", 6); sputl(str, p->a.NM);
        sputsn(str, "\tMD:Z:", 6); sputs(str, p->md);
    }
    if (p->a.score >= 0) { sputsn(str, "\tAS:i:", 6); sputl(str, p->a.score); }
    if (p->a.sub >= 0) { sputsn(str, "\tXS:i:", 6); sputl(str, p->a.sub); }
    if (so->rg_id[0]) { sputsn(str, "\tRG:Z:", 6); sputs(str, so->rg_id); }
    if (!(p->a.flag & 0x100)) {
        for (i = 0; i < n; ++i)
            if (i != which && !(list[i].a.flag & 0x100)) break;
        if (i < n) {
            sputsn(str, "\tSA:Z:", 6);
            for (i = 0; i < n; ++i) {
                const aln_t *r = &list[i];
                if (i == which || (r->a.flag & 0x100)) continue;
                sputs(str, ctg_names + ctg_off[r->a.rid]); sputc(str, ',');
                sputl(str, (long)(r->a.pos + 1)); sputc(str, ',');
                sputc(str, "+-"[r->a.is_rev]); sputc(str, ',');
                for (int k = 0; k < r->a.n_cigar; ++k) { sputl(str, (long)(r->cigar[k] >> 4)); sputc(str, "MIDSH"[r->cigar[k] & 0xf]); }
                sputc(str, ','); sputl(str, r->a.mapq);
                sputc(str, ','); sputl(str, r->a.NM);
                sputc(str, ';');
            }
        }
        if (p->a.alt_sc > 0) {
            char tmp[64];
            snprintf(tmp, sizeof tmp, "\tpa:f:%.3f", (double)p->a.score / p->a.alt_sc);
            sputs(str, tmp);
        }
    }
    if (p->xa_of >= 0) {
        /* the string is produced again here rather than stored: same bytes */
        sbuf_t probe = {0, 0, 0, 0};
        if (gen_alt_for(opt, so, bns, ctg_names, ctg_off, ref_string, regs, n_regs, l_seq, seq, p->xa_of, &probe)) {
            sputsn(str, "\tXA:Z:", 6);
            gen_alt_for(opt, so, bns, ctg_names, ctg_off, ref_string, regs, n_regs, l_seq, seq, p->xa_of, str);
        }
    }
    if (comment) { sputc(str, '\t'); sputs(str, comment); }
    sputc(str, '\n');
}

/* mem_reg2sam(opt, bns, pac, s, a, extra_flag = 0, m = NULL) for one read.  Returns the length of the text (the bytes are
 * written while they fit into cap; -1 - length when they did not). */
int64_t orc_reg2sam_se(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                       const int32_t *ctg_off, const uint8_t *ref_string, int l_seq, const uint8_t *seq, const char *qual,
                       const char *name, const char *comment, const bwams_alnreg_t *regs, int n_regs, char *out, int64_t cap)
{
    sbuf_t str = {out, 0, cap, 0};
    aln_t *aa = (aln_t *)calloc((size_t)(n_regs > 0 ? n_regs : 1), sizeof(aln_t));
    int n_aa = 0, l = 0;
    const int want_xa = !(so->flag & BWAMS_MEM_F_ALL);
    for (int k = 0; k < n_regs; ++k) {
        const bwams_alnreg_t *p = &regs[k];
        aln_t *q;
        if (p->score < so->T) continue;
        if (p->secondary >= 0 && (reg_is_alt(p) || !(so->flag & BWAMS_MEM_F_ALL))) continue;
        if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < regs[p->secondary].score * opt->drop_ratio) continue;
        q = &aa[n_aa++];
        run_reg2aln(opt, bns, ref_string, l_seq, seq, p, q);
        q->xa_of = want_xa ? k : -1;
        if (p->secondary >= 0) q->a.sub = -1;
        if (l && p->secondary < 0) q->a.flag |= (so->flag & BWAMS_MEM_F_NO_MULTI) ? 0x10000 : 0x800;
        if (!(so->flag & BWAMS_MEM_F_KEEP_SUPP_MAPQ) && l && !reg_is_alt(p) && q->a.mapq > aa[0].a.mapq) q->a.mapq = aa[0].a.mapq;
        ++l;
    }
    if (n_aa == 0) {
        aln_t t;
        run_reg2aln(opt, bns, ref_string, l_seq, seq, 0, &t);
        aln2sam(opt, so, bns, ctg_names, ctg_off, ref_string, &str, l_seq, seq, qual, name, comment, 1, &t, 0, regs, n_regs);
        free(t.cigar); free(t.md);
    } else {
        for (int k = 0; k < n_aa; ++k)
            aln2sam(opt, so, bns, ctg_names, ctg_off, ref_string, &str, l_seq, seq, qual, name, comment, n_aa, aa, k, regs, n_regs);
        for (int k = 0; k < n_aa; ++k) { free(aa[k].cigar); free(aa[k].md); }
    }
    free(aa);
    return str.over ? -1 - str.l : str.l;
}
