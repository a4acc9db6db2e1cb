/*
 * sam_oracle.c — CPU restatement of the single-end SAM text step (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 *   mem_reg2sam       /root/reference/src/bwamem.cpp:2091-2150      (which regions become records, supplementary flag, mapq cap)
 *   mem_gen_alt       /root/reference/src/bwamem_extra.cpp:123-187  (XA strings; get_pri_idx)
 *   mem_aln2sam       /root/reference/src/bwamem.cpp:2380-2531      (the record, m = NULL: single-end; V17 build)
 *   kputw / kputl     /root/reference/src/kstring.h:92-141
 * as worker_sam's single-end branch calls them (bwamem.cpp:1836-1844), after mem_mark_primary_se; and the paired-end text:
 *   mem_sam_pe        /root/reference/src/bwamem_pair.cpp:625-833   from mem_pair's result on (= the tail of
 *                     mem_sam_pe_batch_post, :981-1190): multi-hit test, q_pe / q_se, the region edits of the paired branch
 *                     (sub, secondary = -2, the secondary_all switch), the ALT hit, the no_pairing branch with mem_reg2sam.
 *   mem_perfect2sam_cont / mem_aln2sam_perfect   /root/reference/src/bwamem.cpp:2280-2325, :2153-2227   (reads the EMF resolved)
 * mem_reorder_primary5 (MEM_F_PRIMARY5), MEM_F_NOPAIRING and MEM_F_REF_HDR are not restated.
 *
 * PARITY UNPINNED: bwamem.cpp / bwamem_extra.cpp include safestringlib (not buildable here) and the reference ships no
 * SAM fixtures.  Checked through properties (tests/test_oracle_sam.py): every line has the eleven mandatory fields, FLAG /
 * POS / CIGAR / NM / MD equal the mem_reg2aln record, SEQ and QUAL are the read (reverse-complemented / reversed on the
 * reverse strand, trimmed by the hard clips), the CIGAR's query length equals the SEQ length, XA / SA entries name regions
 * of the read.
 */
#include <math.h>
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

static int get_rlen(int n_cigar, const uint32_t *cigar)            /* bwamem.cpp:2639-2648 */
{
    int k, l;
    for (k = l = 0; k < n_cigar; ++k) {
        int op = (int)(cigar[k] & 0xf);
        if (op == 0 || op == 2) l += (int)(cigar[k] >> 4);
    }
    return l;
}

/* mem_aln2sam; m_ = the mate's record or NULL */
/* bntann1_t.anno per sequence for MEM_F_REF_HDR, laid out like the names; set by the test before a call (NULL = none). */
static const char *g_annos;
static const int32_t *g_anno_off;
void orc_set_contig_annos(const char *annos, const int32_t *anno_off) { g_annos = annos; g_anno_off = anno_off; }

static void put_xr(const bwams_sam_opt_t *so, sbuf_t *str, int rid)             /* bwamem.cpp:2522-2529, :2218-2225 */
{
    const char *a;
    if (!(so->flag & BWAMS_MEM_F_REF_HDR) || rid < 0 || !g_annos) return;
    a = g_annos + g_anno_off[rid];
    if (!a[0]) return;
    sputsn(str, "\tXR:Z:", 6);
    for (; *a; ++a) sputc(str, *a == '\t' ? ' ' : *a);
}

static void aln2sam(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                    const int32_t *ctg_off, const uint8_t *ref_string, sbuf_t *str, int l_seq, const uint8_t *seq, const char *qual,
                    const char *name, const char *comment, int n, const aln_t *list, int which, const bwams_alnreg_t *regs, int n_regs,
                    const aln_t *m_)
{
    aln_t ptmp = list[which], *p = &ptmp, mtmp, *m = 0;
    int i;
    if (m_) { mtmp = *m_; m = &mtmp; }
    p->a.flag |= m ? 0x1 : 0;
    p->a.flag |= p->a.rid < 0 ? 0x4 : 0;
    p->a.flag |= m && m->a.rid < 0 ? 0x8 : 0;
    if (p->a.rid < 0 && m && m->a.rid >= 0) { p->a.rid = m->a.rid; p->a.pos = m->a.pos; p->a.is_rev = m->a.is_rev; p->a.n_cigar = 0; }
    if (m && m->a.rid < 0 && p->a.rid >= 0) { m->a.rid = p->a.rid; m->a.pos = p->a.pos; m->a.is_rev = p->a.is_rev; m->a.n_cigar = 0; }
    p->a.flag |= p->a.is_rev ? 0x10 : 0;
    p->a.flag |= m && m->a.is_rev ? 0x20 : 0;
    sputs(str, name); sputc(str, '\t');
    sputl(str, (p->a.flag & 0xffff) | (p->a.flag & 0x10000 ? 0x100 : 0)); sputc(str, '\t');
    if (p->a.rid >= 0) {
        sputs(str, ctg_names + ctg_off[p->a.rid]); sputc(str, '\t');
        sputl(str, (long)(p->a.pos + 1)); sputc(str, '\t');
        sputl(str, p->a.mapq); sputc(str, '\t');
        add_cigar(so, p, str, which);
    } else sputsn(str, "*\t0\t0\t*", 7);
    sputc(str, '\t');
    if (m && m->a.rid >= 0) {
        if (p->a.rid == m->a.rid) sputc(str, '=');
        else sputs(str, ctg_names + ctg_off[m->a.rid]);
        sputc(str, '\t');
        sputl(str, (long)(m->a.pos + 1)); sputc(str, '\t');
        if (p->a.rid == m->a.rid) {
            int64_t p0 = p->a.pos + (p->a.is_rev ? get_rlen(p->a.n_cigar, p->cigar) - 1 : 0);
            int64_t p1 = m->a.pos + (m->a.is_rev ? get_rlen(m->a.n_cigar, m->cigar) - 1 : 0);
            if (m->a.n_cigar == 0 || p->a.n_cigar == 0) sputc(str, '0');
            else sputl(str, (long)(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0))));
        } else sputc(str, '0');
    } else sputsn(str, "*\t0\t0", 5);
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
    if (m && m->a.n_cigar) { sputsn(str, "\tMC:Z:", 6); add_cigar(so, m, str, which); }
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
    put_xr(so, str, p->a.rid);
    sputc(str, '\n');
}

/* mem_reg2sam(opt, bns, pac, s, a, extra_flag, m) for one read, appended to str */
static void reg2sam(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                    const int32_t *ctg_off, const uint8_t *ref_string, int l_seq, const uint8_t *seq, const char *qual,
                    const char *name, const char *comment, const bwams_alnreg_t *regs, int n_regs, int extra_flag, const aln_t *m,
                    sbuf_t *str)
{
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
        q->a.flag |= extra_flag;
        if (p->secondary >= 0) q->a.sub = -1;
        if (l && p->secondary < 0) q->a.flag |= (so->flag & BWAMS_MEM_F_NO_MULTI) ? 0x10000 : 0x800;
        if (!(so->flag & BWAMS_MEM_F_KEEP_SUPP_MAPQ) && l && !reg_is_alt(p) && q->a.mapq > aa[0].a.mapq) q->a.mapq = aa[0].a.mapq;
        ++l;
    }
    if (n_aa == 0) {
        aln_t t;
        run_reg2aln(opt, bns, ref_string, l_seq, seq, 0, &t);
        t.a.flag |= extra_flag;
        aln2sam(opt, so, bns, ctg_names, ctg_off, ref_string, str, l_seq, seq, qual, name, comment, 1, &t, 0, regs, n_regs, m);
        free(t.cigar); free(t.md);
    } else {
        for (int k = 0; k < n_aa; ++k)
            aln2sam(opt, so, bns, ctg_names, ctg_off, ref_string, str, l_seq, seq, qual, name, comment, n_aa, aa, k, regs, n_regs, m);
        for (int k = 0; k < n_aa; ++k) { free(aa[k].cigar); free(aa[k].md); }
    }
    free(aa);
}

/* mem_reg2sam(opt, bns, pac, s, a, 0, NULL) for one read.  Returns the length of the text (the bytes are written while they
 * fit into cap; -1 - length when they did not). */
int64_t orc_reg2sam_se(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                       const int32_t *ctg_off, const uint8_t *ref_string, int l_seq, const uint8_t *seq, const char *qual,
                       const char *name, const char *comment, const bwams_alnreg_t *regs, int n_regs, char *out, int64_t cap)
{
    sbuf_t str = {out, 0, cap, 0};
    reg2sam(opt, so, bns, ctg_names, ctg_off, ref_string, l_seq, seq, qual, name, comment, regs, n_regs, 0, 0, &str);
    return str.over ? -1 - str.l : str.l;
}

/* mem_perfect2sam_cont + mem_aln2sam_perfect (bwamem.cpp:2280-2325, :2153-2227) for one read the EMF resolved.  regs / n = what
 * mem_perfect2reg left for it (orc_perfect2reg: one region per location of get_perfect_locations after perfect_dedup_patch, in
 * that order — the same list mem_perfect2sam_cont walks); seed_len = the table's L (init_mem_aln_perfect shifts the position
 * of a reverse-strand hit of a longer read by l_seq - seed_len).  Returns the text length (-1 - length when cap was short). */
int64_t orc_perfect2sam(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                        const int32_t *ctg_off, int seed_len, int l_seq, const uint8_t *seq, const char *qual, const char *name,
                        const char *comment, const bwams_alnreg_t *regs, int n, char *out, int64_t cap)
{
    sbuf_t str = {out, 0, cap, 0};
    int n_out = 0;
    for (int pass = 0; pass < 2; ++pass) {                  /* primary-assembly hits, then (if none, or with MEM_F_ALL) ALT hits */
        if (pass == 1 && !(n_out == 0 || (so->flag & BWAMS_MEM_F_ALL))) break;
        for (int k = 0; k < n; ++k) {
            const bwams_alnreg_t *r = &regs[k];
            const int is_alt = reg_is_alt(r), is_rev = r->rb >= bns->l_pac, secondary = n_out > 0;
            int64_t loc, pos;
            int flag, i;
            if (is_alt != pass) continue;
            loc = is_rev ? (bns->l_pac << 1) - r->re : r->rb;
            pos = loc;
            if (l_seq != seed_len && is_rev) pos = pos - (l_seq - seed_len);
            pos -= bns->contigs[r->rid].offset;
            flag = (is_rev ? 0x10 : 0) | (secondary ? 0x100 : 0);
            sputs(&str, name); sputc(&str, '\t');
            sputl(&str, flag & 0xffff); sputc(&str, '\t');
            sputs(&str, ctg_names + ctg_off[r->rid]); sputc(&str, '\t');
            sputl(&str, (long)(pos + 1)); sputc(&str, '\t');
            sputl(&str, 60); sputc(&str, '\t');
            sputl(&str, l_seq); sputc(&str, 'M');
            sputc(&str, '\t');
            sputsn(&str, "*\t0\t0", 5);
            sputc(&str, '\t');
            if (flag & 0x100) sputsn(&str, "*\t*", 3);
            else if (!is_rev) {
                for (i = 0; i < l_seq; ++i) sputc(&str, "ACGTN"[seq[i]]);
                sputc(&str, '\t');
                if (qual) { for (i = 0; i < l_seq; ++i) sputc(&str, qual[i]); } else sputc(&str, '*');
            } else {
                for (i = l_seq - 1; i >= 0; --i) sputc(&str, "TGCAN"[seq[i]]);
                sputc(&str, '\t');
                if (qual) { for (i = l_seq - 1; i >= 0; --i) sputc(&str, qual[i]); } else sputc(&str, '*');
            }
            sputsn(&str, "\tNM:i:", 6); sputl(&str, 0);
            sputsn(&str, "\tMD:Z:", 6); sputl(&str, l_seq);
            sputsn(&str, "\tAS:i:", 6); sputl(&str, l_seq * opt->a);
            if (!secondary) { sputsn(&str, "\tXS:i:", 6); sputl(&str, (k == 0 && n > 1) ? l_seq * opt->a : 0); }
            if (so->rg_id[0]) { sputsn(&str, "\tRG:Z:", 6); sputs(&str, so->rg_id); }
            if (comment) { sputc(&str, '\t'); sputs(&str, comment); }
            put_xr(so, &str, r->rid);
            sputc(&str, '\n');
            ++n_out;
            if (!(so->flag & BWAMS_MEM_F_ALL)) break;
        }
    }
    return str.over ? -1 - str.l : str.l;
}

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))            /* bwamem_pair.cpp:432 */

static int infer_dir_(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)          /* mem_infer_dir, bwamem_pair.cpp:57-65 */
{
    int64_t p2;
    int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

/* mem_sam_pe from the call of mem_pair on (bwamem_pair.cpp:686-833): regs[i] / n_regs[i] are end i's regions after mate rescue
 * and mem_mark_primary_se (they are edited in place as the reference edits them), pr = what mem_pair returned.  The texts of the
 * two ends go to out[0] / out[1]; len[i] = their lengths (-1 - length when cap[i] was short). */
void orc_sam_pe(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const orc_bns_t *bns, const char *ctg_names,
                const int32_t *ctg_off, const uint8_t *ref_string, const bwams_pestat_t pes[4], const int32_t l_seq[2],
                const uint8_t *const seq[2], const char *const qual[2], const char *const name[2], const char *const comment[2],
                bwams_alnreg_t *const regs[2], const int32_t n_regs[2], const bwams_pair_t *pr, char *const out[2], const int64_t cap[2],
                int64_t len[2])
{
    int i, j, z[2], o = pr->score, subo = pr->sub, n_sub = pr->n_sub, extra_flag = 1, n_pri[2], n_aa[2] = {0, 0};
    aln_t h[2], g[2], aa[2][2];
    sbuf_t str[2] = {{out[0], 0, cap[0], 0}, {out[1], 0, cap[1], 0}};
    memset(h, 0, sizeof h); memset(g, 0, sizeof g);
    z[0] = pr->z[0]; z[1] = pr->z[1]; n_pri[0] = pr->n_pri[0]; n_pri[1] = pr->n_pri[1];
    if (so->flag & BWAMS_MEM_F_NOPAIRING) goto no_pairing;                       /* bwamem_pair.cpp:1066 */
    if (n_pri[0] && n_pri[1] && o > 0) {
        int is_multi[2], q_pe, score_un, q_se[2];
        for (i = 0; i < 2; ++i) {
            for (j = 1; j < n_pri[i]; ++j)
                if (regs[i][j].secondary < 0 && regs[i][j].score >= so->T) break;
            is_multi[i] = j < n_pri[i] ? 1 : 0;
        }
        if (is_multi[0] || is_multi[1]) goto no_pairing;
        score_un = regs[0][0].score + regs[1][0].score - opt->pen_unpaired;
        subo = subo > score_un ? subo : score_un;
        q_pe = RAW_MAPQ(o - subo, opt->a);
        if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
        if (q_pe < 0) q_pe = 0;
        if (q_pe > 60) q_pe = 60;
        q_pe = (int)(q_pe * (1. - .5 * (regs[0][0].frac_rep + regs[1][0].frac_rep)) + .499);
        if (o > score_un) {
            bwams_alnreg_t *c[2];
            c[0] = &regs[0][z[0]]; c[1] = &regs[1][z[1]];
            for (i = 0; i < 2; ++i) {
                if (c[i]->secondary >= 0) { c[i]->sub = regs[i][c[i]->secondary].score; c[i]->secondary = -2; }
                q_se[i] = orc_approx_mapq_se(opt, c[i]);
            }
            q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
            q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
            extra_flag |= 2;
            q_se[0] = q_se[0] < RAW_MAPQ(c[0]->score - c[0]->csub, opt->a) ? q_se[0] : RAW_MAPQ(c[0]->score - c[0]->csub, opt->a);
            q_se[1] = q_se[1] < RAW_MAPQ(c[1]->score - c[1]->csub, opt->a) ? q_se[1] : RAW_MAPQ(c[1]->score - c[1]->csub, opt->a);
        } else {
            z[0] = z[1] = 0;
            q_se[0] = orc_approx_mapq_se(opt, &regs[0][0]);
            q_se[1] = orc_approx_mapq_se(opt, &regs[1][0]);
        }
        for (i = 0; i < 2; ++i) {
            int k = regs[i][z[i]].secondary_all;
            if (k >= 0 && k < n_pri[i]) {
                for (j = 0; j < n_regs[i]; ++j)
                    if (regs[i][j].secondary_all == k || j == k) regs[i][j].secondary_all = z[i];
                regs[i][z[i]].secondary_all = -1;
            }
        }
        for (i = 0; i < 2; ++i) {
            run_reg2aln(opt, bns, ref_string, l_seq[i], seq[i], &regs[i][z[i]], &h[i]);
            h[i].a.mapq = q_se[i];
            h[i].a.flag |= 0x40 << i | extra_flag;
            h[i].xa_of = !(so->flag & BWAMS_MEM_F_ALL) ? z[i] : -1;
            aa[i][n_aa[i]++] = h[i];
            if (n_pri[i] < n_regs[i]) {
                bwams_alnreg_t *p = &regs[i][n_pri[i]];
                if (p->score < so->T || p->secondary >= 0 || !reg_is_alt(p)) continue;
                run_reg2aln(opt, bns, ref_string, l_seq[i], seq[i], p, &g[i]);
                g[i].a.flag |= 0x800 | 0x40 << i | extra_flag;
                g[i].xa_of = !(so->flag & BWAMS_MEM_F_ALL) ? n_pri[i] : -1;
                aa[i][n_aa[i]++] = g[i];
            }
        }
        for (i = 0; i < 2; ++i)
            for (j = 0; j < n_aa[i]; ++j)
                aln2sam(opt, so, bns, ctg_names, ctg_off, ref_string, &str[i], l_seq[i], seq[i], qual[i], name[i], comment[i], n_aa[i],
                        aa[i], j, regs[i], n_regs[i], &h[!i]);
        for (i = 0; i < 2; ++i) { free(h[i].cigar); free(h[i].md); free(g[i].cigar); free(g[i].md); }
        goto done;
    }
no_pairing:
    for (i = 0; i < 2; ++i) {
        int which = -1;
        if (n_regs[i]) {
            if (regs[i][0].score >= so->T) which = 0;
            else if (n_pri[i] < n_regs[i] && regs[i][n_pri[i]].score >= so->T) which = n_pri[i];
        }
        run_reg2aln(opt, bns, ref_string, l_seq[i], seq[i], which >= 0 ? &regs[i][which] : 0, &h[i]);
    }
    if (!(so->flag & BWAMS_MEM_F_NOPAIRING) && h[0].a.rid == h[1].a.rid && h[0].a.rid >= 0) {
        int64_t dist;
        int d = infer_dir_(bns->l_pac, regs[0][0].rb, regs[1][0].rb, &dist);
        if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
    }
    reg2sam(opt, so, bns, ctg_names, ctg_off, ref_string, l_seq[0], seq[0], qual[0], name[0], comment[0], regs[0], n_regs[0],
            0x41 | extra_flag, &h[1], &str[0]);
    reg2sam(opt, so, bns, ctg_names, ctg_off, ref_string, l_seq[1], seq[1], qual[1], name[1], comment[1], regs[1], n_regs[1],
            0x81 | extra_flag, &h[0], &str[1]);
    free(h[0].cigar); free(h[0].md); free(h[1].cigar); free(h[1].md);
done:
    for (i = 0; i < 2; ++i) len[i] = str[i].over ? -1 - str[i].l : str[i].l;
}
