/*
 * ksw_oracle.c — CPU restatement of the local Smith-Waterman used by mate rescue
 * (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * Follows ksw_align2 / ksw_u8 / ksw_i16 / ksw_qinit,
 * /root/reference/src/ksw.cpp:62-108 (profile, striping, shift), :111-232 (u8), :234-338 (i16),
 * :347-381 (two-pass driver).  PINNED: tests/test_oracle_ksw.py compares it with the
 * reference object in oracle/_ref (ksw.cpp compiled from where it lies).
 *
 * The reference is Farrar's striped SSE2 kernel.  What of its layout is observable, and
 * is therefore restated here on a plain row-by-row DP:
 *   - the query is padded to P = slen * p columns (p = 16 for the byte kernel, 8 for the
 *     16-bit one) and the pad columns score 0 against every base, so they carry the
 *     previous row's last scores diagonally; row maxima are taken over all P columns;
 *   - H is exact after the lazy-F loop (F(i,j) = max over j' < j of Hnf(i,j') - oe_ins -
 *     (j-1-j') e_ins with Hnf = max(diagonal move, E), all floored at 0);
 *   - E(i+1,j) is formed from H before the lazy-F correction; this is unobservable while an
 *     insertion directly followed by a deletion cannot beat a mismatch
 *     (oe_ins + oe_del > max - min of the matrix), which callers must guarantee;
 *   - qe = the smallest column that attains the maximum in the row saved when the global
 *     maximum last improved; score2/te2 from the run-merged list of row maxima >= minsc;
 *   - the byte kernel stops and reports 255 once max + shift reaches 255.
 */
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

#define KSW_XBYTE  0x10000
#define KSW_XSTOP  0x20000
#define KSW_XSUBO  0x40000
#define KSW_XSTART 0x80000

typedef struct { int score, te, qe, score2, te2, tb, qb; } kswr_o;

static kswr_o ksw_core(int size, int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                       const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
    const int m = 5;
    const int p = size == 1 ? 16 : 8;
    const int slen = (qlen + p - 1) / p;
    const int P = slen * p;
    int shift = 127, mx = 0;
    for (int a = 0; a < m * m; ++a) {
        if (mat[a] < shift) shift = mat[a];
        if (mat[a] > mx) mx = mat[a];
    }
    shift = -shift;                        /* (uint8)(256 - min) */
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int minsc = (xtra & KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;
    const int endsc = (xtra & KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
    kswr_o r = { 0, -1, -1, -1, -1, -1, -1 };

    int *H0 = (int *)calloc((size_t)P + 1, sizeof(int));
    int *H1 = (int *)calloc((size_t)P + 1, sizeof(int));
    int *E = (int *)calloc((size_t)P + 1, sizeof(int));
    int *Hmax = (int *)calloc((size_t)P + 1, sizeof(int));
    int64_t *b = NULL;
    int n_b = 0, m_b = 0;
    int gmax = 0, te = -1;

    for (int i = 0; i < tlen; ++i) {
        const int8_t *srow = mat + target[i] * m;
        int imax = 0, best_src = 0;          /* best_src: max over j' < j of (Hnf + j' e_ins) - scaled F source */
        int have_src = 0;
        for (int j = 0; j < P; ++j) {
            const int s = j < qlen ? srow[query[j]] : 0;
            int hd = (j ? H0[j - 1] : 0) + s;
            if (hd < 0) hd = 0;
            int hnf = hd > E[j] ? hd : E[j];
            int f = 0;
            if (have_src) {
                f = best_src - oe_ins - (j - 1) * e_ins;
                if (f < 0) f = 0;
            }
            const int h = hnf > f ? hnf : f;
            H1[j] = h;
            if (h > imax) imax = h;
            /* E for the next row, from the exact H (see header) */
            int e = E[j] - e_del; if (e < 0) e = 0;
            int t = h - oe_del;   if (t < 0) t = 0;
            E[j] = e > t ? e : t;
            /* this column as a gap-open source for the columns to its right */
            const int src = hnf + j * e_ins;
            if (!have_src || src > best_src) { best_src = src; have_src = 1; }
        }
        if (imax >= minsc) {
            if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
                if (n_b == m_b) { m_b = m_b ? m_b << 1 : 8; b = (int64_t *)realloc(b, 8 * (size_t)m_b); }
                b[n_b++] = (int64_t)imax << 32 | i;
            } else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (int64_t)imax << 32 | i;
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            memcpy(Hmax, H1, sizeof(int) * (size_t)P);
            if (size == 1 && gmax + shift >= 255) break;
            if (gmax >= endsc) break;
        }
        int *tmp = H1; H1 = H0; H0 = tmp;
    }
    if (size == 1) r.score = gmax + shift < 255 ? gmax : 255;
    else r.score = gmax;
    r.te = te;
    if (!(size == 1 && r.score == 255)) {
        int max = -1;
        r.qe = -1;
        for (int j = 0; j < P; ++j)
            if (Hmax[j] > max) { max = Hmax[j]; r.qe = j; }       /* smallest column attaining the max */
        if (b) {
            int i = (r.score + mx - 1) / mx;
            int low = te - i, high = te + i;
            for (i = 0; i < n_b; ++i) {
                int e = (int32_t)b[i];
                if ((e < low || e > high) && (int)(b[i] >> 32) > r.score2) {
                    r.score2 = (int)(b[i] >> 32);
                    r.te2 = e;
                }
            }
        }
    }
    free(b); free(H0); free(H1); free(E); free(Hmax);
    return r;
}

/* out[7] = score, te, qe, score2, te2, tb, qb */
void orc_ksw_align2(const bwams_sw_opt_t *o, int qlen, const uint8_t *query, int tlen,
                    const uint8_t *target, int xtra, int *out)
{
    const int size = (xtra & KSW_XBYTE) ? 1 : 2;
    kswr_o r = ksw_core(size, qlen, query, tlen, target, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, xtra);
    if (!((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff)))) {
        /* second pass: both sequences reversed up to the end point; the target keeps its tail */
        uint8_t *rq = (uint8_t *)malloc((size_t)r.qe + 2);
        uint8_t *rt = (uint8_t *)malloc((size_t)tlen + 1);
        for (int j = 0; j <= r.qe; ++j) rq[j] = query[r.qe - j];
        memcpy(rt, target, (size_t)tlen);
        for (int i = 0; i <= r.te; ++i) rt[i] = target[r.te - i];
        kswr_o rr = ksw_core(size, r.qe + 1, rq, tlen, rt, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins,
                             KSW_XSTOP | r.score);
        if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
        free(rq); free(rt);
    }
    out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}
