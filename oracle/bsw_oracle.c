/*
 * bsw_oracle.c — CPU restatement of the banded Smith-Waterman seed extension
 * (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * Follows BandedPairWiseSW::scalarBandedSWA, /root/reference/src/bandedSWA.cpp:116-237
 * (the scalar routine is the specification; the reference's SSE/AVX2/AVX512
 * inter-task kernels produce the same six outputs), and
 * scalarBandedSWAWrapper, /root/reference/src/bandedSWA.cpp:242-260.
 * PINNED: tests/test_oracle_bsw.py compares this file with the reference
 * object built by oracle/Makefile into oracle/_ref/.
 *
 * Recurrence, row i (target) by column j (query), all values >= 0:
 *   M(i,j)   = H(i-1,j-1) ? H(i-1,j-1) + S(t_i, q_j) : 0
 *   H(i,j)   = max(M, E(i,j), F(i,j))
 *   E(i+1,j) = max(E(i,j) - e_del, M - o_del - e_del, 0)
 *   F(i,j+1) = max(F(i,j) - e_ins, M - o_ins - e_ins, 0)
 * The row is evaluated on [beg, end), which is clipped to the band |i-j| <= w
 * and shrinks to the non-zero part of the previous row.
 */
#include <stdlib.h>
#include "bwams_oracle.h"

typedef struct { int32_t h, e; } cell_t;

int orc_bsw_scalar(const bwams_sw_opt_t *o, int qlen, const uint8_t *query,
                   int tlen, const uint8_t *target, int32_t w, int h0,
                   int *qle, int *tle, int *gtle, int *gscore_out, int *max_off_out,
                   int64_t *cells)
{
    const int m = 5;
    const int o_del = o->o_del, e_del = o->e_del, o_ins = o->o_ins, e_ins = o->e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, k;

    int8_t *qp = (int8_t *)malloc((size_t)(qlen > 0 ? qlen : 1) * m);
    cell_t *row = (cell_t *)calloc((size_t)qlen + 1, sizeof(cell_t));

    /* query profile: qp[c][j] = S(c, q_j) */
    for (k = i = 0; k < m; ++k) {
        const int8_t *p = &o->mat[k * m];
        for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
    }

    /* row -1: H decays from h0 by one gap open then extensions */
    row[0].h = h0;
    if (qlen >= 1) row[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && row[j - 1].h > e_ins; ++j)
        row[j].h = row[j - 1].h - e_ins;

    /* clamp the band to the longest gap the score can pay for */
    int max_sc = 0;
    for (i = 0; i < m * m; ++i) max_sc = max_sc > o->mat[i] ? max_sc : o->mat[i];
    int max_ins = (int)((double)(qlen * max_sc + o->end_bonus - o_ins) / e_ins + 1.);
    if (max_ins < 1) max_ins = 1;
    if (w > max_ins) w = max_ins;
    int max_del = (int)((double)(qlen * max_sc + o->end_bonus - o_del) / e_del + 1.);
    if (max_del < 1) max_del = 1;
    if (w > max_del) w = max_del;

    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int beg = 0, end = qlen;
    for (i = 0; i < tlen; ++i) {
        int f = 0, h1, rmax = 0, mj = -1;
        const int8_t *q = &qp[target[i] * qlen];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) {
            h1 = h0 - (o_del + e_del * (i + 1));
            if (h1 < 0) h1 = 0;
        } else h1 = 0;
        for (j = beg; j < end; ++j) {
            cell_t *p = &row[j];
            int M = p->h, e = p->e, h, t;
            p->h = h1;
            M = M ? M + q[j] : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = rmax > h ? mj : j;
            rmax = rmax > h ? rmax : h;
            t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del;     e = e > t ? e : t;
            p->e = e;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins;     f = f > t ? f : t;
            if (cells) (*cells)++;
        }
        row[end].h = h1; row[end].e = 0;
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (rmax == 0) break;
        if (rmax > max) {
            max = rmax; max_i = i; max_j = mj;
            int d = mj - i; if (d < 0) d = -d;
            max_off = max_off > d ? max_off : d;
        } else if (o->zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - rmax - ((i - max_i) - (mj - max_j)) * e_del > o->zdrop) break;
            } else {
                if (max - rmax - ((mj - max_j) - (i - max_i)) * e_ins > o->zdrop) break;
            }
        }
        for (j = beg; j < end && row[j].h == 0 && row[j].e == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && row[j].h == 0 && row[j].e == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(row); free(qp);
    if (qle) *qle = max_j + 1;
    if (tle) *tle = max_i + 1;
    if (gtle) *gtle = max_ie + 1;
    if (gscore_out) *gscore_out = gscore;
    if (max_off_out) *max_off_out = max_off;
    return max;
}

void orc_bsw_pairs(const bwams_sw_opt_t *o, bwams_seqpair_t *pairs,
                   const uint8_t *ref, const uint8_t *qer, int64_t n,
                   int32_t w, int64_t *cells)
{
    for (int64_t i = 0; i < n; ++i) {
        bwams_seqpair_t *p = &pairs[i];
        p->score = orc_bsw_scalar(o, p->len2, qer + p->idq, p->len1, ref + p->idr,
                                  w, p->h0, &p->qle, &p->tle, &p->gtle,
                                  &p->gscore, &p->max_off, cells);
    }
}
