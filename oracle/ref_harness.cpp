/*
 * ref_harness.cpp — extern "C" entry points around the REAL reference objects
 * (TEST INFRASTRUCTURE ONLY).
 *
 * oracle/Makefile compiles /root/reference/src/{bandedSWA,ksw}.cpp from where
 * they lie (they need nothing outside the reference tree and libc) and links
 * them with this file into oracle/_ref/libref_sw_<isa>.so.  No reference source
 * is copied into this repository; this file only calls the reference's public
 * class/functions through their own headers.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bandedSWA.h"   /* /root/reference/src, via -I */
#include "ksw.h"

#include <stddef.h>
#include "../include/bwams_types.h"
/* the records that cross the C-ABI are the reference's own, byte for byte (bandedSWA.h:90-99, ksw.h:43-48) */
#define SAME_FIELD(T, U, f) static_assert(offsetof(T, f) == offsetof(U, f) && sizeof(((T *)0)->f) == sizeof(((U *)0)->f), #f)
static_assert(sizeof(SeqPair) == sizeof(bwams_seqpair_t) && sizeof(SeqPair) == 56, "SeqPair");
SAME_FIELD(SeqPair, bwams_seqpair_t, idr); SAME_FIELD(SeqPair, bwams_seqpair_t, idq); SAME_FIELD(SeqPair, bwams_seqpair_t, id);
SAME_FIELD(SeqPair, bwams_seqpair_t, len1); SAME_FIELD(SeqPair, bwams_seqpair_t, len2); SAME_FIELD(SeqPair, bwams_seqpair_t, h0);
SAME_FIELD(SeqPair, bwams_seqpair_t, seqid); SAME_FIELD(SeqPair, bwams_seqpair_t, regid); SAME_FIELD(SeqPair, bwams_seqpair_t, score);
SAME_FIELD(SeqPair, bwams_seqpair_t, tle); SAME_FIELD(SeqPair, bwams_seqpair_t, gtle); SAME_FIELD(SeqPair, bwams_seqpair_t, qle);
SAME_FIELD(SeqPair, bwams_seqpair_t, gscore); SAME_FIELD(SeqPair, bwams_seqpair_t, max_off);
static_assert(sizeof(kswr_t) == sizeof(bwams_kswr_t) && sizeof(kswr_t) == 28, "kswr_t");
SAME_FIELD(kswr_t, bwams_kswr_t, score); SAME_FIELD(kswr_t, bwams_kswr_t, te); SAME_FIELD(kswr_t, bwams_kswr_t, qe);
SAME_FIELD(kswr_t, bwams_kswr_t, score2); SAME_FIELD(kswr_t, bwams_kswr_t, te2); SAME_FIELD(kswr_t, bwams_kswr_t, tb); SAME_FIELD(kswr_t, bwams_kswr_t, qb);

extern "C" {

struct ref_sw_opt {
    int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus;
    int8_t mat[25];
    int8_t pad_[3];
};

int ref_sizeof_seqpair(void) { return (int)sizeof(SeqPair); }
int ref_simd_width16(void) { return SIMD_WIDTH16; }
int ref_simd_width8(void) { return SIMD_WIDTH8; }

/* BandedPairWiseSW::scalarBandedSWAWrapper (bandedSWA.cpp:242) */
void ref_bsw_scalar(const ref_sw_opt *o, SeqPair *pairs, uint8_t *ref, uint8_t *qer,
                    int n, int w)
{
    BandedPairWiseSW bsw(o->o_del, o->e_del, o->o_ins, o->e_ins, o->zdrop, o->end_bonus,
                         o->mat, o->mat[0], (int8_t)(-o->mat[1]), 1);
    bsw.scalarBandedSWAWrapper(pairs, ref, qer, n, 1, w);
}

/* BandedPairWiseSW::getScores16 / getScores8: the inter-task SIMD kernels of the
 * ISA this object was compiled for.  The pair array must have room for n rounded
 * up to the SIMD width; the sequence buffers must be readable past the last pair. */
void ref_bsw_vec16(const ref_sw_opt *o, SeqPair *pairs, uint8_t *ref, uint8_t *qer,
                   int n, int w)
{
    BandedPairWiseSW bsw(o->o_del, o->e_del, o->o_ins, o->e_ins, o->zdrop, o->end_bonus,
                         o->mat, o->mat[0], (int8_t)(-o->mat[1]), 1);
    bsw.getScores16(pairs, ref, qer, n, 1, w);
}

void ref_bsw_vec8(const ref_sw_opt *o, SeqPair *pairs, uint8_t *ref, uint8_t *qer,
                  int n, int w)
{
    BandedPairWiseSW bsw(o->o_del, o->e_del, o->o_ins, o->e_ins, o->zdrop, o->end_bonus,
                         o->mat, o->mat[0], (int8_t)(-o->mat[1]), 1);
    bsw.getScores8(pairs, ref, qer, n, 1, w);
}

/* ksw_extend2 (ksw.cpp:383): the routine scalarBandedSWA was derived from. */
int ref_ksw_extend2(const ref_sw_opt *o, int qlen, const uint8_t *query, int tlen,
                    const uint8_t *target, int w, int h0, int *qle, int *tle,
                    int *gtle, int *gscore, int *max_off)
{
    return ksw_extend2(qlen, query, tlen, target, 5, o->mat, o->o_del, o->e_del,
                       o->o_ins, o->e_ins, w, o->end_bonus, o->zdrop, h0,
                       qle, tle, gtle, gscore, max_off);
}

/* ksw_align2 (ksw.cpp:347): local SW used by mate rescue.  out[7] =
 * score, te, qe, score2, te2, tb, qb. */
void ref_ksw_align2(const ref_sw_opt *o, int qlen, uint8_t *query, int tlen,
                    uint8_t *target, int xtra, int *out)
{
    kswr_t r = ksw_align2(qlen, query, tlen, target, 5, o->mat, o->o_del, o->e_del,
                          o->o_ins, o->e_ins, xtra, 0);
    out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2;
    out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}

/* ksw_global2 (ksw.cpp:558), score only (n_cigar = cigar = NULL), as mem_patch_reg -> bwa_gen_cigar2 calls it. */
int ref_ksw_global2(const ref_sw_opt *o, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w)
{
    return ksw_global2(qlen, query, tlen, target, 5, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w, 0, 0);
}

/* ksw_global2 with traceback: the CIGAR is copied out (cap entries) and the reference's buffer freed. */
int ref_ksw_global2_cigar(const ref_sw_opt *o, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w,
                          int *n_cigar, uint32_t *cigar_out, int cap)
{
    uint32_t *cigar = 0;
    int n = 0;
    const int sc = ksw_global2(qlen, query, tlen, target, 5, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w, &n, &cigar);
    for (int i = 0; i < n && i < cap; ++i) cigar_out[i] = cigar[i];
    *n_cigar = n;
    free(cigar);
    return sc;
}

} /* extern "C" */
