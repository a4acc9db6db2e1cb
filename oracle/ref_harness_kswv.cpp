/*
 * ref_harness_kswv.cpp — extern "C" entry point around the REAL reference batched mate-rescue kernel
 * (TEST INFRASTRUCTURE ONLY).
 *
 * oracle/Makefile compiles /root/reference/src/kswv.cpp from where it lies (-mavx512bw; it needs nothing outside the
 * reference tree and libc) and links it with this file into oracle/_ref/libref_kswv.so.  No reference source is copied:
 * this file only drives kswv::getScores8 / getScores16 through their own header, in the order mem_sam_pe_batch does
 * (/root/reference/src/bwamem_pair.cpp:903-971): phase 0 on the byte-class and the 16-bit-class tasks, reversal of the
 * prefixes that end at (qe, te) for the tasks that ask for the start, phase 1.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "kswv.h"        /* /root/reference/src, via -I: pulls ksw.h (kswr_t, KSW_X*) and bandedSWA.h (SeqPair) */

extern "C" {

int ref_kswv_available(void) {
#if __AVX512BW__
    return 1;
#else
    return 0;
#endif
}

#if __AVX512BW__
static void rev(int n, uint8_t *s) { for (int i = 0; i < n >> 1; ++i) { uint8_t t = s[i]; s[i] = s[n - 1 - i]; s[n - 1 - i] = t; } }

/* pairs[0, n8) are the byte-class tasks (xtra & KSW_XBYTE), pairs[n8, n8 + n16) the others; pairs[i].regid indexes aln,
 * pairs[i].h0 carries xtra.  ref / qer are modified in place (prefix reversal), as in the reference. */
void ref_kswv_batch(int o_del, int e_del, int o_ins, int e_ins, int a, int b, const SeqPair *pairs_in, int n8, int n16,
                    uint8_t *ref, uint8_t *qer, int max_ref, int max_qer, kswr_t *aln)
{
    const int n = n8 + n16;
    std::vector<SeqPair> arr((size_t)n + 2 * MAX_LINE_LEN + 256);
    memcpy(arr.data(), pairs_in, (size_t)n8 * sizeof(SeqPair));
    SeqPair *p16 = arr.data() + n8 + MAX_LINE_LEN;
    memcpy(p16, pairs_in + n8, (size_t)n16 * sizeof(SeqPair));
    for (int i = 0; i < n; ++i) aln[i].tb = aln[i].qb = -1;
    kswv *k = new kswv(o_del, e_del, o_ins, e_ins, (int8_t)a, (int8_t)(-b), 1, max_ref, max_qer);
    k->getScores8(arr.data(), ref, qer, aln, n8, 1, 0);
    k->getScores16(p16, ref, qer, aln, n16, 1, 0);
    int pos = 0, pos8 = 0, pos16 = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const SeqPair *src = pass ? p16 : arr.data();
        const int cnt = pass ? n16 : n8;
        for (int i = 0; i < cnt; ++i) {
            SeqPair sp = src[i];
            const kswr_t r = aln[sp.regid];
            const int xtra = sp.h0;
            if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) continue;
            sp.h0 = KSW_XSTOP | r.score;
            sp.len2 = r.qe + 1;
            rev(r.qe + 1, qer + sp.idq);
            rev(r.te + 1, ref + sp.idr);
            arr[pos++] = sp;
            if (pass) ++pos16; else ++pos8;
        }
    }
    k->getScores16(arr.data() + pos8, ref, qer, aln, pos16, 1, 1);
    k->getScores8(arr.data(), ref, qer, aln, pos8, 1, 1);
    delete k;
}
#else
void ref_kswv_batch(int, int, int, int, int, int, const void *, int, int, uint8_t *, uint8_t *, int, int, void *) {}
#endif

} /* extern "C" */
