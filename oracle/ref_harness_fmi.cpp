/*
 * ref_harness_fmi.cpp — the pieces of the reference's FM-index code that compile without the un-vendored safestringlib
 * (TEST INFRASTRUCTURE ONLY): FMI_search.h (record layouts, the GET_OCC macro) with the __rdtsc rename ref_harness_emf.cpp
 * already uses, and sais.h (the suffix sorter build_index calls, /root/reference/src/FMI_search.cpp:833-840).
 * FMI_search.cpp itself includes safestringlib and is not buildable here: the search routines stay pinned by first
 * principles (tests/test_oracle_fmi.py).  Compiled where the headers lie: oracle/Makefile -> oracle/_ref/libref_fmi.so.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <x86intrin.h>
#include "sais.h"                /* before macro.h: its one-letter macros (src/macro.h:88-197) would rewrite sais.h's template parameters */
#define __rdtsc __ref_rdtsc      /* utils.h declares its own __rdtsc; GCC >= 11 already has one */
#include "FMI_search.h"
#include "../include/bwams_types.h"

/* the records that cross the C-ABI, against the reference's own declarations (src/FMI_search.h:64-135) */
static_assert(sizeof(CP_OCC) == 64 && sizeof(bwams_cp_occ_t) == 64 && offsetof(CP_OCC, cp_count) == offsetof(bwams_cp_occ_t, cp_count) &&
              offsetof(CP_OCC, one_hot_bwt_str) == 32 && offsetof(bwams_cp_occ_t, one_hot_bwt_str) == 32, "CP_OCC");
static_assert(sizeof(SMEM) == 40 && sizeof(bwams_smem_t) == 40 && offsetof(SMEM, rid) == offsetof(bwams_smem_t, rid) &&
              offsetof(SMEM, m) == offsetof(bwams_smem_t, m) && offsetof(SMEM, n) == offsetof(bwams_smem_t, n) &&
              offsetof(SMEM, k) == 16 && offsetof(bwams_smem_t, k) == 16 && offsetof(SMEM, l) == offsetof(bwams_smem_t, l) &&
              offsetof(SMEM, s) == 32 && offsetof(bwams_smem_t, s) == 32, "SMEM");
#ifdef SMEM_ACCEL
static_assert(sizeof(all_smem_t) == 128 && sizeof(last_smem_t) == 16 && offsetof(last_smem_t, kls) == 4, "FMA table entries");
#endif

extern "C" {

/* Occ(c, pp) through the reference's own GET_OCC macro (src/FMI_search.h:76-83) over caller-supplied blocks;
 * one_hot_mask_array as load_index fills it (src/FMI_search.cpp:1253-1261) */
int64_t ref_get_occ(const void *blocks, int64_t pp, int c) {
    const CP_OCC *cp_occ = (const CP_OCC *)blocks;
    uint64_t one_hot_mask_array[64];
    one_hot_mask_array[0] = 0;
    uint64_t base = 0x8000000000000000L;
    one_hot_mask_array[1] = base;
    for (int64_t i = 2; i < 64; i++) one_hot_mask_array[i] = (one_hot_mask_array[i - 1] >> 1) | base;
    GET_OCC(pp, c, occ_id_pp, y_pp, occ_pp, one_hot_bwt_str_c_pp, match_mask_pp);
    return occ_pp;
}

/* the suffix array as build_index gets it: saisxx over the letters of the text (fw + rc strands, 'A' 'C' 'G' 'T'),
 * sa[0] = n for the terminator, sa[1..n] from saisxx (src/FMI_search.cpp:833-840) */
int ref_sais(const char *text, int64_t n, int64_t *sa) {
    const int status = saisxx<const char *, int64_t *, int64_t>(text, sa + 1, n);
    sa[0] = n;
    return status;
}

int ref_fmi_sizes(int which) {
    switch (which) {
        case 0: return (int)sizeof(CP_OCC);
        case 1: return (int)sizeof(SMEM);
#ifdef SMEM_ACCEL
        case 2: return (int)sizeof(all_smem_t);
        case 3: return (int)sizeof(last_smem_t);
#endif
        default: return -1;
    }
}

}
