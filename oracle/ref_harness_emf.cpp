/*
 * ref_harness_emf.cpp — extern "C" entry points around the REAL inline primitives of the
 * reference's exact-match filter, /root/reference/src/perfect.h (TEST INFRASTRUCTURE ONLY).
 *
 * perfect.h is header-only for these functions (hash, canonical-strand test, ordered compare,
 * tail match) and needs nothing outside the reference tree, so this file is compiled against it
 * where it lies (oracle/Makefile -> oracle/_ref/libref_emf.so).  perfect_map.cpp itself (the
 * probe loop) includes the un-vendored safestringlib and is not buildable here.
 */
#include <stdint.h>
#include <string.h>
#include <assert.h>
#include <x86intrin.h>
#define __rdtsc __ref_rdtsc      /* utils.h declares its own __rdtsc; GCC >= 11 already has one */
#include "macro.h"
#include "perfect.h"

extern "C" {

int64_t ref_emf_hash_fw(uint32_t num_seed_entry, const uint8_t *s, int len) {
    perfect_table_t pt; memset(&pt, 0, sizeof pt); pt.num_seed_entry = num_seed_entry;
    return __get_hash_idx_fw(&pt, s, len);
}
int64_t ref_emf_hash_rc(uint32_t num_seed_entry, const uint8_t *s, int len) {
    perfect_table_t pt; memset(&pt, 0, sizeof pt); pt.num_seed_entry = num_seed_entry;
    return __get_hash_idx_rc(&pt, s, len);
}
int ref_emf_compare_fw_rc(uint8_t *s, int len) { return __compare_fw_rc(s, len); }
int ref_emf_seedcmp(uint8_t *a, int afl, uint8_t *b, int bfl, int len) { return __seedcmp(a, afl, b, bfl, len); }
int ref_emf_match_further(uint8_t *ref, uint32_t seq_len, int seed_len, uint32_t loc, uint8_t *seed, int is_rev, int len) {
    perfect_table_t pt; memset(&pt, 0, sizeof pt);
    pt.ref_string = ref; pt.seq_len = seq_len; pt.seed_len = seed_len;
    return __seedmatch_further(&pt, loc, seed, is_rev, len);
}
int ref_emf_sizeof_table_header(void) { return (int)sizeof(perfect_table_t); }
int ref_emf_sizeof_seed_entry(void) { return (int)sizeof(seed_entry_t); }

}
