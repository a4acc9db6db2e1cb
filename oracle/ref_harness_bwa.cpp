/*
 * ref_harness_bwa.cpp — the reference's own bseq1_t (src/bwa.h:76-86) in the `scale` build's configuration (OPT_RW: the per-work-item
 * SAM string; PERFECT_MATCH: the EMF's record), compiled where the header lies (TEST INFRASTRUCTURE ONLY).  bwa.h needs nothing the
 * image lacks; bwamem.h (mem_opt_t, mem_pestat_t, worker_t) reaches safestringlib and stays pinned to SURVEY §8(b)'s measurements.
 * The mirror the product's host layer compiles against (bwa-mem-scale_amd/host/bwamem_hip.h) asserts the same numbers; the
 * test compares this library's report with the numpy record the Python harness builds (tests/test_oracle_fmi_ref.py).
 */
#include <stddef.h>
#include <stdint.h>
#include <x86intrin.h>
#define __rdtsc __ref_rdtsc      /* utils.h declares its own __rdtsc; GCC >= 11 already has one */
#include "bwa.h"

static_assert(sizeof(bseq1_t) == 64, "bseq1_t of the scale build");
static_assert(offsetof(bseq1_t, l_seq) == 0 && offsetof(bseq1_t, id) == 4 && offsetof(bseq1_t, name) == 16 && offsetof(bseq1_t, comment) == 24 &&
              offsetof(bseq1_t, seq) == 32 && offsetof(bseq1_t, qual) == 40 && offsetof(bseq1_t, sam) == 48 && offsetof(bseq1_t, perfect) == 56,
              "bseq1_t field offsets (host/bwamem_hip.h mirrors them)");
static_assert(sizeof(bseq1_perfect_t) == 8, "bseq1_perfect_t");

extern "C" {
/* {sizeof, offsets of l_seq, id, strbuf, name, comment, seq, qual, sam, perfect, sizeof(perfect), offsets of perfect.flags, .location} */
void ref_bseq1_layout(int *out) {
    bseq1_t s;
    out[0] = (int)sizeof(bseq1_t);
    out[1] = (int)offsetof(bseq1_t, l_seq); out[2] = (int)offsetof(bseq1_t, id); out[3] = (int)offsetof(bseq1_t, strbuf);
    out[4] = (int)offsetof(bseq1_t, name); out[5] = (int)offsetof(bseq1_t, comment); out[6] = (int)offsetof(bseq1_t, seq);
    out[7] = (int)offsetof(bseq1_t, qual); out[8] = (int)offsetof(bseq1_t, sam); out[9] = (int)offsetof(bseq1_t, perfect);
    out[10] = (int)sizeof(s.perfect);
    out[11] = (int)((char *)&s.perfect.flags - (char *)&s.perfect); out[12] = (int)((char *)&s.perfect.location - (char *)&s.perfect);
}
}
