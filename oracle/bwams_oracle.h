/*
 * bwams_oracle.h — CPU restatement of the reference's seed-and-extend hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product path (bwa-mem-scale_amd/) never links
 * or calls into this directory.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - banded-SW extension (orc_bsw_*): PINNED against the reference's own
 *     src/bandedSWA.cpp compiled from where it lies (oracle/_ref, Makefile here).
 *   - FM-index seeding (orc_fmi_*, orc_collect_smem, orc_sa_*): PARITY UNPINNED.
 *     src/FMI_search.cpp and src/bwamem.cpp include the un-vendored safestringlib
 *     (ext/safestringlib is an empty submodule) and cannot be compiled here
 *     without writing stand-in headers, which this project does not do; the
 *     reference ships no tests or golden vectors.  The restatement is instead
 *     cross-checked against a from-definition brute-force model
 *     (tests/test_oracle_fmi.py): Occ by counting, intervals by naive suffix
 *     sorting, SMEM maximality by exhaustive substring search.
 */
#ifndef BWAMS_ORACLE_H
#define BWAMS_ORACLE_H

#include <stdint.h>
#include "../include/bwams_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* In-memory FM-index exactly as FMI_search holds it after load_index
 * (src/FMI_search.cpp:875-906): count[] already carries the +1. */
typedef struct orc_fmi {
    int64_t ref_seq_len;              /* 2*l_pac + 1 */
    int64_t count[5];                 /* file values + 1 */
    const bwams_cp_occ_t *cp_occ;     /* (ref_seq_len >> 6) + 1 blocks */
    const int8_t  *sa_ms_byte;        /* (ref_seq_len >> 3) + 1 */
    const uint32_t *sa_ls_word;       /* (ref_seq_len >> 3) + 1 */
    int64_t sentinel_index;
    /* FMA direct-lookup tables (src/FMI_search.h:101-135); NULL = FM-index only.
     * all_bp / last_bp are ALL_SMEM_MAX_BP (11) / LAST_SMEM_MAX_BP (13) in the reference;
     * smaller depths exist only so that tests can build tables quickly. */
    const void *all_smem;             /* 4^all_bp entries of 128 B (all_smem_t) */
    const void *last_smem;            /* 4^last_bp entries of 16 B (last_smem_t) */
    int32_t all_bp, last_bp;
} orc_fmi_t;

/* FMA table builders (src/FMI_search.cpp:78-227). */
void orc_build_all_smem(const orc_fmi_t *f, int bp, void *table);
void orc_build_last_smem(const orc_fmi_t *f, int bp, void *table);

/* Event counters used to derive the algorithmic byte count of a workload
 * (SURVEY.md §8d): one CP_OCC block is 64 B. */
typedef struct orc_counters {
    int64_t n_ext;            /* backwardExt calls */
    int64_t n_ext_blocks;     /* distinct CP_OCC blocks those calls touch (1 or 2 each) */
    int64_t n_sa_lookups;     /* SA entries produced */
    int64_t n_lf_steps;       /* LF-mapping steps walked by SA lookups */
    int64_t n_smem[3];        /* SMEMs emitted by round 1, 2, 3 */
} orc_counters_t;

/* Occ / backwardExt (src/FMI_search.h:76-83, src/FMI_search.cpp:2029-2056). */
int64_t orc_fmi_occ(const orc_fmi_t *f, int64_t pos, int c);
void orc_backward_ext(const orc_fmi_t *f, const bwams_smem_t *in, int a,
                      bwams_smem_t *out, orc_counters_t *ctr);

/* forward extension by base a (src/FMI_search.cpp:1475-1485) */
void orc_forward_ext(const orc_fmi_t *f, const bwams_smem_t *in, int a, bwams_smem_t *out);

/* getSMEMsOnePosOneThread (src/FMI_search.cpp:1372-1606), FM-index only path
 * (no all_smem table).  query_pos[] is in/out; returns SMEMs appended. */
int64_t orc_smem_one_pos(const orc_fmi_t *f, const uint8_t *enc_qdb,
                         int16_t *query_pos, const int32_t *min_intv,
                         const int32_t *rid, int32_t num_reads,
                         const int64_t *cum_len, int32_t min_seed_len,
                         bwams_smem_t *out, orc_counters_t *ctr);

/* getSMEMsAllPosOneThread (src/FMI_search.cpp:1608-1660). rid[]/min_intv[] are
 * clobbered exactly as in the reference. */
int64_t orc_smem_all_pos(const orc_fmi_t *f, const uint8_t *enc_qdb,
                         int32_t *min_intv, int32_t *rid, int32_t num_reads,
                         const int64_t *cum_len, int32_t min_seed_len,
                         bwams_smem_t *out, orc_counters_t *ctr);

/* bwtSeedStrategyAllPosOneThread (src/FMI_search.cpp:1662-1816), no last_smem
 * table.  skip[i] != 0 mirrors seq_[i].perfect.exist. */
int64_t orc_seed_strategy(const orc_fmi_t *f, const uint8_t *enc_qdb,
                          const uint8_t *skip, int32_t max_intv,
                          int32_t num_reads, const int64_t *cum_len,
                          int32_t min_seed_len, bwams_smem_t *out,
                          orc_counters_t *ctr);

/* mem_collect_smem (src/bwamem.cpp:648-786): three rounds, then the
 * (rid, m, n) ordering left by sortSMEMs + the per-read introsort.
 * cum_len has nseq+1 entries.  Returns the SMEM count or -1 on overflow. */
int64_t orc_collect_smem(const orc_fmi_t *f, const bwams_seed_opt_t *opt,
                         const uint8_t *enc_qdb, const int64_t *cum_len,
                         const uint8_t *skip, int32_t nseq,
                         bwams_smem_t *out, int64_t cap, orc_counters_t *ctr);

/* call_one_step driven to completion (src/FMI_search.cpp:2206-2259): the SA
 * value of BWT row pos, with the sentinel quirk (returns 0). */
int64_t orc_sa_entry(const orc_fmi_t *f, int64_t pos, orc_counters_t *ctr);

/* get_sa_entries_prefetch as used by mem_chain_seeds
 * (src/FMI_search.cpp:2261-2379, src/bwamem.cpp:861-873): for each SMEM the
 * rows k, k+step, ... (at most max_occ).  sa_off gets n+1 prefix offsets.
 * Returns the number of coordinates or -1 on overflow. */
int64_t orc_sa_lookup(const orc_fmi_t *f, const bwams_smem_t *smem, int64_t n,
                      int32_t max_occ, int64_t *coord, int64_t cap,
                      int64_t *sa_off, orc_counters_t *ctr);

/* scalarBandedSWA (src/bandedSWA.cpp:116-237). Returns the score. */
int orc_bsw_scalar(const bwams_sw_opt_t *o, int qlen, const uint8_t *query,
                   int tlen, const uint8_t *target, int32_t w, int h0,
                   int *qle, int *tle, int *gtle, int *gscore, int *max_off,
                   int64_t *cells);

/* scalarBandedSWAWrapper (src/bandedSWA.cpp:242-260). */
void orc_bsw_pairs(const bwams_sw_opt_t *o, bwams_seqpair_t *pairs,
                   const uint8_t *ref, const uint8_t *qer, int64_t n,
                   int32_t w, int64_t *cells);

/* Exact-match filter (EMF): the mapped perfect table (src/perfect.h:93-213) */
typedef struct { uint32_t flags, location, left, right; } orc_seed_entry_t;
typedef struct orc_emf {
    int32_t seed_len;
    uint32_t num_loc_entry, num_seed_entry, seq_len;
    const uint32_t *loc_table;
    const orc_seed_entry_t *seed_table;
    const uint8_t *ref;                /* .0123 */
} orc_emf_t;
int64_t orc_emf_hash(uint32_t num_seed_entry, const uint8_t *s, int len, int fw);
int orc_emf_seedcmp(const uint8_t *a, int afl, const uint8_t *b, int bfl, int len);
int orc_emf_compare_fw_rc(const uint8_t *seed, int len);
int orc_emf_match_further(const orc_emf_t *t, uint32_t loc, const uint8_t *seed, int is_rev, int len);
/* find_perfect_match_entry (src/perfect_map.cpp:638-659): returns the FIND_PERFECT_* code */
int orc_emf_probe(const orc_emf_t *t, const uint8_t *seed, int len, uint32_t *flags, uint32_t *location);

/* mem_perfect2reg with get_perfect_locations and perfect_dedup_patch (src/perfect_map.cpp:659-869): the regions of a
 * read that the EMF resolved (flags / location = its bseq1_perfect_t).  Returns the count (-1: cap too small). */
struct orc_bns;
int orc_perfect2reg(const bwams_mem_opt_t *opt, const orc_emf_t *t, const struct orc_bns *bns, const uint8_t *seq, int l_seq,
                    uint32_t flags, uint32_t location, bwams_alnreg_t *out, int cap, int *first_is_rev);

/* ksw_align2 (src/ksw.cpp:347-381) over ksw_u8 / ksw_i16: local SW of mate rescue.
 * out[7] = score, te, qe, score2, te2, tb, qb. */
void orc_ksw_align2(const bwams_sw_opt_t *o, int qlen, const uint8_t *query, int tlen,
                    const uint8_t *target, int xtra, int *out);

/* ---- chaining, chain filtering, chain -> alignment regions (chain_oracle.c) ----
 * Driver logic PARITY UNPINNED (bwamem.cpp is not buildable here); the B-tree and the
 * introsort it relies on are PINNED against the reference's kbtree.h / ksort.h
 * (oracle/_ref/libref_chain.so, tests/test_oracle_chain.py). */
typedef struct orc_bns {            /* the fields of bntseq_t the path reads */
    int64_t l_pac;
    int32_t n_seqs;
    const bwams_contig_t *contigs;
} orc_bns_t;

/* test hooks for the pinned pieces */
/* SAM-side alignment (aln_oracle.c): ksw_global2 with traceback (pinned), mem_approx_mapq_se, mem_reg2aln (unpinned) */
int orc_ksw_global2_cigar(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del,
                          int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t *cigar);
int orc_approx_mapq_se(const bwams_mem_opt_t *opt, const bwams_alnreg_t *a);
int orc_reg2aln(const bwams_mem_opt_t *opt, const struct orc_bns *bns, const uint8_t *ref_string, int l_query, const uint8_t *query,
                const bwams_alnreg_t *ar, bwams_aln_t *a, uint32_t *cigar, char *md);
/* single-end SAM text (sam_oracle.c): mem_reg2sam + mem_gen_alt + mem_aln2sam for one read (PARITY UNPINNED).  ctg_names:
 * NUL-terminated names back to back, ctg_off[rid] = start of a name.  Returns the text length, or -1 - length if cap was short. */
void orc_set_contig_annos(const char *annos, const int32_t *anno_off);      /* MEM_F_REF_HDR: bntann1_t.anno table, NULL = none */
int64_t orc_reg2sam_se(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const struct orc_bns *bns, const char *ctg_names,
                       const int32_t *ctg_off, const uint8_t *ref_string, int l_seq, const uint8_t *seq, const char *qual,
                       const char *name, const char *comment, const bwams_alnreg_t *regs, int n_regs, char *out, int64_t cap);
/* mem_perfect2sam_cont for one read the EMF resolved, from the regions mem_perfect2reg leaves for it */
int64_t orc_perfect2sam(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const struct orc_bns *bns, const char *ctg_names,
                        const int32_t *ctg_off, int seed_len, int l_seq, const uint8_t *seq, const char *qual, const char *name,
                        const char *comment, const bwams_alnreg_t *regs, int n, char *out, int64_t cap);
/* mem_sam_pe from mem_pair's result on, for one pair (regs are edited in place as the reference edits them) */
void orc_sam_pe(const bwams_mem_opt_t *opt, const bwams_sam_opt_t *so, const struct orc_bns *bns, const char *ctg_names,
                const int32_t *ctg_off, const uint8_t *ref_string, const bwams_pestat_t pes[4], const int32_t l_seq[2],
                const uint8_t *const seq[2], const char *const qual[2], const char *const name[2], const char *const comment[2],
                bwams_alnreg_t *const regs[2], const int32_t n_regs[2], const bwams_pair_t *pr, char *const out[2], const int64_t cap[2],
                int64_t len[2]);
/* read input (fastq_oracle.c): kseq_read + trim_readno + kseq2bseq1 + the base encoding over a memory buffer (PARITY UNPINNED) */
int64_t orc_fastq_parse(const char *buf, int64_t n, int64_t max_reads, char *names, int64_t *name_off, char *comments,
                        int64_t *comment_off, uint8_t *seq, char *qual, int64_t *cum, uint8_t *has_qual);
uint64_t orc_hash_64(uint64_t key);
int64_t orc_depos(int64_t l_pac, int64_t pos, int *is_rev);
int64_t orc_kbt_script(int64_t n, const int64_t *pos, const uint8_t *do_put, int32_t *lower, int32_t *order);
void orc_flt_sort(int64_t n, const uint32_t *w, int32_t *order);

int orc_chain_flt(const bwams_mem_opt_t *opt, int n_chn, bwams_chain_t *a, const bwams_chain_seed_t *seeds);
int64_t orc_chain_seeds(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const bwams_smem_t *smem, int64_t num_smem,
                        const int64_t *sa_coord, const int64_t *sa_off, const int64_t *cum_len, int32_t nseq, int do_flt,
                        bwams_chain_t *chains, int64_t chain_cap, bwams_chain_seed_t *seeds, int64_t seed_cap,
                        int64_t *chain_off, int64_t *n_seeds_out, const uint8_t *ref_string, const uint8_t *enc_qdb);

/* ERT mode (a13): ks_introsort(mem_smem_sort_lt) + mem_chain_new + mem_chain_flt + mem_flt_chained_seeds over the
 * MEMs / hits an ERT walk produced (bwamem.cpp:961-1050, :1193-1203).  Same pinning status as orc_chain_seeds. */
int64_t orc_chain_new_ert(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const bwams_ert_mem_t *mems, const int64_t *mem_off,
                          const uint64_t *hits, const int64_t *hit_off, const int64_t *cum_len, int32_t nseq, int do_flt,
                          bwams_chain_t *chains, int64_t chain_cap, bwams_chain_seed_t *seeds, int64_t seed_cap,
                          int64_t *chain_off, int64_t *n_seeds_out, const uint8_t *ref_string, const uint8_t *enc_qdb);

typedef struct orc_task_dump {      /* the extension task lists as mem_chain2aln_across_reads_V2 builds them */
    int32_t build_only;             /* in: stop after building (regions hold the pre-extension state) */
    int32_t pad_;
    int64_t n_left, n_right;
    bwams_seqpair_t *left, *right;
    uint8_t *left_ref, *left_qer, *right_ref, *right_qer;
    int64_t left_ref_bytes, left_qer_bytes, right_ref_bytes, right_qer_bytes;
} orc_task_dump_t;
int64_t orc_chain2aln(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const uint8_t *enc_qdb,
                      const int64_t *cum_len, int32_t nseq, const bwams_chain_t *chains, const int64_t *chain_off,
                      bwams_chain_seed_t *seeds, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off,
                      orc_task_dump_t *dump);
void orc_task_dump_free(orc_task_dump_t *d);

/* ---- the tail of mem_kernel2_core (dedup_oracle.c): purged regions dropped, mem_sort_dedup_patch, ALT mark ----
 * ksw_global2 and the two sorts PINNED (reference ksw.cpp object / ksort.h); the driver logic PARITY UNPINNED. */
int orc_ksw_global2_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del,
                          int e_del, int o_ins, int e_ins, int w);
void orc_ars_sort(int64_t n, int which, const int64_t *k0, const int64_t *k1, const int64_t *k2, int32_t *order);
/* mem_pestat (bwamem_pair.cpp:89-156) over the final regions; reads 2i, 2i+1 = pair i.  PARITY UNPINNED. */
void orc_pestat(const bwams_mem_opt_t *opt, int64_t l_pac, int n, const bwams_alnreg_t *regs, const int64_t *reg_off,
                bwams_pestat_t pes[4]);
int64_t orc_pestat_keys(const bwams_mem_opt_t *opt, int64_t l_pac, int n, const bwams_alnreg_t *regs, const int64_t *reg_off,
                        uint64_t *keys);
int64_t orc_regs_finish(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const uint8_t *enc_qdb,
                        const int64_t *cum_len, int32_t nseq, bwams_alnreg_t *regs, const int64_t *reg_off, int64_t *out_off);

/* ---- paired-end tail up to the pairing decision (pair_oracle.c): mate rescue (mem_matesw*), mem_mark_primary_se,
 * mem_pair.  Driver logic PARITY UNPINNED; ksw_align2 and the tie-sensitive sorts PINNED as above. ---- */
int orc_mark_primary_se(const bwams_mem_opt_t *opt, int n, bwams_alnreg_t *a, int64_t id);
int64_t orc_pair_pe(const bwams_mem_opt_t *opt, const orc_bns_t *bns, const uint8_t *ref_string, const uint8_t *enc_qdb,
                    const int64_t *cum_len, int32_t n_pairs, const bwams_alnreg_t *regs, const int64_t *reg_off,
                    const bwams_pestat_t pes[4], int64_t id_base, int flags /* 1: no rescue, 2: useErt, 4: no pairing */,
                    int primary5_T /* < 0: off */, bwams_alnreg_t *out, int64_t out_cap, int64_t *out_off, bwams_pair_t *pairs);
void orc_reorder_primary5(int T, int n, bwams_alnreg_t *a);

/* ---- ERT index (ert_oracle.c): writer, decoder, seeding.  PARITY UNPINNED (see the file header). ---- */
typedef struct orc_ert {
    int32_t kmer, xmer, read_len, hit_threshold;   /* kmerSize 15, xmerSize 4, READ_LEN, HIT_THRESHOLD 256 in the reference */
    const uint64_t *kmer_table;                    /* 4^kmer entries */
    const uint8_t *mlt;                            /* <prefix>.mlt_table */
    int64_t mlt_bytes;
    const uint8_t *ref;                            /* .0123, both strands */
    int64_t ref_len;                               /* 2 * l_pac */
} orc_ert_t;
/* buildKmerTrees (src/ertindex.cpp:773-943): fills kmer_table (4^kmer entries) and returns the malloc'ed tree bytes */
uint8_t *orc_ert_build(const orc_fmi_t *f, int kmer, int xmer, int read_len, int hit_threshold, uint64_t *kmer_table,
                       int64_t *mlt_bytes);
void orc_ert_free(uint8_t *p);
/* L[m-1] = longest prefix of read[i..) with at least m occurrences, m = 1..M (M <= 20) */
void orc_ert_profile(const orc_ert_t *e, const uint8_t *q, int len, int i, int M, uint8_t *L);
/* occurrences of read[i, i+mlen) in right-context order; returns the count */
int64_t orc_ert_hits(const orc_ert_t *e, const uint8_t *q, int len, int i, int mlen, int64_t *hits, int64_t cap);
/* the three seeding rounds + hit sampling of mem_kernel1_core_ert, in the layout of orc_collect_smem + orc_sa_lookup */
int64_t orc_ert_collect(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *enc, const int64_t *cum,
                        const uint8_t *skip, int32_t nseq, bwams_smem_t *out, int64_t cap, int64_t *sa_coord,
                        int64_t sa_cap, int64_t *sa_off);

/* ---- the reference's own ERT walk (ert_walk_oracle.c): get_seeds[_prefix] / reseed[_prefix] / last and every tree-walk
 * variant of src/ertseeding.cpp restated function by function, i.e. the MEM records (forward / fetch_leaves /
 * end_correction) and the hit array in the order the walk pushes them: what mem_kernel1_core_ert hands mem_chain_new. ---- */
enum {
    ORC_ERTW_ASSERT    = 1,     /* one of the reference's asserts would have fired */
    ORC_ERTW_LEP_RANGE = 2,     /* a LEP bit beyond the 320-bit vector */
    ORC_ERTW_STACK     = 4      /* the visited-node stack popped empty / overflowed */
};
int64_t orc_ert_walk(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *enc, const int64_t *cum, const uint8_t *skip,
                     int32_t nseq, bwams_ert_mem_t *mems, int64_t mem_cap, int64_t *mem_off, uint64_t *hits, int64_t hit_cap,
                     int64_t *hit_off, int32_t *flags_out);
int64_t orc_ert_walk_collect(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *enc, const int64_t *cum,
                             const uint8_t *skip, int32_t nseq, bwams_smem_t *out, int64_t cap, int64_t *sa_coord, int64_t sa_cap,
                             int64_t *sa_off, uint8_t *cls, int32_t *flags_out);

#ifdef __cplusplus
}
#endif
#endif
