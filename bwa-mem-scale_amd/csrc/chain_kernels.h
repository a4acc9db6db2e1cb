// chain_kernels.h — device-side argument blocks and launchers of the chaining and
// chain-to-alignment stages (chain.hip, ext_aln.hip).
#pragma once

#include "common.h"

namespace bwams {

// bntseq_t as the kernels see it
struct DevBns {
    const bwams_contig_t *contigs;
    int32_t n_seqs;
    int64_t l_pac;
};

// Scratch of the chaining kernel.  Every array except `nodes` and the per-read ones is indexed
// like sa_coord (one slot per SA hit = per seed); a read owns the slice of its SMEMs' hits.
struct ChainArgs {
    const bwams_smem_t *smem;      // (rid, m, n)-sorted
    int64_t n_smem;
    const int64_t *sa_off;         // n_smem + 1
    const int64_t *sa_coord;
    const int64_t *cum;            // nseq + 1
    int64_t nseq;
    DevBns bns;
    bwams_mem_opt_t opt;
    // per seed
    int32_t *s_next;               // next seed of the same chain, -1 = last
    int2 *s_ql;                    // {qbeg, len}
    void *crec;                    // per chain (slot of its first seed): 64-byte chain record
    // filter work arrays (slot i of a read = i-th chain in sorted order)
    uint2 *flt;                    // {w | kept << 29 | is_alt << 31, chain id}
    uint4 *f_rec;                  // {chn_beg, chn_end, w | is_alt << 31, first}
    int32_t *f_first, *f_kept, *f_sel;
    void *nodes;                   // B-tree nodes
    // per read
    int32_t *n_kept, *n_kept_seeds, *n_chn;
    int32_t *heavy;                // reads handed to the wave-per-read kernel (ctr->n_heavy of them)
    int64_t *read_base;
    int64_t *slice;                // 2 per read: its [beg, end) in smem
    const uint32_t *order;         // read ids by descending seed count
    float *frac_rep;
    int32_t *redo;                 // reads to chain again with the B-tree (a chain position repeated)
    int32_t seed_batch;            // wave tier: 64 seeds per pass (chain_seeds_batch) instead of one
    DevCounters *ctr;
};

size_t chain_node_bytes(int64_t n_sa, int64_t nseq);
size_t chain_rec_bytes(int64_t n_sa);
void launch_chain_count(const ChainArgs &A, uint32_t *keys, uint32_t *vals, hipStream_t st);
int launch_chain(const ChainArgs &A, const uint32_t *n_seeds, int cu_count, hipStream_t st, hipStream_t *aux,
                 hipEvent_t fork, hipEvent_t *join);
void launch_chain_emit(const ChainArgs &A, const int64_t *chain_off, const int64_t *seed_off, bwams_chain_t *chains,
                       bwams_chain_seed_t *seeds, hipStream_t st);

// ---- chain -> alignment regions (ext_aln.hip) ----
struct ExtArgs {
    const bwams_chain_t *chains;
    int64_t n_chains;
    bwams_chain_seed_t *seeds;     // .aln is written
    int64_t n_seeds;               // == number of regions
    const int64_t *chain_off;      // nseq + 1
    const int64_t *seed_off;       // nseq + 1: first seed (= first region) of each read
    const uint8_t *enc;
    const int64_t *cum;
    int64_t nseq;
    const uint8_t *ref;            // 2 * l_pac bases
    DevBns bns;
    bwams_mem_opt_t opt;
    bwams_alnreg_t *regs;
    uint32_t *srt;                 // per region slot: the seed's index within its chain, visit order reversed (srtg)
    int64_t *rmax;                 // 2 per chain
    int32_t *cnt;                  // 6 x n_seeds: has_left, lq, lr, has_right, rq, rr
    int32_t *state;                // per seed: kept / purged / requested / extended (extension rounds)
    void *kreg;                    // per read (at its first region's slot): the regions kept so far, 32 B each
    int32_t *cur, *lim;            // per read: seeds decided so far, regions kept so far
    int32_t *sel_heavy;            // reads with many regions (the selection's wave tier)
    unsigned long long *n_sel_heavy, *sel_ticket;     // sel_ticket[0..2]: the small class, the big class' two passes
    DevCounters *ctr;
};
void launch_ext_plan(const ExtArgs &A, int extend_all, hipStream_t st);
void launch_ext_widen(const ExtArgs &A, int64_t *wide, hipStream_t st);
void launch_ext_build(const ExtArgs &A, const int64_t *offs, bwams_seqpair_t *left, uint8_t *lref, uint8_t *lqer,
                      bwams_seqpair_t *right, uint8_t *rref, uint8_t *rqer, int64_t *lsrc, int64_t *rsrc, int cu_count, hipStream_t st);
// after one extension attempt at band width w: settle finished tasks, queue the others for the next width
void launch_ext_post(const ExtArgs &A, int right, const bwams_seqpair_t *pairs, int64_t n, int w, int last_try,
                     bwams_seqpair_t *retry, unsigned long long *n_retry, hipStream_t st);
void launch_ext_right_h0(const ExtArgs &A, bwams_seqpair_t *right, int64_t n, hipStream_t st);
void launch_ext_heavy_list(const ExtArgs &A, hipStream_t st);
int launch_ext_select(const ExtArgs &A, int cu_count, hipStream_t st, hipStream_t *aux, hipEvent_t fork, hipEvent_t *join);
void launch_ext_request_rest(const ExtArgs &A, hipStream_t st);

// ---- mem_flt_chained_seeds for long reads (seed_sw.hip) ----
struct SeedSwArgs {
    bwams_chain_t *chains;
    int64_t n_chains;
    bwams_chain_seed_t *seeds;
    int64_t n_seeds;
    const uint8_t *enc;
    const int64_t *cum;
    int64_t nseq;
    const uint8_t *ref;
    DevBns bns;
    bwams_mem_opt_t opt;
    int32_t *cnt;                  // 3 x n_seeds: runs the SW, query window length, reference window length
    int32_t *win_qb;               // per seed: start of the query window
    int64_t *win_rb;               // per seed: start of the reference window
    int32_t *seed_read;            // per seed: its read
};
void launch_seedsw_plan(const SeedSwArgs &A, int64_t *wide, hipStream_t st);
void launch_seedsw_build(const SeedSwArgs &A, const int64_t *offs, bwams_seqpair_t *pairs, uint8_t *ref, uint8_t *qer,
                         int cu_count, hipStream_t st);
void launch_seedsw_apply(const SeedSwArgs &A, const int64_t *offs, const bwams_kswr_t *res, int32_t *new_n, int64_t *wide,
                         hipStream_t st);
void launch_seedsw_repack(const SeedSwArgs &A, const int32_t *new_n, const int64_t *new_off, bwams_chain_seed_t *out,
                          const int64_t *chain_off, int64_t *seed_off, hipStream_t st);

// ---- the tail of mem_kernel2_core: mem_sort_dedup_patch (dedup.hip) ----
struct DedupArgs {
    bwams_alnreg_t *regs;          // a working copy of the extension's regions (modified in place)
    const int64_t *seed_off;       // nseq + 1: a read's slots
    const uint8_t *enc;
    const int64_t *cum;
    int64_t nseq;
    const uint8_t *ref;
    DevBns bns;
    bwams_mem_opt_t opt;
    int32_t *ord;                  // per slot: the read's surviving regions in their current order
    void *srt;                     // per slot: 24-byte sort records (reads that do not use LDS)
    int2 *eh;                      // (h, e) rows of the global alignment: max_read_len + 2 cells per lane
    int64_t eh_lanes;              // strips [0, eh_lanes) belong to dedup_kernel's lanes, the following ones to the wave kernel
    int32_t max_read_len;
    int32_t *n_out;                // per read: regions left
    int32_t *heavy, *light;        // reads with many / few regions that need the full procedure (listed by the triage kernel)
    unsigned long long *n_heavy_ctr, *n_light_ctr, *ticket;
    int32_t force_seq;             // debug: lane 0 runs the one-lane form for every read
    unsigned long long *ticket2, *ticket3;   // work cursors of the wave tier's smaller instances
    unsigned long long *dbg;       // BWAMS_VERBOSE: cycles per phase of the largest wave instance (8 words), else nullptr
};
size_t dedup_sortrec_bytes(int64_t n);
int launch_dedup(const DedupArgs &A, int64_t n_lanes, int64_t n_waves, int64_t n_waves_small, hipStream_t st, hipStream_t aux,
                 hipStream_t aux2, hipStream_t aux3, hipEvent_t fork, hipEvent_t join, hipEvent_t join2, hipEvent_t join3);
void launch_pestat(const bwams_alnreg_t *regs, const int64_t *reg_off, int64_t n_pairs, int64_t l_pac, const bwams_mem_opt_t &opt,
                   unsigned long long *keys, hipStream_t st);
void launch_dedup_gather(const DedupArgs &A, const int64_t *out_off, bwams_alnreg_t *out, hipStream_t st);
// test hook (bwams_debug_sort): the wave tier's sort of n <= 1024 records on the current device
int launch_sort_test(const int64_t *k, const int32_t *s, const int32_t *q, int n, int by_score, int mode, int32_t *order);

// ---- regions of the reads the exact-match filter resolved (emf_regs.hip) ----
struct EmfRegArgs {
    DevEmf t;
    const uint32_t *perfect;       // 2 per read: flags, location (bseq1_perfect_t)
    const uint8_t *code;           // FIND_PERFECT_* per read
    const uint8_t *enc;
    const int64_t *cum;
    int64_t nseq;
    DevBns bns;
    bwams_mem_opt_t opt;
    void *scratch;                 // mem_aln_perfect_t-like records, a slice per read
};
size_t emfregs_scratch_bytes(int64_t n);
void launch_emfregs_count(const EmfRegArgs &A, int64_t *wide, hipStream_t st);
void launch_emfregs_merge_count(const int64_t *off_a, const int64_t *off_b, int64_t nseq, int64_t *wide, hipStream_t st);
void launch_emfregs_merge(const bwams_alnreg_t *a, const int64_t *off_a, const bwams_alnreg_t *b, const int64_t *off_b, int64_t nseq,
                          const int64_t *off_o, bwams_alnreg_t *out, hipStream_t st);
void launch_emfregs_fill(const EmfRegArgs &A, const int64_t *scr_off, int32_t *n_final, uint8_t *first_is_rev, int64_t *wide,
                         hipStream_t st);
void launch_emfregs_emit(const EmfRegArgs &A, const int64_t *scr_off, const int32_t *n_final, const int64_t *out_off,
                         bwams_alnreg_t *out, hipStream_t st);

// ---- paired-end tail: mate rescue, mem_mark_primary_se, mem_pair (pair.hip) ----
struct PairArgs {
    const bwams_alnreg_t *regs;    // final regions after de-duplication, grouped by read
    const int64_t *reg_off;
    const uint8_t *enc;
    const int64_t *cum;
    int64_t nseq;                  // even: reads 2p, 2p + 1 are the ends of pair p
    const uint8_t *ref;
    DevBns bns;
    bwams_mem_opt_t opt;
    bwams_pestat_t pes[4];
    int64_t id_base;               // n_processed >> 1 of the chunk (mem_mark_primary_se / mem_pair hash their ids)
    int32_t no_rescue, pass;
    int32_t drop_plan;             // test knob: the first pass plans nothing, so every rescue goes through the second
    int32_t use_ert;               // mem_sam_pe_batch_post's useErt branch: mem_matesw_batch_post_ert
    int32_t single_end;            // mem_reg2sam's form: every read on its own, id = id_base + read, no rescue, no pairing
    int32_t no_pairing;            // MEM_F_NOPAIRING: mem_pair is not called (score 0, z = -1; n_pri still reported)
    int32_t primary5_T;            // MEM_F_PRIMARY5: mem_reorder_primary5(T, a) of every read after the marking; < 0 = off
    int32_t *na;                   // per read: anchors it provides
    const int64_t *aoff, *ooff;    // per read: first anchor slot, first pool slot
    int32_t *anchor, *slot_read;   // per anchor slot: region index within its read, the read
    int64_t n_slots;
    int32_t *task;                 // per slot x 4 orientations: rescue alignment index or -1
    int64_t *trb;                  // window start
    int32_t *tl1;                  // window length or -1
    const int32_t *aln;            // 7 per task (kswr_t)
    bwams_alnreg_t *pool;          // per read: its regions, then room for the rescued ones
    int32_t *ord, *zbuf;           // per pool slot
    void *srt;                     // per pool slot: a 24-byte sort record
    int32_t *n_fin, *n_pri, *n_sw; // per read
    uint8_t *full;                 // per read: redo with every orientation planned
    int32_t *heavy;                // reads whose list mem_mark_primary_se handles with a wavefront
    DevCounters *ctr;
};
void launch_pair_count(const PairArgs &A, int64_t *wide, hipStream_t st);
void launch_pair_cap(const PairArgs &A, int64_t *wide, hipStream_t st);
void launch_pair_slots(const PairArgs &A, hipStream_t st);
void launch_pair_plan(const PairArgs &A, int64_t *wide, hipStream_t st);
void launch_pair_build(const PairArgs &A, const int64_t *offs, bwams_seqpair_t *pairs, uint8_t *tref, uint8_t *tqer, int cu_count,
                       hipStream_t st);
void launch_pair_post(const PairArgs &A, int cu_count, hipStream_t st);
void launch_pair_mark(const PairArgs &A, int cu_count, hipStream_t st);
void launch_pair_widen(const PairArgs &A, int64_t *wide, hipStream_t st);
void launch_pair_gather(const PairArgs &A, const int64_t *out_off, bwams_alnreg_t *out, hipStream_t st);
void launch_pair_reorder5(const PairArgs &A, const int64_t *out_off, bwams_alnreg_t *out, hipStream_t st);
void launch_pair_pair(const PairArgs &A, const int64_t *out_off, const bwams_alnreg_t *out, bwams_pair_t *res, hipStream_t st);

// ---- ERT mode: MEMs + hits of the reference's ERT walk -> the chaining kernels' input (ert_chain.hip) ----
struct ErtArgs {
    const bwams_ert_mem_t *mems;   // all reads, grouped by read
    const int64_t *mem_off;        // nseq + 1
    const uint64_t *hits;
    const int64_t *hit_off;        // nseq + 1 (mem.hitbeg is relative to the read's slice)
    int64_t nseq, n_mems, l_pac;
    int32_t max_occ, pad_;
    bwams_smem_t *smem_out;        // one record per MEM, sorted within the read
    int64_t *cnt;                  // n_mems + 1: positions each record contributes
    void *srt;                     // n_mems sort records
};
void launch_ert_sort(const ErtArgs &A, hipStream_t st);
void launch_ert_pick(const ErtArgs &A, const int64_t *sa_off, int64_t *coord, hipStream_t st);

// ---- SAM-side alignment: mem_reg2aln -> bwa_gen_cigar2 -> ksw_global2 with traceback (reg2aln.hip) ----
struct RegAlnArgs {
    const bwams_alnreg_t *regs;    // final regions, grouped by read
    const int64_t *reg_off;        // nseq + 1
    int64_t n_regs, nseq;
    const uint8_t *enc;
    const int64_t *cum;
    const uint8_t *ref;            // .0123
    DevBns bns;
    bwams_mem_opt_t opt;
    int64_t *need;                 // per region: bytes of scratch
    int32_t *cls;                  // per region: -2 rejected, -1 no DP, 0 / 1 LDS ring class, 2 HBM row
    const int64_t *scr_off;        // per region: offset into scr
    uint8_t *scr;
    int32_t *list;                 // 3 * n_regs: regions that need DP, by class
    unsigned long long *n_list;    // 3 counters
    bwams_aln_t *rec;
    const uint8_t *only;           // per region: non-zero = align it; null = every region (bwams_reg2aln_run)
};
void launch_aln_plan(const RegAlnArgs &A, hipStream_t st);
void launch_aln_run(const RegAlnArgs &A, int cu_count, hipStream_t st);
void launch_aln_sizes(const RegAlnArgs &A, int64_t *wide, hipStream_t st);
void launch_aln_gather(const RegAlnArgs &A, const int64_t *offs, uint32_t *cig, char *md, hipStream_t st);

// single-end SAM text (sam_text.hip)
struct SamArgs {
    const bwams_alnreg_t *regs;    // final regions after mem_mark_primary_se, grouped by read
    const int64_t *reg_off;        // nseq + 1
    int64_t n_regs, nseq;
    const bwams_aln_t *rec;        // mem_reg2aln records of the same regions, CIGAR and MD pools
    const uint32_t *cig;
    const char *md;
    const uint8_t *enc;
    const int64_t *cum;
    const char *names;             // read names back to back, name_off[nseq + 1]
    const int64_t *name_off;
    const char *quals;             // laid out like enc, or null
    const char *comments;          // back to back, comment_off[nseq + 1] (empty = none), or null
    const int64_t *comment_off;
    const char *ctg_names;         // NUL-terminated sequence names back to back, ctg_off[rid]
    const int32_t *ctg_off;
    const char *ctg_annos;         // MEM_F_REF_HDR: bntann1_t.anno of every sequence, NUL-terminated back to back, ctg_anno_off[rid]
    const int32_t *ctg_anno_off;
    bwams_mem_opt_t opt;
    bwams_sam_opt_t sopt;
    const double *logtab;          // log(i) from the host's C library
    int32_t logtab_n;
    double coef_fac;               // log(mapQ_coef_len)
    const bwams_alnreg_t *er_regs; // reads the EMF resolved (single-end): their mem_perfect2reg regions, er_off[nseq + 1]; null = none
    const int64_t *er_off;
    int32_t er_seed_len;           // the table's L
    const bwams_contig_t *contigs; // (offsets of the sequences: the exact-match record's POS)
    const bwams_pair_t *pairs;     // paired-end: mem_pair's result per pair (null: single-end)
    bwams_pestat_t pes[4];
    int64_t bns_l_pac;
    int32_t *mapq;                 // per region
    unsigned long long *bad;       // regions whose lengths fall outside logtab
    int64_t *len;                  // per read: bytes of text
    const int64_t *out_off;
    char *out;
};
void launch_sam_need(const SamArgs &A, uint8_t *need, int cu_count, hipStream_t st);
void launch_sam_mapq(const SamArgs &A, hipStream_t st);
void launch_sam_text(const SamArgs &A, bool emit, int cu_count, hipStream_t st);

}  // namespace bwams
