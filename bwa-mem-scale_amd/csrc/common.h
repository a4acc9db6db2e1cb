// common.h — internal declarations shared by the HIP translation units.
#pragma once

#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/bwams.h"

namespace bwams {

void set_last_error(const std::string &s);

// Every device allocation of the library goes through this: BWAMS_POISON=1 (debugging aid) fills the fresh block with 0xAB bytes and waits
// for the fill, so that a kernel that reads what nothing wrote misbehaves in every run — not only when the allocator hands back a block that
// another chunk left dirty (a fresh process gets zeros).  tests: the whole `-m gpu` suite passes under it.
// Debugging aids and A-B switches, all of them result-neutral: read from the environment ONCE (at the first call into the library; again
// on bwams_debug_reload(), which the tests call after changing a variable) — never per call on the hot host path.  include/bwams.h lists them.
struct Knobs {
    int verbose = 0;               // BWAMS_VERBOSE: stage times and counters per run on stderr
    int debug = 0;                 // BWAMS_DEBUG: diagnostic ablations of the SMEM search (bit 0: no SMEM is written)
    int poison = 0;                // BWAMS_POISON: fresh device allocations are filled with 0xAB
    int bwd_min_list = 40, bwd_cols = 24, bwd_late_list = 8;                     // BWAMS_BWD_MIN_LIST / _COLS / _LATE_LIST   (fmi_seed.hip: bwd_hand_over)
    int bwd_dry_min_list = 24, bwd_dry_cols = 8, bwd_dry_late_list = 12;         // BWAMS_BWD_DRY_*: the same once the work queue is dry
    int bwd_fused = 1;             // BWAMS_BWD_FUSED=0: the two roles behind a search kernel as two launches
    int bwd_cap_mul = 1;           // BWAMS_BWD_CAP_MUL: hand-over buffers x this (experiments that hand every backward phase over)
    int r3_beside = 1;             // BWAMS_SEED_R3_BESIDE: 1 (default) = SMEM round 3 beside round 2; 0 = behind it; 2 = from the START of the stage into a
                                   // pool of its own (measured: its workgroups do get placed beside round 1's — round 1 16.1 -> 18.3 ms, stage 35.3 -> 35.8)
    int ext_max_rounds = 0;        // BWAMS_EXT_MAX_ROUNDS: cap of the extension rounds (tests force the extend-the-rest fallback)
    int ext_all_rounds = 0;        // BWAMS_EXT_ALL_ROUNDS: never cut the rounds short
    int ext_inplace = 1;           // BWAMS_EXT_INPLACE=0: extension tasks copied into flat buffers
    int dedup_seq = 0;             // BWAMS_DEDUP_SEQ=1: every read through de-duplication's one-lane form
    int pair_drop_plan = 0;        // BWAMS_PAIR_DROP_PLAN: exercise mate rescue's second pass
    int trace_pair = 0;            // BWAMS_TRACE_PAIR: a synchronisation and a line per launch of the paired-end tail
    int bsw_pk = 1;                // BWAMS_BSW_PK=0: the 32-bit eight-task banded-SW kernel
    int chain_batch = 1;           // BWAMS_CHAIN_BATCH=0: chaining's wave tier takes one seed at a time (chain.hip: chain_seeds_batch)
    int fwd_bpc = 8, bwdl_bpc = 6; // BWAMS_FWD_BPC / BWAMS_BWDL_BPC: workgroups per CU of the forward / backward lane kernels (lab)
    int seed_split = 0;            // BWAMS_SEED_SPLIT=1: SMEM rounds 1 and 2 as a forward kernel + a backward kernel (fmi_seed.hip)
    int cp2 = 2;                   // BWAMS_CP2: the table the SMEM search kernels read — 2 (default): the INTERLEAVED form of CP_OCC (piece b = count and
                                   // string of base b: an extension reads half a block per end, fetched by a pair of lanes); 1: the compact 128-rows-per-block
                                   // form (measured: no gain); 0: CP_OCC itself
    int ert_fat = 1;               // BWAMS_ERT_FAT=0: the ERT walk reads the reference's two tables only (no entry + tree-head table)
    int ert_grid = -1, ert_ticket = 1;   // BWAMS_ERT_GRID (blocks per CU, 0 = one block per 256 bases) / BWAMS_ERT_TICKET=0 (round robin)
};
const Knobs &knobs();
void knobs_reload();

template <class T> static inline hipError_t dev_malloc(T **p, size_t bytes) {
    const bool poison = knobs().poison != 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes);
    if (e == hipSuccess && poison && bytes) {
        e = hipMemset(*p, 0xAB, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();     // hipMemset on the null stream does not order with non-blocking streams
    }
    return e;
}

#define BWAMS_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            char buf_[512];                                                              \
            snprintf(buf_, sizeof buf_, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,    \
                     hipGetErrorString(e_));                                             \
            bwams::set_last_error(buf_);                                                 \
            return (e_ == hipErrorOutOfMemory) ? BWAMS_ERR_NOMEM : BWAMS_ERR_DEVICE;     \
        }                                                                                \
    } while (0)

// FM-index as the kernels see it (passed by value in the kernarg segment).
// cp points at the reference's CP_OCC array unchanged: block b occupies four
// 16-byte pieces  [cnt0 cnt1] [cnt2 cnt3] [hot0 hot1] [hot2 hot3].
struct DevFmi {
    const uint4 *cp;
    const uint4 *cp2;          // the search kernels' resident form of cp (fmi_seed.hip; BWAMS_CP2), or nullptr
    int32_t tab_kind;          // ... 1: compact, 128 rows per block; 2: interleaved, piece b = count and string of base b
    const int8_t *sa_ms;
    const uint32_t *sa_ls;
    const uint8_t *ref;        // .0123 or nullptr
    // FMA direct-lookup tables (reference layouts, src/FMI_search.h:101-135) or nullptr
    const uint32_t *all_smem;  // 4^all_bp entries x 32 words: last_avail, 10 x {k32, l32, s32}, pad
    const uint4 *last_smem;    // 4^last_bp entries x 16 B: bp | kms<<8 | lms<<16 | sms<<24, kls, lls, sls
    int32_t all_bp, last_bp;
    int64_t count[5];
    int64_t sentinel;
    int64_t ref_seq_len;
};

// EMF table as the probe kernel sees it
struct DevEmf {
    const uint4 *seed_table;     // {flags, location, left, right}
    const uint32_t *loc_table;
    const uint8_t *ref;          // .0123
    uint32_t num_seed_entry, num_loc_entry, seq_len;
    int32_t seed_len;
};

// ERT index as the kernels see it: the reference's two files, resident (src/ertindex.cpp writes them)
struct DevErt {
    const uint64_t *kmer;      // <prefix>.kmer_table: 4^K entries
    const uint8_t *mlt;        // <prefix>.mlt_table, padded by 16 bytes
    const uint8_t *ref;        // .0123, both strands
    int64_t ref_len;           // 2 * l_pac
    int32_t K, X, read_len;    // kmerSize, xmerSize, READ_LEN of the build (src/macro.h:204-206, :66)
    uint64_t *cnt_tab;         // hit counts of the subtrees with 20 hits or more: {node address + 1, hits} pairs, open addressing
    int32_t cnt_bits;          // log2 of the number of pairs
    const uint8_t *fat;        // resident, derived once per index (ert_seed.hip: ert_fat_kernel), or null: 64 B per k-mer = its table entry + the
                               // first 56 bytes of its tree, so that a walk's entry and first records are ONE line
};

// device-side counters of one seed run
struct DevCounters {
    unsigned long long n_ext, n_ext_blocks, n_sa_lookups, n_lf_steps;
    unsigned long long n_smem_total;     // append cursor of the SMEM pool (slots handed out, holes included)
    unsigned long long n_smem_valid;     // real SMEMs written
    unsigned long long valid_after[3];   // n_smem_valid when round 1, 2, 3 ended
    unsigned long long n_after_r1, n_after_r2;
    unsigned long long work_head;        // dynamic work queue cursor (reset per kernel)
    unsigned long long n_work2;          // round-2 work items
    unsigned long long overflow;         // SMEM pool overflow flag / needed size
    unsigned long long bsw_cells;
    unsigned long long bsw_head[4];      // ticket counters of the one-task-per-wave banded-SW kernels
    unsigned long long bsw_cls_cnt[6], bsw_cls_head[6];   // banded SW: tasks per query-length class, ticket counters of the class launches
    unsigned long long ext_after[3], blk_after[3];
    unsigned long long emf_nodes, emf_cmp_bytes;     // EMF probe: entries visited, reference bytes compared   // n_ext / n_ext_blocks when round 1, 2, 3 ended
    unsigned long long chain_overflow;   // chaining: B-tree node region exhausted (never expected)
    unsigned long long chain_longread;   // chaining: reads long enough for mem_flt_chained_seeds to re-score seeds
    unsigned long long n_heavy;          // chaining: reads handed to the wave-per-read filter kernel
    unsigned long long chain_class[10];  // chaining: reads with more seeds than the L, L1, M, M1, S, lane-tier, XL, L2, M2 and XL2 limits
    unsigned long long chain_ticket[10]; // chaining: work cursors of the wave kernels
    unsigned long long heavy_tickets[6]; // chaining: work cursors of chain_heavy_kernel's size classes (five used)
    unsigned long long n_retry;          // extension: tasks queued for the next band width
    unsigned long long n_req;            // extension: seeds requested by the last selection
    unsigned long long sel_heavy, sel_ticket, sel_ticket2, sel_ticket3;   // extension: reads of the selection's wave tier; work cursors: the small class, the big class' two passes
    unsigned long long dedup_heavy, dedup_ticket, dedup_light;   // dedup: reads for the wave tier, its work cursor, reads for the lane tier
    unsigned long long chain_redo, chain_redo_ticket;   // chaining: reads the ordered-array attempt gave up on, work cursor
    unsigned long long pair_heavy, pair_ticket;   // mem_mark_primary_se: reads of the wave tier, its work cursor
    unsigned long long dbg[80];          // diagnostics printed under BWAMS_VERBOSE (chain_heavy_kernel: size histogram, cycles)
    unsigned long long pair_ticket2;     // mem_mark_primary_se: work cursor of the wave tier's small-LDS instance
    unsigned long long dedup_ticket2, dedup_ticket3;    // dedup: work cursors of the wave tier's smaller instances
    unsigned long long ert_kmer, ert_nodes, ert_ref;   // ERT profile kernel: k-mer entries read, tree records decoded, text bytes compared
    unsigned long long work_head3, n_ext3, n_blk3, n_smem3;   // SMEM round 3 (it may run beside round 2): its own cursor and counts, folded in by mark_kernel(3)
    unsigned long long n_rest;           // extension: slots behind the requests of the last selection (an upper bound of the undecided seeds)
    unsigned long long ert_ticket;       // ERT walk: work cursor of ert_profile_kernel (groups of 64 read positions)
    unsigned long long bwd_items, bwd_entries, bwd_ticket;   // SMEM search: backward phases handed to the wave kernel, their list entries, its work cursor
    unsigned long long f_items, f_ticket, f_overflow, f_items_r[2];        // SMEM search split by role: item slots handed out, the backward kernel's cursor, pivots without room (the caller re-runs unsplit)
    unsigned long long bwd_items_s, bwd_ticket_s;            // ... the short lists (smem_bwd_group_kernel): slots handed out, work cursor
    unsigned long long pair_full, pair_fail;   // mate rescue: reads redone with every orientation planned; reads the second pass could not finish (never expected)
};

struct ChainState;                       // chain / extension buffers of a batch (api_chain.hip)
void chain_state_free(ChainState *s);

// banded-SW parameters in kernel form (max_sc = max entry of mat)
struct SwParams {
    int o_del, e_del, o_ins, e_ins, zdrop, end_bonus, max_sc;
    int8_t mat[25];
};

struct Round2Work {
    uint32_t rid;
    int32_t x;
    int32_t min_intv;
};

// A backward phase handed from the lane-per-read SMEM search to the wave-per-pivot kernel (fmi_seed.hip): the pivot and the
// interval list its forward phase left, `num_prev` packed 16-byte entries from `off` of the entry buffer (0 = slot not used).
struct BwdItem {
    uint32_t rid;
    int32_t x;
    int32_t min_intv;
    int32_t num_prev;
    int64_t off;
};

int launch_bsw(bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, const uint8_t *qer, int w, const SwParams &prm, int qmax,
               DevCounters *ctr, int cu_count, hipStream_t st, int32_t *list, hipStream_t *aux = nullptr, hipEvent_t fork = nullptr,
               hipEvent_t *join = nullptr, const int64_t *src = nullptr, int dir = 1);
size_t bsw_list_bytes(int64_t n_tasks);          // scratch `list` of launch_bsw
size_t bsw_lds_bytes(int qmax);
void launch_emf_probe(const DevEmf &t, const uint8_t *enc, const int64_t *cum, int64_t nseq, uint32_t *out,
                      uint8_t *code, uint8_t *skip, DevCounters *ctr, hipStream_t st);
constexpr int kKswMaxTarget = 20000;     // longest local-SW target: its row-maxima list must fit the LDS of a 4-wave block
int launch_ksw(const bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, const uint8_t *qer, const SwParams &prm,
                int pmax, int tmax, void *out, DevCounters *ctr, int cu_count, hipStream_t st);

// read input helpers shared with the outer boundary (fastq.hip)
struct SegMove { const char *src; char *dst; int64_t len; };     // one contiguous piece of device memory to copy
int segment_copy(const std::vector<SegMove> &moves, hipStream_t st);
}  // namespace bwams
struct bwams_fastq;
namespace bwams {
int fastq_classify(bwams_fastq *f, std::vector<uint8_t> *which);                       // bseq_classify: 1 = an end of a pair
int fastq_subset(bwams_fastq *f, const std::vector<int64_t> &ids, bwams_fastq **out);  // those reads as a chunk of their own
int fastq_interleave(bwams_fastq *f1, bwams_fastq *f2, bwams_fastq **out);             // read k of f1, read k of f2, ...
}  // namespace bwams

struct bwams_index {
    int device = 0;
    bwams::DevFmi fmi{};
    bool owns = true;
    int64_t bytes = 0;
    int64_t n_blk = 0, n_sa = 0;
    void *d_cp = nullptr, *d_ms = nullptr, *d_ls = nullptr, *d_ref = nullptr;
    int cp2_kind = 0;
    void *d_cp2 = nullptr;                       // compact search table derived from d_cp on first use (BWAMS_CP2=1; always owned)
    void *d_all = nullptr, *d_last = nullptr;    // FMA tables (owned)
    void *d_contigs = nullptr;                   // bwams_contig_t[n_seqs] (owned); null = one sequence [0, l_pac)
    int32_t n_seqs = 0;
    void *d_ctg_annos = nullptr, *d_ctg_anno_off = nullptr;   // bntann1_t.anno for MEM_F_REF_HDR (bwams_index_set_contig_annos)
    void *d_ctg_names = nullptr, *d_ctg_off = nullptr;   // sequence names for the SAM text (bwams_index_set_contig_names)
};

namespace bwams {
int bsw_list_ensure(bwams_batch *b, int64_t n_tasks);   // grows b->d_bsw_list (synchronises the stream when it must reallocate)
}

struct bwams_ert {
    bwams_index *idx = nullptr;
    bwams::DevErt t{};
    void *d_kmer = nullptr, *d_mlt = nullptr, *d_cnt = nullptr, *d_fat = nullptr;
    int64_t bytes = 0, mlt_bytes = 0, n_big = 0;
    float build_ms[3] = {0, 0, 0};       // bwams_ert_build: sizes, scan + allocation, bytes
};

struct bwams_emf {
    bwams_index *idx = nullptr;
    bwams::DevEmf t{};
    bool owns = true;
    void *d_seeds = nullptr, *d_loc = nullptr;
    int64_t bytes = 0;
    int64_t n_used = 0, n_key = 0, n_other = 0, build_ms = 0;     // bwams_emf_build: distinct L-mers, buckets used, nodes outside their bucket
};

namespace bwams {
int emf_build_device(bwams_emf *e, const uint8_t *ref, int64_t l_pac, int seed_len, double slack, int cu_count, int verbose,
                     int64_t stats[4]);
}

struct bwams_batch {
    bwams_index *idx = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t seed_aux = nullptr;      // SMEM round 3 runs beside round 2
    hipEvent_t seed_fork = nullptr, seed_join = nullptr;
    int64_t max_reads = 0, max_bases = 0, max_smem = 0, max_sa = 0;
    int64_t pool_cap = 0;                // max_smem + chunk slack
    int cu_count = 0;

    // reads
    uint8_t *d_enc = nullptr;
    int64_t *d_cum = nullptr;
    uint8_t *d_skip = nullptr;
    bool has_skip = false;
    int64_t nseq = 0, nbases = 0;
    int max_read_len = 0;
    uint32_t *d_packed = nullptr;        // packed reads (2-bit codes + N mask)
    int64_t packed_cap = 0;              // words
    int read_w = 0, read_cw = 0;

    // seeding buffers
    bwams_smem_t *d_pool = nullptr;      // unsorted SMEM pool (append order)
    bwams_smem_t *d_pool3 = nullptr;     // round 3's own pool while it runs from the start of the stage (appended to d_pool behind round 2)
    int64_t pool3_cap = 0;
    bwams::DevCounters *d_ctr3 = nullptr;   // ... and its own counters
    bwams_smem_t *d_sorted = nullptr;    // (rid, m, n) order
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_vals2 = nullptr;
    bwams::Round2Work *d_work2 = nullptr;
    int64_t *d_sa_off = nullptr;         // max_smem + 1
    int64_t *d_sa_cnt = nullptr;         // max_smem + 1
    int64_t *d_sa_coord = nullptr;
    void *d_tmp = nullptr;               // rocPRIM temporary storage
    size_t tmp_bytes = 0;
    bwams::DevCounters *d_ctr = nullptr;
    bwams::DevCounters *h_ctr = nullptr;  // pinned host mirror
    // per-lane scratch of the SMEM search (previous-interval lists)
    uint4 *d_prev = nullptr;
    bwams::BwdItem *d_f_items = nullptr;        // SMEM search split by role: pivots between the forward and the backward kernel ...
    uint4 *d_fl_ent = nullptr;                  // ... and their interval lists
    int64_t f_items_cap = 0, fl_cap = 0;
    int split_parity = 0, split_dbl = 1;                       // lab (BWAMS_SEED_SPLIT=2): which half of the doubled buffers the forward kernel writes
    int64_t f_items_prev[2][2] = {{-1, -1}, {-1, -1}};   // ... items per round the previous runs left in each half
    bool seed_split_failed = false;             // a chunk whose pivots did not fit: this batch searches with the one-kernel form from then on
    bwams::BwdItem *d_bwd_items = nullptr;      // SMEM search: backward phases with long interval lists (wave-per-pivot kernel)
    uint4 *d_bwd_ent = nullptr;
    int64_t bwd_items_cap = 0, bwd_ent_cap = 0;
    int64_t prev_threads = 0;
    int prev_cap = 0;

    int64_t n_smem = 0, n_sa = 0;
    int64_t n_pool_slots = 0;            // SMEM pool slots the last seeding pass handed out (holes included)
    bool seed_done = false, with_sa = false;
    bwams_ert *seed_ert = nullptr;       // the last seed run went over this ERT (nullptr: FM-index)
    uint8_t *d_ert_prof = nullptr;       // ERT seeding: match-length planes, (M + 1) x nbases bytes
    int64_t cap_ert_prof = 0;
    uint64_t *d_ert_stk = nullptr;       // ERT seeding: stacks of the leaf walks (ert_walk_threads x frames words)
    int ert_stk_frames = 0;
    uint32_t *d_ert_redo = nullptr;      // ERT seeding: seeds whose hits the rank descent could not list (bit per seed)
    int64_t cap_ert_redo = 0;            // seeds it is sized for
    bwams_seed_opt_t last_seed_opt{};    // of the last bwams_seed_run (a grown SA buffer re-runs the lookup)

    // extension buffers
    bwams_seqpair_t *d_pairs = nullptr;
    uint8_t *d_ref = nullptr, *d_qer = nullptr;
    int64_t cap_pairs = 0, cap_ref = 0, cap_qer = 0, n_pairs = 0;
    int max_qlen = 0, max_tlen = 0;
    uint32_t *d_emf_out = nullptr;
    uint8_t *d_emf_code = nullptr;
    int64_t cap_emf = 0;
    void *d_ksw_out = nullptr;
    int64_t cap_ksw = 0;
    int32_t *d_bsw_list = nullptr;       // task lists of the banded-SW length classes (launch_bsw)
    int64_t cap_bsw_list = 0;            // tasks it is sized for

    bwams::ChainState *chain = nullptr;

    hipEvent_t ev[16] = {};
    hipEvent_t ev_emf[2] = {};
    unsigned long long emf_nodes = 0, emf_cmp_bytes = 0;
    bwams_stats_t stats{};
};
