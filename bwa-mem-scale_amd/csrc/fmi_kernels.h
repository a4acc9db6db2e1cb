// fmi_kernels.h — launch wrappers of the FM-index kernels (fmi_seed.hip, fmi_sal.hip).
#pragma once
#include "common.h"

namespace bwams {

struct SeedLaunch {
    DevFmi fmi;
    const uint8_t *enc;
    const int64_t *cum;
    const uint8_t *skip;          // may be nullptr
    const uint32_t *packed;       // reads packed by launch_pack_reads: read_w words per read
    int read_w, read_cw;          // words per read (multiple of 4) / code words
    int reads_in_lds;             // 1: each lane keeps its current read in LDS
    int debug;                    // diagnostic ablations (env BWAMS_DEBUG); 0 in production
    int64_t nseq;
    int min_seed_len;
    bwams_smem_t *pool;
    int64_t pool_cap;
    DevCounters *ctr;
    uint4 *prev;                  // per-lane previous-interval lists: prev_cap entries x 16 B, lane-contiguous
    int prev_cap;                 // entries per lane
    int64_t prev_threads;         // lanes the scratch was sized for
    // backward phases whose interval list has at least bwd_min_list entries go to smem_bwd_wave_kernel (0 = never)
    // rounds 1 and 2 as a forward kernel and a backward kernel (smem_fwd_kernel / smem_bwdl_kernel): the pivots between them and their lists
    BwdItem *f_items;             // f_items_cap slots, reserved 64 at a time per wave (num_prev = 0: unused)
    int64_t f_items_cap;
    int64_t f_items_fixed;        // >= 0: the backward kernel takes this many items instead of the counter (lab: overlap experiment)
    uint4 *fl_ent;                // the lists: fl_cap 16-byte entries
    int64_t fl_cap;
    int32_t fl_item_stride;       // round 2: entries per work item (max read length + 2)
    BwdItem *bwd_items, *bwd_items_s;   // lists beyond / up to kBwdShortMax entries; bwd_items_cap slots each
    uint4 *bwd_ent;
    int64_t bwd_items_cap, bwd_ent_cap;
    int bwd_min_list;
    int bwd_dry_min_list, bwd_dry_cols, bwd_dry_late_list;   // the same three thresholds once the work queue has run dry
    int bwd_cols, bwd_late_list;  // ... or, later: after bwd_cols columns with bwd_late_list entries still alive
};

// grid sizing shared by batch_create (scratch) and the launches
int seed_block_threads();
int64_t seed_max_threads(int cu_count);
// round 3's records (a pool and counters of its own while it ran from the start of the stage) behind the main pool's, its counts into n_smem3 / n_ext3 / n_blk3
void launch_append_r3(bwams_smem_t *pool, int64_t pool_cap, const bwams_smem_t *pool3, int64_t pool3_cap, DevCounters *ctr, const DevCounters *ctr3, hipStream_t st);
int64_t seed_pool_slack(int cu_count);   // pool slots the launches of one seeding pass can leave unused in partly filled chunks

// enc_qdb bytes -> 2-bit codes + N mask, W words per read
void launch_pack_reads(const uint8_t *enc, const int64_t *cum, int64_t nseq, int W, int cw, uint32_t *packed,
                       hipStream_t st);
// between rounds: snapshot counters (which = 1, 2, 3 after that round; 0 = start) and reset the work cursor
void launch_mark(DevCounters *ctr, int which, hipStream_t st);
// round 1: every pivot of every read (getSMEMsAllPosOneThread)
void launch_smem_round1(const SeedLaunch &a, int cu_count, hipStream_t st);
// round 2: build the work list from round-1 SMEMs, then one pivot per item (getSMEMsOnePosOneThread)
void launch_round2_work(const SeedLaunch &a, Round2Work *work, int64_t work_cap, int split_len,
                        int split_width, int cu_count, hipStream_t st);
void launch_smem_round2(const SeedLaunch &a, const Round2Work *work, int cu_count, hipStream_t st);
// rounds 1 / 2 split by role: forward phases of every read (work == nullptr) or work item, then the backward phase of every pivot
void launch_smem_fwd(const SeedLaunch &a, const Round2Work *work, int cu_count, hipStream_t st);
void launch_smem_bwdl(const SeedLaunch &a, int cu_count, hipStream_t st);
// the backward phases rounds 1 / 2 set aside (lists of bwd_min_list entries and more): one wavefront per pivot, one lane per entry
void launch_smem_bwd_wave(const SeedLaunch &a, int cu_count, hipStream_t st);
// round 3: forward-only seeds (bwtSeedStrategyAllPosOneThread)
void launch_smem_round3(const SeedLaunch &a, int max_intv, int cu_count, hipStream_t st);

// FMA table builders (__build_all_smem_table / __build_last_smem_table): one lane per table entry
void launch_build_fma(const DevFmi &f, int all_bp, uint32_t *all_tab, int last_bp, uint4 *last_tab, hipStream_t st);

// sort keys / gather / SA lookup
void launch_make_keys(const bwams_smem_t *pool, int64_t n, uint64_t *keys, uint32_t *vals, uint32_t hole_key_rid,
                      hipStream_t st);
void launch_gather_sorted(const bwams_smem_t *pool, const uint32_t *order, int64_t n, bwams_smem_t *sorted,
                          int64_t *sa_cnt, int max_occ, hipStream_t st);
void launch_sa_lookup(const DevFmi &f, const bwams_smem_t *sorted, int64_t n_smem, const int64_t *sa_off,
                      int64_t *coord, int64_t coord_cap, int max_occ, DevCounters *ctr, int cu_count,
                      hipStream_t st);

// the search kernels' compact Occ table (CpOcc2, BWAMS_CP2=1): (n_blk + 1) / 2 blocks of 64 B
size_t cp2_bytes(int64_t n_blk, int kind);          // kind 1: compact (128 rows per block); 2: interleaved (piece b = count and string of base b)
void launch_cp2_build(const uint4 *cp, int64_t n_blk, uint4 *cp2, int kind, hipStream_t st);

}  // namespace bwams
