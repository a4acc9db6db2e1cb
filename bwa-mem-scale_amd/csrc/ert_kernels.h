// ert_kernels.h — launch wrappers of the ERT seeding kernels (ert_seed.hip).
#pragma once
#include "common.h"

namespace bwams {

// planes of `prof`: (M + 1) x nbases bytes, zeroed by the caller; [0] = N flag, [m] = L_m
void launch_ert_profile(const DevErt &e, const uint8_t *enc, const int64_t *cum, const uint8_t *skip, int64_t nseq,
                        int64_t nbases, int M, uint8_t *prof, DevCounters *ctr, unsigned long long *part, hipStream_t st);
size_t ert_count_bytes();      // `part`: zeroed once; the launch leaves it zeroed
void launch_ert_select(const uint8_t *prof, const int64_t *cum, const uint8_t *skip, int64_t nseq, int64_t nbases, int M,
                       const bwams_seed_opt_t &opt, bwams_smem_t *pool, int64_t pool_cap, DevCounters *ctr, hipStream_t st);
// the leaf walks keep their stack in `stk`: ert_walk_threads(cu_count) lanes x max_frames words, frame-major
int64_t ert_walk_threads(int cu_count);
void launch_ert_locate(const DevErt &e, const uint8_t *enc, const int64_t *cum, bwams_smem_t *sorted, int64_t n,
                       int64_t *sa_cnt, int max_occ, DevCounters *ctr, uint64_t *stk, int max_frames, int cu_count,
                       hipStream_t st);
void launch_ert_gather(const DevErt &e, bwams_smem_t *sorted, int64_t n, const int64_t *sa_off, int64_t *coord,
                       int64_t coord_cap, int max_occ, DevCounters *ctr, uint64_t *stk, int max_frames, int cu_count,
                       hipStream_t st);

// ert_build.hip: the two tables of `bwa-mem2 index -a ert` from the resident FM-index, into buffers owned by *e
int ert_build_device(bwams_ert *e, const DevFmi &f, int K, int X, int read_len, int hit_threshold, int cu_count, int verbose);

}  // namespace bwams
