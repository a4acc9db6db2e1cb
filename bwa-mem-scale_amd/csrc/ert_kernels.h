// ert_kernels.h — launch wrappers of the ERT seeding kernels (ert_seed.hip).
#pragma once
#include "common.h"

namespace bwams {

// hit-count table of the big subtrees (see ert_seed.hip): open addressing, 32 probes
static __device__ __forceinline__ int64_t cnt_lookup(const DevErt &e, int64_t node) {
    if (!e.cnt_tab) return -1;
    const uint64_t key = (uint64_t)node + 1, mask = (1ull << e.cnt_bits) - 1;
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64 - e.cnt_bits);
    for (int p = 0; p < 32; ++p) {
        const uint64_t k = __atomic_load_n(&e.cnt_tab[2 * h], __ATOMIC_RELAXED);
        if (k == key) {
            const uint64_t v = __atomic_load_n(&e.cnt_tab[2 * h + 1], __ATOMIC_RELAXED);
            return v ? (int64_t)v : -1;
        }
        if (k == 0) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}
static __device__ __forceinline__ void cnt_insert(const DevErt &e, int64_t node, int64_t val) {
    if (!e.cnt_tab) return;
    const uint64_t key = (uint64_t)node + 1, mask = (1ull << e.cnt_bits) - 1;
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64 - e.cnt_bits);
    for (int p = 0; p < 32; ++p) {
        const unsigned long long old = atomicCAS((unsigned long long *)&e.cnt_tab[2 * h], 0ull, (unsigned long long)key);
        if (old == 0ull || old == key) {
            __atomic_store_n(&e.cnt_tab[2 * h + 1], (uint64_t)val, __ATOMIC_RELAXED);
            return;
        }
        h = (h + 1) & mask;
    }
}



// the 64-byte-per-k-mer table of entry + tree head (DevErt::fat) from the two tables; fat: 64 << (2 K) bytes
void launch_ert_fat(const DevErt &e, int64_t mlt_bytes, uint8_t *fat, hipStream_t st);
// prof: ert_prof_bytes(nbases) bytes, a 24-byte record per read position ([0] = N flag, [m] = L_m, M <= 23), written whole by the walk
size_t ert_prof_bytes(int64_t nbases);
void launch_ert_profile(const DevErt &e, const uint8_t *enc, const int64_t *cum, const uint8_t *skip, int64_t nseq,
                        int64_t nbases, int M, uint8_t *prof, DevCounters *ctr, unsigned long long *part, int cu_count,
                        hipStream_t st);
size_t ert_count_bytes();      // `part`: zeroed once; the launch leaves it zeroed
void launch_ert_select(const uint8_t *prof, const int64_t *cum, const uint8_t *skip, int64_t nseq, int64_t nbases, int M,
                       const bwams_seed_opt_t &opt, bwams_smem_t *pool, int64_t pool_cap, DevCounters *ctr, int cu_count,
                       hipStream_t st);
// the leaf walks keep their stack in `stk`: ert_walk_threads(cu_count) lanes x max_frames words, frame-major
int64_t ert_walk_threads(int cu_count);
void launch_ert_locate(const DevErt &e, const uint8_t *enc, const int64_t *cum, bwams_smem_t *sorted, int64_t n,
                       int64_t *sa_cnt, int max_occ, DevCounters *ctr, uint64_t *stk, int max_frames, int cu_count,
                       hipStream_t st);
// hits by rank descent (lane per sampled hit); seeds whose counts the table lacks are redone by the serial leaf walk.
// redo: (n + 31) / 32 words; n_coord_hint: an upper bound of the hits to list (sizes the grid)
void launch_ert_gather(const DevErt &e, bwams_smem_t *sorted, int64_t n, const int64_t *sa_off, int64_t *coord,
                       int64_t coord_cap, int max_occ, DevCounters *ctr, uint64_t *stk, int max_frames, uint32_t *redo,
                       int64_t n_coord_hint, int cu_count, hipStream_t st);
void launch_ert_clear(bwams_smem_t *sorted, int64_t n, hipStream_t st);

// ert_build.hip: the two tables of `bwa-mem2 index -a ert` from the resident FM-index, into buffers owned by *e
int ert_build_device(bwams_ert *e, const DevFmi &f, int K, int X, int read_len, int hit_threshold, int cu_count, int verbose);

}  // namespace bwams
