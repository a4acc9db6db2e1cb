// ert_chain.hip — ERT mode's way into the chaining stage (/root/reference/src/bwamem.cpp:961-1050, :1193-1203).
// The reference's ERT walk (get_seeds / reseed / last, ertseeding.cpp — not built here, it stays on the host) leaves,
// per read, a list of MEMs (mem_t) and one array of hits; mem_kernel1_core_ert then sorts the MEMs
// (ks_introsort(mem_smem_sort_lt)) and chains them with mem_chain_new, which is mem_chain_seeds' procedure on a
// different representation of the seeds: the positions come from hits[hitbeg + k] (k = 0, step, ..; mapped back to
// the match when the MEM was found by backward search) instead of get_sa_entries, hitcount plays the part of the
// interval size, and the `pos < num_smem - 1` guard does not exist.  So the two kernels below only translate: MEMs
// -> SMEM-like records in sorted order, picked hits -> the coordinate array the chaining kernels read; chaining,
// filtering and everything after are the FM-index path's kernels.
#include "common.h"
#include "chain_kernels.h"
#include "region_sort.h"

namespace bwams {
namespace {

// lane per read: the reference's introsort of the read's MEMs by (start, end) — unstable, and equal MEMs occur
// (reseeding finds a MEM again), so operation by operation — then one record per MEM in that order
__global__ void ert_sort_kernel(ErtArgs A) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) A.cnt[A.n_mems] = 0;
    if (r >= A.nseq) return;
    const int64_t m0 = A.mem_off[r];
    const int n = (int)(A.mem_off[r + 1] - m0);
    SortRec *srt = reinterpret_cast<SortRec *>(A.srt) + m0;
    for (int i = 0; i < n; ++i) {
        const bwams_ert_mem_t *p = &A.mems[m0 + i];
        SortRec x; x.k = p->start; x.s = p->end; x.q = 0; x.idx = i; x.pad_ = 0;
        srt[i] = x;
    }
    sort_records(srt, n, 5);
    for (int i = 0; i < n; ++i) {
        const int64_t g = m0 + srt[i].idx;
        const bwams_ert_mem_t *p = &A.mems[g];
        bwams_smem_t o;
        o.rid = (uint32_t)r; o.m = (uint32_t)p->start; o.n = (uint32_t)(p->end - 1); o.pad_ = 0;
        o.k = g;                                             // which MEM this record stands for (ert_pick_kernel)
        o.l = 0; o.s = p->hitcount;
        A.smem_out[m0 + i] = o;
        const int step = p->hitcount > A.max_occ ? p->hitcount / A.max_occ : 1;
        int64_t picks = p->hitcount > 0 ? ((int64_t)p->hitcount + step - 1) / step : 0;    // k = 0, step, .. < hitcount
        A.cnt[m0 + i] = picks < A.max_occ ? picks : A.max_occ;
    }
}

// lane per record: the positions mem_chain_new would visit
__global__ void ert_pick_kernel(ErtArgs A, const int64_t *sa_off, int64_t *coord) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_mems) return;
    const bwams_smem_t rec = A.smem_out[i];
    const bwams_ert_mem_t p = A.mems[rec.k];
    const uint64_t *hh = A.hits + A.hit_off[rec.rid] + p.hitbeg;
    const int slen = p.end - p.start;
    const int step = p.hitcount > A.max_occ ? p.hitcount / A.max_occ : 1;
    int64_t *out = coord + sa_off[i];
    int count = 0;
    for (int64_t k = 0; k < p.hitcount && count < A.max_occ; k += step, ++count) {
        const int64_t h = (int64_t)hh[k];
        out[count] = (p.forward || p.fetch_leaves) ? h : (A.l_pac << 1) - (h + slen - p.end_correction);
    }
}

}  // namespace

void launch_ert_sort(const ErtArgs &A, hipStream_t st) {
    ert_sort_kernel<<<(unsigned)((A.nseq + 1 + 63) / 64), 64, 0, st>>>(A);
}
void launch_ert_pick(const ErtArgs &A, const int64_t *sa_off, int64_t *coord, hipStream_t st) {
    if (A.n_mems > 0) ert_pick_kernel<<<(unsigned)((A.n_mems + 255) / 256), 256, 0, st>>>(A, sa_off, coord);
}

}  // namespace bwams
