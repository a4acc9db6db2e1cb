// fmi_sal.hip — suffix-array lookup of SMEM occurrences for gfx950.
//
// Reference semantics: get_sa_entries_prefetch + call_one_step
// (/root/reference/src/FMI_search.cpp:2261-2379, :2206-2259) as driven by
// mem_chain_seeds (/root/reference/src/bwamem.cpp:861-873): SMEM i contributes
// the BWT rows k, k+step, ... (step = s/max_occ when s > max_occ, at most max_occ
// rows); each row is walked with LF-mapping until a multiple of 8, where the
// sampled SA (int8 high byte + uint32 low word) is read and the walk length added.
// A walk that meets the sentinel row yields 0 (reference quirk, :2234-2237).
//
// Mapping: every lookup is an independent chain of at most 7 dependent block
// reads, so one LANE per lookup; the owning SMEM is found by binary search in the
// prefix offsets, which stay L2-resident.
#include "fmi_kernels.h"

namespace bwams {
namespace {

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) {
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__global__ __launch_bounds__(256) void sa_lookup_kernel(DevFmi f, const bwams_smem_t *__restrict__ sm,
                                                        int64_t n_smem, const int64_t *__restrict__ sa_off,
                                                        int64_t *__restrict__ coord, int64_t coord_cap,
                                                        int max_occ, DevCounters *ctr) {
    const int64_t total = sa_off[n_smem] < coord_cap ? sa_off[n_smem] : coord_cap;
    unsigned long long lf = 0;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        // upper_bound(sa_off, g) - 1
        int64_t lo = 0, hi = n_smem;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (sa_off[mid] <= g) lo = mid; else hi = mid;
        }
        const int64_t k = sm[lo].k, s = sm[lo].s;
        const int64_t step = s > (int64_t)max_occ ? s / max_occ : 1;
        int64_t sp = k + (g - sa_off[lo]) * step;
        int64_t off = 0, val = 0;
        while (true) {
            if ((sp & 7) == 0) {
                val = ((int64_t)f.sa_ms[sp >> 3] << 32) + (int64_t)f.sa_ls[sp >> 3] + off;
                break;
            }
            const uint4 *p = f.cp + ((sp >> 6) << 2);
            // the whole block in one round trip: the count of the row's base would otherwise be a second, dependent load
            const uint4 c01 = p[0], c23 = p[1], h01 = p[2], h23 = p[3];
            const int sh = 63 - (int)(sp & 63);
            const uint64_t h0 = mk64(h01.x, h01.y), h1 = mk64(h01.z, h01.w);
            const uint64_t h2 = mk64(h23.x, h23.y), h3 = mk64(h23.z, h23.w);
            int b = 4;
            uint64_t hb = 0;
            if ((h0 >> sh) & 1) { b = 0; hb = h0; }
            else if ((h1 >> sh) & 1) { b = 1; hb = h1; }
            else if ((h2 >> sh) & 1) { b = 2; hb = h2; }
            else if ((h3 >> sh) & 1) { b = 3; hb = h3; }
            if (b == 4) { val = 0; break; }
            const int y = (int)(sp & 63);
            const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
            const int64_t cnt = (int64_t)(b == 0 ? mk64(c01.x, c01.y) : b == 1 ? mk64(c01.z, c01.w) : b == 2 ? mk64(c23.x, c23.y) : mk64(c23.z, c23.w));
            sp = (b == 0 ? f.count[0] : b == 1 ? f.count[1] : b == 2 ? f.count[2] : f.count[3]) + cnt +
                 __popcll(hb & mask);
            off++;
            lf++;
        }
        coord[g] = val;
    }
    for (int o = 32; o > 0; o >>= 1)
        lf += mk64(__shfl_down((uint32_t)lf, o), __shfl_down((uint32_t)(lf >> 32), o));
    if ((threadIdx.x & 63) == 0 && lf) atomicAdd(&ctr->n_lf_steps, lf);
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_sa_lookups = (unsigned long long)sa_off[n_smem];
}

}  // namespace

void launch_sa_lookup(const DevFmi &f, const bwams_smem_t *sorted, int64_t n_smem, const int64_t *sa_off,
                      int64_t *coord, int64_t coord_cap, int max_occ, DevCounters *ctr, int cu_count,
                      hipStream_t st) {
    if (n_smem <= 0) return;
    sa_lookup_kernel<<<cu_count * 8, 256, 0, st>>>(f, sorted, n_smem, sa_off, coord, coord_cap, max_occ, ctr);
}

}  // namespace bwams
