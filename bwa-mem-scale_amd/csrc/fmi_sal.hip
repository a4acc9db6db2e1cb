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
    // The owner of position g is the last SMEM whose first position is <= g: a binary search over 9 M offsets, i.e. 23
    // DEPENDENT loads per lookup, more than the LF walk behind it.  A wave takes a contiguous range of positions instead:
    // one search for its first position, and from there the owners only move forward — 64 consecutive positions belong
    // to at most 64 consecutive SMEMs (every SMEM has at least one), whose offsets are one coalesced load into LDS and a
    // six-step search there.
    __shared__ int64_t win_s[4][65];
    int64_t *const win = win_s[threadIdx.x >> 6];
    const int lane = (int)(threadIdx.x & 63);
    const int64_t n_waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t per = ((total + n_waves - 1) / n_waves + 63) & ~(int64_t)63;
    const int64_t start = wave * per, end = start + per < total ? start + per : total;
    int64_t lo0 = 0;
    if (start < end) {
        int64_t a = 0, b = n_smem;
        while (b - a > 1) {
            const int64_t mid = (a + b) >> 1;
            if (sa_off[mid] <= start) a = mid; else b = mid;
        }
        lo0 = a;
    }
    for (int64_t g0 = start; g0 < end; g0 += 64) {
        const int64_t g = g0 + lane;
        {
            const int64_t c = lo0 + 1 + lane;
            win[lane] = c <= n_smem ? sa_off[c] : INT64_MAX;
            if (lane == 0) win[64] = INT64_MAX;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int t = 0;                                            // first candidate whose offset is beyond g
        for (int stp = 32; stp > 0; stp >>= 1)
            if (win[t + stp - 1] <= g) t += stp;
        if (t == 63 && win[63] <= g) t = 64;
        int64_t lo = lo0 + t;
        if (t == 64 && g < end) {                             // not within 64 SMEMs (never, while every SMEM has a position): search
            int64_t a = lo, b = n_smem;
            while (b - a > 1) {
                const int64_t mid = (a + b) >> 1;
                if (sa_off[mid] <= g) a = mid; else b = mid;
            }
            lo = a;
        }
        {   // the next trip starts from the owner of this trip's last position
            const int last = (int)((end - g0 < 64 ? end - g0 : 64) - 1);
            const uint32_t l_lo = (uint32_t)lo, l_hi = (uint32_t)((uint64_t)lo >> 32);
            lo0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)l_hi, last) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)l_lo, last));
        }
        __builtin_amdgcn_wave_barrier();
        if (g >= end) continue;
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
