// emf_build.hip — construction of the exact-match filter (EMF) table on the GPU.
//
// The table format is the reference's `<prefix>.perfect.<L>` (/root/reference/src/perfect.h:188-213, :772-822): a hash
// table of every L-mer of the forward strand in canonical orientation (perfect.h:362-368: forward if forward <= reverse
// complement on the first half), bucket = fmix64(XOR of the 2-bit packed words) % num_seed_entry (:541-707), one binary
// search tree per bucket ordered by the canonical L-mer (:273-360), the tree's root in the bucket's own slot, its other
// nodes in free slots and flagged (bit 1), L-mers that occur more than once with their further locations in loc_table
// (:170-186).  The reference builds it with `bwa-mem2 perfect-index` (perfect_index.cpp, host, ~20 min for the human
// genome, README.md:30-34) and places collision nodes by an order-dependent probe; every table that satisfies the
// format's invariants is probed identically (emf_probe.hip, emf_regs.hip), and this builder keeps each bucket's tree as
// a chain in ascending order (a bucket holds 0.9 L-mers on average).
//
// Pipeline, sized for 3.2 G windows (GRCh38):
//   1. emf_key_kernel: one lane per window: canonical orientation, hash; sort key = bucket << 32 | 31 high hash bits | orientation.
//   2. one radix sort of (key, position) pairs: a bucket's windows become a run, equal L-mers neighbours inside it.
//   3. emf_bucket_kernel<1>: the lane at the head of each run marks its bucket used and counts what pass 2 allocates.
//   4. the free slots (buckets without a run), listed by a stream compaction.
//   5. emf_bucket_kernel<2>: per run: distinct L-mers (verified base by base: hash words can collide), ascending order,
//      root into the bucket's slot, the others into free slots, multi-location lists into loc_table.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace bwams {
namespace {

constexpr uint32_t kNoEntry = 0xffffffffu;
constexpr int kMaxU = 48;                 // distinct L-mers one bucket may hold (Poisson mean 0.9: never reached)

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ uint64_t ld8(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
// eight base codes (one per byte, first base in the low byte) -> 16 bits, first base in the TOP bits
__device__ __forceinline__ uint64_t pack8_msb(uint64_t v) {
    v = __builtin_bswap64(v) & 0x0303030303030303ull;
    v = (v | (v >> 6)) & 0x000F000F000F000Full;
    v = (v | (v >> 12)) & 0x000000FF000000FFull;
    return (v | (v >> 24)) & 0xFFFFull;
}

struct EmfBuild {
    const uint8_t *ref;        // .0123, forward strand first (padded)
    int64_t l_pac, n_win;      // windows = l_pac - L + 1
    int32_t L;
    uint64_t n_entry;
};

// bases [8t, 8t + 8) of window p in canonical orientation, one per byte, first base lowest; bytes beyond the window are
// not defined (the callers mask).  Reverse complement of the window: base i = 3 - ref[p + L - 1 - i].
__device__ __forceinline__ uint64_t canon_chunk(const EmfBuild &B, int64_t p, bool fw, int t) {
    if (fw) return ld8(B.ref + p + 8 * t);
    const int64_t at = p + B.L - 8 - 8 * t;              // the eight forward bases that chunk t of the reverse complement mirrors
    if (at >= 0) return __builtin_bswap64(0x0303030303030303ull - (ld8(B.ref + at) & 0x0303030303030303ull));
    uint64_t v = 0;                                       // the last, partial chunk of a window at the very start of the text
    for (int i = 0; i < 8; ++i) {
        const int64_t q = p + B.L - 1 - (8 * t + i);
        if (q >= p) v |= (uint64_t)(3 - (B.ref[q] & 3)) << (8 * i);
    }
    return v;
}
// canonical strand (perfect.h:362-368): forward unless the reverse complement is smaller on the first (L + 1) / 2 bases
__device__ __forceinline__ bool fw_less(const EmfBuild &B, int64_t p) {
    const int half = (B.L + 1) / 2;
    for (int t = 0; 8 * t < half; ++t) {
        const uint64_t a = ld8(B.ref + p + 8 * t) & 0x0303030303030303ull, b = canon_chunk(B, p, false, t);
        uint64_t x = a ^ b;
        const int n = half - 8 * t;
        if (n < 8) x &= (1ull << (8 * n)) - 1;
        if (x) {
            const int sh = __builtin_ctzll(x) & ~7;
            return ((a >> sh) & 0xff) < ((b >> sh) & 0xff);
        }
    }
    return true;
}
// lexicographic compare of the canonical L-mers of two windows: -1, 0, 1
__device__ __forceinline__ int canon_cmp(const EmfBuild &B, int64_t pa, bool fa, int64_t pb, bool fb) {
    for (int t = 0; 8 * t < B.L; ++t) {
        uint64_t x = canon_chunk(B, pa, fa, t), y = canon_chunk(B, pb, fb, t);
        const int n = B.L - 8 * t;
        if (n < 8) { const uint64_t m = (1ull << (8 * n)) - 1; x &= m; y &= m; }
        const uint64_t d = x ^ y;
        if (d) {
            const int sh = __builtin_ctzll(d) & ~7;
            return ((x >> sh) & 0xff) < ((y >> sh) & 0xff) ? -1 : 1;
        }
    }
    return 0;
}

__global__ __launch_bounds__(256) void emf_key_kernel(EmfBuild B, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < B.n_win; p += (int64_t)gridDim.x * 256) {
        const bool fw = fw_less(B, p);
        uint64_t h = 0, w = 0;
        int in_w = 0;                                      // bases in w
        for (int t = 0; 8 * t < B.L; ++t) {
            const uint64_t c = pack8_msb(canon_chunk(B, p, fw, t));
            const int n = B.L - 8 * t < 8 ? B.L - 8 * t : 8;
            w = (w << (2 * n)) | (c >> (2 * (8 - n)));
            in_w += n;
            if (in_w == 32) { h ^= w; w = 0; in_w = 0; }
        }
        if (in_w) h ^= w;
        const uint64_t hf = fmix64(h);
        keys[p] = ((hf % B.n_entry) << 32) | ((hf >> 33) << 1) | (fw ? 1ull : 0ull);     // low bit: this window's orientation
        vals[p] = (uint32_t)p;
    }
}

struct BucketArgs {
    EmfBuild B;
    const uint64_t *keys;      // sorted
    const uint32_t *pos;       // sorted along
    uint32_t *used;            // bit per bucket
    const uint32_t *free_list; // pass 2: the buckets without a run, ascending
    uint4 *seeds;              // pass 2
    uint32_t *loc;             // pass 2
    unsigned long long *ctr;   // [0] nodes outside their bucket, [1] loc_table words (starts at 1), [2] distinct L-mers,
                               // [3] buckets used, [4] errors (a list or table region too small), [5] free-slot cursor,
                               // [6] runs handed to the big-bucket kernels
    unsigned long long free_cap, loc_cap;
    ulonglong2 *big;           // pass 1: runs with more than kMaxU distinct L-mers {first sorted window, windows}, ctr[6] of them
    unsigned long long big_cap;
};

// lane = one sorted window; the lane at the head of a bucket's run does the bucket.  Counters and free-slot ranges are
// handed out once per wavefront (a per-bucket atomic on one address would be two billion of them).
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1)
        v += ((unsigned long long)(uint32_t)__shfl_down((int)(v >> 32), o) << 32 | (uint32_t)__shfl_down((int)v, o));
    return v;
}
template <int PASS>
__global__ __launch_bounds__(256) void emf_bucket_kernel(BucketArgs A) {
    const EmfBuild &B = A.B;
    const int lane = threadIdx.x & 63;
    const int64_t n_round = (B.n_win + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * 256) {
        uint32_t key = 0;
        bool head = false;
        if (i < B.n_win) {
            key = (uint32_t)(A.keys[i] >> 32);
            head = i == 0 || (uint32_t)(A.keys[i - 1] >> 32) != key;
        }
        // distinct L-mers of the run: first position, orientation, members in either orientation
        // per L-mer: a witness window (u_pos, its orientation in u_fw) for the comparisons; per orientation the smallest
        // position and the number of windows.  The entry's location is the smallest position of all (as the reference's
        // builder, which meets the windows in text order, stores the first one).
        uint32_t u_pos[kMaxU], u_hi[kMaxU], u_min[kMaxU][2], u_cnt[kMaxU][2];
        uint64_t u_fw = 0;
        int c = 0;
        int64_t j = i;
        bool too_many = false;
        if (head) {
            j = i + 1;
            while (j < B.n_win && (uint32_t)(A.keys[j] >> 32) == key) ++j;
            for (int64_t e = i; e < j; ++e) {
                const uint32_t hi = (uint32_t)A.keys[e];
                const int64_t p = A.pos[e];
                const bool f = hi & 1u;
                int hit = -1;
                for (int k = 0; k < c && hit < 0; ++k)
                    if (((u_hi[k] ^ hi) >> 1) == 0 && canon_cmp(B, u_pos[k], (u_fw >> k) & 1, p, f) == 0) hit = k;
                if (hit >= 0) {
                    u_cnt[hit][f]++;
                    if ((uint32_t)p < u_min[hit][f]) u_min[hit][f] = (uint32_t)p;
                } else if (c < kMaxU) {
                    u_pos[c] = (uint32_t)p; u_hi[c] = hi;
                    u_cnt[c][0] = u_cnt[c][1] = 0; u_min[c][0] = u_min[c][1] = kNoEntry;
                    u_cnt[c][f] = 1; u_min[c][f] = (uint32_t)p;
                    if (f) u_fw |= 1ull << c;
                    ++c;
                } else too_many = true;
            }
        }
        if (too_many) {                                    // emf_big_sort_kernel / emf_big_emit_kernel take this run
            if (PASS == 1) {
                const unsigned long long t = atomicAdd(&A.ctr[6], 1ull);
                if (t < A.big_cap) A.big[t] = make_ulonglong2((unsigned long long)i, (unsigned long long)(j - i));
            }
            head = false; c = 0;
        }
        if (PASS == 1) {
            unsigned long long words = 0;
            if (head) {
                atomicOr(&A.used[key >> 5], 1u << (key & 31));
                for (int k = 0; k < c; ++k) {
                    const int f0 = u_min[k][1] < u_min[k][0] ? 1 : 0;
                    const uint32_t same = u_cnt[k][f0] - 1, other = u_cnt[k][1 - f0];
                    if (same + other) words += (same < 256 && other < 256 ? 1 : 3) + same + other;
                }
            }
            const unsigned long long s0 = wave_sum(head && c > 1 ? (unsigned long long)(c - 1) : 0ull), s1 = wave_sum(words),
                                     s2 = wave_sum((unsigned long long)c), s3 = wave_sum(head ? 1ull : 0ull);
            if (lane == 0) {
                if (s0) atomicAdd(&A.ctr[0], s0);
                if (s1) atomicAdd(&A.ctr[1], s1);
                if (s2) atomicAdd(&A.ctr[2], s2);
                if (s3) atomicAdd(&A.ctr[3], s3);
            }
            continue;
        }
        // ---- pass 2: free-slot ranges per wavefront
        const unsigned int need = head && c > 1 ? (unsigned int)(c - 1) : 0u;
        unsigned int incl = need;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int v = (unsigned int)__shfl_up((int)incl, o);
            if (lane >= o) incl += v;
        }
        const unsigned int total = (unsigned int)__shfl((int)incl, 63);
        unsigned long long wbase = 0;
        if (total) {
            if (lane == 0) wbase = atomicAdd(&A.ctr[5], (unsigned long long)total);
            wbase = ((unsigned long long)(uint32_t)__shfl((int)(wbase >> 32), 0) << 32) | (uint32_t)__shfl((int)wbase, 0);
        }
        if (!head) continue;
        const unsigned long long fbase = wbase + incl - need;
        if (need && fbase + need > A.free_cap) { atomicAdd(&A.ctr[4], 1ull); continue; }
        // ascending order (insertion sort over a handful of L-mers)
        int ord[kMaxU];
        for (int k = 0; k < c; ++k) {
            int at = k;
            while (at > 0 && canon_cmp(B, u_pos[ord[at - 1]], (u_fw >> ord[at - 1]) & 1, u_pos[k], (u_fw >> k) & 1) > 0) { ord[at] = ord[at - 1]; --at; }
            ord[at] = k;
        }
        uint32_t slot_of_next = kNoEntry;
        for (int r = c - 1; r >= 0; --r) {                 // from the tail of the chain: a node's `right` is the next larger L-mer
            const int k = ord[r];
            const uint32_t slot = r == 0 ? key : A.free_list[fbase + (unsigned long long)(r - 1)];
            const int f0 = u_min[k][1] < u_min[k][0] ? 1 : 0;             // orientation of the first (smallest) location
            const uint32_t rep = u_min[k][f0], same = u_cnt[k][f0] - 1, other = u_cnt[k][1 - f0];
            uint32_t flags = (f0 ? 1u : 0u) | (r == 0 ? 0u : 2u);
            const uint32_t m = same + other;
            if (m) {
                const bool shortf = same < 256 && other < 256;
                const unsigned long long at = atomicAdd(&A.ctr[1], (unsigned long long)((shortf ? 1 : 3) + m));
                if (at + (shortf ? 1 : 3) + m <= A.loc_cap) {
                    unsigned long long wpos;
                    if (shortf) { A.loc[at] = (same << 16) | other; wpos = at + 1; }
                    else { A.loc[at] = 0x80000000u | (uint32_t)(at + 1); A.loc[at + 1] = same; A.loc[at + 2] = other; wpos = at + 3; }
                    unsigned long long ws = wpos, wo = wpos + same;
                    const bool kf = (u_fw >> k) & 1;
                    for (int64_t e = i; e < j; ++e) {       // the run again: the other windows of this L-mer, positions ascending
                        const int64_t p = A.pos[e];
                        const uint32_t hi = (uint32_t)A.keys[e];
                        if ((uint32_t)p == rep || ((hi ^ u_hi[k]) >> 1) != 0) continue;
                        const bool f = hi & 1u;
                        if (canon_cmp(B, u_pos[k], kf, p, f) != 0) continue;
                        if ((int)f == f0) A.loc[ws++] = (uint32_t)p; else A.loc[wo++] = (uint32_t)p;
                    }
                    flags |= (uint32_t)at << 2;
                } else atomicAdd(&A.ctr[4], 1ull);
            }
            A.seeds[slot] = make_uint4(flags, rep, kNoEntry, slot_of_next);
            slot_of_next = slot;
        }
    }
}

// ---- buckets with more than kMaxU distinct L-mers.  The hash is an XOR of the window's 32-base words (the reference's, perfect.h:541-707,
// and part of the table format), so it is blind to WHERE in a periodic stretch something sits: slide a window along (GA)n with a poly-T run
// inside it and every position of the run gives another L-mer with the same XOR — dozens of distinct L-mers in one bucket on a genome with
// microsatellites, where the Poisson mean is 0.9.  The reference's builder just grows that bucket's tree; here such a run (they are few) is
// sorted by (canonical L-mer, position) with a bitonic network in HBM by one workgroup, and a wavefront then walks the sorted run: an L-mer is
// a group of neighbours, its first member the smallest position.
struct BigRun { unsigned long long start; unsigned int len, padded; unsigned long long off; };
struct BigArgs {
    BucketArgs A;
    const BigRun *runs;
    uint32_t *ord;             // per run `padded` words: indices into the run, sorted; bit 31 = first member of its L-mer
};
constexpr uint32_t kBigPad = 0x7fffffffu;
constexpr unsigned long long kBigCap = 1ull << 16;       // runs the big-bucket path lists

__device__ __forceinline__ bool big_less(const BucketArgs &A, unsigned long long start, uint32_t a, uint32_t b) {
    if (a == kBigPad || b == kBigPad) return a != kBigPad;                 // padding sorts last
    const int64_t pa = A.pos[start + a], pb = A.pos[start + b];
    const int c = canon_cmp(A.B, pa, A.keys[start + a] & 1ull, pb, A.keys[start + b] & 1ull);
    return c ? c < 0 : pa < pb;
}

__global__ __launch_bounds__(256) void emf_big_sort_kernel(BigArgs G) {
    const BigRun r = G.runs[blockIdx.x];
    uint32_t *ord = G.ord + r.off;
    for (uint32_t e = threadIdx.x; e < r.padded; e += 256) ord[e] = e < r.len ? e : kBigPad;
    __threadfence(); __syncthreads();
    for (uint32_t k = 2; k <= r.padded; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t e = threadIdx.x; e < r.padded; e += 256) {
                const uint32_t x = e ^ j;
                if (x > e) {
                    const uint32_t a = ord[e], b = ord[x];
                    const bool up = (e & k) == 0;
                    if (big_less(G.A, r.start, b, a) == up) { ord[e] = b; ord[x] = a; }
                }
            }
            __threadfence(); __syncthreads();
        }
    // group heads: a window whose canonical L-mer differs from its predecessor's (bit 31 of its own word; neighbours mask it)
    for (uint32_t e = threadIdx.x; e < r.len; e += 256) {
        bool h = e == 0;
        if (!h) {
            const uint32_t a = ord[e - 1] & kBigPad, b = ord[e] & kBigPad;
            h = canon_cmp(G.A.B, G.A.pos[r.start + a], G.A.keys[r.start + a] & 1ull, G.A.pos[r.start + b], G.A.keys[r.start + b] & 1ull) != 0;
        }
        if (h) atomicOr(&ord[e], 0x80000000u);
    }
}

template <int PASS>
__global__ __launch_bounds__(64) void emf_big_emit_kernel(BigArgs G) {
    const BucketArgs &A = G.A;
    const BigRun r = G.runs[blockIdx.x];
    const uint32_t *ord = G.ord + r.off;
    const int lane = threadIdx.x;
    const uint32_t key = (uint32_t)(A.keys[r.start] >> 32);
    // distinct L-mers of the run
    unsigned int c = 0;
    for (uint32_t e0 = 0; e0 < r.len; e0 += 64) {
        const uint32_t e = e0 + lane;
        c += (unsigned int)__popcll(__ballot(e < r.len && (ord[e] >> 31)));
    }
    unsigned long long fbase = 0;
    if (PASS == 2) {
        if (lane == 0) fbase = atomicAdd(&A.ctr[5], (unsigned long long)(c - 1));
        fbase = ((unsigned long long)(uint32_t)__shfl((int)(fbase >> 32), 0) << 32) | (uint32_t)__shfl((int)fbase, 0);
        if (fbase + (c - 1) > A.free_cap) { if (lane == 0) atomicAdd(&A.ctr[4], 1ull); return; }
    }
    unsigned long long words = 0;
    unsigned int gbase = 0;                                 // groups before this chunk
    for (uint32_t e0 = 0; e0 < r.len; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool head = e < r.len && (ord[e] >> 31);
        const unsigned long long hm = __ballot(head);
        const unsigned int g = gbase + (unsigned int)__popcll(hm & ((1ull << lane) - 1ull));      // this group's rank: ascending L-mers
        gbase += (unsigned int)__popcll(hm);
        if (!head) continue;
        // the group: members e .. e1 - 1, positions ascending; the first is the entry's location
        const uint32_t m0 = ord[e] & kBigPad;
        const uint32_t rep = A.pos[r.start + m0];
        const uint32_t f0 = (uint32_t)(A.keys[r.start + m0] & 1ull);
        uint32_t same = 0, other = 0, e1 = e + 1;
        for (; e1 < r.len && !(ord[e1] >> 31); ++e1) {
            if ((uint32_t)(A.keys[r.start + ord[e1]] & 1ull) == f0) ++same; else ++other;
        }
        const uint32_t m = same + other;
        const bool shortf = same < 256 && other < 256;
        if (PASS == 1) { if (m) words += (shortf ? 1 : 3) + m; continue; }
        const uint32_t slot = g == 0 ? key : A.free_list[fbase + (g - 1)];
        const uint32_t next = g + 1 < c ? A.free_list[fbase + g] : kNoEntry;
        uint32_t flags = (f0 ? 1u : 0u) | (g == 0 ? 0u : 2u);
        if (m) {
            const unsigned long long at = atomicAdd(&A.ctr[1], (unsigned long long)((shortf ? 1 : 3) + m));
            if (at + (shortf ? 1 : 3) + m <= A.loc_cap) {
                unsigned long long wpos;
                if (shortf) { A.loc[at] = (same << 16) | other; wpos = at + 1; }
                else { A.loc[at] = 0x80000000u | (uint32_t)(at + 1); A.loc[at + 1] = same; A.loc[at + 2] = other; wpos = at + 3; }
                unsigned long long ws = wpos, wo = wpos + same;
                for (uint32_t x = e + 1; x < e1; ++x) {
                    const uint32_t mm = ord[x];
                    const uint32_t p = A.pos[r.start + mm];
                    if ((uint32_t)(A.keys[r.start + mm] & 1ull) == f0) A.loc[ws++] = p; else A.loc[wo++] = p;
                }
                flags |= (uint32_t)at << 2;
            } else atomicAdd(&A.ctr[4], 1ull);
        }
        A.seeds[slot] = make_uint4(flags, rep, kNoEntry, next);
    }
    if (PASS == 1) {
        words = wave_sum(words);
        if (lane == 0) {
            atomicOr(&A.used[key >> 5], 1u << (key & 31));
            atomicAdd(&A.ctr[0], (unsigned long long)(c - 1));
            if (words) atomicAdd(&A.ctr[1], words);
            atomicAdd(&A.ctr[2], (unsigned long long)c);
            atomicAdd(&A.ctr[3], 1ull);
        }
    }
}

__global__ void emf_fill_kernel(uint4 *seeds, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        seeds[i] = make_uint4(0u, kNoEntry, kNoEntry, kNoEntry);
}

struct IsFree {
    const uint32_t *used;
    __host__ __device__ bool operator()(uint32_t k) const { return !((used[k >> 5] >> (k & 31)) & 1u); }
};

}  // namespace

// Builds the table of the resident forward reference into device buffers owned by *e.
int emf_build_device(bwams_emf *e, const uint8_t *ref, int64_t l_pac, int seed_len, double slack, int cu_count, int verbose,
                     int64_t stats[4]) {
    hipStream_t st = nullptr;
    EmfBuild B;
    B.ref = ref; B.l_pac = l_pac; B.L = seed_len; B.n_win = l_pac - seed_len + 1;
    double ne = slack * (double)l_pac;
    B.n_entry = ne < 16 ? 16 : (uint64_t)ne;
    if (B.n_entry >= 0xffffffffull || l_pac >= 0xffffffffll || B.n_win <= 0) {
        set_last_error("emf_build: the table's 32-bit locations and slots hold references of fewer than 2^32 / slack bases");
        return BWAMS_ERR_UNSUPPORTED;
    }
    void *d_k = nullptr, *d_k2 = nullptr, *d_v = nullptr, *d_v2 = nullptr, *d_tmp = nullptr, *d_used = nullptr, *d_free = nullptr, *d_ctr = nullptr,
         *d_nsel = nullptr, *d_big = nullptr, *d_runs = nullptr, *d_ord = nullptr;
    auto cleanup = [&]() {
        for (void *p : {d_k, d_k2, d_v, d_v2, d_tmp, d_used, d_free, d_ctr, d_nsel, d_big, d_runs, d_ord})
            if (p) (void)hipFree(p);
    };
#define EMF_HIP(call)                                                       \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            set_last_error(std::string("emf_build: " #call " -> ") + hipGetErrorString(e_)); \
            cleanup();                                                      \
            return e_ == hipErrorOutOfMemory ? BWAMS_ERR_NOMEM : BWAMS_ERR_DEVICE; \
        }                                                                   \
    } while (0)
    const size_t n = (size_t)B.n_win;
    hipEvent_t e0, e1, e2, e3;
    EMF_HIP(hipEventCreate(&e0)); EMF_HIP(hipEventCreate(&e1)); EMF_HIP(hipEventCreate(&e2)); EMF_HIP(hipEventCreate(&e3));
    EMF_HIP(dev_malloc(&d_k, n * 8)); EMF_HIP(dev_malloc(&d_k2, n * 8));
    EMF_HIP(dev_malloc(&d_v, n * 4)); EMF_HIP(dev_malloc(&d_v2, n * 4));
    EMF_HIP(dev_malloc(&d_ctr, 8 * 8));
    EMF_HIP(hipMemsetAsync(d_ctr, 0, 8 * 8, st));
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)((n + 255) / 256), (int64_t)cu_count * 32);
    EMF_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(emf_key_kernel, dim3(grid), dim3(256), 0, st, B, (uint64_t *)d_k, (uint32_t *)d_v);
    EMF_HIP(hipGetLastError());
    {
        rocprim::double_buffer<uint64_t> kb((uint64_t *)d_k, (uint64_t *)d_k2);
        rocprim::double_buffer<uint32_t> vb((uint32_t *)d_v, (uint32_t *)d_v2);
        size_t tb = 0;
        EMF_HIP(rocprim::radix_sort_pairs(nullptr, tb, kb, vb, n, 0, 64, st));
        EMF_HIP(dev_malloc(&d_tmp, tb ? tb : 8));
        EMF_HIP(rocprim::radix_sort_pairs(d_tmp, tb, kb, vb, n, 0, 64, st));
        EMF_HIP(hipStreamSynchronize(st));
        if (kb.current() != (uint64_t *)d_k) std::swap(d_k, d_k2);
        if (vb.current() != (uint32_t *)d_v) std::swap(d_v, d_v2);
        (void)hipFree(d_tmp); d_tmp = nullptr;
        (void)hipFree(d_k2); d_k2 = nullptr;
        (void)hipFree(d_v2); d_v2 = nullptr;
    }
    EMF_HIP(hipEventRecord(e1, st));
    const size_t used_words = (size_t)((B.n_entry + 31) / 32);
    EMF_HIP(dev_malloc(&d_used, used_words * 4 + 4));
    EMF_HIP(hipMemsetAsync(d_used, 0, used_words * 4 + 4, st));
    BucketArgs A;
    memset(&A, 0, sizeof A);
    A.B = B; A.keys = (const uint64_t *)d_k; A.pos = (const uint32_t *)d_v; A.used = (uint32_t *)d_used;
    A.ctr = (unsigned long long *)d_ctr;
    EMF_HIP(dev_malloc(&d_big, kBigCap * sizeof(ulonglong2)));
    A.big = (ulonglong2 *)d_big; A.big_cap = kBigCap;
    hipLaunchKernelGGL(emf_bucket_kernel<1>, dim3(grid), dim3(256), 0, st, A);
    EMF_HIP(hipGetLastError());
    unsigned long long c[8];
    EMF_HIP(hipMemcpyAsync(c, d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
    EMF_HIP(hipStreamSynchronize(st));
    // runs with more distinct L-mers than a lane keeps: sorted and counted by their own kernels
    const unsigned long long n_big = c[6];
    BigArgs G;
    memset(&G, 0, sizeof G);
    if (n_big > kBigCap) {
        set_last_error("emf_build: " + std::to_string(n_big) + " hash buckets hold more than " + std::to_string(kMaxU) + " distinct L-mers (the builder lists " +
                       std::to_string(kBigCap) + " of them)");
        cleanup();
        return BWAMS_ERR_UNSUPPORTED;
    }
    unsigned long long big_windows = 0;
    if (n_big) {
        std::vector<ulonglong2> lst(n_big);
        EMF_HIP(hipMemcpy(lst.data(), d_big, n_big * sizeof(ulonglong2), hipMemcpyDeviceToHost));
        std::sort(lst.begin(), lst.end(), [](const ulonglong2 &a, const ulonglong2 &b) { return a.x < b.x; });
        std::vector<BigRun> runs(n_big);
        unsigned long long off = 0;
        for (size_t t = 0; t < n_big; ++t) {
            if (lst[t].y >= (1ull << 30)) {
                set_last_error("emf_build: a hash bucket holds 2^30 windows or more");
                cleanup();
                return BWAMS_ERR_UNSUPPORTED;
            }
            unsigned int pad = 2;
            while (pad < lst[t].y) pad <<= 1;
            runs[t].start = lst[t].x; runs[t].len = (unsigned int)lst[t].y; runs[t].padded = pad; runs[t].off = off;
            off += pad;
            big_windows += lst[t].y;
        }
        EMF_HIP(dev_malloc(&d_runs, n_big * sizeof(BigRun)));
        EMF_HIP(dev_malloc(&d_ord, off * 4));
        EMF_HIP(hipMemcpyAsync(d_runs, runs.data(), n_big * sizeof(BigRun), hipMemcpyHostToDevice, st));
        G.A = A; G.runs = (const BigRun *)d_runs; G.ord = (uint32_t *)d_ord;
        hipLaunchKernelGGL(emf_big_sort_kernel, dim3((unsigned)n_big), dim3(256), 0, st, G);
        hipLaunchKernelGGL(emf_big_emit_kernel<1>, dim3((unsigned)n_big), dim3(64), 0, st, G);
        EMF_HIP(hipGetLastError());
        EMF_HIP(hipMemcpyAsync(c, d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
        EMF_HIP(hipStreamSynchronize(st));                  // (runs[] stays alive until here)
    }
    const unsigned long long n_other = c[0], n_loc = c[1] + 1, n_used = c[2], n_key = c[3];
    // the free slots, ascending
    const size_t n_free = (size_t)(B.n_entry - n_key);
    EMF_HIP(dev_malloc(&d_free, (n_free ? n_free : 1) * 4));
    EMF_HIP(dev_malloc(&d_nsel, 8));
    {
        IsFree pred;
        pred.used = (const uint32_t *)d_used;
        rocprim::counting_iterator<uint32_t> it(0u);
        size_t tb = 0;
        EMF_HIP(rocprim::select(nullptr, tb, it, (uint32_t *)d_free, (size_t *)d_nsel, (size_t)B.n_entry, pred, st));
        EMF_HIP(dev_malloc(&d_tmp, tb ? tb : 8));
        EMF_HIP(rocprim::select(d_tmp, tb, it, (uint32_t *)d_free, (size_t *)d_nsel, (size_t)B.n_entry, pred, st));
    }
    EMF_HIP(hipEventRecord(e2, st));
    EMF_HIP(dev_malloc(&e->d_seeds, (size_t)B.n_entry * 16));
    EMF_HIP(dev_malloc(&e->d_loc, (size_t)n_loc * 4));
    EMF_HIP(hipMemsetAsync(e->d_loc, 0, (size_t)n_loc * 4, st));
    hipLaunchKernelGGL(emf_fill_kernel, dim3((unsigned)cu_count * 32), dim3(256), 0, st, (uint4 *)e->d_seeds, B.n_entry);
    const unsigned long long init[8] = {0, 1, 0, 0, 0, 0, 0, 0};        // loc_table[0] is unused
    EMF_HIP(hipMemcpyAsync(d_ctr, init, sizeof init, hipMemcpyHostToDevice, st));
    A.free_list = (const uint32_t *)d_free; A.seeds = (uint4 *)e->d_seeds; A.loc = (uint32_t *)e->d_loc;
    A.free_cap = n_free; A.loc_cap = n_loc;
    hipLaunchKernelGGL(emf_bucket_kernel<2>, dim3(grid), dim3(256), 0, st, A);
    if (n_big) {
        G.A = A;
        hipLaunchKernelGGL(emf_big_emit_kernel<2>, dim3((unsigned)n_big), dim3(64), 0, st, G);
    }
    EMF_HIP(hipGetLastError());
    EMF_HIP(hipEventRecord(e3, st));
    EMF_HIP(hipMemcpyAsync(c, d_ctr, sizeof c, hipMemcpyDeviceToHost, st));
    EMF_HIP(hipStreamSynchronize(st));
    if (c[4] || c[5] != n_other || c[1] != n_loc) {
        set_last_error("emf_build: internal error, the two passes disagree");
        cleanup();
        return BWAMS_ERR_DEVICE;
    }
    float ms1 = 0, ms2 = 0, ms3 = 0;
    (void)hipEventElapsedTime(&ms1, e0, e1); (void)hipEventElapsedTime(&ms2, e1, e2); (void)hipEventElapsedTime(&ms3, e2, e3);
    if (verbose)
        fprintf(stderr, "[bwams] emf_build: L = %d, %llu windows, %llu distinct L-mers in %llu buckets of %llu, %llu nodes outside their bucket, "
                        "%llu location words, %llu buckets (%llu windows) beyond %d L-mers; keys + sort %.1f ms, count + free list %.1f ms, table %.1f ms\n",
                seed_len, (unsigned long long)n, n_used, n_key, (unsigned long long)B.n_entry, n_other, n_loc, n_big, big_windows, kMaxU, ms1, ms2, ms3);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
    cleanup();
#undef EMF_HIP
    e->t.seed_table = reinterpret_cast<const uint4 *>(e->d_seeds);
    e->t.loc_table = reinterpret_cast<const uint32_t *>(e->d_loc);
    e->t.ref = ref;
    e->t.num_seed_entry = (uint32_t)B.n_entry;
    e->t.num_loc_entry = (uint32_t)n_loc;
    e->t.seq_len = (uint32_t)l_pac;
    e->t.seed_len = seed_len;
    e->bytes = (int64_t)B.n_entry * 16 + (int64_t)n_loc * 4;
    stats[0] = (int64_t)n_used; stats[1] = (int64_t)n_key; stats[2] = (int64_t)n_other; stats[3] = (int64_t)(ms1 + ms2 + ms3);
    return BWAMS_OK;
}

}  // namespace bwams
