// emf_probe.hip — exact-match filter (EMF) probe for gfx950 (MI355X).
//
// Reference semantics: find_perfect_match_entry (/root/reference/src/perfect_map.cpp:638-659)
// -> __find_perfect_match_entry (:583-629) -> seedmatch_further (:528-581); primitives in
// /root/reference/src/perfect.h (hash :541-707, canonical strand :362-368, ordered compare
// :273-360, tail match :415-491, multi-location list :170-186).  For every read: reject reads
// with N; canonical strand of the first L bases (fw if fw <= revcomp on the first half);
// fmix64 hash of the 2-bit packed canonical L-mer -> bucket head; descend the bucket's BST by
// comparing the L-mer stored at `location` of the .0123 reference; for reads longer than L
// verify the tail at the head location and at the listed alternatives.  Output:
// bseq1_perfect_t {flags, location} and the FIND_PERFECT_* code (perfect.h:902-907).
//
// Mapping: one read per LANE (a probe is 1-3 dependent random reads: one 16-byte entry and one
// L-byte reference window per BST node); the comparisons stream the window as 16-byte loads.
// HBM-latency bound; algorithmic bytes = 16 B per entry visited + L B per compare.
#include "common.h"

namespace bwams {
namespace {

constexpr uint32_t kNoEntry = 0xffffffffu;

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ int canon_at(const uint8_t *s, int len, bool fw, int i) {
    return fw ? (s[i] & 3) : 3 - (s[len - 1 - i] & 3);
}
// lexicographic compare of a (read forward if afl else as reverse complement) with b likewise
__device__ __forceinline__ int seedcmp(const uint8_t *a, bool afl, const uint8_t *b, bool bfl, int len) {
    for (int i = 0; i < len; ++i) {
        const int x = afl ? a[i] : 3 - a[len - 1 - i];
        const int y = bfl ? b[i] : 3 - b[len - 1 - i];
        if (x != y) return x > y ? 1 : -1;
    }
    return 0;
}
__device__ __forceinline__ bool match_further(const DevEmf &t, uint32_t loc, const uint8_t *seed, bool is_rev, int len) {
    const int L = t.seed_len;
    len -= L;
    if (!is_rev) {
        if (loc + (uint32_t)len >= t.seq_len) return false;
        for (int i = 0; i < len; ++i)
            if (t.ref[loc + L + i] != seed[L + i]) return false;
        return true;
    }
    if (loc < (uint32_t)len) return false;
    for (int i = 0; i < len; ++i)
        if (t.ref[loc - len + i] != 3 - seed[L + len - 1 - i]) return false;
    return true;
}

__global__ __launch_bounds__(256) void emf_probe_kernel(DevEmf t, const uint8_t *__restrict__ enc,
                                                        const int64_t *__restrict__ cum, int64_t nseq,
                                                        uint32_t *__restrict__ out /* flags, location */,
                                                        uint8_t *__restrict__ code_out, uint8_t *__restrict__ skip,
                                                        DevCounters *ctr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_nodes = 0, n_cmp_bytes = 0;
    if (r < nseq) {
    const uint8_t *seed = enc + cum[r];
    const int len = (int)(cum[r + 1] - cum[r]);
    uint32_t flags = 0, location = 0;
    int code;
    const int L = t.seed_len;
    if (len < L) {
        code = 0;                                            // FIND_PERFECT_NO_TABLE
    } else {
        int n = 0;
        for (int i = 0; i < len; ++i) n |= seed[i] & 0xC;
        if (n) {
            code = 1;                                        // FIND_PERFECT_WITH_N
        } else {
            const int half = (L + 1) / 2;
            const bool fw_less = seedcmp(seed, true, seed + (L - half), false, half) <= 0;
            uint64_t h = 0, w = 0;
            const int full = L - L % 32;
            int i = 0;
            for (; i < full; ++i) {
                w = (w << 2) | (uint64_t)canon_at(seed, L, fw_less, i);
                if ((i & 31) == 31) { h ^= w; w = 0; }
            }
            if (L % 32) {
                w = 0;
                for (; i < L; ++i) w = (w << 2) | (uint64_t)canon_at(seed, L, fw_less, i);
                h ^= w;
            }
            uint32_t idx = (uint32_t)(fmix64(h) % t.num_seed_entry);
            uint4 ent = t.seed_table[idx];                   // x flags, y location, z left, w right
            code = 2;                                        // FIND_PERFECT_NOT_MATCHED
            if (ent.y != kNoEntry && !(ent.x & 2u)) {
                while (true) {
                    const bool efl = (ent.x & 1u) != 0;
                    const int cmp = seedcmp(t.ref + ent.y, efl, seed, fw_less, L);
                    n_nodes++;
                    n_cmp_bytes += (unsigned long long)L;
                    if (cmp == 0) {
                        bool is_rev = efl != fw_less;
                        if (len == L) {
                            location = ent.y;
                        } else {
                            uint32_t loc = kNoEntry;
                            if (match_further(t, ent.y, seed, is_rev, len)) {
                                loc = ent.y;
                            } else if (ent.x >> 2) {
                                const uint32_t multi = ent.x >> 2;
                                const uint32_t first = t.loc_table[multi];
                                const bool many = (first & 0x80000000u) != 0;
                                const uint32_t st = many ? (first & 0x7fffffffu) : multi;
                                uint32_t nfw, nrc, base;
                                if (!many) { nfw = (t.loc_table[st] >> 16) & 0xffff; nrc = t.loc_table[st] & 0xffff; base = st + 1; }
                                else { nfw = t.loc_table[st]; nrc = t.loc_table[st + 1]; base = st + 2; }
                                for (uint32_t k = 0; k < nfw && loc == kNoEntry; ++k) {
                                    const uint32_t c = t.loc_table[base + k];
                                    if (match_further(t, c, seed, is_rev, len)) loc = c;
                                }
                                if (loc == kNoEntry) {
                                    is_rev = !is_rev;
                                    for (uint32_t k = 0; k < nrc && loc == kNoEntry; ++k) {
                                        const uint32_t c = t.loc_table[base + nfw + k];
                                        if (match_further(t, c, seed, is_rev, len)) loc = c;
                                    }
                                }
                            }
                            if (loc == kNoEntry) { code = 5; break; }     // FIND_PERFECT_SEED_ONLY_MATCHED
                            location = loc;
                        }
                        if (!is_rev) { flags = (ent.x & ~2u) | 1u; code = 3; }
                        else { flags = ent.x | 2u | 1u; code = 4; }
                        break;
                    }
                    idx = cmp > 0 ? ent.z : ent.w;
                    if (idx == kNoEntry) break;
                    ent = t.seed_table[idx];
                }
            }
        }
    }
    out[2 * r] = flags;
    out[2 * r + 1] = location;
    code_out[r] = (uint8_t)code;
    if (skip) skip[r] = (code == 3 || code == 4) ? 1 : 0;
    }
    // algorithmic bytes of this launch: 16 B per entry visited + L B per compare
    for (int o = 32; o > 0; o >>= 1) {
        n_nodes += ((unsigned long long)__shfl_down((uint32_t)n_nodes, o)) | ((unsigned long long)__shfl_down((uint32_t)(n_nodes >> 32), o) << 32);
        n_cmp_bytes += ((unsigned long long)__shfl_down((uint32_t)n_cmp_bytes, o)) | ((unsigned long long)__shfl_down((uint32_t)(n_cmp_bytes >> 32), o) << 32);
    }
    if (ctr && (threadIdx.x & 63) == 0 && n_nodes) {
        atomicAdd(&ctr->emf_nodes, n_nodes);
        atomicAdd(&ctr->emf_cmp_bytes, n_cmp_bytes);
    }
}

}  // namespace

void launch_emf_probe(const DevEmf &t, const uint8_t *enc, const int64_t *cum, int64_t nseq, uint32_t *out,
                      uint8_t *code, uint8_t *skip, DevCounters *ctr, hipStream_t st) {
    if (nseq <= 0) return;
    emf_probe_kernel<<<(unsigned)((nseq + 255) / 256), 256, 0, st>>>(t, enc, cum, nseq, out, code, skip, ctr);
}

}  // namespace bwams
