// fmi_build.hip — FM-index construction on the GPU, for texts beyond 2^32 rows (GRCh38: 6.4 G rows).
//
// Replaces FMI_search::build_index + build_fm_index (/root/reference/src/FMI_search.cpp:774-849, :611-771):
//   text  = fw || revcomp(fw) over {0,1,2,3}                          (pac2nt, :774-829; also the .0123 file)
//   SA    = suffix array of text + terminator, SA[0] = |text|          (saisxx, :833-840)
//   BWT   -> CP_OCC blocks of 64 rows (4 counts + 4 one-hot strings)  (:640-713)
//   SA samples every 8 rows, split into int8 high byte + uint32 low word (:719-737)
// The outputs are defined by the suffix array alone, so any correct suffix sorter reproduces the reference's
// files byte for byte.  The reference runs single-threaded SA-IS on the host (51 GB of int64 for GRCh38);
// here the sort is prefix doubling laid out for one MI355X:
//
//   1. the text is packed to 2 bits per base (1.6 GB for GRCh38: it stays in the Infinity Cache / L2 while the keys
//      are formed); the 64-bit key of suffix i = its first 29 bases (58 bits, zero padded) | min(|suffix|, 29)
//      (6 bits: a suffix cut short by the terminator sorts in front of a longer one with the same padded bases).
//   2. MSD partition: a histogram over the keys' top 14 bits cuts the key space into chunks of <= chunk_rows
//      suffixes; per chunk the (key, position) pairs are collected (block-aggregated append) and sorted with
//      rocPRIM's radix sort, which leaves SA in 29-order; heads of equal-key groups go to a bitmap, and
//      ISA[i] = first SA row of i's group.
//   3. doubling rounds h = 29, 58, 116, ...: only rows of groups with more than one member take part
//      (Larsson-Sadakane); key = (dense group number << rbits) | ISA[i + h]; one radix sort per round over the
//      unresolved rows; new group heads, ISA of the moved suffixes.  On a genome-like text the first round
//      handles the repeats' share of the rows and the later ones next to nothing.
//   4. BWT / CP_OCC: one wavefront per 64-row block, lane j = row j; the one-hot strings are ballots; block
//      counts by a scan over 32-byte {A,C,G,T} records.  SA samples by a strided copy.
//
// HBM at GRCh38 size (L = 6.42 G rows): SA + ISA 2 x 51 GB, text 6.4 + 1.6 GB, head bitmap 0.8 GB, sort
// buffers 4 x 8 B x chunk_rows (34 GB at the default 2^30): ~150 GB of the 288.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <vector>

#include "common.h"

namespace bwams {
namespace {

constexpr int kKeyBases = 29;
constexpr int kBinBits = 14;                       // histogram over the first 7 bases
constexpr int kBins = 1 << kBinBits;
constexpr int kBlk = 256;

// ---- text --------------------------------------------------------------------------------------------

// ref[0, n) = fw, ref[n, 2n) = reverse complement; per-block base counts of fw into cnt[4]
__global__ void fwrc_kernel(const uint8_t *__restrict__ fw, int64_t n, uint8_t *__restrict__ ref, unsigned long long *cnt,
                            unsigned long long *bad) {
    __shared__ unsigned int c[4];
    if (threadIdx.x < 4) c[threadIdx.x] = 0;
    __syncthreads();
    unsigned int mine[4] = {0, 0, 0, 0};
    unsigned int nbad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t b = fw[i];
        if (b > 3) { nbad++; b &= 3; }
        ref[i] = b;
        ref[2 * n - 1 - i] = (uint8_t)(3 - b);
        mine[0] += b == 0; mine[1] += b == 1; mine[2] += b == 2; mine[3] += b == 3;
    }
    for (int k = 0; k < 4; ++k) atomicAdd(&c[k], mine[k]);
    if (nbad) atomicAdd(bad, (unsigned long long)nbad);
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&cnt[threadIdx.x], (unsigned long long)c[threadIdx.x]);
}

// word w = bases 32w .. 32w+31, base j in bits [62 - 2(j & 31), 63 - 2(j & 31)] (first base most significant), zero past N
__global__ void pack_text_kernel(const uint8_t *__restrict__ ref, int64_t N, uint64_t *__restrict__ packed, int64_t nwords) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const int64_t b0 = w << 5;
    uint64_t v = 0;
    if (b0 + 32 <= N) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(ref + b0), hi = *reinterpret_cast<const uint4 *>(ref + b0 + 16);
        const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = q[k];       // four bases, first in the low byte
            const uint64_t four = ((uint64_t)(x & 3) << 6) | ((uint64_t)((x >> 8) & 3) << 4) | ((uint64_t)((x >> 16) & 3) << 2) |
                                  (uint64_t)((x >> 24) & 3);
            v |= four << (56 - 8 * k);
        }
    } else {
        for (int j = 0; j < 32 && b0 + j < N; ++j) v |= (uint64_t)(ref[b0 + j] & 3) << (62 - 2 * j);
    }
    packed[w] = v;
}

__device__ __forceinline__ uint64_t suffix_key(const uint64_t *__restrict__ packed, int64_t i, int64_t N) {
    const int64_t w = i >> 5;
    const int sh = (int)(i & 31) * 2;
    const uint64_t a = packed[w], b = packed[w + 1];
    const uint64_t x = sh ? ((a << sh) | (b >> (64 - sh))) : a;
    const int64_t rem = N - i;
    const uint64_t len = rem >= kKeyBases ? (uint64_t)kKeyBases : (uint64_t)rem;
    return (x & ~0x3Full) | len;
}

// ---- step 2: MSD partition ------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void key_hist_kernel(const uint64_t *__restrict__ packed, int64_t N,
                                                        unsigned long long *__restrict__ hist) {
    __shared__ unsigned int h[kBins];
    for (int k = threadIdx.x; k < kBins; k += blockDim.x) h[k] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= N; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[suffix_key(packed, i, N) >> (64 - kBinBits)], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < kBins; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[k], (unsigned long long)h[k]);
}

// (key, i) of every suffix whose bin lies in [bin_lo, bin_hi), appended in arbitrary order (one atomic per block and pass)
__global__ __launch_bounds__(1024) void key_collect_kernel(const uint64_t *__restrict__ packed, int64_t N, uint32_t bin_lo,
                                                           uint32_t bin_hi, uint64_t *__restrict__ keys, int64_t *__restrict__ vals,
                                                           unsigned long long *cursor) {
    __shared__ unsigned int wave_cnt[16];
    __shared__ unsigned long long blk_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 <= N; i0 += stride) {
        const int64_t i = i0 + threadIdx.x;
        uint64_t key = 0;
        bool sel = false;
        if (i <= N) {
            key = suffix_key(packed, i, N);
            const uint32_t bin = (uint32_t)(key >> (64 - kBinBits));
            sel = bin >= bin_lo && bin < bin_hi;
        }
        const unsigned long long m = __ballot(sel);
        if (lane == 0) wave_cnt[wave] = (unsigned int)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            for (int w = 0; w < nw; ++w) { const unsigned int c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
            blk_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        if (sel) {
            const unsigned long long slot = blk_base + wave_cnt[wave] + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            keys[slot] = key;
            vals[slot] = i;
        }
        __syncthreads();
    }
}

// head position of each sorted element's group (0 where the element is not a head), for the max-scan
struct HeadPos {
    const uint64_t *keys;
    int64_t base;
    __device__ int64_t operator()(int64_t j) const { return (j == 0 || keys[j] != keys[j - 1]) ? base + j : (int64_t)0; }
};
struct MaxOp {
    __device__ int64_t operator()(int64_t a, int64_t b) const { return a > b ? a : b; }
};

// set bits [P0, P0 + 64) of the head bitmap from a wave's ballot (bit b of word w = row 64 w + b)
__device__ __forceinline__ void bitmap_or_wave(unsigned long long *bm, int64_t P0, unsigned long long m) {
    if ((threadIdx.x & 63) == 0 && m) {
        const int sh = (int)(P0 & 63);
        atomicOr(&bm[P0 >> 6], m << sh);
        if (sh && (m >> (64 - sh))) atomicOr(&bm[(P0 >> 6) + 1], m >> (64 - sh));
    }
}

// after a chunk's sort: SA rows [base, base + cnt), head bits, ISA of its suffixes
__global__ void chunk_finish_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ vals,
                                    const int64_t *__restrict__ rank, int64_t cnt, int64_t base, int64_t *__restrict__ sa,
                                    int64_t *__restrict__ isa, unsigned long long *bm) {
    const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool head = false;
    if (j < cnt) {
        const int64_t i = vals[j];
        sa[base + j] = i;
        isa[i] = rank[j];
        head = j == 0 || keys[j] != keys[j - 1];
    }
    bitmap_or_wave(bm, base + j0, __ballot(head));
}

// ---- step 3: doubling rounds ---------------------------------------------------------------------------------

// row p is unresolved unless it is a head and row p + 1 is a head too (row L counts as one)
__device__ __forceinline__ unsigned long long unresolved_mask(const unsigned long long *__restrict__ bm, int64_t w, int64_t nw,
                                                              int64_t L) {
    const unsigned long long h = bm[w];
    unsigned long long nxt = (w + 1 < nw) ? bm[w + 1] : 0ull;
    unsigned long long h1 = (h >> 1) | (nxt << 63);            // head bit of row p + 1
    unsigned long long valid = ~0ull;
    const int64_t r0 = w << 6;
    if (r0 + 64 > L) valid = (L - r0 >= 64) ? ~0ull : ((1ull << (L - r0)) - 1ull);
    // row L (one past the end) is a head
    if (L >= r0 + 1 && L <= r0 + 64) h1 |= 1ull << (L - 1 - r0);
    return ~(h & h1) & valid;
}

__global__ void unres_count_kernel(const unsigned long long *__restrict__ bm, int64_t nw, int64_t L, int64_t *__restrict__ cnt) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) cnt[w] = __popcll(unresolved_mask(bm, w, nw, L));
}

// Upos[off[w] ...] = the unresolved rows of word w, ascending; ishead[j] = that row is a group head
__global__ void unres_fill_kernel(const unsigned long long *__restrict__ bm, int64_t nw, int64_t L, const int64_t *__restrict__ off,
                                  int64_t *__restrict__ upos, int64_t *__restrict__ ishead) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    unsigned long long m = unresolved_mask(bm, w, nw, L);
    const unsigned long long h = bm[w];
    int64_t o = off[w];
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        upos[o] = (w << 6) + b;
        ishead[o] = (int64_t)((h >> b) & 1ull);
        ++o;
    }
}

// key[j] = (dense group number << rbits) | ISA[SA[p] + h]; val[j] = SA[p]
__global__ void round_keys_kernel(const int64_t *__restrict__ upos, const int64_t *__restrict__ gsum, int64_t M,
                                  const int64_t *__restrict__ sa, const int64_t *__restrict__ isa, int64_t h, int64_t N, int rbits,
                                  uint64_t *__restrict__ keys, int64_t *__restrict__ vals) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int64_t i = sa[upos[j]];
    const int64_t r = (i + h <= N) ? isa[i + h] : (int64_t)0;      // a suffix that short is alone in its group already
    keys[j] = ((uint64_t)(gsum[j] - 1) << rbits) | (uint64_t)r;
    vals[j] = i;
}

struct RoundHeadPos {
    const uint64_t *keys;
    const int64_t *upos;
    __device__ int64_t operator()(int64_t j) const { return (j == 0 || keys[j] != keys[j - 1]) ? upos[j] : (int64_t)0; }
};

__global__ void round_finish_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ vals,
                                    const int64_t *__restrict__ upos, const int64_t *__restrict__ rank, int64_t M,
                                    int64_t *__restrict__ sa, int64_t *__restrict__ isa, unsigned long long *bm) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int64_t p = upos[j], i = vals[j];
    sa[p] = i;
    isa[i] = rank[j];
    // rows are scattered here: per-row atomics, skipped for rows that were heads already (a stale 0 only costs a redundant atomic)
    if ((j == 0 || keys[j] != keys[j - 1]) && !((bm[p >> 6] >> (p & 63)) & 1ull)) atomicOr(&bm[p >> 6], 1ull << (p & 63));
}

// ---- step 4: BWT -> CP_OCC, SA samples -------------------------------------------------------------------------

struct Cnt4 {
    int64_t c[4];
};
struct Cnt4Add {
    __device__ Cnt4 operator()(const Cnt4 &a, const Cnt4 &b) const {
        Cnt4 r;
        r.c[0] = a.c[0] + b.c[0]; r.c[1] = a.c[1] + b.c[1]; r.c[2] = a.c[2] + b.c[2]; r.c[3] = a.c[3] + b.c[3];
        return r;
    }
};

// one wavefront per block of 64 rows: one-hot strings (bit 63 - j = row j) into cp[blk].hot, the block's base counts into blkcnt.
// Grid-stride over the blocks: a launch may not hold 2^32 threads, and GRCh38 has 6.4 G rows.
__global__ void bwt_block_kernel(const int64_t *__restrict__ sa, const uint8_t *__restrict__ ref, int64_t L, int64_t filled,
                                 uint64_t *__restrict__ cp /* 8 words per block */, Cnt4 *__restrict__ blkcnt) {
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t blk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; blk < filled; blk += n_waves) {
        const int64_t p = (blk << 6) + lane;
        int c = 6;                                            // DUMMY_CHAR past the last row
        if (p < L) {
            const int64_t s = sa[p];
            c = s == 0 ? 4 : (int)ref[s - 1];
        }
        unsigned long long m[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) m[b] = __brevll(__ballot(c == b));
        if (lane < 4) {
            const unsigned long long mine = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
            cp[blk * 8 + 4 + lane] = mine;
            blkcnt[blk].c[lane] = __popcll(mine);
        }
    }
}
__global__ void cp_counts_kernel(const Cnt4 *__restrict__ pre, int64_t filled, uint64_t *__restrict__ cp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= filled * 4) return;
    cp[(t >> 2) * 8 + (t & 3)] = (uint64_t)pre[t >> 2].c[t & 3];
}
__global__ void sa_sample_kernel(const int64_t *__restrict__ sa, int64_t L, int64_t n_sa, int8_t *__restrict__ ms,
                                 uint32_t *__restrict__ ls) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_sa) return;
    const int64_t v = (t << 3) < L ? sa[t << 3] : (int64_t)0;
    ms[t] = (int8_t)((v >> 32) & 0xff);
    ls[t] = (uint32_t)(v & 0xffffffff);
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) {
        if (p) { (void)hipFree(p); p = nullptr; }
        return dev_malloc(&p, bytes ? bytes : 8);
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// blocks of `per` threads covering n items; every caller keeps n below 2^32 (HIP's bound on the threads of one launch)
inline unsigned int nblk(int64_t n, int per) { return (unsigned int)((n + per - 1) / per); }

int bits_for(uint64_t v) {          // bits needed to hold values 0..v
    int b = 1;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

}  // namespace

// Builds every array of the index into freshly allocated device buffers owned by *ix.  fw: l_pac codes (0..3) in device memory.
int fmi_build_device(bwams_index *ix, const uint8_t *d_fw, int64_t l_pac, int keep_ref, int64_t chunk_rows, int verbose,
                     bwams_build_stats_t *bs) {
    const int64_t N = 2 * l_pac, L = N + 1;
    if (L >= ((int64_t)1 << 36)) {
        set_last_error("text longer than 2^36 rows is not supported by the 36-bit interval packing");
        return BWAMS_ERR_UNSUPPORTED;
    }
    if (chunk_rows <= 0) chunk_rows = (int64_t)1 << 30;
    if (chunk_rows > ((int64_t)1 << 31)) chunk_rows = (int64_t)1 << 31;        // one launch per chunk: fewer than 2^32 threads
    hipStream_t st = nullptr;
    hipEvent_t e0, e1;
    BWAMS_HIP(hipEventCreate(&e0));
    BWAMS_HIP(hipEventCreate(&e1));
    BWAMS_HIP(hipEventRecord(e0, st));

    // ---- text: fw || rc, packed form, base counts
    DevBuf ref, packed, small;
    BWAMS_HIP(ref.alloc((size_t)N + 64));
    const int64_t nwords = (N >> 5) + 3;
    BWAMS_HIP(packed.alloc((size_t)nwords * 8));
    BWAMS_HIP(small.alloc(8 * 8 + (size_t)kBins * 8));
    unsigned long long *d_cnt = small.as<unsigned long long>();          // [0..3] base counts, [4] bad codes, [5] cursor
    unsigned long long *d_hist = d_cnt + 8;
    BWAMS_HIP(hipMemsetAsync(small.p, 0, 8 * 8 + (size_t)kBins * 8, st));
    BWAMS_HIP(hipMemsetAsync(ref.as<uint8_t>() + N, 0, 64, st));
    hipLaunchKernelGGL(fwrc_kernel, dim3(std::min<int64_t>(nblk(l_pac, 1024), 8192)), dim3(1024), 0, st, d_fw, l_pac, ref.as<uint8_t>(),
                       d_cnt, d_cnt + 4);
    hipLaunchKernelGGL(pack_text_kernel, dim3(nblk(nwords, kBlk)), dim3(kBlk), 0, st, ref.as<uint8_t>(), N, packed.as<uint64_t>(), nwords);
    hipLaunchKernelGGL(key_hist_kernel, dim3(4096), dim3(1024), 0, st, packed.as<uint64_t>(), N, d_hist);
    BWAMS_HIP(hipGetLastError());
    std::vector<unsigned long long> h_small(8 + kBins);
    BWAMS_HIP(hipMemcpyAsync(h_small.data(), small.p, h_small.size() * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    if (h_small[4]) {
        set_last_error("bwams_index_build: the sequence holds codes > 3 (replace N before indexing, as bns_fasta2bntseq does)");
        return BWAMS_ERR_ARG;
    }
    int64_t count[5];
    {
        const int64_t a = (int64_t)h_small[0], c = (int64_t)h_small[1], g = (int64_t)h_small[2], t = (int64_t)h_small[3];
        const int64_t tot[4] = {a + t, c + g, g + c, t + a};          // fw + its reverse complement
        count[0] = 0;
        for (int k = 0; k < 4; ++k) count[k + 1] = count[k] + tot[k];
    }

    // ---- chunks of the key space
    std::vector<uint32_t> cut{0};
    {
        int64_t acc = 0;
        for (int b = 0; b < kBins; ++b) {
            const int64_t c = (int64_t)h_small[8 + b];
            if (c > chunk_rows) {
                set_last_error("bwams_index_build: one 7-base prefix alone exceeds chunk_rows suffixes (degenerate text); raise chunk_rows");
                return BWAMS_ERR_UNSUPPORTED;
            }
            if (acc + c > chunk_rows) { cut.push_back((uint32_t)b); acc = 0; }
            acc += c;
        }
        cut.push_back((uint32_t)kBins);
    }
    int64_t max_chunk = 0;
    for (size_t k = 0; k + 1 < cut.size(); ++k) {
        int64_t c = 0;
        for (uint32_t b = cut[k]; b < cut[k + 1]; ++b) c += (int64_t)h_small[8 + b];
        max_chunk = std::max(max_chunk, c);
    }

    DevBuf sa, isa, bm;
    const int64_t bm_words = (L >> 6) + 2;
    BWAMS_HIP(sa.alloc((size_t)L * 8));
    BWAMS_HIP(isa.alloc((size_t)(L + 1) * 8));
    BWAMS_HIP(bm.alloc((size_t)bm_words * 8));
    BWAMS_HIP(hipMemsetAsync(bm.p, 0, (size_t)bm_words * 8, st));

    DevBuf k0, k1, v0, v1, tmp;
    size_t tmp_bytes = 0;
    auto need_tmp = [&](size_t b) -> hipError_t {
        if (b <= tmp_bytes) return hipSuccess;
        tmp_bytes = b + b / 8 + 256;
        return tmp.alloc(tmp_bytes);
    };
    {
        BWAMS_HIP(k0.alloc((size_t)max_chunk * 8));
        BWAMS_HIP(k1.alloc((size_t)max_chunk * 8));
        BWAMS_HIP(v0.alloc((size_t)max_chunk * 8));
        BWAMS_HIP(v1.alloc((size_t)max_chunk * 8));
        int64_t base = 0;
        for (size_t k = 0; k + 1 < cut.size(); ++k) {
            int64_t cnt = 0;
            for (uint32_t b = cut[k]; b < cut[k + 1]; ++b) cnt += (int64_t)h_small[8 + b];
            if (!cnt) continue;
            BWAMS_HIP(hipMemsetAsync(d_cnt + 5, 0, 8, st));
            hipLaunchKernelGGL(key_collect_kernel, dim3(4096), dim3(1024), 0, st, packed.as<uint64_t>(), N, cut[k], cut[k + 1],
                               k0.as<uint64_t>(), v0.as<int64_t>(), d_cnt + 5);
            BWAMS_HIP(hipGetLastError());
            rocprim::double_buffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
            rocprim::double_buffer<int64_t> dv(v0.as<int64_t>(), v1.as<int64_t>());
            size_t tb = 0;
            // the chunk's keys differ from bit 0 (length field) up to the top of the bin field
            BWAMS_HIP(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, (size_t)cnt, 0u, 64u, st));
            BWAMS_HIP(need_tmp(tb));
            tb = tmp_bytes;
            BWAMS_HIP(rocprim::radix_sort_pairs(tmp.p, tb, dk, dv, (size_t)cnt, 0u, 64u, st));
            // rank of every element = row of its group's head: inclusive max-scan of the head rows, into the free key buffer
            int64_t *rank = reinterpret_cast<int64_t *>(dk.alternate());
            auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), HeadPos{dk.current(), base});
            tb = 0;
            BWAMS_HIP(rocprim::inclusive_scan(nullptr, tb, in, rank, (size_t)cnt, MaxOp(), st));
            BWAMS_HIP(need_tmp(tb));
            tb = tmp_bytes;
            BWAMS_HIP(rocprim::inclusive_scan(tmp.p, tb, in, rank, (size_t)cnt, MaxOp(), st));
            hipLaunchKernelGGL(chunk_finish_kernel, dim3(nblk(cnt, kBlk)), dim3(kBlk), 0, st, dk.current(), dv.current(), rank, cnt, base,
                               sa.as<int64_t>(), isa.as<int64_t>(), bm.as<unsigned long long>());
            BWAMS_HIP(hipGetLastError());
            base += cnt;
            // the buffers are reused by the next chunk in their original roles
            BWAMS_HIP(hipStreamSynchronize(st));
        }
        if (base != L) {
            set_last_error("bwams_index_build: internal error, chunks do not cover the text");
            return BWAMS_ERR_DEVICE;
        }
        k0.release(); k1.release(); v0.release(); v1.release();
    }
    BWAMS_HIP(hipEventRecord(e1, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    float ms_first = 0;
    BWAMS_HIP(hipEventElapsedTime(&ms_first, e0, e1));

    // ---- doubling rounds
    DevBuf wcnt, woff, upos, gsum;
    BWAMS_HIP(wcnt.alloc((size_t)bm_words * 8));
    BWAMS_HIP(woff.alloc((size_t)(bm_words + 1) * 8));
    int64_t cap_m = 0;
    int rounds = 0;
    int64_t first_m = 0;
    const int64_t nw = (L + 63) >> 6;
    const int rbits = bits_for((uint64_t)L);
    for (int64_t h = kKeyBases;; h *= 2) {
        hipLaunchKernelGGL(unres_count_kernel, dim3(nblk(nw, kBlk)), dim3(kBlk), 0, st, bm.as<unsigned long long>(), nw, L, wcnt.as<int64_t>());
        size_t tb = 0;
        BWAMS_HIP(rocprim::exclusive_scan(nullptr, tb, wcnt.as<int64_t>(), woff.as<int64_t>(), (int64_t)0, (size_t)nw + 1,
                                          rocprim::plus<int64_t>(), st));
        BWAMS_HIP(need_tmp(tb));
        tb = tmp_bytes;
        // (one element past the counts is read: wcnt has bm_words >= nw + 1 entries; its value does not matter)
        BWAMS_HIP(rocprim::exclusive_scan(tmp.p, tb, wcnt.as<int64_t>(), woff.as<int64_t>(), (int64_t)0, (size_t)nw + 1,
                                          rocprim::plus<int64_t>(), st));
        int64_t M = 0;
        BWAMS_HIP(hipMemcpyAsync(&M, woff.as<int64_t>() + nw, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        if (verbose) fprintf(stderr, "[bwams_index_build] h = %lld: %lld unresolved rows\n", (long long)h, (long long)M);
        if (M == 0) break;
        if (rounds == 0) first_m = M;
        if (M >= ((int64_t)1 << 32)) {
            set_last_error("bwams_index_build: more than 2^32 rows tied after a pass (degenerate text)");
            return BWAMS_ERR_UNSUPPORTED;
        }
        if (h > 2 * N + 64) {
            set_last_error("bwams_index_build: internal error, suffixes still tied beyond the text length");
            return BWAMS_ERR_DEVICE;
        }
        ++rounds;
        if (M > cap_m) {
            cap_m = M;
            BWAMS_HIP(upos.alloc((size_t)M * 8));
            BWAMS_HIP(gsum.alloc((size_t)M * 8));
            BWAMS_HIP(k0.alloc((size_t)M * 8));
            BWAMS_HIP(k1.alloc((size_t)M * 8));
            BWAMS_HIP(v0.alloc((size_t)M * 8));
            BWAMS_HIP(v1.alloc((size_t)M * 8));
        }
        // rows of the unresolved groups (ascending) and their head flags (into v1, summed into gsum)
        hipLaunchKernelGGL(unres_fill_kernel, dim3(nblk(nw, kBlk)), dim3(kBlk), 0, st, bm.as<unsigned long long>(), nw, L, woff.as<int64_t>(),
                           upos.as<int64_t>(), v1.as<int64_t>());
        tb = 0;
        BWAMS_HIP(rocprim::inclusive_scan(nullptr, tb, v1.as<int64_t>(), gsum.as<int64_t>(), (size_t)M, rocprim::plus<int64_t>(), st));
        BWAMS_HIP(need_tmp(tb));
        tb = tmp_bytes;
        BWAMS_HIP(rocprim::inclusive_scan(tmp.p, tb, v1.as<int64_t>(), gsum.as<int64_t>(), (size_t)M, rocprim::plus<int64_t>(), st));
        int64_t n_groups = 0;
        BWAMS_HIP(hipMemcpyAsync(&n_groups, gsum.as<int64_t>() + (M - 1), 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        const int gbits = bits_for((uint64_t)(n_groups > 0 ? n_groups - 1 : 0));
        if (rbits + gbits > 64) {
            set_last_error("bwams_index_build: too many tied groups for a 64-bit (group, rank) key");
            return BWAMS_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(round_keys_kernel, dim3(nblk(M, kBlk)), dim3(kBlk), 0, st, upos.as<int64_t>(), gsum.as<int64_t>(), M,
                           sa.as<int64_t>(), isa.as<int64_t>(), h, N, rbits, k0.as<uint64_t>(), v0.as<int64_t>());
        BWAMS_HIP(hipGetLastError());
        rocprim::double_buffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
        rocprim::double_buffer<int64_t> dv(v0.as<int64_t>(), v1.as<int64_t>());
        tb = 0;
        BWAMS_HIP(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, (size_t)M, 0u, (unsigned)(rbits + gbits), st));
        BWAMS_HIP(need_tmp(tb));
        tb = tmp_bytes;
        BWAMS_HIP(rocprim::radix_sort_pairs(tmp.p, tb, dk, dv, (size_t)M, 0u, (unsigned)(rbits + gbits), st));
        int64_t *rank = reinterpret_cast<int64_t *>(dk.alternate());
        auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), RoundHeadPos{dk.current(), upos.as<int64_t>()});
        tb = 0;
        BWAMS_HIP(rocprim::inclusive_scan(nullptr, tb, in, rank, (size_t)M, MaxOp(), st));
        BWAMS_HIP(need_tmp(tb));
        tb = tmp_bytes;
        BWAMS_HIP(rocprim::inclusive_scan(tmp.p, tb, in, rank, (size_t)M, MaxOp(), st));
        hipLaunchKernelGGL(round_finish_kernel, dim3(nblk(M, kBlk)), dim3(kBlk), 0, st, dk.current(), dv.current(), upos.as<int64_t>(), rank,
                           M, sa.as<int64_t>(), isa.as<int64_t>(), bm.as<unsigned long long>());
        BWAMS_HIP(hipGetLastError());
        BWAMS_HIP(hipStreamSynchronize(st));
    }
    upos.release(); gsum.release(); k0.release(); k1.release(); v0.release(); v1.release(); wcnt.release(); woff.release();
    bm.release(); packed.release();
    int64_t sentinel = -1;
    BWAMS_HIP(hipMemcpy(&sentinel, isa.as<int64_t>(), 8, hipMemcpyDeviceToHost));         // row of suffix 0
    isa.release();
    BWAMS_HIP(hipEventRecord(e0, st));

    // ---- BWT -> CP_OCC, SA samples
    const int64_t n_blk = (L >> 6) + 1, filled = (L + 63) >> 6, n_sa = (L >> 3) + 1;
    DevBuf cp, ms, ls, blkcnt;
    BWAMS_HIP(cp.alloc((size_t)n_blk * 64));
    BWAMS_HIP(ms.alloc((size_t)n_sa));
    BWAMS_HIP(ls.alloc((size_t)n_sa * 4));
    BWAMS_HIP(blkcnt.alloc((size_t)(filled + 1) * sizeof(Cnt4)));
    BWAMS_HIP(hipMemsetAsync(cp.p, 0, (size_t)n_blk * 64, st));
    hipLaunchKernelGGL(bwt_block_kernel, dim3((unsigned)std::min<int64_t>(nblk(filled * 64, kBlk), 1 << 20)), dim3(kBlk), 0, st, sa.as<int64_t>(), ref.as<uint8_t>(), L, filled,
                       cp.as<uint64_t>(), blkcnt.as<Cnt4>());
    {
        size_t tb = 0;
        Cnt4 zero{{0, 0, 0, 0}};
        BWAMS_HIP(rocprim::exclusive_scan(nullptr, tb, blkcnt.as<Cnt4>(), blkcnt.as<Cnt4>(), zero, (size_t)filled, Cnt4Add(), st));
        BWAMS_HIP(need_tmp(tb));
        tb = tmp_bytes;
        BWAMS_HIP(rocprim::exclusive_scan(tmp.p, tb, blkcnt.as<Cnt4>(), blkcnt.as<Cnt4>(), zero, (size_t)filled, Cnt4Add(), st));
    }
    hipLaunchKernelGGL(cp_counts_kernel, dim3(nblk(filled * 4, kBlk)), dim3(kBlk), 0, st, blkcnt.as<Cnt4>(), filled, cp.as<uint64_t>());
    hipLaunchKernelGGL(sa_sample_kernel, dim3(nblk(n_sa, kBlk)), dim3(kBlk), 0, st, sa.as<int64_t>(), L, n_sa, ms.as<int8_t>(), ls.as<uint32_t>());
    BWAMS_HIP(hipGetLastError());
    BWAMS_HIP(hipEventRecord(e1, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    float ms_out = 0;
    BWAMS_HIP(hipEventElapsedTime(&ms_out, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    sa.release(); blkcnt.release(); tmp.release();

    ix->owns = true;
    ix->n_blk = n_blk;
    ix->n_sa = n_sa;
    ix->d_cp = cp.p; cp.p = nullptr;
    ix->d_ms = ms.p; ms.p = nullptr;
    ix->d_ls = ls.p; ls.p = nullptr;
    if (keep_ref) { ix->d_ref = ref.p; ref.p = nullptr; }
    ix->bytes = n_blk * 64 + n_sa * 5 + (keep_ref ? N : 0);
    ix->fmi.cp = reinterpret_cast<const uint4 *>(ix->d_cp);
    ix->fmi.sa_ms = reinterpret_cast<const int8_t *>(ix->d_ms);
    ix->fmi.sa_ls = reinterpret_cast<const uint32_t *>(ix->d_ls);
    ix->fmi.ref = reinterpret_cast<const uint8_t *>(ix->d_ref);
    for (int k = 0; k < 5; ++k) ix->fmi.count[k] = count[k] + 1;           // as the loader leaves them (FMI_search.cpp:880-883)
    ix->fmi.sentinel = sentinel;
    ix->fmi.ref_seq_len = L;
    if (bs) {
        bs->rows = L;
        bs->chunks = (int32_t)(cut.size() - 1);
        bs->rounds = rounds;
        bs->unresolved_after_first = first_m;
        bs->ms_first_pass = ms_first;
        bs->ms_outputs = ms_out;
    }
    return BWAMS_OK;
}

}  // namespace bwams
