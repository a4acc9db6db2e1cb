// ert_seed.hip — seeding over the ERT index (enumerated radix trees) for gfx950.
//
// Replaces the per-read block of mem_kernel1_core_ert (/root/reference/src/bwamem.cpp:1122-1193: get_seeds[_prefix],
// reseed[_prefix], last, ks_introsort) and the hit sampling of mem_chain_new (:993-1004).  The index bytes are the
// reference's own (<prefix>.kmer_table, <prefix>.mlt_table as src/ertindex.cpp writes them; entry and node fields
// as src/ertseeding.cpp:2142-2305, :836-975, :485-497, :521-587 read them).
//
// The reference walks a read serially: one forward walk from a pivot marks, in a bit vector (LEP), the end positions
// where the hit set shrinks, and a backward walk (the same trees, reverse-complemented read) is started from each
// marked end.  That order is a chain of dependent, divergent pointer chases per read.  On the GPU every start position
// of every read gets its own lane and one forward walk (about 150 independent walks per read), which records
//     L_m(i) = the longest prefix of read[i..) that occurs at least m times,  m = 1 .. M (M <= 20),
// m bounded by what the trees store: child pointers carry the hit count of their subtree while it is below 20
// (src/ertindex.cpp:455-461), which is all reseeding (min_intv <= split_width + 1) and `last` (max_mem_intv) ask.
// The three seeding rounds then are arithmetic on those profiles (ert_select_kernel):
//     round 1  [i, i+L_1(i)) is an SMEM iff L_1(i) >= min_seed_len and it ends behind the match from i-1;
//     round 2  for an SMEM with <= split_width hits and >= split_len bases: the same test on L_m, m = hits + 1,
//              for the matches that cover the SMEM's middle;
//     round 3  from x, the first length >= min_seed_len + 1 with fewer than max_mem_intv hits; continue behind it.
// The result is what mem_collect_smem gives over the FM-index (the reference's ERT mode is written to reproduce it:
// src/ertseeding.cpp:2891, :3444-3461), so the seeds go through the same sort and the chaining stage unchanged.
// Hits come out of the trees in A, C, G, T order = suffix order, as the SA interval would list them.
#include "fmi_kernels.h"
#include "ert_kernels.h"
#include "wave_ops.h"

namespace bwams {
namespace {

enum { N_EMPTY = 0, N_LEAF = 1, N_UNIFORM = 2, N_DIVERGE = 3 };          // node_type_t, ertindex.h:12
enum { E_INVALID = 0, E_SINGLE = 1, E_INFREQUENT = 2, E_FREQUENT = 3 };  // macro.h:216-219
constexpr int kMany = 255;           // 20 hits or more: the trees do not say how many
constexpr int kCntSlots = 4096;      // partial event counters of the walk kernel
constexpr int kCoopRounds = 6;       // diagonals a wave expands cooperatively before the lanes go on alone

// n <= 8 bytes at any address, little endian: one unaligned 8-byte load (gfx950 global loads need no alignment; the
// tables, the reads and the text are padded so that the 8 bytes exist)
__device__ __forceinline__ uint64_t ld8(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint64_t ld_le(const uint8_t *p, int n) {
    const uint64_t v = ld8(p);
    return n == 8 ? v : v & ((1ull << (8 * n)) - 1);
}

// eight base codes (one per byte, first base in the low byte) -> 16 bits, first base lowest
__device__ __forceinline__ uint64_t squeeze8(uint64_t v) {
    v &= 0x0303030303030303ull;
    v = (v | (v >> 6)) & 0x000F000F000F000Full;
    v = (v | (v >> 12)) & 0x000000FF000000FFull;
    return (v | (v >> 24)) & 0xFFFFull;
}

struct WalkCnt { uint32_t kmer, nodes, ref; };      // k-mer entries read, tree records decoded, text bytes compared

struct Pend { int64_t leaf_pos; int d, cur; };      // a walk that reached a leaf: text position of the match start, depth, hits

struct Where {
    int kind;          // 0 nothing, 1 one position, 2 multi-hit list at `at`, 3 subtree at `at`
    int w;
    int64_t at, root;
};

// One forward walk from read position i (oracle: ert_walk).  PROFILE: store L_m into the planes; otherwise stop at
// stop_len and describe where the hits of read[i, i+stop_len) are.  Returns the matched length.
// the profile record of a read position: byte 0 = the base is N, byte m = L_m, m = 1 .. M <= 23, in three 64-bit words.  L_m is a step
// function of m and every m is set once (the words start at zero): a range of m takes a mask, not a loop of byte stores
constexpr int kProfRec = 24;
__device__ __forceinline__ void prof_set(uint64_t *acc, int lo, int hi, int d) {                   // bytes lo .. hi (inclusive) = d
    const uint64_t v = 0x0101010101010101ull * (uint64_t)(uint8_t)d;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = lo > 8 * k ? lo - 8 * k : 0, b = hi < 8 * k + 7 ? hi - 8 * k : 7;
        if (a <= b) acc[k] |= v & ((~0ull >> (8 * (7 - b))) & (~0ull << (8 * a)));
    }
}
template <bool PROFILE>
__device__ int ert_walk(const DevErt &e, const uint8_t *__restrict__ q, int len, int i, int M, uint64_t *__restrict__ acc,
                        int stop_len, Where *wh, WalkCnt &wc, Pend *pend = nullptr) {
    const int K = e.K, X = e.X;
    if (i + K > len) return 0;
    // 16 bases as two 8-byte loads (the read buffer is padded); 2-bit codes gathered first base lowest, as getHashKey does
    const uint64_t b0 = ld_le(q + i, 8), b1 = ld_le(q + i + 8, 8);
    const uint64_t nmask = K >= 8 ? (K >= 16 ? ~0ull : ((1ull << (8 * (K - 8))) - 1)) : 0ull;
    const uint64_t n0 = K >= 8 ? ~0ull : ((1ull << (8 * K)) - 1);
    if (((b0 & n0) | (b1 & nmask)) & 0xFCFCFCFCFCFCFCFCull) return 0;
    const uint64_t key = (squeeze8(b0) | (squeeze8(b1) << 16)) & ((1ull << (2 * K)) - 1);
    // with the entry + tree-head table the entry and the first 56 bytes of the k-mer's tree are one line (at() below)
    const uint8_t *__restrict__ fe = e.fat ? e.fat + (key << 6) : nullptr;
    const uint64_t ent = fe ? ld8(fe) : e.kmer[key];
    wc.kmer++;
    int code = (int)(ent & 3);
    if (code == E_INVALID) return 0;
    const int64_t root = (int64_t)(ent >> 24);
    const int w = ((ent >> 22) & 3) == 0 ? 4 : (int)((ent >> 22) & 3);
    int cur = (int)((ent >> 17) & 31);
    if (cur == 0) cur = kMany;
    int d = K;
    int64_t node = -1, leaf_pos = -1, mh_at = -1;
    const uint8_t *__restrict__ mlt = e.mlt;
    // eight bytes of the tree table at offset a: from the k-mer's line when they lie within the copied head of its tree
    auto at = [&](int64_t a) -> const uint8_t * {
        const uint64_t o = (uint64_t)(a - root);
        return (fe && o <= 48) ? fe + 8 + o : mlt + a;
    };
    if (code == E_SINGLE) {
        leaf_pos = (int64_t)(ld_le(at(root + 1), 5) >> 1);
        cur = 1;
    } else if (code == E_INFREQUENT) {
        node = root + 4;
    } else {
        if (i + K + X > len) return 0;
        uint32_t xk = 0;
        for (int j = 0; j < X; ++j) {
            const uint32_t b = q[i + K + j];
            if (b > 3) return 0;
            xk |= b << (2 * j);
        }
        const uint64_t xe = ld_le(at(root + 4 + 8 * (int64_t)xk), 8);
        wc.nodes++;
        code = (int)(xe & 3);
        if (code == E_INVALID) return 0;
        d = K + X;
        cur = (int)((xe >> 17) & 31);
        if (cur == 0) cur = kMany;
        if (code == E_SINGLE) {
            leaf_pos = (int64_t)(ld_le(at(root + (int64_t)(xe >> 24) + 1), 5) >> 1);
            cur = 1;
        } else {
            node = root + (int64_t)(xe >> 24);
        }
    }
    int64_t mh_base = -1;
    auto drop_to = [&](int nc) {
        if (PROFILE) {
            const int hi = cur < M ? cur : M;
            if (nc + 1 <= hi) prof_set(acc, nc + 1, hi, d);
        }
        cur = nc;
    };
    while (leaf_pos < 0) {
        if (!PROFILE && d >= stop_len) break;
        if (i + d >= len) break;
        const uint32_t b = q[i + d];
        if (b > 3) break;
        const int c = 3 - (int)b;
        const uint64_t head = ld_le(at(node), 8);       // code byte and the first 7 bytes behind it
        wc.nodes++;
        const uint32_t cd = (uint32_t)(head & 0xff);
        const int t = (cd >> (c << 1)) & 3;
        if (t == N_EMPTY) break;
        if (t == N_UNIFORM) {
            const int nbp = (int)((head >> 8) & 0xff);
            // the run, eight bases per step: the run's bytes hold four bases each, first base in the top bits, as 3 - base;
            // the read's eight bytes are squeezed to 2-bit codes, first base in the low bits
            int lim_j = nbp;
            if (!PROFILE && stop_len - d < lim_j) lim_j = stop_len - d;
            if (len - i - d < lim_j) lim_j = len - i - d;
            int j = 0;
            while (j < lim_j) {
                const uint64_t q8 = ld8(q + i + d + j);
                const uint32_t rb = j < 24 ? (uint32_t)((head >> (16 + 2 * j)) & 0xffff)      // bytes 2.. of the head word (j is a multiple of 8)
                                           : (uint32_t)ld_le(at(node + 2 + (j >> 2)), 2);
                // reverse the four 2-bit groups of each byte, then complement: codes of eight bases, first base lowest
                uint32_t r = ((rb & 0x0303u) << 6) | ((rb & 0x0C0Cu) << 2) | ((rb & 0x3030u) >> 2) | ((rb & 0xC0C0u) >> 6);
                r = ~r & 0xffffu;
                const uint32_t x = ((uint32_t)squeeze8(q8) ^ r) & 0xffffu;
                const uint64_t nn = q8 & 0xFCFCFCFCFCFCFCFCull;
                int ok = x ? (__builtin_ctz(x) >> 1) : 8;
                const int okn = nn ? (__builtin_ctzll(nn) >> 3) : 8;
                if (okn < ok) ok = okn;
                if (ok > lim_j - j) ok = lim_j - j;
                j += ok;
                if (ok < 8) break;
            }
            d += j;
            node = node + 2 + ((nbp + 3) >> 2);
            if (j < nbp) break;
            continue;
        }
        // types of the four children: pointers first, then the leaf records, both in A, C, G, T order (c = 3 .. 0)
        const uint32_t is_div = (cd & (cd >> 1)) & 0x55, is_leaf = (cd & ~(cd >> 1)) & 0x55;
        const uint32_t above = c == 3 ? 0u : (0xffu << ((c + 1) << 1)) & 0xff;
        const int n_ptr = __popc(is_div), before_ptr = __popc(is_div & above), before_leaf = __popc(is_leaf & above);
        if (t == N_LEAF) {
            const uint64_t rec = ld_le(at(node + 1 + n_ptr * w + 5 * before_leaf), 5);
            wc.nodes++;
            int nc = 1;
            if (rec & 1) {
                if (mh_base < 0) mh_base = root + (int64_t)ld_le(at(root), 4);
                mh_at = mh_base + (int64_t)(rec >> 1);
                const uint64_t h = ld_le(at(mh_at), 7);
                nc = (int)(h & 0xffff);
                leaf_pos = (int64_t)((h >> 16) >> 1);
                if (nc >= 20) nc = kMany;
            } else {
                leaf_pos = (int64_t)(rec >> 1);
            }
            drop_to(nc);
            d += 1;
        } else {
            const uint64_t v = ld_le(at(node + 1 + before_ptr * w), w);
            int nc = (int)(v & 63);
            if (nc == 0) nc = kMany;
            drop_to(nc);
            d += 1;
            node = node + (int64_t)(v >> 6);
        }
    }
    if (PROFILE && pend) {              // the caller expands the leaf (wave-cooperatively) and stores the final lengths
        pend->leaf_pos = leaf_pos; pend->d = d; pend->cur = cur;
        return d;
    }
    if (leaf_pos >= 0) {
        // the rest of the suffix is not in the tree: compare with the text (get_seeds_prefix :2940-2965)
        const int lim = PROFILE ? len - i : (stop_len < len - i ? stop_len : len - i);
        const uint8_t *__restrict__ rf = e.ref + leaf_pos;
        const int64_t room = e.ref_len - leaf_pos;
        const int stop = (int64_t)lim < room ? lim : (int)room;
        const int d_from = d;
        bool open = true;
        while (d + 8 <= stop) {                    // eight bases per step; an N in the read (4) never equals a text base
            const uint64_t x = ld_le(rf + d, 8) ^ ld_le(q + i + d, 8);
            if (x) { d += __builtin_ctzll(x) >> 3; open = false; break; }
            d += 8;
        }
        while (open && d < stop) {
            if (rf[d] != q[i + d]) break;
            d++;
        }
        wc.ref += (uint32_t)(d - d_from + (d < stop));
    }
    if (PROFILE) {
        const int hi = cur < M ? cur : M;
        if (hi >= 1) prof_set(acc, 1, hi, d);
    } else {
        wh->root = root; wh->w = w;
        if (leaf_pos >= 0 && mh_at >= 0) { wh->kind = 2; wh->at = mh_at; }
        else if (leaf_pos >= 0) { wh->kind = 1; wh->at = leaf_pos; }
        else { wh->kind = 3; wh->at = node; }
    }
    return d;
}

// entry + tree head, 64 bytes per k-mer: {the table entry, the first 56 bytes of its tree (what lies at its pointer; zero past the table)}.
// The trees lie in k-mer order, so neighbouring threads read neighbouring bytes.
__global__ __launch_bounds__(256) void ert_fat_kernel(const uint64_t *__restrict__ kmer, const uint8_t *__restrict__ mlt, int64_t mlt_bytes,
                                                      int64_t n, uint8_t *__restrict__ fat) {
    for (int64_t key = (int64_t)blockIdx.x * 256 + threadIdx.x; key < n; key += (int64_t)gridDim.x * 256) {
        const uint64_t ent = kmer[key];
        uint64_t w[8];
        w[0] = ent;
        const int64_t root = (int64_t)(ent >> 24);
        for (int k = 0; k < 7; ++k) {
            const int64_t a = root + 8 * k;
            w[k + 1] = ((ent & 3) != E_INVALID && a + 8 <= mlt_bytes + 16) ? ld8(mlt + a) : 0ull;      // the table is padded by 16 bytes
        }
        uint4 *o = reinterpret_cast<uint4 *>(fat + (key << 6));
        for (int k = 0; k < 4; ++k) o[k] = make_uint4((uint32_t)w[2 * k], (uint32_t)(w[2 * k] >> 32), (uint32_t)w[2 * k + 1], (uint32_t)(w[2 * k + 1] >> 32));
    }
}

constexpr int kErtTicket = 4;
// lane = one base of the batch = one start position of one read.  prof: a 24-byte record per position ([0] = the base is N, [m] = L_m),
// written whole (round 3: (M + 1) byte planes, a byte store per m and lane — 158 M store instructions and 7 GB of partial lines per launch).
// Persistent waves, 64 consecutive bases per trip: a walk lives for some ten microseconds, and a grid of one short-lived
// workgroup per 256 bases kept only five waves per CU in flight (workgroup launch rate), where the walk needs dozens
// of them to cover its chain of dependent HBM reads.
__global__ __launch_bounds__(256) void ert_profile_kernel(DevErt e, const uint8_t *__restrict__ enc,
                                                          const int64_t *__restrict__ cum, const uint8_t *__restrict__ skip,
                                                          int64_t nseq, int64_t nbases, int M, uint8_t *__restrict__ prof,
                                                          unsigned long long *__restrict__ part, unsigned long long *__restrict__ ticket) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    WalkCnt wc = {0, 0, 0};
    // work: groups of 64 consecutive read positions; with a cursor (ticket != null) a wave takes kErtTicket groups per atomic,
    // otherwise the groups are dealt out round robin
    int64_t g_next = ticket ? 0 : wave * 64, g_end = 0;
    for (;;) {
        int64_t g0;
        if (ticket) {
            if (g_next >= g_end) {
                g_next = (int64_t)wave_ticket(ticket, (unsigned long long)kErtTicket) * 64;
                g_end = g_next + (int64_t)kErtTicket * 64;
            }
            g0 = g_next; g_next += 64;
            if (g0 >= nbases) break;
        } else {
            g0 = g_next; g_next += n_waves * 64;
            if (g0 >= nbases) break;
        }
        {
        int64_t r = 0;
        if (lane == 0) {
            // last r with cum[r] <= g0, one search per wave: reads are about equally long, so look next to the proportional
            // guess first and bisect only what is left
            int64_t lo = 0, hi = nseq;
            const int64_t gs = (int64_t)((double)g0 * (double)nseq / (double)nbases);
            int64_t a = gs - 2 < 0 ? 0 : gs - 2, b = gs + 3 > nseq ? nseq : gs + 3;
            if (cum[a] <= g0) lo = a; else hi = a;
            if (hi > b) { if (cum[b] <= g0) lo = b; else hi = b; }
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (cum[mid] <= g0) lo = mid; else hi = mid;
            }
            r = lo;
        }
        r = ((int64_t)__builtin_amdgcn_readfirstlane((uint32_t)(r >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)r);
        const int64_t g = g0 + lane;
        Pend pd = {-1, 0, 0};
        uint64_t acc[3] = {0ull, 0ull, 0ull};
        int64_t c0 = 0;
        int len = 0, i = 0;
        if (g < nbases) {
            while (r + 1 < nseq && g >= cum[r + 1]) r++;
            c0 = cum[r];
            len = (int)(cum[r + 1] - c0); i = (int)(g - c0);
            const uint8_t *q = enc + c0;
            acc[0] = q[i] > 3 ? 1ull : 0ull;
            if (!(skip && skip[r])) ert_walk<true>(e, q, len, i, M, acc, 0, nullptr, wc, &pd);
        }
        // ---- leaf expansion: the rest of the suffix is not in the tree, compare with the text (get_seeds_prefix :2940-2965)
        bool open = pd.leaf_pos >= 0;
        int d = pd.d;
        const int64_t D = pd.leaf_pos - i;                    // text position of read position 0 on this lane's diagonal
        auto probe = [&]() {                                 // eight bases; an N in the read (4) never equals a text base
            const int64_t room = e.ref_len - pd.leaf_pos;
            const int stop = (int64_t)(len - i) < room ? len - i : (int)room;
            const int n = stop - d;
            if (n <= 0) { open = false; return; }
            uint64_t x = ld8(e.ref + pd.leaf_pos + d) ^ ld8(enc + c0 + i + d);
            if (n < 8) x |= 0xFFull << (8 * n);
            wc.ref += n < 8 ? n : 8;
            if (x) { d += __builtin_ctzll(x) >> 3; open = false; } else d += 8;
        };
        if (open) probe();
        // Lanes on a long match mostly share their diagonal (consecutive read positions, one locus): the wave compares that
        // diagonal once, 64 bases per load, and every lane of the group reads its answer from the mismatch bits.  A lane
        // for lane compare costs one divergent load per lane and 8 bases, and the address unit serialises those.
        for (int round = 0; round < kCoopRounds; ++round) {
            const uint64_t todo = __ballot(open);
            if (!todo) break;
            const int leader = __builtin_ctzll(todo);
            const int64_t Dl = ((int64_t)__builtin_amdgcn_readlane((int)(D >> 32), leader) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)D, leader);
            const int64_t cl = ((int64_t)__builtin_amdgcn_readlane((int)(c0 >> 32), leader) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)c0, leader);
            const int lenl = __builtin_amdgcn_readlane(len, leader);
            uint64_t m0 = ~0ull, m1 = ~0ull, m2 = ~0ull, m3 = ~0ull;
            const int nch = (lenl + 63) >> 6;
            for (int k = 0; k < nch && k < 4; ++k) {
                const int j = 64 * k + lane;
                const int64_t pos = Dl + j;
                bool mm = true;
                if (j < lenl && pos >= 0 && pos < e.ref_len) mm = e.ref[pos] != enc[cl + j];
                const uint64_t bits = __ballot(mm);
                if (k == 0) m0 = bits; else if (k == 1) m1 = bits; else if (k == 2) m2 = bits; else m3 = bits;
            }
            if (lane == 0) wc.ref += 64 * (nch < 4 ? nch : 4);
            if (open && D == Dl && c0 == cl) {
                const int x = i + d;                         // first mismatch at or behind read position x
                int k = x >> 6;
                uint64_t m = (k == 0 ? m0 : k == 1 ? m1 : k == 2 ? m2 : m3) & (~0ull << (x & 63));
                while (m == 0 && k < 3) { k++; m = k == 1 ? m1 : k == 2 ? m2 : m3; }
                d = (m ? 64 * k + __builtin_ctzll(m) : 256) - i;
                if (d > len - i) d = len - i;
                open = false;
            }
        }
        while (open) probe();                                // more diagonals in the wave than rounds: each lane for itself
        if (pd.leaf_pos >= 0 || pd.cur > 0) {
            const int hi = pd.cur < M ? pd.cur : M;
            if (hi >= 1) prof_set(acc, 1, hi, d);
        }
        if (g < nbases) {                                    // the position's record, whole: 24 bytes, neighbours adjacent
            uint64_t *o = reinterpret_cast<uint64_t *>(prof + g * kProfRec);
            o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
        }
        }
    }
    // event counts of the launch (8 B per k-mer entry, a 32-B sector per tree record, the text bytes)
    uint32_t a = wc.kmer, b = wc.nodes, c = wc.ref;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); c += __shfl_down(c, o); }
    if (lane == 0) {
        unsigned long long *p = part + 3 * (wave & (kCntSlots - 1));
        if (a) atomicAdd(p, (unsigned long long)a);
        if (b) atomicAdd(p + 1, (unsigned long long)b);
        if (c) atomicAdd(p + 2, (unsigned long long)c);
    }
}

__global__ void ert_count_kernel(unsigned long long *part, DevCounters *ctr) {
    __shared__ unsigned long long acc[3];
    if (threadIdx.x < 3) acc[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < kCntSlots; i += blockDim.x)
        for (int k = 0; k < 3; ++k) {
            const unsigned long long v = part[3 * i + k];
            if (v) { atomicAdd(&acc[k], v); part[3 * i + k] = 0; }
        }
    __syncthreads();
    if (threadIdx.x == 0) { ctr->ert_kmer = acc[0]; ctr->ert_nodes = acc[1]; ctr->ert_ref = acc[2]; }
}

struct SelectArgs {
    const uint8_t *prof;
    const int64_t *cum;
    const uint8_t *skip;
    int64_t nseq, nbases;
    int M, msl, split_len, split_width, max_intv;
    bwams_smem_t *pool;
    int64_t pool_cap;
    DevCounters *ctr;
};

// wave = one read at a time, lane = read position (64 positions per chunk, up to four chunks): the records of a read's
// positions are consecutive, so a wave's load covers 1.5 KB and a position's other lengths come from the same line
// (a lane per read made every load 64 lines: 5.2 ms per million reads).  Seeds are staged in LDS per wave and appended to the pool in batches.
constexpr int kSelStage = 320;        // staged seeds per wave (flushed when fewer than 64 + kSelSlack slots are free)
constexpr int kSelSlack = 64;

struct SelWave {
    uint2 *st;                        // {rid, start | len << 8 | hits << 16}
    int n;
};

__device__ __forceinline__ void sel_flush(const SelectArgs &A, SelWave &W, int lane) {
    if (W.n == 0) return;
    const unsigned long long base = wave_ticket(&A.ctr->n_smem_total, (unsigned long long)W.n);
    for (int j = lane; j < W.n; j += 64) {
        const uint2 rec = W.st[j];
        const int64_t at = (int64_t)base + j;
        if (at < A.pool_cap) {
            bwams_smem_t o;
            const int stt = rec.y & 0xff, ln = (rec.y >> 8) & 0xff, cnt = (rec.y >> 16) & 0xff;
            o.rid = rec.x; o.m = (uint32_t)stt; o.n = (uint32_t)(stt + ln - 1); o.pad_ = 0;
            o.k = 0; o.l = 0; o.s = cnt == kMany ? -1 : cnt;
            A.pool[at] = o;
        }
    }
    __builtin_amdgcn_wave_barrier();
    W.n = 0;
}
// lanes with `want` append one seed each
__device__ __forceinline__ void sel_emit(const SelectArgs &A, SelWave &W, int lane, bool want, uint32_t rid, int stt, int ln, int cnt) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    if (W.n + __popcll(m) > kSelStage) sel_flush(A, W, lane);
    if (want) W.st[W.n + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(rid, (uint32_t)stt | ((uint32_t)ln << 8) | ((uint32_t)cnt << 16));
    W.n += __popcll(m);
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void ert_select_kernel(SelectArgs A) {
    __shared__ uint2 stage[4 * kSelStage];
    const int lane = threadIdx.x & 63;
    SelWave W;
    W.st = stage + (threadIdx.x >> 6) * kSelStage;
    W.n = 0;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < A.nseq; r += n_waves) {
        if (A.skip && A.skip[r]) continue;
        const int64_t c0 = A.cum[r];
        const int len = (int)(A.cum[r + 1] - c0);
        if (len <= 0) continue;
        const uint8_t *P = A.prof + c0 * kProfRec;             // the read's records: 24 bytes per position, byte m = L_m, byte 0 = N
        const int nch = (len + 63) >> 6;                       // <= 4 (reads are at most 255 long)
        auto L = [&](int m, int i) -> int { return P[i * kProfRec + m]; };
        // hits of read[i, i+ln): the largest m with L_m(i) >= ln (L_m falls with m)
        auto count_of = [&](int i, int ln) -> int {
            int c = 1;
            for (int m = 2; m <= A.M; ++m) {
                if (L(m, i) >= ln) c = m; else break;
            }
            return c >= A.M ? kMany : c;
        };
        int l1[4], lx[4];
        unsigned long long nm[4];
        for (int k = 0; k < 4; ++k) {
            const int p = 64 * k + lane;
            const bool v = k < nch && p < len;
            l1[k] = v ? L(1, p) : 0;
            lx[k] = v && A.max_intv > 0 ? L(A.max_intv, p) : 0;
            nm[k] = __ballot(v && P[p * kProfRec] != 0);
        }
        // ---- round 1, with reseeding (bwamem.cpp:1165-1181) of the SMEMs that qualify
        int prev_e_carry = -1;                                // position - 1 + L_1 of the lane in front of this chunk
#pragma unroll
        for (int k = 0; k < 4; ++k) {                         // unrolled: l1[] stays in registers
            if (k >= nch) break;
            const int p = 64 * k + lane;
            const int e = p + l1[k];
            int prev_e = __shfl_up(e, 1);
            if (lane == 0) prev_e = prev_e_carry;
            prev_e_carry = __shfl(e, 63);
            const bool smem = p < len && l1[k] >= A.msl && (p == 0 || e > prev_e);
            int cnt = 0;
            if (smem) cnt = count_of(p, l1[k]);
            sel_emit(A, W, lane, smem, (uint32_t)r, p, l1[k], cnt);
            unsigned long long todo = __ballot(smem && l1[k] >= A.split_len && cnt <= A.split_width);
            while (todo) {
                const int ld = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int i0 = 64 * k + ld, en0 = i0 + __shfl(l1[k], ld), m = __shfl(cnt, ld) + 1;
                const int x = (i0 + en0) >> 1;                 // matches with at least m hits that cover position x
                int carry = 0;                                 // L_m of the position in front of the chunk
                for (int kk = 0; kk <= (x >> 6); ++kk) {
                    const int a = 64 * kk + lane;
                    const int la = a <= x && a < len ? L(m, a) : 0;
                    int lp = __shfl_up(la, 1);
                    if (lane == 0) lp = carry;
                    carry = __shfl(la, 63);
                    const bool ok = a <= x && la >= A.msl && a + la > x && (a == 0 || a + la > a - 1 + lp);
                    int c2 = 0;
                    if (ok) c2 = count_of(a, la);
                    sel_emit(A, W, lane, ok, (uint32_t)r, a, la, c2);
                }
            }
        }
        // ---- round 3: `last` (ertseeding.cpp:3425-3511 = bwtSeedStrategyAllPosOneThread); x is wave-uniform
        if (A.max_intv > 0) {
            auto first_n = [&](int lo, int hi) -> int {      // first N in [lo, hi), -1 if none
                for (int wd = lo >> 6; wd <= (hi - 1) >> 6 && wd < 4; ++wd) {
                    unsigned long long mk = nm[wd];
                    if (wd == (lo >> 6)) mk &= ~0ull << (lo & 63);
                    if (wd == ((hi - 1) >> 6) && (hi & 63)) mk &= ~0ull >> (64 - (hi & 63));
                    if (mk) return wd * 64 + __builtin_ctzll(mk);
                }
                return -1;
            };
            int x = 0;
            while (x < len) {
                const int k = x >> 6, j = x & 63;
                if ((nm[k] >> j) & 1) { x++; continue; }
                const int lxx = __shfl(k == 0 ? lx[0] : k == 1 ? lx[1] : k == 2 ? lx[2] : lx[3], j);
                const int l1x = __shfl(k == 0 ? l1[0] : k == 1 ? l1[1] : k == 2 ? l1[2] : l1[3], j);
                int want = lxx + 1;
                if (want < A.msl + 1) want = A.msl + 1;
                const int hi = x + want < len ? x + want : len;
                const int nn = hi > x + 1 ? first_n(x + 1, hi) : -1;
                if (nn >= 0) { x = nn + 1; continue; }
                if (x + want > len) break;
                if (l1x >= want) {
                    // hits: lane m holds L_m(x); the number of leading planes that reach `want`
                    const int lm = lane >= 1 && lane <= A.M ? L(lane, x) : 0;
                    const unsigned long long ge = __ballot(lane >= 1 && lane <= A.M && lm >= want);
                    int c = __builtin_ctzll(~(ge >> 1));
                    c = c >= A.M ? kMany : c;
                    sel_emit(A, W, lane, lane == 0, (uint32_t)r, x, want, c);
                }
                x += want;
            }
        }
        if (W.n > kSelStage - 64 - kSelSlack) sel_flush(A, W, lane);
    }
    sel_flush(A, W, lane);
}

// ---- hit counts of the big subtrees -------------------------------------------------------------------------------
// A child pointer carries its subtree's hit count only below 20 (src/ertindex.cpp:455-461); for the others the reference
// walks the subtree and counts leaves.  Here those counts live in an open-addressing table in HBM (key = address of the
// four-way node + 1, value = hits): bwams_ert_build fills it from the FM-index intervals it has at hand, a loaded index
// fills it as subtrees are counted for the first time.  With every count known a hit of given rank is found by descent.
// hits of the leaf record `li` (0-based among the node's leaf records) of a four-way node: one position or a list
__device__ __forceinline__ int64_t leaf_rec(const uint8_t *__restrict__ mlt, int64_t node, int n_ptr, int w, int li, int64_t mh_base,
                                            int64_t *list_at) {
    const uint64_t rec = ld_le(mlt + node + 1 + n_ptr * w + 5 * li, 5);
    if (rec & 1) {
        *list_at = mh_base + (int64_t)(rec >> 1);
        return (int64_t)ld_le(mlt + *list_at, 2);
    }
    *list_at = -(int64_t)(rec >> 1) - 1;       // a single position, encoded negative
    return 1;
}

// the four-way node a pointer target stands for: runs (UNIFORM) in front of it are skipped
__device__ __forceinline__ int64_t skip_runs(const uint8_t *__restrict__ mlt, int64_t node) {
    for (;;) {
        const uint64_t head = ld_le(mlt + node, 2);
        const uint32_t cd = (uint32_t)(head & 0xff);
        const uint32_t is_uni = (~cd & (cd >> 1)) & 0x55;
        if (!is_uni) return node;
        node = node + 2 + (((int)(head >> 8) + 3) >> 2);
    }
}

// Number of hits below `node` (a four-way node).  Post-order walk: counts of big subtrees come from the table when it has
// them and go into it when they had to be summed.  Returns -1 when the explicit stack is exhausted (a corrupt index).
__device__ int64_t ert_count(const DevErt &e, int64_t node, int64_t mh_base, int w, uint64_t *__restrict__ stk, int64_t stride,
                             int max_frames) {
    const uint8_t *__restrict__ mlt = e.mlt;
    int sp = 0, c = 3;
    int64_t acc = 0;
    for (;;) {
        if (c < 0) {
            if (acc >= 20) cnt_insert(e, node, acc);
            if (sp == 0) return acc;
            sp -= 2;
            const uint64_t top = stk[(int64_t)sp * stride];
            const int64_t pacc = (int64_t)stk[(int64_t)(sp + 1) * stride];
            node = (int64_t)(top >> 3);
            c = (int)(top & 7) - 1;
            acc += pacc;
            continue;
        }
        const uint32_t cd = mlt[node];
        const int ty = (cd >> (c << 1)) & 3;
        if (ty == N_EMPTY) { c--; continue; }
        const uint32_t is_div = (cd & (cd >> 1)) & 0x55, is_leaf = (cd & ~(cd >> 1)) & 0x55;
        const uint32_t above = c == 3 ? 0u : (0xffu << ((c + 1) << 1)) & 0xff;
        const int n_ptr = __popc(is_div);
        if (ty == N_LEAF) {
            int64_t at;
            acc += leaf_rec(mlt, node, n_ptr, w, __popc(is_leaf & above), mh_base, &at);
            c--;
            continue;
        }
        const uint64_t v = ld_le(mlt + node + 1 + __popc(is_div & above) * w, w);
        if (v & 63) { acc += (int64_t)(v & 63); c--; continue; }
        const int64_t child = skip_runs(mlt, node + (int64_t)(v >> 6));
        const int64_t known = cnt_lookup(e, child);
        if (known >= 0) { acc += known; c--; continue; }
        if (sp + 2 > max_frames) return -1;
        stk[(int64_t)sp * stride] = ((uint64_t)node << 3) | (uint64_t)c;
        stk[(int64_t)(sp + 1) * stride] = (uint64_t)acc;
        sp += 2;
        node = child;
        c = 3;
        acc = 0;
    }
}

// Leaves below a node in A, C, G, T order (getNextByteIdx_dfs).  COUNT: number of hits (subtrees whose pointer
// carries the count are not entered).  Otherwise: hit t goes to out[t / step] when t % step == 0 and t / step < lim.
// Returns the number of hits, or -1 when the explicit stack is exhausted (a corrupt index).
template <bool COUNT>
__device__ int64_t ert_leaves(const DevErt &e, int64_t node, int64_t mh_base, int w, int64_t step, int64_t lim,
                              int64_t *__restrict__ out, uint64_t *__restrict__ stk, int64_t stride, int max_frames) {
    // explicit stack in HBM, frame-major / lane-minor (stk already points at this lane's column): node << 3 | next child
    int sp = 0;
    int64_t t = 0;
    int c = 3;
    const uint8_t *__restrict__ mlt = e.mlt;
    auto put = [&](int64_t pos) {
        if (!COUNT) {
            const int64_t k = t / step;
            if (k * step == t && k < lim) out[k] = pos;
        }
        t++;
    };
    for (;;) {
        if (c < 0) {
            if (sp == 0) break;
            const uint64_t top = stk[(int64_t)(--sp) * stride];
            node = (int64_t)(top >> 3);
            c = (int)(top & 7) - 1;
            continue;
        }
        const uint64_t head = ld_le(mlt + node, 2);
        const uint32_t cd = (uint32_t)(head & 0xff);
        const int ty = (cd >> (c << 1)) & 3;
        if (ty == N_EMPTY) { c--; continue; }
        if (ty == N_UNIFORM) {
            const int nbp = (int)(head >> 8);
            node = node + 2 + ((nbp + 3) >> 2);          // a run is the only child of its node: nothing to come back to
            c = 3;
            continue;
        }
        const uint32_t is_div = (cd & (cd >> 1)) & 0x55, is_leaf = (cd & ~(cd >> 1)) & 0x55;
        const uint32_t above = c == 3 ? 0u : (0xffu << ((c + 1) << 1)) & 0xff;
        const int n_ptr = __popc(is_div);
        if (ty == N_LEAF) {
            const uint64_t rec = ld_le(mlt + node + 1 + n_ptr * w + 5 * __popc(is_leaf & above), 5);
            if (rec & 1) {
                const int64_t at = mh_base + (int64_t)(rec >> 1);
                const int nc = (int)ld_le(mlt + at, 2);
                if (COUNT) t += nc;
                else
                    for (int k = 0; k < nc; ++k) put((int64_t)(ld_le(mlt + at + 2 + 5 * k, 5) >> 1));
            } else {
                put((int64_t)(rec >> 1));
            }
            c--;
            continue;
        }
        const uint64_t v = ld_le(mlt + node + 1 + __popc(is_div & above) * w, w);
        if (v & 63) {
            // the pointer carries the subtree's hit count (below 20): counting needs no visit, and sampling only when
            // one of the wanted ranks (multiples of step, fewer than lim of them) falls inside [t, t + count)
            const int64_t cnt = (int64_t)(v & 63);
            bool skip = COUNT;
            if (!COUNT) {
                const int64_t k = (t + step - 1) / step;
                skip = k * step >= t + cnt || k >= lim;
            }
            if (skip) { t += cnt; c--; continue; }
        }
        if (!COUNT && t > (lim - 1) * step) break;       // every wanted rank has been written
        if (sp == max_frames) return -1;
        stk[(int64_t)(sp++) * stride] = ((uint64_t)node << 3) | (uint64_t)c;   // popped as c - 1: resume with the next child
        node = node + (int64_t)(v >> 6);
        c = 3;
    }
    return t;
}

// lane = one sorted seed: walk again to its depth, remember where its hits are (k = address, l = kind | w << 8 |
// root << 16) and count them when the profile could not (20 or more)
__global__ __launch_bounds__(256) void ert_locate_kernel(DevErt e, const uint8_t *__restrict__ enc,
                                                         const int64_t *__restrict__ cum, bwams_smem_t *__restrict__ sm,
                                                         int64_t n, int64_t *__restrict__ sa_cnt, int max_occ,
                                                         DevCounters *ctr, uint64_t *__restrict__ stk, int max_frames) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    for (int64_t g = tid; g < n; g += nt) {
        bwams_smem_t s = sm[g];
        const int64_t c0 = cum[s.rid];
        const int len = (int)(cum[s.rid + 1] - c0), mlen = (int)(s.n - s.m + 1);
        Where wh;
        wh.kind = 0; wh.w = 0; wh.at = 0; wh.root = 0;
        WalkCnt wc = {0, 0, 0};
        const int d = ert_walk<false>(e, enc + c0, len, (int)s.m, 0, nullptr, mlen, &wh, wc);
        int64_t cnt = s.s;
        if (d < mlen) { wh.kind = 0; cnt = 0; }                 // cannot happen with a consistent index
        if (wh.kind == 3) wh.at = skip_runs(e.mlt, wh.at);      // hits are listed from the four-way node behind a run
        if (cnt < 0) {
            if (wh.kind == 1) cnt = 1;
            else if (wh.kind == 2) cnt = (int64_t)ld_le(e.mlt + wh.at, 2);
            else if (wh.kind == 3) {
                cnt = cnt_lookup(e, wh.at);
                if (cnt < 0) cnt = ert_count(e, wh.at, wh.root + (int64_t)ld_le(e.mlt + wh.root, 4), wh.w, stk + tid, nt, max_frames);
                if (cnt < 0) { cnt = 0; wh.kind = 0; atomicAdd(&ctr->overflow, 1ull); }
            }
        }
        s.s = cnt;
        s.k = wh.at;
        s.l = (int64_t)wh.kind | ((int64_t)wh.w << 8) | (wh.root << 16);
        sm[g] = s;
        if (sa_cnt) sa_cnt[g] = cnt < (int64_t)max_occ ? cnt : (int64_t)max_occ;
    }
}

// lane = one sampled hit: seed g's hits of rank 0, step, 2 step, ... (step = s / max_occ as mem_chain_new samples them,
// src/bwamem.cpp:993-1004) are found by descent: at a four-way node the children's hit counts (leaf records, pointer
// bits, the table) say which child holds the wanted rank.  A count the table does not have marks the seed for the
// serial walk below.
__global__ __launch_bounds__(256) void ert_hits_kernel(DevErt e, const bwams_smem_t *__restrict__ sm, int64_t n,
                                                       const int64_t *__restrict__ sa_off, int64_t *__restrict__ coord,
                                                       int64_t coord_cap, int max_occ, DevCounters *ctr,
                                                       uint32_t *__restrict__ redo) {
    const int64_t total = sa_off[n] < coord_cap ? sa_off[n] : coord_cap;
    const uint8_t *__restrict__ mlt = e.mlt;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_sa_lookups = (unsigned long long)sa_off[n];
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = n;                    // upper_bound(sa_off, g) - 1
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (sa_off[mid] <= g) lo = mid; else hi = mid;
        }
        const bwams_smem_t s = sm[lo];
        const int kind = (int)(s.l & 0xff), w = (int)((s.l >> 8) & 0xff);
        const int64_t root = s.l >> 16;
        const int64_t step = s.s > (int64_t)max_occ ? s.s / max_occ : 1;
        int64_t rho = (g - sa_off[lo]) * step, pos = -1;
        if (kind == 1) {
            pos = s.k;
        } else if (kind == 2) {
            pos = (int64_t)(ld_le(mlt + s.k + 2 + 5 * rho, 5) >> 1);
        } else if (kind == 3) {
            const int64_t mh_base = root + (int64_t)ld_le(mlt + root, 4);
            int64_t node = s.k;
            bool fail = false;
            for (int guard = 0; guard < 1024 && pos < 0 && !fail; ++guard) {
                const uint32_t cd = mlt[node];
                const uint32_t is_div = (cd & (cd >> 1)) & 0x55;
                const int n_ptr = __popc(is_div);
                int li = 0, pi = 0;
                int64_t next = -1;
                for (int c = 3; c >= 0 && pos < 0 && next < 0 && !fail; --c) {
                    const int ty = (cd >> (c << 1)) & 3;
                    if (ty == N_EMPTY) continue;
                    if (ty == N_LEAF) {
                        int64_t at;
                        const int64_t nc = leaf_rec(mlt, node, n_ptr, w, li++, mh_base, &at);
                        if (rho < nc) pos = at < 0 ? -at - 1 : (int64_t)(ld_le(mlt + at + 2 + 5 * rho, 5) >> 1);
                        else rho -= nc;
                    } else if (ty == N_DIVERGE) {
                        const uint64_t v = ld_le(mlt + node + 1 + (pi++) * w, w);
                        const int64_t child = skip_runs(mlt, node + (int64_t)(v >> 6));
                        int64_t cnt = (int64_t)(v & 63);
                        if (cnt == 0) {
                            cnt = cnt_lookup(e, child);
                            if (cnt < 0) { fail = true; break; }
                        }
                        if (rho < cnt) next = child; else rho -= cnt;
                    } else {
                        fail = true;          // a run cannot be met here: s.k and every child are behind their runs
                    }
                }
                if (pos < 0 && next < 0) fail = true;
                node = next;
            }
            if (fail || pos < 0) { atomicOr(&redo[lo >> 5], 1u << (lo & 31)); continue; }
        }
        coord[g] = pos;
    }
}

// lane = one seed marked by ert_hits_kernel: its hits by the serial leaf walk (the reference's leaf_gather order)
__global__ __launch_bounds__(256) void ert_gather_kernel(DevErt e, const bwams_smem_t *__restrict__ sm, int64_t n,
                                                         const int64_t *__restrict__ sa_off, int64_t *__restrict__ coord,
                                                         int64_t coord_cap, int max_occ, const uint32_t *__restrict__ redo,
                                                         uint64_t *__restrict__ stk, int max_frames) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    for (int64_t g = tid; g < n; g += nt) {
        if (!((redo[g >> 5] >> (g & 31)) & 1)) continue;
        const bwams_smem_t s = sm[g];
        const int kind = (int)(s.l & 0xff), w = (int)((s.l >> 8) & 0xff);
        const int64_t root = s.l >> 16, at = s.k, off = sa_off[g];
        const int64_t step = s.s > (int64_t)max_occ ? s.s / max_occ : 1;
        const int64_t lim = s.s < (int64_t)max_occ ? s.s : (int64_t)max_occ;
        if (kind == 3 && off + lim <= coord_cap)
            ert_leaves<false>(e, at, root + (int64_t)ld_le(e.mlt + root, 4), w, step, lim, coord + off, stk + tid, nt, max_frames);
    }
}

// the FM-index interval has no meaning for these seeds: k = l = 0 once the hits are listed
__global__ void ert_clear_kernel(bwams_smem_t *__restrict__ sm, int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) { sm[g].k = 0; sm[g].l = 0; }
}

}  // namespace

size_t ert_prof_bytes(int64_t nbases) { return (size_t)(nbases > 0 ? nbases : 1) * kProfRec + 8; }
void launch_ert_fat(const DevErt &e, int64_t mlt_bytes, uint8_t *fat, hipStream_t st) {
    const int64_t n = (int64_t)1 << (2 * e.K);
    ert_fat_kernel<<<(unsigned)((n + 255) / 256 < 65536 * 16 ? (n + 255) / 256 : 65536 * 16), 256, 0, st>>>(e.kmer, e.mlt, mlt_bytes, n, fat);
}
void launch_ert_profile(const DevErt &e, const uint8_t *enc, const int64_t *cum, const uint8_t *skip, int64_t nseq,
                        int64_t nbases, int M, uint8_t *prof, DevCounters *ctr, unsigned long long *part, int cu_count,
                        hipStream_t st) {
    if (nbases <= 0) return;
    int64_t blocks = (nbases + 255) / 256;
    // persistent waves that take their groups of read positions from a cursor (kErtTicket groups per atomic): trips differ a lot in
    // length (repeats).  Round robin over many more waves than fit at once measured 17.2 ms (32 blocks per CU; 20.9 ms for one
    // residency), the cursor 16.2 ms with 12 blocks per CU (GRCh38 size, 1 M reads; profiles/r03_notes.md 89).
    const int dyn = knobs().ert_ticket;                                          // experiments: 0 = round robin
    const int per_cu = knobs().ert_grid >= 0 ? knobs().ert_grid : (dyn ? 12 : 32);    // experiments: blocks per CU, 0 = one block per 256 bases
    if (per_cu > 0 && blocks > (int64_t)cu_count * per_cu) blocks = (int64_t)cu_count * per_cu;
    ert_profile_kernel<<<(unsigned)blocks, 256, 0, st>>>(e, enc, cum, skip, nseq, nbases, M, prof, part, dyn ? &ctr->ert_ticket : nullptr);
    ert_count_kernel<<<1, 256, 0, st>>>(part, ctr);
}

void launch_ert_select(const uint8_t *prof, const int64_t *cum, const uint8_t *skip, int64_t nseq, int64_t nbases, int M,
                       const bwams_seed_opt_t &opt, bwams_smem_t *pool, int64_t pool_cap, DevCounters *ctr, int cu_count,
                       hipStream_t st) {
    if (nseq <= 0) return;
    SelectArgs A;
    A.prof = prof; A.cum = cum; A.skip = skip; A.nseq = nseq; A.nbases = nbases; A.M = M;
    A.msl = opt.min_seed_len;
    A.split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    A.split_width = opt.split_width;
    A.max_intv = opt.max_mem_intv;
    A.pool = pool; A.pool_cap = pool_cap; A.ctr = ctr;
    int64_t blocks = (nseq + 3) / 4;
    if (blocks > (int64_t)cu_count * 16) blocks = (int64_t)cu_count * 16;
    ert_select_kernel<<<(unsigned)blocks, 256, 0, st>>>(A);
}

int64_t ert_walk_threads(int cu_count) { return (int64_t)cu_count * 8 * 256; }
size_t ert_count_bytes() { return (size_t)kCntSlots * 3 * 8; }

void launch_ert_locate(const DevErt &e, const uint8_t *enc, const int64_t *cum, bwams_smem_t *sorted, int64_t n,
                       int64_t *sa_cnt, int max_occ, DevCounters *ctr, uint64_t *stk, int max_frames, int cu_count,
                       hipStream_t st) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)cu_count * 8) blocks = (int64_t)cu_count * 8;
    ert_locate_kernel<<<(unsigned)blocks, 256, 0, st>>>(e, enc, cum, sorted, n, sa_cnt, max_occ, ctr, stk, max_frames);
}

void launch_ert_gather(const DevErt &e, bwams_smem_t *sorted, int64_t n, const int64_t *sa_off, int64_t *coord,
                       int64_t coord_cap, int max_occ, DevCounters *ctr, uint64_t *stk, int max_frames, uint32_t *redo,
                       int64_t n_coord_hint, int cu_count, hipStream_t st) {
    if (n <= 0) return;
    (void)hipMemsetAsync(redo, 0, (size_t)((n + 31) / 32) * 4, st);
    int64_t blocks = (n_coord_hint + 255) / 256;
    if (blocks > (int64_t)cu_count * 16) blocks = (int64_t)cu_count * 16;
    if (blocks < 1) blocks = 1;
    ert_hits_kernel<<<(unsigned)blocks, 256, 0, st>>>(e, sorted, n, sa_off, coord, coord_cap, max_occ, ctr, redo);
    blocks = (n + 255) / 256;
    if (blocks > (int64_t)cu_count * 8) blocks = (int64_t)cu_count * 8;
    ert_gather_kernel<<<(unsigned)blocks, 256, 0, st>>>(e, sorted, n, sa_off, coord, coord_cap, max_occ, redo, stk, max_frames);
}

void launch_ert_clear(bwams_smem_t *sorted, int64_t n, hipStream_t st) {
    if (n <= 0) return;
    ert_clear_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(sorted, n);
}

}  // namespace bwams
