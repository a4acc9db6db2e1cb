// ert_build.hip — ERT index construction on the GPU, byte for byte what `bwa-mem2 index -a ert` writes.
//
// Replaces buildKmerTrees / buildIndex (/root/reference/src/ertindex.cpp:773-943, :490-771) with
// ert_build_kmertree / handleDivergence / handleLeaf (:88-211), ert_build_table (:213-311) and
// ert_traverse_kmertree (:380-476).  The reference enumerates the 4^15 k-mers over the classic bwt with pthreads
// (2.2 h for the human genome on 40 cores, README.md:30-34), builds each k-mer's radix tree as a linked node
// structure and then serialises it, twice (sizes first, bytes second).
//
// Here one lane owns one k-mer.  Tree construction and serialisation are one depth-first walk over FM-index
// intervals (the index built by fmi_build.hip, resident): a node's bytes are written when the walk enters it, the
// pointer to a child when the walk descends into it, so no node structure is materialised.  The walk runs twice,
// as in the reference: pass B measures (for pointer widths 2, 3 and 4 at once, which decides the width exactly as
// the reference's retry does), a scan turns sizes into offsets, pass C writes.  The explicit stack of the walk (one
// frame per ancestor that still has children to visit: 40 B) lives in HBM, interleaved across lanes.
//
// Occurrence counts (`hits < 20`), LEP bits, the x-mer table of k-mers above HIT_THRESHOLD, multi-hit leaves and the
// 16-bit truncation of their counts are reproduced as written; tests/test_gpu_ert_build.py compares the bytes with the
// CPU restatement of the writer (oracle/ert_oracle.c).
// A k-mer with very many occurrences (satellite repeats) is still walked by a single lane: correct, but the tail of
// such lanes bounds the build time on real genomes; synthetic genomes have none.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <string>
#include "common.h"
#include "ert_kernels.h"

namespace bwams {
namespace {

enum { N_EMPTY = 0, N_LEAF = 1, N_UNIFORM = 2, N_DIVERGE = 3 };          // node_type_t, ertindex.h:12
enum { E_INVALID = 0, E_SINGLE = 1, E_INFREQUENT = 2, E_FREQUENT = 3 };  // macro.h:216-219

struct Iv { int64_t k, l, s; };

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

__device__ __forceinline__ void occ4(const DevFmi &f, int64_t pos, int64_t o[4]) {
    const uint4 *p = f.cp + ((pos >> 6) << 2);
    const uint4 c01 = p[0], c23 = p[1], h01 = p[2], h23 = p[3];
    const int y = (int)(pos & 63);
    const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
    o[0] = (int64_t)mk64(c01.x, c01.y) + __popcll(mk64(h01.x, h01.y) & mask);
    o[1] = (int64_t)mk64(c01.z, c01.w) + __popcll(mk64(h01.z, h01.w) & mask);
    o[2] = (int64_t)mk64(c23.x, c23.y) + __popcll(mk64(h23.x, h23.y) & mask);
    o[3] = (int64_t)mk64(c23.z, c23.w) + __popcll(mk64(h23.z, h23.w) & mask);
}

// bwt_extend(bwt, ik, ok, 0): ok[i] = interval of the pattern followed by base 3 - i
__device__ void ext4(const DevFmi &f, const Iv &ik, Iv ok[4]) {
    int64_t a[4], b[4];
    occ4(f, ik.l, a);
    occ4(f, ik.l + ik.s, b);
    const int64_t sent = (ik.l <= f.sentinel && ik.l + ik.s > f.sentinel) ? 1 : 0;
    int64_t ll = ik.k + sent;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        ok[i].s = b[i] - a[i];
        ok[i].l = f.count[i] + a[i];
        ok[i].k = ll;
        ll += ok[i].s;
    }
}

// bwt_sa: text position of BWT row `row` (the classic index has no sentinel quirk)
__device__ int64_t sa_true(const DevFmi &f, int64_t row) {
    int64_t sp = row, off = 0;
    for (;;) {
        if ((sp & 7) == 0) return ((int64_t)f.sa_ms[sp >> 3] << 32) + (int64_t)f.sa_ls[sp >> 3] + off;
        const uint4 *p = f.cp + ((sp >> 6) << 2);
        const uint4 h01 = p[2], h23 = p[3];
        const int sh = 63 - (int)(sp & 63);
        const uint64_t h0 = mk64(h01.x, h01.y), h1 = mk64(h01.z, h01.w), h2 = mk64(h23.x, h23.y), h3 = mk64(h23.z, h23.w);
        int c = 4;
        uint64_t hb = 0;
        if ((h0 >> sh) & 1) { c = 0; hb = h0; }
        else if ((h1 >> sh) & 1) { c = 1; hb = h1; }
        else if ((h2 >> sh) & 1) { c = 2; hb = h2; }
        else if ((h3 >> sh) & 1) { c = 3; hb = h3; }
        if (c == 4) return off;
        const int y = (int)(sp & 63);
        const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
        sp = f.count[c] + reinterpret_cast<const int64_t *>(p)[c] + __popcll(hb & mask);
        off++;
    }
}

struct BuildArgs {
    DevFmi f;
    int K, X, max_depth, hit_threshold;
    uint64_t n_kmers;
    uint64_t *kmer;          // entries: pass B writes the low 24 bits, pass C adds the offset
    uint64_t *meta;          // pass B: tree bytes << 32 | blob bytes
    const uint64_t *off;     // pass C: blob offsets (exclusive scan of the blob bytes)
    uint8_t *mlt;
    uint64_t *stk;           // 5 words per frame, frame-major, lane-minor
    int64_t n_threads;
    int max_frames;
    unsigned long long *err; // [0] stack exhausted, [1] four-way nodes with 20 hits or more (pass B)
    DevErt cnt;              // pass C: only cnt_tab / cnt_bits are used (the hit-count table of those nodes)
};

// One k-mer's walk.  EMIT = false: cur[0..2] are the byte cursors under pointer widths 2, 3, 4; EMIT = true: cur[0]
// is the cursor under the chosen width and bytes go to `out`.
template <bool EMIT>
struct Walk {
    const BuildArgs &A;
    int64_t tid;
    uint8_t *out;
    int w;
    uint32_t cur[3], maxp[3];
    uint32_t mh, mh_base, n_big;
    int64_t blob_off;
    bool failed, mh_over;

    __device__ Walk(const BuildArgs &a, int64_t t) : A(a), tid(t), out(nullptr), w(2), mh(0), mh_base(0), n_big(0), blob_off(0), failed(false), mh_over(false) {
        cur[0] = cur[1] = cur[2] = 0;
        maxp[0] = maxp[1] = maxp[2] = 0;
    }
    __device__ void put(uint32_t at, uint64_t v, int n) {
        for (int i = 0; i < n; ++i) out[at + i] = (uint8_t)(v >> (8 * i));
    }
    __device__ void adv(int fixed, int per_ptr) {            // advance by fixed + per_ptr * width
        if (EMIT) cur[0] += fixed + per_ptr * w;
        else { cur[0] += fixed + per_ptr * 2; cur[1] += fixed + per_ptr * 3; cur[2] += fixed + per_ptr * 4; }
    }
    // a multi-hit leaf: 5-byte pointer into the multi-hit area, there a 16-bit count and the positions
    // The count is 16 bits in the format (ertindex.cpp:336-352; the reference's own writer never returns from a leaf of 65536 hits or
    // more: its loop variable is a uint16_t): a text in which one read-length string occurs that often has no ERT index.
    __device__ void put_mh(const Iv &iv) {
        if (iv.s > 0xffff) mh_over = true;
        const uint32_t n16 = (uint32_t)(iv.s & 0xffff);
        if (EMIT) {
            put(cur[0], ((uint64_t)mh << 1) | 1ull, 5);
            put(mh_base + mh, (uint64_t)iv.s, 2);
            for (uint32_t j = 0; j < n16; ++j) put(mh_base + mh + 2 + 5 * j, ((uint64_t)sa_true(A.f, iv.k + j) << 1) | 1ull, 5);
        }
        adv(5, 0);
        mh += 2 + 5 * n16;
    }
    __device__ uint64_t &frame(int sp, int word) { return A.stk[((int64_t)sp * 5 + word) * A.n_threads + tid]; }

    // serialise the subtree of the pattern with interval ik0 and `depth0` matched bases (ert_build_kmertree +
    // ert_traverse_kmertree on the node that owns the children of ik0)
    __device__ void subtree(Iv ik, int depth) {
        int sp = 0;
        Iv ok[4];
        uint32_t start[3] = {0, 0, 0};
        int next_i = 3, pidx = 0;
        bool resume = false;
        for (;;) {
            if (!resume) {
                ext4(A.f, ik, ok);
                int nb = 0, ub = 0;
                for (int i = 0; i < 4; ++i)
                    if (ok[i].s > 0) { nb++; ub = i; }
                bool four = true;
                if (nb == 1) {
                    const int c = ub;
                    four = false;
                    if (depth < A.max_depth) {
                        const Iv ok_init = ok[ub];
                        Iv ikn = ok[ub];
                        uint8_t run[64];
                        for (int t = 0; t < 64; ++t) run[t] = 0;
                        int nbp = 1;
                        run[0] = (uint8_t)(ub << 6);
                        bool leaf = false;
                        for (;;) {
                            depth += 1;
                            ext4(A.f, ikn, ok);
                            nb = 0; ub = 0;
                            for (int i = 0; i < 4; ++i)
                                if (ok[i].s > 0) { nb++; ub = i; }
                            if (nb != 1) break;
                            ikn = ok[ub];
                            if (nbp < 256) run[nbp >> 2] |= (uint8_t)(ub << ((~nbp & 3) << 1));
                            nbp++;
                            if (depth == A.max_depth) { leaf = true; break; }
                        }
                        if (leaf) {
                            if (EMIT) put(cur[0], (uint64_t)N_LEAF << (c << 1), 1);
                            adv(1, 0);
                            put_mh(ok_init);
                        } else {
                            const int nby = (nbp + 3) >> 2;
                            if (EMIT) {
                                put(cur[0], (uint64_t)N_UNIFORM << (c << 1), 1);
                                put(cur[0] + 1, (uint64_t)(uint8_t)nbp, 1);
                                for (int t = 0; t < nby; ++t) out[cur[0] + 2 + t] = run[t];
                            }
                            adv(2 + nby, 0);
                            ik = ikn;
                            four = true;        // the run's node owns the four children of `ok` at `depth`
                        }
                    } else {
                        if (EMIT) put(cur[0], (uint64_t)N_LEAF << (c << 1), 1);
                        adv(1, 0);
                        put_mh(ok[ub]);
                    }
                }
                if (four) {
                    uint32_t code = 0;
                    int n_ptr = 0;
                    for (int i = 0; i < 4; ++i) {
                        if (ok[i].s == 0) continue;
                        if (ok[i].s > 1 && depth != A.max_depth) { code |= (uint32_t)N_DIVERGE << (i << 1); n_ptr++; }
                        else code |= (uint32_t)N_LEAF << (i << 1);
                    }
                    start[0] = cur[0]; start[1] = cur[1]; start[2] = cur[2];
                    if (ik.s >= 20) {          // its hit count is in no pointer: the seeding kernels look it up by address
                        if (EMIT) cnt_insert(A.cnt, blob_off + cur[0], ik.s);
                        else n_big++;
                    }
                    if (EMIT) put(cur[0], code, 1);
                    adv(1, n_ptr);
                    for (int i = 3; i >= 0; --i) {
                        if (ok[i].s == 0 || (ok[i].s > 1 && depth != A.max_depth)) continue;
                        if (ok[i].s == 1) {
                            if (EMIT) put(cur[0], (uint64_t)sa_true(A.f, ok[i].k) << 1, 5);
                            adv(5, 0);
                        } else {
                            put_mh(ok[i]);
                        }
                    }
                    next_i = 3; pidx = 0;
                } else {
                    next_i = -1;                 // a leaf: nothing below
                }
            }
            resume = false;
            // next DIVERGE child of the current node
            int i = next_i;
            while (i >= 0 && !(ok[i].s > 1 && depth != A.max_depth)) i--;
            if (i < 0) {
                if (sp == 0) return;
                sp--;
                ik.k = (int64_t)frame(sp, 0); ik.l = (int64_t)frame(sp, 1); ik.s = (int64_t)frame(sp, 2);
                const uint64_t w3 = frame(sp, 3), w4 = frame(sp, 4);
                start[0] = (uint32_t)w3; start[1] = (uint32_t)(w3 >> 32); start[2] = (uint32_t)w4;
                depth = (int)((w4 >> 32) & 0xff); next_i = (int)((w4 >> 40) & 0xff) - 1; pidx = (int)((w4 >> 48) & 0xff);
                ext4(A.f, ik, ok);
                resume = true;
                continue;
            }
            // pointer to the child = its offset from the node's code byte, and its hit count while below 20
            const uint64_t cnt = ok[i].s < 20 ? (uint64_t)ok[i].s : 0;
            if (EMIT) {
                const uint32_t p = cur[0] - start[0];
                put(start[0] + 1 + pidx * w, ((uint64_t)p << 6) | cnt, w);
            } else {
                for (int t = 0; t < 3; ++t) {
                    const uint32_t p = cur[t] - start[t];
                    if (p > maxp[t]) maxp[t] = p;
                }
            }
            int more = i - 1;
            while (more >= 0 && !(ok[more].s > 1 && depth != A.max_depth)) more--;
            if (more >= 0) {                        // come back for the remaining children
                if (sp >= A.max_frames) { failed = true; return; }
                frame(sp, 0) = (uint64_t)ik.k; frame(sp, 1) = (uint64_t)ik.l; frame(sp, 2) = (uint64_t)ik.s;
                frame(sp, 3) = (uint64_t)start[0] | ((uint64_t)start[1] << 32);
                frame(sp, 4) = (uint64_t)start[2] | ((uint64_t)depth << 32) | ((uint64_t)i << 40) | ((uint64_t)(pidx + 1) << 48);
                sp++;
            }
            ik = ok[i];
            depth += 1;
        }
    }
};

// the k-mer loop of buildIndex (:528-552): interval, LEP bits, hits
__device__ bool kmer_search(const BuildArgs &A, uint64_t idx, Iv &ik, uint64_t &lep, int64_t &num_hits) {
    const int a0 = (int)(idx & 3);
    ik.k = A.f.count[a0]; ik.l = A.f.count[3 - a0]; ik.s = A.f.count[a0 + 1] - A.f.count[a0];
    int64_t prev = ik.s;
    lep = 0;
    num_hits = ik.s;
    Iv ok[4];
    for (int i = 1; i < A.K; ++i) {
        const int c = 3 - (int)((idx >> (2 * i)) & 3);
        ext4(A.f, ik, ok);
        if (ok[c].s != prev) lep |= 1ull << (i - 1);
        num_hits = ok[c].s;
        if (ok[c].s >= 1) { prev = ok[c].s; ik = ok[c]; }
        else return false;
    }
    return num_hits >= 1;
}

// the chosen pointer width: the reference starts at 2 and retries with 3, then 4 (:628-645)
__device__ int pick_width(const uint32_t maxp[3]) {
    if (maxp[0] >= 1024 && maxp[0] < 262144) return maxp[1] >= 262144 ? 4 : 3;
    if (maxp[0] >= 262144) return 4;
    return 2;
}

template <bool EMIT>
__device__ void xmer_table(Walk<EMIT> &W, const BuildArgs &A, const Iv &ik0) {
    const int n_x = 1 << (2 * A.X);
    W.adv(4 + 8 * n_x, 0);
    uint64_t lep1 = 0;                     // not reset between x-mers in the reference either (:227)
    for (int x = 0; x < n_x && !W.failed; ++x) {
        Iv ik = ik0, ok[4];
        int64_t prev = ik0.s;
        int j, c = 0;
        for (j = 0; j < A.X; ++j) {
            c = 3 - ((x >> (2 * j)) & 3);
            ext4(A.f, ik, ok);
            if (ok[c].s != prev) lep1 |= 1ull << j;
            if (ok[c].s >= 1) { prev = ok[c].s; ik = ok[c]; } else break;
        }
        const int64_t num_hits = ok[c].s;
        const uint32_t mlt_offset = W.cur[0];
        uint32_t xdata;
        if (num_hits == 0) {
            xdata = (uint32_t)(((lep1 & 0x3FFF) << 2) | E_INVALID) & 0xffff;
        } else if (num_hits == 1) {
            xdata = (uint32_t)(((lep1 & 0x3FFF) << 2) | E_SINGLE) & 0xffff;
            if (EMIT) {
                W.put(W.cur[0], 0, 1);
                W.put(W.cur[0] + 1, (uint64_t)sa_true(A.f, ok[c].k) << 1, 5);
            }
            W.adv(6, 0);
        } else {
            xdata = (uint32_t)(((lep1 & 0x3FFF) << 2) | E_INFREQUENT) & 0xffff;
            W.subtree(ik, A.K + j);
        }
        if (EMIT) {
            uint64_t e = ((uint64_t)mlt_offset << 24) | xdata;
            if (num_hits < 20) e |= (uint64_t)num_hits << 17;
            e |= (uint64_t)(W.w < 4 ? W.w : 0) << 22;
            W.put(4 + 8 * x, e, 8);
        }
    }
}

// pass B: sizes, pointer width, the low bits of the entry
__global__ __launch_bounds__(256) void ert_size_kernel(BuildArgs A) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t lep_mask = (1ull << (A.K - 1)) - 1;
    for (uint64_t idx = (uint64_t)tid; idx < A.n_kmers; idx += (uint64_t)A.n_threads) {
        Iv ik;
        uint64_t lep;
        int64_t num_hits;
        const bool alive = kmer_search(A, idx, ik, lep, num_hits);
        uint64_t lo = (lep & lep_mask) << 2, meta = 0;
        if (!alive) {
            lo |= E_INVALID;
        } else if (num_hits == 1) {
            lo |= E_SINGLE | (1ull << 17);
            meta = (6ull << 32) | 6ull;
        } else {
            Walk<false> W(A, tid);
            if (num_hits <= A.hit_threshold) {
                lo |= E_INFREQUENT;
                W.adv(4, 0);
                W.subtree(ik, A.K);
            } else {
                lo |= E_FREQUENT;
                xmer_table<false>(W, A, ik);
            }
            if (W.failed) atomicAdd(&A.err[0], 1ull);
            if (W.mh_over) atomicAdd(&A.err[2], 1ull);
            if (W.n_big) atomicAdd(&A.err[1], (unsigned long long)W.n_big);
            const int w = pick_width(W.maxp);
            const uint64_t tree = W.cur[w - 2];
            // a pointer is 26 bits of offset beside 6 bits of hit count, and the reference asserts both that and the size of a k-mer's
            // bytes below 2^26 (ertindex.cpp:452, :607, :651): a k-mer with more hits than 64 MiB of leaves has no tree in this format
            if (W.maxp[w - 2] >= (1u << 26) || tree + W.mh >= (1ull << 26) || num_hits >= (1ll << 26) / 5) atomicAdd(&A.err[3], 1ull);
            if (num_hits < 20) lo |= (uint64_t)num_hits << 17;
            lo |= (uint64_t)(w < 4 ? w : 0) << 22;
            meta = (tree << 32) | (tree + W.mh);
        }
        A.kmer[idx] = lo;
        A.meta[idx] = meta;
    }
}

// pass C: bytes
__global__ __launch_bounds__(256) void ert_emit_kernel(BuildArgs A) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (uint64_t idx = (uint64_t)tid; idx < A.n_kmers; idx += (uint64_t)A.n_threads) {
        const uint64_t lo = A.kmer[idx], meta = A.meta[idx], off = A.off[idx];
        A.kmer[idx] = (off << 24) | lo;
        const int code = (int)(lo & 3);
        if (code == E_INVALID) continue;
        Iv ik;
        uint64_t lep;
        int64_t num_hits;
        kmer_search(A, idx, ik, lep, num_hits);
        Walk<true> W(A, tid);
        W.out = A.mlt + off;
        W.blob_off = (int64_t)off;
        if (code == E_SINGLE) {
            W.put(0, 0, 1);
            W.put(1, (uint64_t)sa_true(A.f, ik.k) << 1, 5);
            continue;
        }
        W.w = ((lo >> 22) & 3) == 0 ? 4 : (int)((lo >> 22) & 3);
        W.mh_base = (uint32_t)(meta >> 32);
        W.put(0, meta >> 32, 4);
        if (code == E_INFREQUENT) {
            W.adv(4, 0);
            W.subtree(ik, A.K);
        } else {
            xmer_table<true>(W, A, ik);
        }
    }
}

struct LowWord {
    __host__ __device__ uint64_t operator()(uint64_t v) const { return v & 0xffffffffull; }
};

}  // namespace

// Builds <prefix>.kmer_table / <prefix>.mlt_table of the resident index into device buffers owned by *e.
int ert_build_device(bwams_ert *e, const DevFmi &f, int K, int X, int read_len, int hit_threshold, int cu_count, int verbose) {
    hipStream_t st = nullptr;
    const uint64_t n_kmers = 1ull << (2 * K);
    BuildArgs A;
    A.f = f; A.K = K; A.X = X; A.max_depth = read_len - 1; A.hit_threshold = hit_threshold;
    A.n_kmers = n_kmers;
    int64_t blocks = (int64_t)cu_count * 8;
    if ((uint64_t)blocks * 256 > n_kmers) blocks = (int64_t)((n_kmers + 255) / 256);
    A.n_threads = blocks * 256;
    A.max_frames = read_len - K + 2;
    void *d_meta = nullptr, *d_off = nullptr, *d_stk = nullptr, *d_err = nullptr, *d_tmp = nullptr;
    auto cleanup = [&]() {
        for (void *p : {d_meta, d_off, d_stk, d_err, d_tmp})
            if (p) (void)hipFree(p);
    };
#define ERT_HIP(call)                                                       \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            set_last_error(std::string("ert_build: " #call " -> ") + hipGetErrorString(e_)); \
            cleanup();                                                      \
            return e_ == hipErrorOutOfMemory ? BWAMS_ERR_NOMEM : BWAMS_ERR_DEVICE; \
        }                                                                   \
    } while (0)
    hipEvent_t e0, e1, e2, e3;
    ERT_HIP(hipEventCreate(&e0)); ERT_HIP(hipEventCreate(&e1)); ERT_HIP(hipEventCreate(&e2)); ERT_HIP(hipEventCreate(&e3));
    ERT_HIP(dev_malloc(&e->d_kmer, n_kmers * 8));
    ERT_HIP(dev_malloc(&d_meta, n_kmers * 8));
    ERT_HIP(dev_malloc(&d_off, n_kmers * 8));
    ERT_HIP(dev_malloc(&d_stk, (size_t)A.n_threads * (size_t)A.max_frames * 40));
    ERT_HIP(dev_malloc(&d_err, 32));
    ERT_HIP(hipMemsetAsync(d_err, 0, 32, st));
    A.kmer = (uint64_t *)e->d_kmer;
    A.meta = (uint64_t *)d_meta;
    A.off = (const uint64_t *)d_off;
    A.mlt = nullptr;
    A.stk = (uint64_t *)d_stk;
    A.err = (unsigned long long *)d_err;
    ERT_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(ert_size_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A);
    ERT_HIP(hipGetLastError());
    ERT_HIP(hipEventRecord(e1, st));
    {
        auto in = rocprim::make_transform_iterator((const uint64_t *)d_meta, LowWord());
        size_t tb = 0;
        ERT_HIP(rocprim::exclusive_scan(nullptr, tb, in, (uint64_t *)d_off, (uint64_t)0, (size_t)n_kmers, rocprim::plus<uint64_t>(), st));
        ERT_HIP(dev_malloc(&d_tmp, tb ? tb : 8));
        ERT_HIP(rocprim::exclusive_scan(d_tmp, tb, in, (uint64_t *)d_off, (uint64_t)0, (size_t)n_kmers, rocprim::plus<uint64_t>(), st));
    }
    uint64_t last_off = 0, last_meta = 0;
    unsigned long long err[4] = {0, 0, 0, 0};
    ERT_HIP(hipMemcpyAsync(&last_off, (uint64_t *)d_off + (n_kmers - 1), 8, hipMemcpyDeviceToHost, st));
    ERT_HIP(hipMemcpyAsync(&last_meta, (uint64_t *)d_meta + (n_kmers - 1), 8, hipMemcpyDeviceToHost, st));
    ERT_HIP(hipMemcpyAsync(err, d_err, 32, hipMemcpyDeviceToHost, st));
    ERT_HIP(hipStreamSynchronize(st));
    if (err[2]) {
        set_last_error("ert_build: " + std::to_string(err[2]) + " k-mer tree(s) hold a string of read_len bases that occurs 65536 times or more; the ERT format "
                       "counts the hits of such a leaf in 16 bits (src/ertindex.cpp:336-352), so this text has no ERT index");
        cleanup();
        return BWAMS_ERR_UNSUPPORTED;
    }
    if (err[3]) {
        set_last_error("ert_build: the trees of " + std::to_string(err[3]) + " k-mer(s) reach 64 MiB; child pointers carry 26 bits of offset and the reference's "
                       "writer asserts every k-mer's bytes below 2^26 (src/ertindex.cpp:452, :607, :651), so this text has no ERT index with this k");
        cleanup();
        return BWAMS_ERR_UNSUPPORTED;
    }
    if (err[0]) {
        set_last_error("ert_build: a radix tree is deeper than the read length allows (corrupt index?)");
        cleanup();
        return BWAMS_ERR_UNSUPPORTED;
    }
    const int64_t mlt_bytes = (int64_t)(last_off + (last_meta & 0xffffffffull));
    ERT_HIP(dev_malloc(&e->d_mlt, (size_t)mlt_bytes + 16));
    ERT_HIP(hipMemsetAsync(e->d_mlt, 0, (size_t)mlt_bytes + 16, st));
    A.mlt = (uint8_t *)e->d_mlt;
    int bits = 10;
    while (((uint64_t)1 << bits) < 2 * err[1] + 16) bits++;
    ERT_HIP(dev_malloc(&e->d_cnt, ((size_t)16) << bits));
    ERT_HIP(hipMemsetAsync(e->d_cnt, 0, ((size_t)16) << bits, st));
    memset(&A.cnt, 0, sizeof A.cnt);
    A.cnt.cnt_tab = (uint64_t *)e->d_cnt;
    A.cnt.cnt_bits = bits;
    ERT_HIP(hipEventRecord(e2, st));
    hipLaunchKernelGGL(ert_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A);
    ERT_HIP(hipGetLastError());
    ERT_HIP(hipEventRecord(e3, st));
    ERT_HIP(hipStreamSynchronize(st));
    float msB = 0, msS = 0, msC = 0;
    (void)hipEventElapsedTime(&msB, e0, e1);
    (void)hipEventElapsedTime(&msS, e1, e2);
    (void)hipEventElapsedTime(&msC, e2, e3);
    if (verbose)
        fprintf(stderr, "[bwams] ert_build: %llu k-mers, trees %.3f GB, %llu nodes with 20+ hits (count table 2^%d); sizes %.1f ms, scan + alloc %.1f ms, bytes %.1f ms\n",
                (unsigned long long)n_kmers, mlt_bytes / 1e9, err[1], bits, msB, msS, msC);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
    cleanup();
#undef ERT_HIP
    e->t.kmer = (const uint64_t *)e->d_kmer;
    e->t.mlt = (const uint8_t *)e->d_mlt;
    e->t.ref = f.ref;
    e->t.ref_len = f.ref_seq_len - 1;
    e->t.K = K; e->t.X = X; e->t.read_len = read_len;
    e->t.cnt_tab = (uint64_t *)e->d_cnt;
    e->t.cnt_bits = bits;
    e->n_big = (int64_t)err[1];
    e->bytes = (int64_t)(n_kmers * 8) + mlt_bytes + 16 + ((int64_t)16 << bits);
    e->mlt_bytes = mlt_bytes;
    e->build_ms[0] = msB; e->build_ms[1] = msS; e->build_ms[2] = msC;
    return BWAMS_OK;
}

}  // namespace bwams
