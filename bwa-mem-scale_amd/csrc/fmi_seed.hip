// fmi_seed.hip — FM-index SMEM search for gfx950 (MI355X).
//
// What is computed (reference semantics, /root/reference):
//   round 1  getSMEMsAllPosOneThread        src/FMI_search.cpp:1608-1660
//   round 2  getSMEMsOnePosOneThread        src/FMI_search.cpp:1372-1606 on the pivots chosen
//            by mem_collect_smem            src/bwamem.cpp:721-751
//   round 3  bwtSeedStrategyAllPosOneThread src/FMI_search.cpp:1662-1816
//   each step is one backwardExt            src/FMI_search.cpp:2029-2056 over CP_OCC blocks
//
// How it is mapped to the machine (DESIGN.md §"SMEM kernel"):
//   The search is a chain of dependent random 64-byte block reads (one or two per
//   extension), ~460 per read.  Throughput therefore comes from the number of
//   independent chains in flight, not from lanes cooperating on one chain: every
//   LANE owns one read and runs a small state machine whose every iteration performs
//   exactly one extension, whatever phase (forward / backward / round 3) the lane is
//   in, so divergent phases still share one memory round trip.  Lanes pull the next
//   read from a global cursor when they finish (no tail of idle lanes), the grid is
//   persistent and sized to the chip, and the per-lane list of "previous" intervals
//   lives in a lane-interleaved HBM scratch that stays L2-resident.
#include "fmi_kernels.h"

namespace bwams {

namespace {

constexpr int kBlock = 256;
constexpr int kBlocksPerCU = 6;

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) {
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

struct Occ4 {
    int64_t v[4];
};

// Both ends of an interval.  When k and k+s fall into one block the block is read once.
__device__ __forceinline__ void occ4_pair(const uint4 *__restrict__ cp, int64_t sp, int64_t ep,
                                          Occ4 &osp, Occ4 &oep) {
    const uint4 *p = cp + ((sp >> 6) << 2);
    uint4 c01 = p[0], c23 = p[1], h01 = p[2], h23 = p[3];
    {
        const int y = (int)(sp & 63);
        const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
        osp.v[0] = (int64_t)mk64(c01.x, c01.y) + __popcll(mk64(h01.x, h01.y) & mask);
        osp.v[1] = (int64_t)mk64(c01.z, c01.w) + __popcll(mk64(h01.z, h01.w) & mask);
        osp.v[2] = (int64_t)mk64(c23.x, c23.y) + __popcll(mk64(h23.x, h23.y) & mask);
        osp.v[3] = (int64_t)mk64(c23.z, c23.w) + __popcll(mk64(h23.z, h23.w) & mask);
    }
    if ((ep >> 6) != (sp >> 6)) {
        const uint4 *q = cp + ((ep >> 6) << 2);
        c01 = q[0]; c23 = q[1]; h01 = q[2]; h23 = q[3];
    }
    {
        const int y = (int)(ep & 63);
        const uint64_t mask = y ? (~0ull << (64 - y)) : 0ull;
        oep.v[0] = (int64_t)mk64(c01.x, c01.y) + __popcll(mk64(h01.x, h01.y) & mask);
        oep.v[1] = (int64_t)mk64(c01.z, c01.w) + __popcll(mk64(h01.z, h01.w) & mask);
        oep.v[2] = (int64_t)mk64(c23.x, c23.y) + __popcll(mk64(h23.x, h23.y) & mask);
        oep.v[3] = (int64_t)mk64(c23.z, c23.w) + __popcll(mk64(h23.z, h23.w) & mask);
    }
}

// backwardExt of the interval (k, l, s) by base a.
__device__ __forceinline__ void backward_ext(const DevFmi &f, int64_t k, int64_t l, int64_t s, int a,
                                             int64_t &nk, int64_t &nl, int64_t &ns) {
    Occ4 osp, oep;
    occ4_pair(f.cp, k, k + s, osp, oep);
    const int64_t s0 = oep.v[0] - osp.v[0], s1 = oep.v[1] - osp.v[1];
    const int64_t s2 = oep.v[2] - osp.v[2], s3 = oep.v[3] - osp.v[3];
    const int64_t l3 = l + ((k <= f.sentinel && k + s > f.sentinel) ? 1 : 0);
    const int64_t l2 = l3 + s3, l1 = l2 + s2, l0 = l1 + s1;
    nk = (a == 0 ? f.count[0] + osp.v[0] : a == 1 ? f.count[1] + osp.v[1]
          : a == 2 ? f.count[2] + osp.v[2] : f.count[3] + osp.v[3]);
    ns = a == 0 ? s0 : a == 1 ? s1 : a == 2 ? s2 : s3;
    nl = a == 0 ? l0 : a == 1 ? l1 : a == 2 ? l2 : l3;
}

// count[i] without dynamic indexing of the kernel argument (keeps it in SGPRs)
__device__ __forceinline__ int64_t cnt_at(const DevFmi &f, int i) {
    return i == 0 ? f.count[0] : i == 1 ? f.count[1] : i == 2 ? f.count[2] : i == 3 ? f.count[3] : f.count[4];
}

// wave-aggregated fetch of one item index per requesting lane
__device__ __forceinline__ unsigned long long take_ticket(unsigned long long *head, bool want) {
    const unsigned long long m = __ballot(want);
    unsigned long long base = 0;
    if (m) {
        const int lane = (int)(threadIdx.x & 63);
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(head, (unsigned long long)__popcll(m));
        const uint32_t lo = __shfl((uint32_t)base, leader), hi = __shfl((uint32_t)(base >> 32), leader);
        base = mk64(lo, hi) + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    }
    return base;
}

__device__ __forceinline__ void emit_smem(const SeedLaunch &a, uint32_t rid, uint32_t m, uint32_t n,
                                          int64_t k, int64_t l, int64_t s) {
    const unsigned long long pos = atomicAdd(&a.ctr->n_smem_total, 1ull);
    if ((int64_t)pos < a.pool_cap) {
        bwams_smem_t r;
        r.rid = rid; r.m = m; r.n = n; r.pad_ = 0;
        r.k = k; r.l = l; r.s = s;
        a.pool[pos] = r;
    }
}

__device__ __forceinline__ void flush_counters(DevCounters *ctr, unsigned long long n_ext,
                                               unsigned long long n_blk) {
    for (int o = 32; o > 0; o >>= 1) {
        n_ext += mk64(__shfl_down((uint32_t)n_ext, o), __shfl_down((uint32_t)(n_ext >> 32), o));
        n_blk += mk64(__shfl_down((uint32_t)n_blk, o), __shfl_down((uint32_t)(n_blk >> 32), o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr->n_ext, n_ext);
        atomicAdd(&ctr->n_ext_blocks, n_blk);
    }
}

enum : int { PH_FETCH = 0, PH_PIVOT, PH_FWD, PH_FWD_END, PH_BWD, PH_BWD_END, PH_EXIT };

// Rounds 1 and 2.  ALL_POS: work item = read, walk every pivot (round 1).
// !ALL_POS: work item = (read, pivot, min_intv), one pivot (round 2).
template <bool ALL_POS>
__global__ __launch_bounds__(kBlock) void smem_search_kernel(SeedLaunch a, const Round2Work *work) {
    const DevFmi &f = a.fmi;
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t nt = a.prev_threads;
    const int cap = a.prev_cap;
    const int64_t n_work = ALL_POS ? a.nseq : (int64_t)a.ctr->n_work2;

    int phase = PH_FETCH;
    uint32_t rid = 0;
    int64_t qoff = 0;
    int len = 0, x = 0, next_x = 0, min_intv = 1;
    int64_t ck = 0, cl = 0, cs = 0;      // current interval (forward phase)
    int cn = 0;                           // its end position n
    int j = 0;                            // position being extended to
    int num_prev = 0, base = 0, p = 0, num_curr = 0, cur_m = 0;
    int32_t curr_s = -1;
    bool first = true;
    int bwd_a = 0;
    unsigned long long n_ext = 0, n_blk = 0;

    while (true) {
        // ---- leave a finished pivot -------------------------------------------------
        if (phase == PH_BWD_END) {
            if (num_prev != 0) {
                const int64_t e = (int64_t)base * nt + slot;
                const int pn = a.prev_n[e];
                if (pn - cur_m + 1 >= a.min_seed_len)
                    emit_smem(a, rid, (uint32_t)cur_m, (uint32_t)pn, a.prev_k[e], a.prev_l[e], a.prev_s[e]);
            }
            x = next_x;
            phase = ALL_POS ? PH_PIVOT : PH_FETCH;
        }
        // ---- take the next work item ------------------------------------------------
        {
            const bool want = phase == PH_FETCH;
            const unsigned long long t = take_ticket(&a.ctr->work_head, want);
            if (want) {
                if ((int64_t)t >= n_work) {
                    phase = PH_EXIT;
                } else {
                    if (ALL_POS) {
                        rid = (uint32_t)t;
                        x = 0;
                        min_intv = 1;
                    } else {
                        const Round2Work wk = work[t];
                        rid = wk.rid;
                        x = wk.x;
                        min_intv = wk.min_intv;
                    }
                    qoff = a.cum[rid];
                    len = (int)(a.cum[rid + 1] - qoff);
                    phase = PH_PIVOT;
                    if (ALL_POS && a.skip && a.skip[rid]) phase = PH_FETCH;
                }
            }
        }
        if (__all(phase == PH_EXIT)) break;

        // ---- open a pivot -----------------------------------------------------------
        if (phase == PH_PIVOT) {
            if (x >= len) {
                phase = PH_FETCH;
            } else {
                const int c = a.enc[qoff + x];
                if (c >= 4) {
                    x = x + 1;                          // query_pos = next_x = x + 1
                    if (!ALL_POS) phase = PH_FETCH;
                } else {
                    ck = cnt_at(f, c);
                    cl = cnt_at(f, 3 - c);
                    cs = cnt_at(f, c + 1) - ck;
                    cn = x;
                    j = x + 1;
                    next_x = x + 1;
                    num_prev = 0;
                    phase = PH_FWD;
                }
            }
        }

        bool do_ext = false;
        int64_t ek = 0, el = 0, es = 0;
        int ea = 0;
        int64_t pk = 0, pl = 0, ps = 0;
        int pn = 0;

        // ---- forward phase: pre ------------------------------------------------------
        if (phase == PH_FWD) {
            phase = PH_FWD_END;
            if (j < len) {
                const int c = a.enc[qoff + j];
                next_x = j + 1;
                if (c < 4) {
                    phase = PH_FWD;
                    do_ext = true;
                    ek = cl; el = ck; es = cs;         // forward = backward on the other strand
                    ea = 3 - c;
                }
            }
        }
        if (phase == PH_FWD_END) {
            if (cs >= min_intv) {
                const int64_t e = (int64_t)(cap - 1 - num_prev) * nt + slot;
                a.prev_k[e] = ck; a.prev_l[e] = cl; a.prev_s[e] = cs; a.prev_n[e] = cn;
                num_prev++;
            }
            base = cap - num_prev;                      // entry p lives at base + p, longest first
            j = x - 1;
            p = 0; num_curr = 0; curr_s = -1; first = true;
            cur_m = x;
            phase = PH_BWD;
        }
        // ---- backward phase: pre -----------------------------------------------------
        if (phase == PH_BWD && !do_ext) {
            bool go = true;
            if (p == 0) {
                go = false;
                if (num_prev != 0 && j >= 0) {
                    bwd_a = a.enc[qoff + j];
                    go = bwd_a < 4;
                }
            }
            if (!go) {
                phase = PH_BWD_END;
            } else {
                const int64_t e = (int64_t)(base + p) * nt + slot;
                pk = a.prev_k[e]; pl = a.prev_l[e]; ps = a.prev_s[e]; pn = a.prev_n[e];
                do_ext = true;
                ek = pk; el = pl; es = ps; ea = bwd_a;
            }
        }

        // ---- the one extension of this iteration -------------------------------------
        int64_t nk = 0, nl = 0, ns = 0;
        if (do_ext) {
            backward_ext(f, ek, el, es, ea, nk, nl, ns);
            n_ext++;
            n_blk += ((ek >> 6) == ((ek + es) >> 6)) ? 1 : 2;
        }

        // ---- post ---------------------------------------------------------------------
        if (do_ext && phase == PH_FWD) {
            // the extended interval is (k, l) = (nl, nk) after swapping strands back
            if (ns != cs) {
                const int64_t e = (int64_t)(cap - 1 - num_prev) * nt + slot;
                a.prev_k[e] = ck; a.prev_l[e] = cl; a.prev_s[e] = cs; a.prev_n[e] = cn;
                num_prev++;
            }
            if (ns < min_intv) {
                next_x = j;
                phase = PH_FWD_END;
                // FWD_END pushes cur only if cs >= min_intv: cur is still the old interval
            } else {
                ck = nl; cl = nk; cs = ns; cn = j;
                j++;
            }
            if (phase == PH_FWD_END) {
                if (cs >= min_intv) {
                    const int64_t e = (int64_t)(cap - 1 - num_prev) * nt + slot;
                    a.prev_k[e] = ck; a.prev_l[e] = cl; a.prev_s[e] = cs; a.prev_n[e] = cn;
                    num_prev++;
                }
                base = cap - num_prev;
                j = x - 1;
                p = 0; num_curr = 0; curr_s = -1; first = true;
                cur_m = x;
                phase = PH_BWD;
            }
        } else if (do_ext && phase == PH_BWD) {
            bool keep = false;
            if (first) {
                if (ns < min_intv && (pn - cur_m + 1) >= a.min_seed_len) {
                    emit_smem(a, rid, (uint32_t)cur_m, (uint32_t)pn, pk, pl, ps);
                    first = false;
                } else if (ns >= min_intv && ns != (int64_t)curr_s) {
                    keep = true;
                    first = false;
                }
            } else {
                keep = ns >= min_intv && ns != (int64_t)curr_s;
            }
            if (keep) {
                curr_s = (int32_t)ns;
                const int64_t e = (int64_t)(base + num_curr) * nt + slot;
                a.prev_k[e] = nk; a.prev_l[e] = nl; a.prev_s[e] = ns; a.prev_n[e] = pn;
                num_curr++;
            }
            p++;
            if (p == num_prev) {                         // this column is done
                num_prev = num_curr;
                if (num_curr == 0) {
                    phase = PH_BWD_END;
                } else {
                    cur_m = j;
                    j--;
                    p = 0; num_curr = 0; curr_s = -1; first = true;
                }
            }
        }
    }
    flush_counters(a.ctr, n_ext, n_blk);
}

// Select round-2 pivots from the round-1 SMEMs (src/bwamem.cpp:721-738).
__global__ void round2_work_kernel(const bwams_smem_t *pool, DevCounters *ctr, Round2Work *work,
                                   int64_t work_cap, int split_len, int split_width) {
    const int64_t n1 = (int64_t)ctr->n_after_r1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1;
         i += (int64_t)gridDim.x * blockDim.x) {
        const bwams_smem_t s = pool[i];
        const int start = (int)s.m, end = (int)s.n + 1;
        if (end - start < split_len || s.s > split_width) continue;
        const unsigned long long pos = atomicAdd(&ctr->n_work2, 1ull);
        if ((int64_t)pos < work_cap) {
            Round2Work w;
            w.rid = s.rid;
            w.x = (end + start) >> 1;
            w.min_intv = (int32_t)(s.s + 1);
            work[pos] = w;
        }
    }
}

// bookkeeping between rounds (single thread): snapshot the pool cursor, reset the queue
__global__ void mark_kernel(DevCounters *ctr, int which) {
    if (which == 1) ctr->n_after_r1 = ctr->n_smem_total;
    if (which == 2) ctr->n_after_r2 = ctr->n_smem_total;
    if (which >= 1 && which <= 3) {
        ctr->ext_after[which - 1] = ctr->n_ext;
        ctr->blk_after[which - 1] = ctr->n_ext_blocks;
    }
    ctr->work_head = 0;
}

// Round 3: forward-only seeds.
__global__ __launch_bounds__(kBlock) void seed_strategy_kernel(SeedLaunch a, int max_intv) {
    const DevFmi &f = a.fmi;
    int phase = PH_FETCH;
    uint32_t rid = 0;
    int64_t qoff = 0;
    int len = 0, x = 0, next_x = 0, j = 0;
    int64_t ck = 0, cl = 0, cs = 0;
    unsigned long long n_ext = 0, n_blk = 0;

    while (true) {
        {
            const bool want = phase == PH_FETCH;
            const unsigned long long t = take_ticket(&a.ctr->work_head, want);
            if (want) {
                if ((int64_t)t >= a.nseq) {
                    phase = PH_EXIT;
                } else {
                    rid = (uint32_t)t;
                    qoff = a.cum[rid];
                    len = (int)(a.cum[rid + 1] - qoff);
                    x = 0;
                    phase = PH_PIVOT;
                    if (a.skip && a.skip[rid]) phase = PH_FETCH;
                }
            }
        }
        if (__all(phase == PH_EXIT)) break;

        if (phase == PH_PIVOT) {
            if (x >= len) {
                phase = PH_FETCH;
            } else {
                const int c = a.enc[qoff + x];
                next_x = x + 1;
                if (c >= 4) {
                    x = next_x;
                } else {
                    ck = cnt_at(f, c);
                    cl = cnt_at(f, 3 - c);
                    cs = cnt_at(f, c + 1) - ck;
                    j = x + 1;
                    phase = PH_FWD;
                }
            }
        }
        if (phase == PH_FWD) {
            bool stop = true;
            if (j < len) {
                const int c = a.enc[qoff + j];
                next_x = j + 1;
                if (c < 4) {
                    int64_t nk, nl, ns;
                    backward_ext(f, cl, ck, cs, 3 - c, nk, nl, ns);
                    n_ext++;
                    n_blk += ((cl >> 6) == ((cl + cs) >> 6)) ? 1 : 2;
                    ck = nl; cl = nk; cs = ns;
                    stop = false;
                    if (cs < max_intv && (j - x + 1) >= a.min_seed_len) {
                        if (cs > 0) emit_smem(a, rid, (uint32_t)x, (uint32_t)j, ck, cl, cs);
                        stop = true;
                    }
                    j++;
                }
            }
            if (stop) {
                x = next_x;
                phase = PH_PIVOT;
            }
        }
    }
    flush_counters(a.ctr, n_ext, n_blk);
}

// (rid, m, n) sort key of each pooled SMEM
__global__ void make_keys_kernel(const bwams_smem_t *pool, int64_t n, uint64_t *keys, uint32_t *vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bwams_smem_t s = pool[i];
        keys[i] = ((uint64_t)s.rid << 32) | ((uint64_t)(s.m & 0xffff) << 16) | (uint64_t)(s.n & 0xffff);
        vals[i] = (uint32_t)i;
    }
}

__global__ void gather_sorted_kernel(const bwams_smem_t *pool, const uint32_t *order, int64_t n,
                                     bwams_smem_t *sorted, int64_t *sa_cnt, int max_occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bwams_smem_t s = pool[order[i]];
        sorted[i] = s;
        if (sa_cnt) sa_cnt[i] = s.s < (int64_t)max_occ ? s.s : (int64_t)max_occ;
    }
}

int grid_for(int64_t n_items, int cu_count) {
    int64_t blocks = (n_items + kBlock - 1) / kBlock;
    const int64_t maxb = (int64_t)cu_count * kBlocksPerCU;
    if (blocks > maxb) blocks = maxb;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

int seed_block_threads() { return kBlock; }
int64_t seed_max_threads(int cu_count) { return (int64_t)cu_count * kBlocksPerCU * kBlock; }

void launch_mark(DevCounters *ctr, int which, hipStream_t st) { mark_kernel<<<1, 1, 0, st>>>(ctr, which); }

void launch_smem_round1(const SeedLaunch &a, int cu_count, hipStream_t st) {
    smem_search_kernel<true><<<grid_for(a.nseq, cu_count), kBlock, 0, st>>>(a, nullptr);
}

void launch_round2_work(const SeedLaunch &a, Round2Work *work, int64_t work_cap, int split_len,
                        int split_width, int cu_count, hipStream_t st) {
    round2_work_kernel<<<cu_count * 4, 256, 0, st>>>(a.pool, a.ctr, work, work_cap, split_len, split_width);
}

void launch_smem_round2(const SeedLaunch &a, const Round2Work *work, int cu_count, hipStream_t st) {
    // the number of items is only known on the device: launch the persistent grid at chip size
    smem_search_kernel<false><<<grid_for(a.nseq, cu_count), kBlock, 0, st>>>(a, work);
}

void launch_smem_round3(const SeedLaunch &a, int max_intv, int cu_count, hipStream_t st) {
    seed_strategy_kernel<<<grid_for(a.nseq, cu_count), kBlock, 0, st>>>(a, max_intv);
}

void launch_make_keys(const bwams_smem_t *pool, int64_t n, uint64_t *keys, uint32_t *vals, hipStream_t st) {
    if (n <= 0) return;
    make_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pool, n, keys, vals);
}

void launch_gather_sorted(const bwams_smem_t *pool, const uint32_t *order, int64_t n, bwams_smem_t *sorted,
                          int64_t *sa_cnt, int max_occ, hipStream_t st) {
    if (n <= 0) return;
    gather_sorted_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pool, order, n, sorted, sa_cnt, max_occ);
}

}  // namespace bwams
