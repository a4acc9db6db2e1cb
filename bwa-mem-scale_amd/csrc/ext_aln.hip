// ext_aln.hip — from filtered chains to alignment regions, on the device.
//
// Replaces mem_chain2aln_across_reads_V2 (/root/reference/src/bwamem.cpp:2773-3760) for a whole
// chunk: per chain the reference window (cal_max_gap :94-104, strand and contig clip
// :2896-2925, bns_fetch_seq_v2 bntseq.cpp:484-520), per seed one region and up to two extension
// tasks in the reference's SeqPair layout (:2953-3188), the post-extension bookkeeping with the
// band-retry rule (:3240-3274 and its five copies), and the purge of seeds already covered by an
// earlier region (:3648-3755).  The banded Smith-Waterman itself is bsw_extend.hip.
//
// Regions share the index space of the flat seed array: region p of a read is the p-th seed
// visited (chain by chain, seeds by descending score then index), as in the reference's av->a.
#include "common.h"
#include "chain_kernels.h"
#include "wave_ops.h"

namespace bwams {
namespace {

constexpr int H0_ = -99;         // macro.h:56
// per-seed state of the extension rounds
constexpr int kExtKept = 1, kExtPurged = 2, kExtReq = 4, kExtDone = 8;

__device__ __forceinline__ int cal_max_gap(const bwams_mem_opt_t &o, int qlen) {
    const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
    const int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < (o.w << 1) ? l : (o.w << 1);
}

__device__ __forceinline__ int pos2rid(const DevBns &b, int64_t pos_f) {
    int left = 0, mid = 0, right = b.n_seqs;
    if (pos_f >= b.l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= b.contigs[mid].offset) {
            if (mid == b.n_seqs - 1) break;
            if (pos_f < b.contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

__device__ __forceinline__ int seedcov(const bwams_alnreg_t &a, const bwams_chain_t &c, const bwams_chain_seed_t *seeds) {
    int cov = 0;
    for (int i = 0; i < c.n; ++i) {
        const bwams_chain_seed_t *t = &seeds[c.seed_off + i];
        if (t->qbeg >= a.qb && t->qbeg + t->len <= a.qe && t->rbeg >= a.rb && t->rbeg + t->len <= a.re) cov += t->len;
    }
    return cov;
}

// lane per chain: window, seed order, regions, task sizes
__global__ void ext_plan_kernel(ExtArgs A, int extend_all) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n_chains) return;
    const bwams_chain_t c = A.chains[j];
    const int r = c.seqid;
    const int l_query = (int)(A.cum[r + 1] - A.cum[r]);
    const int64_t l_pac = A.bns.l_pac;
    bwams_chain_seed_t *cs = A.seeds + c.seed_off;
    if (c.n == 0) return;

    int64_t r0 = l_pac << 1, r1 = 0;
    for (int i = 0; i < c.n; ++i) {
        const int64_t rb = cs[i].rbeg;
        const int qb = cs[i].qbeg, ln = cs[i].len;
        const int64_t b = rb - (qb + cal_max_gap(A.opt, qb));
        const int64_t e = rb + ln + ((l_query - qb - ln) + cal_max_gap(A.opt, l_query - qb - ln));
        r0 = r0 < b ? r0 : b;
        r1 = r1 > e ? r1 : e;
    }
    r0 = r0 > 0 ? r0 : 0;
    r1 = r1 < (l_pac << 1) ? r1 : (l_pac << 1);
    const int64_t rbeg0 = cs[0].rbeg;
    if (r0 < l_pac && l_pac < r1) {
        if (rbeg0 < l_pac) r1 = l_pac;
        else r0 = l_pac;
    }
    {   // bns_fetch_seq_v2: clip to the reference sequence holding the first seed
        const bool is_rev = rbeg0 >= l_pac;
        const int rid = pos2rid(A.bns, is_rev ? (l_pac << 1) - 1 - rbeg0 : rbeg0);
        int64_t far_beg = A.bns.contigs[rid].offset, far_end = far_beg + A.bns.contigs[rid].len;
        if (is_rev) { const int64_t t0 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t0; }
        r0 = r0 > far_beg ? r0 : far_beg;
        r1 = r1 < far_end ? r1 : far_end;
    }
    A.rmax[2 * j] = r0; A.rmax[2 * j + 1] = r1;

    // srt: seed indices by ascending (score, index) — ks_introsort_64 over distinct keys
    uint32_t *srt = A.srt + c.seed_off;
    for (int i = 0; i < c.n; ++i) {
        const int sc = cs[i].score;
        int p = i;
        while (p > 0 && cs[srt[p - 1]].score > sc) { srt[p] = srt[p - 1]; --p; }
        srt[p] = (uint32_t)i;
    }

    const int64_t reg0 = A.seed_off[r];
    const int64_t N = A.n_seeds;
    for (int k = c.n - 1; k >= 0; --k) {
        const int64_t p = c.seed_off + (c.n - 1 - k);
        bwams_chain_seed_t *s = &cs[srt[k]];
        s->aln = (int32_t)(p - reg0);
        bwams_alnreg_t a;
        a.rb = a.re = H0_; a.qb = a.qe = H0_;
        a.rid = c.rid; a.pad0_ = 0;
        a.chain = j;
        a.score = a.truesc = -1;
        a.sub = a.alt_sc = a.csub = a.sub_n = 0;
        a.w = A.opt.w; a.seedcov = 0; a.secondary = a.secondary_all = 0;
        a.seedlen0 = s->len; a.n_comp_is_alt = 0;
        a.frac_rep = c.frac_rep; a.pad1_ = 0; a.hash = 0; a.flg = 0; a.pad2_ = 0;
        int nl = 0, lq = 0, lr = 0, nr = 0, rq = 0, rr = 0;
        if (s->qbeg) {
            nl = 1; lq = s->qbeg; lr = (int)(s->rbeg - r0);
            a.qb = s->qbeg; a.rb = s->rbeg;
        } else {
            a.score = a.truesc = s->len * A.opt.a; a.qb = 0; a.rb = s->rbeg;
        }
        if (s->qbeg + s->len != l_query) {
            const int64_t qe = s->qbeg + s->len;
            const int64_t re = s->rbeg + s->len - r0;
            nr = 1; rq = (int)(l_query - qe); rr = (int)(r1 - r0 - re);
            a.qe = (int32_t)qe; a.re = r0 + re;
        } else {
            a.qe = l_query; a.re = s->rbeg + s->len;
            a.seedcov = seedcov(a, c, A.seeds);       // rb, qb are always set at this point
        }
        A.regs[p] = a;
        A.cnt[0 * N + p] = nl; A.cnt[1 * N + p] = lq; A.cnt[2 * N + p] = lr;
        A.cnt[3 * N + p] = nr; A.cnt[4 * N + p] = rq; A.cnt[5 * N + p] = rr;
        // the first round extends the first seed visited of every chain (every seed when extend_all):
        // those are the seeds most likely to survive the containment test
        int st = 0;
        if (!nl && !nr) st = kExtDone;                       // nothing to extend
        else if (extend_all || k == c.n - 1) st = kExtReq;
        A.state[p] = st;
    }
}

// task sizes of the seeds requested this round, widened for the scans
__global__ void ext_widen_kernel(const int32_t *cnt, const int32_t *state, int64_t n, int64_t *wide) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 6 * (n + 1)) return;
    const int64_t row = g / (n + 1), i = g - row * (n + 1);
    wide[g] = (i < n && (state[i] & kExtReq)) ? (int64_t)cnt[row * n + i] : 0;
}

// SeqPair records and sequence copies.  A wavefront takes 64 slots at a time: every lane looks at one slot's state (most slots are
// not requested in a round), a requested slot's lane fetches what the slot needs — chain, place in the chain's order, seed, offsets:
// dependent loads, in flight for all the requested slots of the 64 together — and writes its SeqPair records; then the wave copies
// the bytes of one requested slot after the other, the slot's fields broadcast from its lane.  (A wave per slot paid the chain of
// dependent loads once per slot: 5.2 ms per step for 3.6 M tasks among 21.9 M slots.)
__device__ __forceinline__ int64_t shfl64(int64_t v, int src) {
    return ((int64_t)__shfl((int)(v >> 32), src) << 32) | (uint32_t)__shfl((int)v, src);
}
// lsrc / rsrc != nullptr: the tasks are extended IN PLACE (bwams_extend_run) — a task's sequences are where they lie, the read in the
// chunk's base codes and the window in the resident .0123 text, read backwards for a left extension; what is written per task is the two
// start offsets {query, target} and no byte is copied (the copies were 2.2 of the 3.9 ms this stage took per million reads).  The flat
// buffers of the SeqPair boundary are still built for bwams_extend_build / bwams_extend_tasks_fetch.
__global__ __launch_bounds__(256) void ext_build_kernel(ExtArgs A, const int64_t *__restrict__ offs, bwams_seqpair_t *left,
                                                        uint8_t *lref, uint8_t *lqer, bwams_seqpair_t *right, uint8_t *rref,
                                                        uint8_t *rqer, int64_t *lsrc, int64_t *rsrc) {
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t N = A.n_seeds, n1 = N + 1;
    for (int64_t p0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; p0 < N; p0 += n_waves * 64) {
        const int64_t p = p0 + lane;
        const int st = p < N ? A.state[p] : 0;
        const bool req = (st & kExtReq) != 0;
        // left: query bytes enc[l_qsrc - t], t < l_ql, to lqer[l_qo + t]; reference bytes ref[l_rsrc - t], t < l_rl, to lref[l_ro + t]
        int64_t l_qsrc = 0, l_rsrc = 0, l_qo = 0, l_ro = 0, r_qsrc = 0, r_rsrc = 0, r_qo = 0, r_ro = 0;
        int l_ql = 0, l_rl = 0, r_ql = 0, r_rl = 0;
        if (req) {
            const int nl = A.cnt[0 * N + p], nr = A.cnt[3 * N + p];
            A.state[p] = (st & ~kExtReq) | kExtDone;      // extended by the time the next selection runs
            const int64_t j = A.regs[p].chain;
            const bwams_chain_t c = A.chains[j];
            const int k = c.n - 1 - (int)(p - c.seed_off);
            const bwams_chain_seed_t sd = A.seeds[c.seed_off + A.srt[c.seed_off + k]];
            const int r = c.seqid;
            const int64_t qoff = A.cum[r];
            const int l_query = (int)(A.cum[r + 1] - qoff);
            const int64_t r0 = A.rmax[2 * j];
            if (nl) {
                const int64_t ti = offs[0 * n1 + p];
                l_qo = offs[1 * n1 + p]; l_ro = offs[2 * n1 + p];
                l_ql = sd.qbeg; l_rl = (int)(sd.rbeg - r0);
                l_qsrc = qoff + sd.qbeg - 1; l_rsrc = sd.rbeg - 1;
                bwams_seqpair_t sp;
                sp.idr = (int32_t)l_ro; sp.idq = (int32_t)l_qo; sp.id = (int32_t)ti;
                sp.len1 = l_rl; sp.len2 = l_ql; sp.h0 = sd.len * A.opt.a; sp.seqid = r; sp.regid = sd.aln;
                sp.score = sp.tle = sp.gtle = sp.qle = sp.gscore = sp.max_off = 0;
                left[ti] = sp;
                if (lsrc) { lsrc[2 * ti] = l_qsrc; lsrc[2 * ti + 1] = l_rsrc; }
            }
            if (nr) {
                const int64_t ti = offs[3 * n1 + p];
                r_qo = offs[4 * n1 + p]; r_ro = offs[5 * n1 + p];
                const int qe = sd.qbeg + sd.len;
                r_ql = l_query - qe; r_rl = A.cnt[5 * N + p];
                r_qsrc = qoff + qe; r_rsrc = sd.rbeg + sd.len;
                bwams_seqpair_t sp;
                sp.idr = (int32_t)r_ro; sp.idq = (int32_t)r_qo; sp.id = (int32_t)ti;
                sp.len1 = r_rl; sp.len2 = r_ql; sp.h0 = H0_; sp.seqid = r; sp.regid = sd.aln;
                sp.score = sp.tle = sp.gtle = sp.qle = sp.gscore = sp.max_off = 0;
                right[ti] = sp;
                if (rsrc) { rsrc[2 * ti] = r_qsrc; rsrc[2 * ti + 1] = r_rsrc; }
            }
        }
        unsigned long long m = lsrc ? 0ull : __ballot(req);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int ql = __shfl(l_ql, src), rl = __shfl(l_rl, src), qr = __shfl(r_ql, src), rr = __shfl(r_rl, src);
            if (ql | rl) {
                const int64_t qs = shfl64(l_qsrc, src), rs = shfl64(l_rsrc, src), qo = shfl64(l_qo, src), ro = shfl64(l_ro, src);
                for (int t = lane; t < ql; t += 64) lqer[qo + t] = A.enc[qs - t];
                for (int t = lane; t < rl; t += 64) lref[ro + t] = A.ref[rs - t];
            }
            if (qr | rr) {
                const int64_t qs = shfl64(r_qsrc, src), rs = shfl64(r_rsrc, src), qo = shfl64(r_qo, src), ro = shfl64(r_ro, src);
                for (int t = lane; t < qr; t += 64) rqer[qo + t] = A.enc[qs + t];
                for (int t = lane; t < rr; t += 64) rref[ro + t] = A.ref[rs + t];
            }
        }
    }
}

// lane per task, after one extension attempt
__global__ void ext_post_kernel(ExtArgs A, int right, const bwams_seqpair_t *__restrict__ pairs, int64_t n, int w, int last_try,
                                bwams_seqpair_t *retry, unsigned long long *n_retry) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const bwams_seqpair_t sp = pairs[l];
    bwams_alnreg_t *ap = &A.regs[A.seed_off[sp.seqid] + sp.regid];
    bwams_alnreg_t a = *ap;
    const int prev = a.score;
    a.score = sp.score;
    if (a.score == prev || sp.max_off < (w >> 1) + (w >> 2) || last_try) {
        if (!right) {
            if (sp.gscore <= 0 || sp.gscore <= a.score - A.opt.pen_clip5) {
                a.qb -= sp.qle; a.rb -= sp.tle;
                a.truesc = a.score;
            } else {
                a.qb = 0; a.rb -= sp.gtle;
                a.truesc = sp.gscore;
            }
        } else {
            if (sp.gscore <= 0 || sp.gscore <= a.score - A.opt.pen_clip3) {
                a.qe += sp.qle; a.re += sp.tle;
                a.truesc += a.score - sp.h0;
            } else {
                a.qe = (int32_t)(A.cum[sp.seqid + 1] - A.cum[sp.seqid]); a.re += sp.gtle;
                a.truesc += sp.gscore - sp.h0;
            }
        }
        a.w = a.w > w ? a.w : w;
        if (a.rb != H0_ && a.qb != H0_ && a.qe != H0_ && a.re != H0_) a.seedcov = seedcov(a, A.chains[a.chain], A.seeds);
        *ap = a;
    } else {
        ap->score = a.score;
        const unsigned long long slot = atomicAdd(n_retry, 1ull);
        retry[slot] = sp;
    }
}

// the right extension starts from the score the left one reached (bwamem.cpp:3425-3430)
__global__ void ext_right_h0_kernel(ExtArgs A, bwams_seqpair_t *right, int64_t n) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    right[l].h0 = A.regs[A.seed_off[right[l].seqid] + right[l].regid].score;
}

// ---- purge: drop seeds (and their regions) that an earlier region already explains ----------
// (bwamem.cpp:3648-3755).  Sequential per read; the scan over the read's regions is quadratic, so
// reads with few regions take one lane each and the others one wave each (the scan over regions
// then runs 64 at a time, with ballots standing in for the loop's counters and its break).
constexpr int kLightRegs = 32;

// does region p "explain" seed s?  0: p is purged (skipped), 1: no (v++), 2: yes (break)
__device__ __forceinline__ int purge_class(const bwams_mem_opt_t &opt, const bwams_chain_seed_t &s, int l_query, int64_t prb,
                                           int64_t pre, int pqb, int pqe, int pseedlen0, int pw) {
    if (pqb == -1 && pqe == -1) return 0;
    if (s.rbeg < prb || s.rbeg + s.len > pre || s.qbeg < pqb || s.qbeg + s.len > pqe) return 1;
    if ((double)(s.len - pseedlen0) > .1 * (double)l_query) return 1;
    int qd = s.qbeg - pqb;
    int64_t rd = s.rbeg - prb;
    int max_gap = cal_max_gap(opt, (int)(qd < rd ? qd : rd));
    int w = max_gap < pw ? max_gap : pw;
    if (qd - rd < w && rd - qd < w) return 2;
    qd = pqe - (s.qbeg + s.len); rd = pre - (s.rbeg + s.len);
    max_gap = cal_max_gap(opt, (int)(qd < rd ? qd : rd));
    w = max_gap < pw ? max_gap : pw;
    if (qd - rd < w && rd - qd < w) return 2;
    return 1;
}

// is there a longer-or-similar seed later in the visiting order that overlaps s off-diagonal?
__device__ __forceinline__ bool purge_keep_anyway(const bwams_chain_seed_t &s, const bwams_chain_seed_t *cs, const uint32_t *srt2,
                                                  int k, int n) {
    int v;
    for (v = k + 1; v < n; ++v) {
        const uint32_t sv = __hip_atomic_load(&srt2[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sv == 0xffffffffu) continue;
        const bwams_chain_seed_t t = cs[sv];
        if ((double)t.len < (double)s.len * .95) continue;
        if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= (s.len >> 2) && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
        if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= (s.len >> 2) && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
    }
    return v != n;
}

// Selection (one round): walk the read's seeds in visiting order from where the last round stopped.
// A seed explained by a region kept so far — and not rescued by the overlap test — is purged without
// ever being extended; the reference extends it first and throws the result away (the purged region
// is dropped by mem_kernel2_core, bwamem.cpp:1446-1456), and its test reads nothing but regions kept
// EARLIER (its scan stops after `lim` unpurged regions, which are exactly those).  A seed that must
// be kept needs its own extension: if that is not done yet it is requested and the read waits for
// the next round.
struct KReg { int64_t rb, re; int32_t qb, qe, seedlen0, w; };      // what the test reads of a kept region

__device__ __forceinline__ void select_seed_of_slot(const ExtArgs &A, int64_t p, bwams_chain_t &c, int &k, bwams_chain_seed_t &s) {
    c = A.chains[A.regs[p].chain];
    k = c.n - 1 - (int)(p - c.seed_off);
    const uint32_t sk = __hip_atomic_load(&A.srt[c.seed_off + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = A.seeds[c.seed_off + sk];
}

__global__ __launch_bounds__(64) void ext_select_kernel(ExtArgs A) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool req = false;
    int rest = 0;
    if (r < A.nseq) {
        const int64_t reg0 = A.seed_off[r];
        const int av_n = (int)(A.seed_off[r + 1] - reg0);
        int t = A.cur[r];
        if (av_n <= kLightRegs && t < av_n) {
            const int l_query = (int)(A.cum[r + 1] - A.cum[r]);
            KReg *kreg = reinterpret_cast<KReg *>(A.kreg) + reg0;
            int lim = A.lim[r];
            for (; t < av_n; ++t) {
                const int64_t p = reg0 + t;
                bwams_chain_t c;
                bwams_chain_seed_t s;
                int k;
                select_seed_of_slot(A, p, c, k, s);
                bool brk = false;
                for (int i = 0; i < lim && !brk; ++i) {
                    const KReg q = kreg[i];
                    brk = purge_class(A.opt, s, l_query, q.rb, q.re, q.qb, q.qe, q.seedlen0, q.w) == 2;
                }
                const int st = A.state[p];
                if (brk && !purge_keep_anyway(s, A.seeds + c.seed_off, A.srt + c.seed_off, k, c.n)) {
                    A.regs[p].qb = -1; A.regs[p].qe = -1;
                    A.srt[c.seed_off + k] = 0xffffffffu;
                    A.state[p] = st | kExtPurged;
                    continue;
                }
                if (!(st & kExtDone)) { A.state[p] = st | kExtReq; req = true; rest = av_n - t - 1; break; }
                const bwams_alnreg_t *a = &A.regs[p];
                KReg q;
                q.rb = a->rb; q.re = a->re; q.qb = a->qb; q.qe = a->qe; q.seedlen0 = a->seedlen0; q.w = a->w;
                kreg[lim++] = q;
                A.state[p] = st | kExtKept;
            }
            A.cur[r] = t;
            A.lim[r] = lim;
        }
    }
    const unsigned long long m = __ballot(req);
    if (m) {
        for (int o = 32; o > 0; o >>= 1) rest += __shfl_xor(rest, o);
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) {
            atomicAdd(&A.ctr->n_req, (unsigned long long)__popcll(m));
            if (rest) atomicAdd(&A.ctr->n_rest, (unsigned long long)rest);
        }
    }
}

// lane per read, once per run: the reads the wave tier of the selection handles
__global__ void ext_heavy_list_kernel(ExtArgs A) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    if (A.seed_off[r + 1] - A.seed_off[r] > kLightRegs) A.sel_heavy[atomicAdd(A.n_sel_heavy, 1ull)] = (int32_t)r;
}

// purge_keep_anyway over the wavefront: the later seeds of the chain 64 at a time (the scalar loop stops at the first hit and
// reports whether there was one — an "any")
__device__ __forceinline__ bool purge_keep_anyway_w(const bwams_chain_seed_t &s, const bwams_chain_seed_t *cs, const uint32_t *srt2, int k, int n,
                                                    int lane) {
    bool found = false;
    for (int v0 = k + 1; v0 < n && !found; v0 += 64) {
        const int v = v0 + lane;
        bool hit = false;
        if (v < n) {
            const uint32_t sv = __hip_atomic_load(&srt2[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sv != 0xffffffffu) {
                const bwams_chain_seed_t t = cs[sv];
                if (!((double)t.len < (double)s.len * .95)) {
                    hit = (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= (s.len >> 2) && t.qbeg - s.qbeg != t.rbeg - s.rbeg) ||
                          (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= (s.len >> 2) && s.qbeg - t.qbeg != s.rbeg - t.rbeg);
                }
            }
        }
        found = __any(hit);
    }
    return found;
}

// Wave per read.  The decisions are sequential in the visiting order, but what a slot needs to be decided — its chain, its place
// in the chain's order, its seed, its state — depends on no decision (a purge rewrites only the purged slot's own entries), so the
// wave fetches 64 slots at once, one per lane (four dependent loads each, in flight together), and then walks them with the
// slot's fields broadcast from its lane.  (One slot at a time, every slot paid the four load latencies: the kernel's duration was
// the ~1000 seeds of the heaviest read times ~3 us, the same at any chunk size.)
// Round 3: the kept regions a slot is tested against (KReg, 32 B; 64 per step) live in LDS while a read is walked.  Instrumented
// (-DBWAMS_SELDBG, BWAMS_VERBOSE): the kernel's 3.2 ms were ONE read — 1191 slots, 10.8 k steps of that scan at ~650 cycles each, i.e.
// the L2 round trip of the kreg load, 79 % of its 8.9 M cycles.  The copy in HBM stays the state between rounds (loaded at the start
// of a read, written through).  Three size classes, each its own launch with its own work cursor over the same list: up to 256 regions
// (8 KB of LDS per wave: sixteen waves per CU), up to 640 (20 KB: seven), up to 1280 (40 KB: three; the reads beyond fall back to HBM).
// The top class holds a few dozen reads per million on a genome like the bench's, so its longest walk starts when the kernel does; on
// a repeat-heavy genome (27 k reads per million beyond 128 regions) the middle class is what keeps enough wavefronts on them.
constexpr int kSelCap[3] = {256, 640, 1280};
__global__ __launch_bounds__(64) void ext_select_wave_kernel(ExtArgs A, int lo, int cap, unsigned long long *ticket) {
    extern __shared__ __align__(16) unsigned char l_sel_raw[];
    KReg *lk = reinterpret_cast<KReg *>(l_sel_raw);
    const int lane = threadIdx.x & 63;
    const int64_t n_heavy = (int64_t)*A.n_sel_heavy;
    for (;;) {
        const int64_t ti = (int64_t)wave_ticket(ticket, 1ull);
        if (ti >= n_heavy) break;
        const int64_t r = A.sel_heavy[ti];
        const int64_t reg0 = A.seed_off[r];
        const int av_n = (int)(A.seed_off[r + 1] - reg0);
        if (av_n <= lo || (av_n > cap && cap != kSelCap[2])) continue;        // another class's read
        int t = A.cur[r];
        if (t >= av_n) continue;
        const int l_query = (int)(A.cum[r + 1] - A.cum[r]);
        KReg *kreg = reinterpret_cast<KReg *>(A.kreg) + reg0;
        int lim = A.lim[r];
        const bool in_lds = av_n <= cap;
        __syncthreads();                                                      // (one wavefront per block) the previous read is done with lk
        if (in_lds)
            for (int i = lane; i < lim; i += 64) lk[i] = kreg[i];
        __syncthreads();
        bool stop = false;
#ifdef BWAMS_SELDBG                 // phase timers of the walk (printed by bwams_extend_run under BWAMS_VERBOSE)
        const unsigned long long T0 = __builtin_amdgcn_s_memtime();
        unsigned long long t_scan = 0, n_slots = 0, n_chunks = 0;
#endif
        while (t < av_n && !stop) {
            const int nb = av_n - t < 64 ? av_n - t : 64;
            int64_t my_off = 0, my_rbeg = 0;
            int my_n = 0, my_k = 0, my_st = 0, my_qbeg = 0, my_len = 0;
            if (lane < nb) {
                const int64_t p = reg0 + t + lane;
                bwams_chain_t c;
                bwams_chain_seed_t sd;
                select_seed_of_slot(A, p, c, my_k, sd);
                my_off = c.seed_off; my_n = c.n; my_rbeg = sd.rbeg; my_qbeg = sd.qbeg; my_len = sd.len;
                my_st = A.state[p];
            }
            int j = 0;
            for (; j < nb; ++j) {
                const int64_t p = reg0 + t + j;
                const int64_t c_off = ((int64_t)__shfl((int)(my_off >> 32), j) << 32) | (uint32_t)__shfl((int)my_off, j);
                bwams_chain_seed_t s;
                s.rbeg = ((int64_t)__shfl((int)(my_rbeg >> 32), j) << 32) | (uint32_t)__shfl((int)my_rbeg, j);
                s.qbeg = __shfl(my_qbeg, j); s.len = __shfl(my_len, j);
                const int c_n = __shfl(my_n, j), k = __shfl(my_k, j), st = __shfl(my_st, j);
                bool brk = false;
#ifdef BWAMS_SELDBG
                const unsigned long long Tb = __builtin_amdgcn_s_memtime();
                ++n_slots;
#endif
                for (int base = 0; base < lim && !brk; base += 64) {          // the kept regions, 64 at a time
                    const int i = base + lane;
                    int cls = 0;
                    if (i < lim) {
                        const KReg q = in_lds ? lk[i] : kreg[i];
                        cls = purge_class(A.opt, s, l_query, q.rb, q.re, q.qb, q.qe, q.seedlen0, q.w);
                    }
                    brk = __ballot(cls == 2) != 0;
#ifdef BWAMS_SELDBG
                    ++n_chunks;
#endif
                }
#ifdef BWAMS_SELDBG
                t_scan += __builtin_amdgcn_s_memtime() - Tb;
#endif
                if (brk && !purge_keep_anyway_w(s, A.seeds + c_off, A.srt + c_off, k, c_n, lane)) {
                    if (lane == 0) {
                        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&A.regs[p].qb), 0xffffffffffffffffull,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // qb = qe = -1
                        __hip_atomic_store(&A.srt[c_off + k], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        A.state[p] = st | kExtPurged;
                    }
                    continue;
                }
                if (!(st & kExtDone)) {
                    if (lane == 0) {
                        A.state[p] = st | kExtReq;
                        atomicAdd(&A.ctr->n_req, 1ull);
                        if (av_n - (t + j) - 1 > 0) atomicAdd(&A.ctr->n_rest, (unsigned long long)(av_n - (t + j) - 1));
                    }
                    stop = true;
                    break;
                }
                if (lane == 0) {
                    const bwams_alnreg_t *a = &A.regs[p];
                    KReg q;
                    q.rb = a->rb; q.re = a->re; q.qb = a->qb; q.qe = a->qe; q.seedlen0 = a->seedlen0; q.w = a->w;
                    kreg[lim] = q;
                    if (in_lds) lk[lim] = q;
                    A.state[p] = st | kExtKept;
                }
                ++lim;
                __syncthreads();                                              // the appended region is read by every lane from the next slot on
            }
            t += j;                                                          // a request leaves t at the requested slot
        }
        if (lane == 0) { A.cur[r] = t; A.lim[r] = lim; }
#ifdef BWAMS_SELDBG
        if (lane == 0) {
            const unsigned long long tot = __builtin_amdgcn_s_memtime() - T0;
            unsigned long long *d = A.ctr->dbg;
            atomicAdd(&d[0], 1ull); atomicAdd(&d[1], tot); atomicAdd(&d[3], t_scan); atomicAdd(&d[5], n_slots); atomicAdd(&d[6], n_chunks);
            if (atomicMax(&d[8], tot) < tot) { d[10] = t_scan; d[12] = n_slots; d[13] = n_chunks; d[15] = (unsigned long long)av_n; }
        }
#endif
    }
}

// after too many rounds: request every seed that is still undecided and unextended
__global__ void ext_request_rest_kernel(ExtArgs A) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n_seeds) return;
    const int st = A.state[p];
    if (!(st & (kExtKept | kExtPurged | kExtDone | kExtReq))) A.state[p] = st | kExtReq;
}

}  // namespace

void launch_ext_plan(const ExtArgs &A, int extend_all, hipStream_t st) {
    if (A.n_chains > 0) ext_plan_kernel<<<(unsigned)((A.n_chains + 63) / 64), 64, 0, st>>>(A, extend_all);
}
void launch_ext_widen(const ExtArgs &A, int64_t *wide, hipStream_t st) {
    const int64_t g = 6 * (A.n_seeds + 1);
    ext_widen_kernel<<<(unsigned)((g + 255) / 256), 256, 0, st>>>(A.cnt, A.state, A.n_seeds, wide);
}

void launch_ext_build(const ExtArgs &A, const int64_t *offs, bwams_seqpair_t *left, uint8_t *lref, uint8_t *lqer,
                      bwams_seqpair_t *right, uint8_t *rref, uint8_t *rqer, int64_t *lsrc, int64_t *rsrc, int cu_count, hipStream_t st) {
    if (A.n_seeds <= 0) return;
    int64_t blocks = (A.n_seeds + 255) / 256;              // a wave per 64 slots
    if (blocks > (int64_t)cu_count * 16) blocks = (int64_t)cu_count * 16;
    ext_build_kernel<<<(unsigned)blocks, 256, 0, st>>>(A, offs, left, lref, lqer, right, rref, rqer, lsrc, rsrc);
}

void launch_ext_post(const ExtArgs &A, int right, const bwams_seqpair_t *pairs, int64_t n, int w, int last_try,
                     bwams_seqpair_t *retry, unsigned long long *n_retry, hipStream_t st) {
    if (n <= 0) return;
    ext_post_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(A, right, pairs, n, w, last_try, retry, n_retry);
}

void launch_ext_right_h0(const ExtArgs &A, bwams_seqpair_t *right, int64_t n, hipStream_t st) {
    if (n <= 0) return;
    ext_right_h0_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(A, right, n);
}

void launch_ext_heavy_list(const ExtArgs &A, hipStream_t st) {
    if (A.nseq <= 0) return;
    ext_heavy_list_kernel<<<(unsigned)((A.nseq + 255) / 256), 256, 0, st>>>(A);
}
int launch_ext_select(const ExtArgs &A, int cu_count, hipStream_t st, hipStream_t *aux, hipEvent_t fork, hipEvent_t *join) {
    if (A.nseq <= 0) return 0;
    if (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(aux[0], fork, 0) != hipSuccess ||
        hipStreamWaitEvent(aux[1], fork, 0) != hipSuccess) return -1;
    ext_select_wave_kernel<<<(unsigned)(cu_count * 3), 64, kSelCap[2] * sizeof(KReg), aux[0]>>>(A, kSelCap[1], kSelCap[2], A.sel_ticket + 2);
    ext_select_wave_kernel<<<(unsigned)(cu_count * 7), 64, kSelCap[1] * sizeof(KReg), aux[1]>>>(A, kSelCap[0], kSelCap[1], A.sel_ticket + 1);
    ext_select_kernel<<<(unsigned)((A.nseq + 63) / 64), 64, 0, st>>>(A);
    ext_select_wave_kernel<<<(unsigned)(cu_count * 16), 64, kSelCap[0] * sizeof(KReg), st>>>(A, kLightRegs, kSelCap[0], A.sel_ticket);
    for (int i = 0; i < 2; ++i)
        if (hipEventRecord(join[i], aux[i]) != hipSuccess || hipStreamWaitEvent(st, join[i], 0) != hipSuccess) return -1;
    return 0;
}
void launch_ext_request_rest(const ExtArgs &A, hipStream_t st) {
    if (A.n_seeds <= 0) return;
    ext_request_rest_kernel<<<(unsigned)((A.n_seeds + 255) / 256), 256, 0, st>>>(A);
}

}  // namespace bwams
