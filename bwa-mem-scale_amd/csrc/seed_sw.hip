// seed_sw.hip — mem_flt_chained_seeds on the device (/root/reference/src/bwamem.cpp:491-526).
//
// For reads long enough that 5.5 ln L <= 0.05 L (about 1100 bases and up; never for 150-bp reads)
// the reference re-scores every seed shorter than 200 bases of the kept chains with a local
// Smith-Waterman in a +-50-base window (mem_seed_sw, :425-449) and drops the seeds scoring below
// min_HSP_score.  Here: a plan kernel decides per seed whether the SW runs and lays out its task,
// the tasks go through the mate-rescue local-SW kernel (ksw_local.hip, the same ksw_align2
// semantics), an apply kernel compacts each chain's seed list in place, and the flat seed array is
// re-packed so that a read's regions stay one per remaining seed.
#include "common.h"
#include "chain_kernels.h"

namespace bwams {
namespace {

constexpr int MEM_SHORT_EXT = 50, MEM_SHORT_LEN = 200;
constexpr int KSW_XSTART = 0x80000;

__device__ __forceinline__ bool read_is_long(const bwams_mem_opt_t &o, int L, int *min_hsp) {
    const double min_l = o.min_chain_weight ? (double)(1.1f * (float)o.min_chain_weight) : (double)5.5f * log((double)L);
    *min_hsp = (int)((double)o.a * min_l + .499);
    return !(min_l > (double)(0.05f * (float)L));
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

// lane per chain: the SW window of each of its seeds (win[4 * slot] = qb, qe - qb, tlen; rb separately), or "no SW"
__global__ void seedsw_plan_kernel(SeedSwArgs A) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n_chains) return;
    const bwams_chain_t c = A.chains[j];
    const int L = (int)(A.cum[c.seqid + 1] - A.cum[c.seqid]);
    int min_hsp;
    const bool is_long = read_is_long(A.opt, L, &min_hsp);
    const int64_t l_pac = A.bns.l_pac;
    for (int i = 0; i < c.n; ++i) {
        const int64_t p = c.seed_off + i;
        const bwams_chain_seed_t s = A.seeds[p];
        int need = 0, ql = 0, tl = 0, qb = 0;
        int64_t rb = 0;
        if (is_long && s.len < MEM_SHORT_LEN) {
            int qe;
            int64_t re;
            qb = s.qbeg; qe = s.qbeg + s.len;
            rb = s.rbeg; re = s.rbeg + s.len;
            const int64_t mid = (rb + re) >> 1;
            qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
            qe += MEM_SHORT_EXT; qe = qe < L ? qe : L;
            rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
            re += MEM_SHORT_EXT; re = re < (l_pac << 1) ? re : (l_pac << 1);
            if (rb < l_pac && l_pac < re) {
                if (mid < l_pac) re = l_pac;
                else rb = l_pac;
            }
            if (!(qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN)) {
                const bool is_rev = mid >= l_pac;                       // bns_fetch_seq: clip to the sequence holding mid
                const int rid = pos2rid(A.bns, is_rev ? (l_pac << 1) - 1 - mid : mid);
                int64_t far_beg = A.bns.contigs[rid].offset, far_end = far_beg + A.bns.contigs[rid].len;
                if (is_rev) { const int64_t t0 = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t0; }
                rb = rb > far_beg ? rb : far_beg;
                re = re < far_end ? re : far_end;
                need = 1; ql = qe - qb; tl = (int)(re - rb);
            }
        }
        A.cnt[0 * A.n_seeds + p] = need; A.cnt[1 * A.n_seeds + p] = ql; A.cnt[2 * A.n_seeds + p] = tl;
        A.win_qb[p] = qb;
        A.win_rb[p] = rb;
    }
}

__global__ void seedsw_widen_kernel(const int32_t *cnt, int64_t n, int64_t *wide) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 3 * (n + 1)) return;
    const int64_t row = g / (n + 1), i = g - row * (n + 1);
    wide[g] = i < n ? (int64_t)cnt[row * n + i] : 0;
}

// wave per seed: the local-SW task (query window forward, reference window forward, xtra = KSW_XSTART)
__global__ __launch_bounds__(256) void seedsw_build_kernel(SeedSwArgs A, const int64_t *__restrict__ offs, bwams_seqpair_t *pairs,
                                                           uint8_t *ref, uint8_t *qer) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t N = A.n_seeds, n1 = N + 1;
    for (int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); p < N; p += stride) {
        if (!A.cnt[p]) continue;
        const int64_t ti = offs[p], qo = offs[n1 + p], ro = offs[2 * n1 + p];
        const int ql = A.cnt[N + p], tl = A.cnt[2 * N + p];
        const int r = A.seed_read[p];
        const int64_t qoff = A.cum[r] + A.win_qb[p], rb = A.win_rb[p];
        for (int t = lane; t < ql; t += 64) qer[qo + t] = A.enc[qoff + t];
        for (int t = lane; t < tl; t += 64) ref[ro + t] = A.ref[rb + t];
        if (lane == 0) {
            bwams_seqpair_t sp;
            sp.idr = (int32_t)ro; sp.idq = (int32_t)qo; sp.id = (int32_t)ti;
            sp.len1 = tl; sp.len2 = ql; sp.h0 = KSW_XSTART; sp.seqid = r; sp.regid = 0;
            sp.score = sp.tle = sp.gtle = sp.qle = sp.gscore = sp.max_off = 0;
            pairs[ti] = sp;
        }
    }
}

// lane per chain: which read each seed slot belongs to (the build kernel works per seed)
__global__ void seedsw_owner_kernel(SeedSwArgs A) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n_chains) return;
    const bwams_chain_t c = A.chains[j];
    for (int i = 0; i < c.n; ++i) A.seed_read[c.seed_off + i] = c.seqid;
}

// lane per chain: keep the seeds that pass, in place; new length
__global__ void seedsw_apply_kernel(SeedSwArgs A, const int64_t *__restrict__ offs, const bwams_kswr_t *__restrict__ res,
                                    int32_t *new_n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n_chains) return;
    bwams_chain_t c = A.chains[j];
    const int L = (int)(A.cum[c.seqid + 1] - A.cum[c.seqid]);
    int min_hsp;
    if (!read_is_long(A.opt, L, &min_hsp)) { new_n[j] = c.n; return; }
    int k = 0;
    for (int i = 0; i < c.n; ++i) {
        const int64_t p = c.seed_off + i;
        bwams_chain_seed_t s = A.seeds[p];
        const int sc = A.cnt[p] ? res[offs[p]].score : -1;
        if (sc < 0 || sc >= min_hsp) {
            s.score = sc < 0 ? s.len * A.opt.a : sc;
            A.seeds[c.seed_off + k] = s;
            ++k;
        }
    }
    new_n[j] = k;
}

// lane per chain: move the chain's seeds to their packed place, fix seed_off and n
__global__ void seedsw_repack_kernel(SeedSwArgs A, const int32_t *__restrict__ new_n, const int64_t *__restrict__ new_off,
                                     bwams_chain_seed_t *out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.n_chains) return;
    bwams_chain_t *c = &A.chains[j];
    const int64_t so = c->seed_off, no = new_off[j];
    const int n = new_n[j];
    for (int i = 0; i < n; ++i) out[no + i] = A.seeds[so + i];
    c->seed_off = no;
    c->n = n;
}

// lane per read: the read's first seed slot after repacking
__global__ void seedsw_readoff_kernel(const int64_t *__restrict__ chain_off, const int64_t *__restrict__ new_off, int64_t n_chains,
                                      int64_t nseq, int64_t *seed_off) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nseq) return;
    const int64_t j = chain_off[r];
    seed_off[r] = new_off[j < n_chains ? j : n_chains];
}

__global__ void widen1_kernel(const int32_t *a, int64_t n, int64_t *wide) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n) return;
    wide[g] = g < n ? (int64_t)a[g] : 0;
}

}  // namespace

void launch_seedsw_plan(const SeedSwArgs &A, int64_t *wide, hipStream_t st) {
    if (A.n_chains <= 0) return;
    const unsigned g = (unsigned)((A.n_chains + 255) / 256);
    seedsw_owner_kernel<<<g, 256, 0, st>>>(A);
    seedsw_plan_kernel<<<g, 256, 0, st>>>(A);
    const int64_t w = 3 * (A.n_seeds + 1);
    seedsw_widen_kernel<<<(unsigned)((w + 255) / 256), 256, 0, st>>>(A.cnt, A.n_seeds, wide);
}
void launch_seedsw_build(const SeedSwArgs &A, const int64_t *offs, bwams_seqpair_t *pairs, uint8_t *ref, uint8_t *qer,
                         int cu_count, hipStream_t st) {
    if (A.n_seeds <= 0) return;
    int64_t blocks = (A.n_seeds + 3) / 4;
    if (blocks > (int64_t)cu_count * 16) blocks = (int64_t)cu_count * 16;
    seedsw_build_kernel<<<(unsigned)blocks, 256, 0, st>>>(A, offs, pairs, ref, qer);
}
void launch_seedsw_apply(const SeedSwArgs &A, const int64_t *offs, const bwams_kswr_t *res, int32_t *new_n, int64_t *wide,
                         hipStream_t st) {
    if (A.n_chains <= 0) return;
    seedsw_apply_kernel<<<(unsigned)((A.n_chains + 255) / 256), 256, 0, st>>>(A, offs, res, new_n);
    widen1_kernel<<<(unsigned)((A.n_chains + 256) / 256), 256, 0, st>>>(new_n, A.n_chains, wide);
}
void launch_seedsw_repack(const SeedSwArgs &A, const int32_t *new_n, const int64_t *new_off, bwams_chain_seed_t *out,
                          const int64_t *chain_off, int64_t *seed_off, hipStream_t st) {
    if (A.n_chains > 0)
        seedsw_repack_kernel<<<(unsigned)((A.n_chains + 255) / 256), 256, 0, st>>>(A, new_n, new_off, out);
    seedsw_readoff_kernel<<<(unsigned)((A.nseq + 256) / 256), 256, 0, st>>>(chain_off, new_off, A.n_chains, A.nseq, seed_off);
}

}  // namespace bwams
