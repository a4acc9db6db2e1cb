// seed_tasks.hip — extension tasks straight from seeds, on the device (bwams_tasks_from_seeds).
//
// The reference builds its extension tasks from chains (mem_chain_seeds -> mem_chain_flt ->
// mem_chain2aln_across_reads_V2, /root/reference/src/bwamem.cpp:789-959, :528-646, :2849-3191);
// that path is chain.hip + ext_aln.hip.  This older entry point, kept in the ABI as a cheap
// seeds-only probe of the extension stage, treats
// the longest non-repetitive seed of each read as a one-seed chain and lays out
// its left and right tasks exactly as the reference does for such a chain: window from
// cal_max_gap (bwamem.cpp:94-104, :2880-2905), strand clipping (:2906-2910), left = reversed
// query prefix vs reversed reference window with h0 = seed_len * a (:2953-3060), right = query
// suffix vs the window after the seed (:3061-3188; its h0 is the left score in the
// reference, the seed score here).  bwams/pairs.py is the same rule in numpy and is what
// tests compare this file with.
#include "common.h"

namespace bwams {
namespace {

struct TaskPlan {          // per read
    int32_t qbeg, slen;    // seed span on the read; slen = 0: no task
    int64_t rbeg, r0, r1;
};

__device__ __forceinline__ int64_t max_gap(int qlen, int a, int o, int e, int w) {
    int64_t l = (int64_t)((double)(qlen * a - o) / e + 1.);
    l = l > 1 ? l : 1;
    const int64_t lim = (int64_t)w << 1;
    return l < lim ? l : lim;
}

// lane per read: pick the seed, compute the windows and the byte counts
__global__ void plan_kernel(const bwams_smem_t *__restrict__ sm, int64_t n_smem, const int64_t *__restrict__ sa_off,
                            const int64_t *__restrict__ sa_coord, const int64_t *__restrict__ cum, int64_t nseq,
                            int64_t l_pac, int max_occ, int a, int o_gap, int e_gap, int w, TaskPlan *plan,
                            int32_t *cnt /* 6 x nseq: nl, lq, lr, nr, rq, rr */) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nseq) return;
    // segment of this read in the (rid, m, n)-sorted SMEM array
    int64_t lo = 0, hi = n_smem;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sm[mid].rid < r) lo = mid + 1; else hi = mid; }
    const int64_t beg = lo;
    hi = n_smem;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sm[mid].rid <= r) lo = mid + 1; else hi = mid; }
    const int64_t end = lo;
    int64_t pick = -1;
    int best = 0;
    for (int64_t i = beg; i < end; ++i) {
        const bwams_smem_t s = sm[i];
        if (s.s > (int64_t)max_occ || s.s <= 0) continue;
        const int len = (int)s.n - (int)s.m + 1;
        if (len > best) { best = len; pick = i; }            // ties: the earliest wins
    }
    TaskPlan p;
    p.qbeg = 0; p.slen = 0; p.rbeg = p.r0 = p.r1 = 0;
    int nl = 0, lq = 0, lr = 0, nr = 0, rq = 0, rr = 0;
    if (pick >= 0) {
        const int L = (int)(cum[r + 1] - cum[r]);
        const int qbeg = (int)sm[pick].m, ln = best, qend = qbeg + ln;
        const int64_t rbeg = sa_coord[sa_off[pick]];
        if (!(rbeg < l_pac && rbeg + ln > l_pac)) {
            int64_t r0 = rbeg - (qbeg + max_gap(qbeg, a, o_gap, e_gap, w));
            r0 = r0 > 0 ? r0 : 0;
            int64_t r1 = rbeg + ln + (L - qend) + max_gap(L - qend, a, o_gap, e_gap, w);
            r1 = r1 < 2 * l_pac ? r1 : 2 * l_pac;
            if (rbeg < l_pac) r1 = r1 < l_pac ? r1 : l_pac;
            else r0 = r0 > l_pac ? r0 : l_pac;
            p.qbeg = qbeg; p.slen = ln; p.rbeg = rbeg; p.r0 = r0; p.r1 = r1;
            if (qbeg > 0) { nl = 1; lq = qbeg; lr = (int)(rbeg - r0); }
            if (qend < L) { nr = 1; rq = L - qend; rr = (int)(r1 - rbeg - ln); }
        }
    }
    plan[r] = p;
    cnt[0 * nseq + r] = nl; cnt[1 * nseq + r] = lq; cnt[2 * nseq + r] = lr;
    cnt[3 * nseq + r] = nr; cnt[4 * nseq + r] = rq; cnt[5 * nseq + r] = rr;
}

// wave per read: write the SeqPair records and copy / reverse the sequences
__global__ __launch_bounds__(256) void build_kernel(const TaskPlan *__restrict__ plan, const int32_t *__restrict__ cnt,
                                                    const int64_t *__restrict__ offs /* 6 x (nseq+1) exclusive */,
                                                    const uint8_t *__restrict__ enc, const int64_t *__restrict__ cum,
                                                    const uint8_t *__restrict__ ref0123, int64_t nseq, int a,
                                                    bwams_seqpair_t *pairs, uint8_t *refbuf, uint8_t *qerbuf) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t n1 = nseq + 1;
    const int64_t tot_nl = offs[0 * n1 + nseq], tot_lq = offs[1 * n1 + nseq], tot_lr = offs[2 * n1 + nseq];
    for (int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < nseq; r += stride) {
        const TaskPlan p = plan[r];
        if (p.slen == 0) continue;
        const int64_t qoff = cum[r];
        const int L = (int)(cum[r + 1] - qoff);
        const int qend = p.qbeg + p.slen;
        if (cnt[0 * nseq + r]) {                                        // left task
            const int64_t ti = offs[0 * n1 + r], qo = offs[1 * n1 + r], ro = offs[2 * n1 + r];
            const int ql = p.qbeg, rl = (int)(p.rbeg - p.r0);
            for (int t = lane; t < ql; t += 64) qerbuf[qo + t] = enc[qoff + p.qbeg - 1 - t];
            for (int t = lane; t < rl; t += 64) refbuf[ro + t] = ref0123[p.rbeg - 1 - t];
            if (lane == 0) {
                bwams_seqpair_t s;
                s.idr = (int32_t)ro; s.idq = (int32_t)qo; s.id = (int32_t)ti;
                s.len1 = rl; s.len2 = ql; s.h0 = p.slen * a; s.seqid = (int32_t)r; s.regid = 0;
                s.score = s.tle = s.gtle = s.qle = s.gscore = s.max_off = 0;
                pairs[ti] = s;
            }
        }
        if (cnt[3 * nseq + r]) {                                        // right task
            const int64_t ti = tot_nl + offs[3 * n1 + r], qo = tot_lq + offs[4 * n1 + r], ro = tot_lr + offs[5 * n1 + r];
            const int ql = L - qend, rl = (int)(p.r1 - p.rbeg - p.slen);
            for (int t = lane; t < ql; t += 64) qerbuf[qo + t] = enc[qoff + qend + t];
            for (int t = lane; t < rl; t += 64) refbuf[ro + t] = ref0123[p.rbeg + p.slen + t];
            if (lane == 0) {
                bwams_seqpair_t s;
                s.idr = (int32_t)ro; s.idq = (int32_t)qo; s.id = (int32_t)ti;
                s.len1 = rl; s.len2 = ql; s.h0 = p.slen * a; s.seqid = (int32_t)r; s.regid = 1;
                s.score = s.tle = s.gtle = s.qle = s.gscore = s.max_off = 0;
                pairs[ti] = s;
            }
        }
    }
}

// exclusive scans of the six count rows into 6 x (nseq+1) int64 (last element = total)
__global__ void widen_kernel(const int32_t *cnt, int64_t nseq, int64_t *wide) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 6 * (nseq + 1)) return;
    const int64_t row = g / (nseq + 1), i = g - row * (nseq + 1);
    wide[g] = i < nseq ? (int64_t)cnt[row * nseq + i] : 0;
}

}  // namespace

size_t task_plan_bytes(int64_t nseq) { return (size_t)nseq * sizeof(TaskPlan); }

void launch_task_plan(const bwams_smem_t *sm, int64_t n_smem, const int64_t *sa_off, const int64_t *sa_coord,
                      const int64_t *cum, int64_t nseq, int64_t l_pac, int max_occ, int a, int o_gap, int e_gap, int w,
                      void *plan, int32_t *cnt, int64_t *wide, hipStream_t st) {
    if (nseq <= 0) return;
    plan_kernel<<<(unsigned)((nseq + 255) / 256), 256, 0, st>>>(sm, n_smem, sa_off, sa_coord, cum, nseq, l_pac, max_occ, a,
                                                                o_gap, e_gap, w, reinterpret_cast<TaskPlan *>(plan), cnt);
    const int64_t g = 6 * (nseq + 1);
    widen_kernel<<<(unsigned)((g + 255) / 256), 256, 0, st>>>(cnt, nseq, wide);
}

void launch_task_build(const void *plan, const int32_t *cnt, const int64_t *offs, const uint8_t *enc, const int64_t *cum,
                       const uint8_t *ref0123, int64_t nseq, int a, bwams_seqpair_t *pairs, uint8_t *refbuf,
                       uint8_t *qerbuf, int cu_count, hipStream_t st) {
    if (nseq <= 0) return;
    int64_t blocks = (nseq + 3) / 4;
    if (blocks > (int64_t)cu_count * 16) blocks = (int64_t)cu_count * 16;
    build_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const TaskPlan *>(plan), cnt, offs, enc, cum, ref0123,
                                                   nseq, a, pairs, refbuf, qerbuf);
}

}  // namespace bwams
