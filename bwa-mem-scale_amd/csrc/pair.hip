// pair.hip — the paired-end tail of worker_sam up to the pairing decision, on the device
// (/root/reference/src/bwamem_pair.cpp): mate rescue (mem_sam_pe_batch_pre :838-870 -> mem_matesw_batch_pre
// :1193-1355, the batched ksw_align2 of mem_sam_pe_batch :880-979, mem_sam_pe_batch_post :981-1042 ->
// mem_matesw_batch_post :1497-1601), mem_mark_primary_se (bwamem.cpp:1905-1980) of both ends and mem_pair (:366-427).
//
// Structure.  The anchors of an end are a snapshot of its regions taken before any rescue and the region list they
// rescue INTO is the mate's, so the two ends of a pair are independent: the unit of sequential work is "one read's
// region list, visited by its mate's anchors in order".
//   plan    lane per anchor: which of the four orientations still lack a consistent hit in the mate's list as it
//           is now, and their rescue windows (bns_fetch_seq's clip to the anchor's sequence and strand)
//   build   wave per window: the SeqPair and its two byte strings (the mate reverse-complemented for FR / RF)
//   align   ksw_local.hip (both passes of ksw_align2)
//   post    lane per read: the reference's sequential procedure — re-test the orientations against the list as it
//           has become, insert each rescued region by score, mem_sort_dedup_patch (bns = pac = query = 0: no
//           patching) after every alignment consumed; then mem_mark_primary_se
//   gather  regions in final order;  pair: lane per pair, mem_pair
// The reference's _post can want an alignment that _pre did not batch (a region that made an orientation
// "consistent" was removed by a later de-duplication); it then calls ksw_align2 on the spot (index == -1).  Here
// such a read is flagged, and a second pass plans every non-failed orientation for it and redoes it from scratch.
//
// The region lists are index arrays (`ord`) over a per-read pool: originals first, rescued regions appended.
#include "common.h"
#include "chain_kernels.h"
#include "region_sort.h"
#include "wave_ops.h"

namespace bwams {
namespace {

constexpr int KSW_XBYTE = 0x10000, KSW_XSUBO = 0x40000, KSW_XSTART = 0x80000;

__device__ __forceinline__ uint64_t hash_64(uint64_t key) {          // utils.h:117-128
    key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
    key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
    return key;
}
__device__ __forceinline__ int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist) {      // bwamem_pair.cpp:57-65
    const int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
__device__ __forceinline__ int pos2rid(const DevBns &b, int64_t pos_f) {                             // bntseq.cpp:397-413
    if (pos_f >= b.l_pac) return -1;
    int left = 0, mid = 0, right = b.n_seqs;
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
__device__ __forceinline__ int is_alt(const bwams_alnreg_t &r) { return (r.n_comp_is_alt >> 30) & 3; }

// the rescue window of anchor a for orientation r4 (mem_matesw: window arithmetic + bns_fetch_seq's clip)
__device__ bool rescue_window(const PairArgs &A, const bwams_alnreg_t &a, int r4, int l_ms, int64_t *rb_out, int64_t *re_out) {
    const int64_t l_pac = A.bns.l_pac;
    const bool is_rev = (r4 >> 1) != (r4 & 1), is_larger = !(r4 >> 1);
    const int64_t low = A.pes[r4].low, high = A.pes[r4].high;
    int64_t rb, re;
    if (!is_rev) {
        rb = is_larger ? a.rb + low : a.rb - high;
        re = (is_larger ? a.rb + high : a.rb - low) + l_ms;
    } else {
        rb = (is_larger ? a.rb + low : a.rb - high) - l_ms;
        re = is_larger ? a.rb + high : a.rb - low;
    }
    if (rb < 0) rb = 0;
    if (re > l_pac << 1) re = l_pac << 1;
    if (!(rb < re)) return false;
    const int64_t mid = (rb + re) >> 1;
    const bool mrev = mid >= l_pac;
    const int rid = pos2rid(A.bns, mrev ? (l_pac << 1) - 1 - mid : mid);
    int64_t far_beg = A.bns.contigs[rid].offset, far_end = far_beg + A.bns.contigs[rid].len;
    if (mrev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
    rb = rb > far_beg ? rb : far_beg;
    re = re < far_end ? re : far_end;
    *rb_out = rb; *re_out = re;
    return a.rid == rid && re - rb >= A.opt.min_seed_len;
}

// ---- anchors ---------------------------------------------------------------------------------------------------
// lane per read: how many anchors it provides (regions scoring within pen_unpaired of its best, at most max_matesw)
__global__ void pair_count_kernel(PairArgs A, int64_t *wide) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.nseq) return;
    int na = 0;
    if (r < A.nseq && !A.no_rescue) {
        const int64_t r0 = A.reg_off[r];
        const int n = (int)(A.reg_off[r + 1] - r0);
        for (int j = 0; j < n && na < A.opt.max_matesw; ++j)
            if (A.regs[r0 + j].score >= A.regs[r0].score - A.opt.pen_unpaired) ++na;
    }
    A.na[r] = na;                                            // entry nseq too (0): pair_cap_kernel reads na[r ^ 1], which for the last read of a
                                                             // single-end chunk with an odd number of reads is that entry
    wide[r] = na;                                            // row 0: anchor slots
}
// lane per read: capacity of its pool = its regions + 4 per anchor of its mate
__global__ void pair_cap_kernel(PairArgs A, int64_t *wide) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.nseq) return;
    wide[r] = r < A.nseq ? (A.reg_off[r + 1] - A.reg_off[r]) + 4 * (int64_t)A.na[r ^ 1] : 0;
}
__global__ void pair_slots_kernel(PairArgs A) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    const int64_t r0 = A.reg_off[r], s0 = A.aoff[r];
    const int n = (int)(A.reg_off[r + 1] - r0), na = A.na[r];
    int k = 0;
    for (int j = 0; j < n && k < na; ++j)
        if (A.regs[r0 + j].score >= A.regs[r0].score - A.opt.pen_unpaired) {
            A.anchor[s0 + k] = j;
            A.slot_read[s0 + k] = (int32_t)r;
            ++k;
        }
}

// ---- plan: lane per anchor --------------------------------------------------------------------------------------
__global__ void pair_plan_kernel(PairArgs A, int64_t *wide) {          // wide: 3 rows of 4 * n_slots + 1
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N1 = 4 * A.n_slots + 1;
    if (s == 0) { wide[N1 - 1] = 0; wide[2 * N1 - 1] = 0; wide[3 * N1 - 1] = 0; }
    if (s >= A.n_slots) return;
    const int64_t r = A.slot_read[s], m = r ^ 1;
    const bwams_alnreg_t a = A.regs[A.reg_off[r] + A.anchor[s]];
    const int l_ms = (int)(A.cum[m + 1] - A.cum[m]);
    bool skip[4];
    for (int k = 0; k < 4; ++k) skip[k] = A.pes[k].failed != 0;
    bool active = !(A.pass == 0 && A.drop_plan);
    if (A.pass == 0) {                                       // the mate's list before any rescue
        const int64_t m0 = A.reg_off[m];
        const int mn = (int)(A.reg_off[m + 1] - m0);
        for (int i = 0; i < mn; ++i) {
            int64_t dist;
            const int d = infer_dir(A.bns.l_pac, a.rb, A.regs[m0 + i].rb, &dist);
            if (dist >= A.pes[d].low && dist <= A.pes[d].high) skip[d] = true;
        }
    } else active = A.full[m] != 0;                          // second pass: only the flagged reads, every orientation
    for (int k = 0; k < 4; ++k) {
        int64_t rb = 0, re = 0;
        const bool on = active && !skip[k] && rescue_window(A, a, k, l_ms, &rb, &re);
        const int64_t e = 4 * s + k;
        A.trb[e] = rb;
        A.tl1[e] = on ? (int32_t)(re - rb) : -1;
        wide[e] = on ? 1 : 0;
        wide[N1 + e] = on ? re - rb : 0;
        wide[2 * N1 + e] = on ? l_ms : 0;
    }
}

// ---- build: wave per window ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_build_kernel(PairArgs A, const int64_t *offs, bwams_seqpair_t *pairs, uint8_t *tref,
                                                         uint8_t *tqer) {
    const int lane = threadIdx.x & 63;
    const int64_t N1 = 4 * A.n_slots + 1;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t e = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); e < 4 * A.n_slots; e += stride) {
        const int l1 = A.tl1[e];
        if (lane == 0) A.task[e] = l1 >= 0 ? (int32_t)offs[e] : -1;
        if (l1 < 0) continue;
        const int64_t s = e >> 2;
        const int k = (int)(e & 3);
        const int64_t m = (int64_t)A.slot_read[s] ^ 1;
        const int l_ms = (int)(A.cum[m + 1] - A.cum[m]);
        const int64_t t = offs[e], ro = offs[N1 + e], qo = offs[2 * N1 + e], rb = A.trb[e];
        const bool is_rev = (k >> 1) != (k & 1);
        const uint8_t *ms = A.enc + A.cum[m];
        for (int i = lane; i < l1; i += 64) tref[ro + i] = A.ref[rb + i];
        for (int i = lane; i < l_ms; i += 64) {
            const uint8_t c = is_rev ? ms[l_ms - 1 - i] : ms[i];
            tqer[qo + i] = is_rev ? (c < 4 ? 3 - c : 4) : c;
        }
        if (lane == 0) {
            bwams_seqpair_t p;
            p.idr = (int32_t)ro; p.idq = (int32_t)qo; p.id = (int32_t)e; p.len1 = l1; p.len2 = l_ms;
            p.h0 = KSW_XSUBO | KSW_XSTART | (l_ms * A.opt.a < 250 ? KSW_XBYTE : 0) | (A.opt.min_seed_len * A.opt.a);
            p.seqid = (int32_t)m; p.regid = (int32_t)t;
            p.score = p.tle = p.gtle = p.qle = p.gscore = p.max_off = -1;
            pairs[t] = p;
        }
    }
}

// ---- post: lane per read ------------------------------------------------------------------------------------------
// sort_alnreg_re (by_score = 0) / sort_alnreg_score (1) on the list ord[0, n) over pool: ksort.h's introsort
__device__ void list_sort(const bwams_alnreg_t *pool, int32_t *ord, int n, SortRec *srt, int by_score) {
    if (n < 2) return;
    for (int i = 0; i < n; ++i) {
        const bwams_alnreg_t *p = &pool[ord[i]];
        SortRec x; x.idx = ord[i]; x.pad_ = 0;
        if (by_score) { x.k = p->rb; x.s = p->score; x.q = p->qb; } else { x.k = p->re; x.s = 0; x.q = 0; }
        srt[i] = x;
    }
    sort_records(srt, n, by_score);
    for (int i = 0; i < n; ++i) ord[i] = srt[i].idx;
}
// the pairwise redundancy pass of mem_(sort_)dedup_patch over the list as it stands, then the compaction; without a
// query mem_patch_reg returns 0 (bwamem.cpp:206), so nothing is merged
__device__ int list_pairwise(const PairArgs &A, bwams_alnreg_t *pool, int32_t *ord, int n) {
    for (int i = 0; i < n; ++i) pool[ord[i]].n_comp_is_alt = (pool[ord[i]].n_comp_is_alt & ~0x3fffffff) | 1;
    for (int i = 1; i < n; ++i) {
        bwams_alnreg_t *p = &pool[ord[i]];
        const bwams_alnreg_t *pr = &pool[ord[i - 1]];
        if (p->rid != pr->rid || p->rb >= pr->re + A.opt.max_chain_gap) continue;
        for (int j = i - 1; j >= 0; --j) {
            bwams_alnreg_t *q = &pool[ord[j]];
            if (!(p->rid == q->rid && p->rb < q->re + A.opt.max_chain_gap)) break;
            if (q->qe == q->qb) continue;
            const int64_t or_ = q->re - p->rb;
            const int64_t oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if ((float)or_ > A.opt.mask_level_redun * (float)mr && (float)oq > A.opt.mask_level_redun * (float)mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            }
        }
    }
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (pool[ord[i]].qe > pool[ord[i]].qb) ord[m++] = ord[i];
    return m;
}
// mem_sort_dedup_patch(opt, 0, 0, 0, n, a) on the list
__device__ int list_sort_dedup(const PairArgs &A, bwams_alnreg_t *pool, int32_t *ord, int n, SortRec *srt) {
    if (n <= 1) return n;
    list_sort(pool, ord, n, srt, 0);
    n = list_pairwise(A, pool, ord, n);
    list_sort(pool, ord, n, srt, 1);
    for (int i = 1; i < n; ++i) {
        bwams_alnreg_t *p = &pool[ord[i]];
        const bwams_alnreg_t *pr = &pool[ord[i - 1]];
        if (p->score == pr->score && p->rb == pr->rb && p->qb == pr->qb) p->qe = p->qb;
    }
    int m = n ? 1 : 0;
    for (int i = 1; i < n; ++i)
        if (pool[ord[i]].qe > pool[ord[i]].qb) ord[m++] = ord[i];
    return m;
}
// mem_dedup_patch(opt, 0, 0, 0, n, a) (bwamem.cpp:262-312): no sorting, no identical-hit pass
__device__ int list_dedup(const PairArgs &A, bwams_alnreg_t *pool, int32_t *ord, int n) {
    if (n <= 1) return n;
    return list_pairwise(A, pool, ord, n);
}

// mem_mark_primary_se_core on the list
__device__ void list_mark_core(const PairArgs &A, bwams_alnreg_t *pool, const int32_t *ord, int n, int32_t *z) {
    int tmp = A.opt.a + A.opt.b;
    tmp = A.opt.o_del + A.opt.e_del > tmp ? A.opt.o_del + A.opt.e_del : tmp;
    tmp = A.opt.o_ins + A.opt.e_ins > tmp ? A.opt.o_ins + A.opt.e_ins : tmp;
    int zn = 0;
    z[zn++] = 0;
    for (int i = 1; i < n; ++i) {
        bwams_alnreg_t *ai = &pool[ord[i]];
        int k;
        for (k = 0; k < zn; ++k) {
            bwams_alnreg_t *aj = &pool[ord[z[k]]];
            const int b_max = aj->qb > ai->qb ? aj->qb : ai->qb;
            const int e_min = aj->qe < ai->qe ? aj->qe : ai->qe;
            if (e_min > b_max) {
                const int min_l = ai->qe - ai->qb < aj->qe - aj->qb ? ai->qe - ai->qb : aj->qe - aj->qb;
                if ((float)(e_min - b_max) >= (float)min_l * A.opt.mask_level) {
                    if (aj->sub == 0) aj->sub = ai->score;
                    if (aj->score - ai->score <= tmp && (is_alt(*aj) || !is_alt(*ai))) ++aj->sub_n;
                    break;
                }
            }
        }
        if (k == zn) z[zn++] = i;
        else ai->secondary = z[k];
    }
}
// mem_mark_primary_se on the list; returns n_pri.  The two sorts compare (score, is_alt, hash) keys that cannot tie
// (hash_64 is a bijection on id + i), so their result does not depend on the sorting algorithm.
__device__ int list_mark_primary(const PairArgs &A, bwams_alnreg_t *pool, int32_t *ord, int n, int64_t id, SortRec *srt, int32_t *z) {
    if (n == 0) return 0;
    int n_pri = 0;
    for (int i = 0; i < n; ++i) {
        bwams_alnreg_t *p = &pool[ord[i]];
        p->sub = p->alt_sc = 0; p->secondary = p->secondary_all = -1; p->hash = hash_64((uint64_t)(id + i));
        if (!is_alt(*p)) ++n_pri;
    }
    for (int i = 0; i < n; ++i) {
        const bwams_alnreg_t *p = &pool[ord[i]];
        SortRec x; x.k = (int64_t)p->hash; x.s = p->score; x.q = is_alt(*p); x.idx = ord[i]; x.pad_ = 0;
        srt[i] = x;
    }
    sort_records(srt, n, 2);
    for (int i = 0; i < n; ++i) ord[i] = srt[i].idx;
    list_mark_core(A, pool, ord, n, z);
    for (int i = 0; i < n; ++i) {
        bwams_alnreg_t *p = &pool[ord[i]];
        p->secondary_all = i;
        if (!is_alt(*p) && p->secondary >= 0 && is_alt(pool[ord[p->secondary]])) p->alt_sc = pool[ord[p->secondary]].score;
    }
    if (n_pri < n) {
        if (n_pri > 0) {
            for (int i = 0; i < n; ++i) {
                const bwams_alnreg_t *p = &pool[ord[i]];
                SortRec x; x.k = (int64_t)p->hash; x.s = p->score; x.q = is_alt(*p); x.idx = ord[i]; x.pad_ = 0;
                srt[i] = x;
            }
            sort_records(srt, n, 3);
            for (int i = 0; i < n; ++i) ord[i] = srt[i].idx;
        }
        for (int i = 0; i < n; ++i) z[pool[ord[i]].secondary_all] = i;
        for (int i = 0; i < n; ++i) {
            bwams_alnreg_t *p = &pool[ord[i]];
            if (p->secondary >= 0) {
                p->secondary_all = z[p->secondary];
                if (is_alt(*p)) p->secondary = 0x7fffffff;
            } else p->secondary_all = -1;
        }
        if (n_pri > 0) {
            for (int i = 0; i < n_pri; ++i) { pool[ord[i]].sub = 0; pool[ord[i]].secondary = -1; }
            list_mark_core(A, pool, ord, n_pri, z);
        }
    } else {
        for (int i = 0; i < n; ++i) pool[ord[i]].secondary_all = pool[ord[i]].secondary;
    }
    return n_pri;
}

// the whole rescue of read m's list by one lane, lists and sort records in HBM
__device__ void post_read_seq(const PairArgs &A, int64_t m) {
    const int64_t l_pac = A.bns.l_pac;
    const int64_t o0 = A.ooff[m];
    bwams_alnreg_t *pool = A.pool + o0;
    int32_t *ord = A.ord + o0;
    SortRec *srt = reinterpret_cast<SortRec *>(A.srt) + o0;
    const int64_t m0 = A.reg_off[m];
    int n = (int)(A.reg_off[m + 1] - m0), n_pool = n;
    for (int i = 0; i < n; ++i) { bwams_alnreg_t x = A.regs[m0 + i]; x.flg = 0; pool[i] = x; ord[i] = i; }
    const int64_t r = m ^ 1;                                 // the mate provides the anchors
    const int l_ms = (int)(A.cum[m + 1] - A.cum[m]);
    const int na = A.na[r];
    int n_sw = 0, last_cnt = 0;
    bool need_full = false;
    // useErt (mem_sam_pe_batch_post, bwamem_pair.cpp:1017-1041): the list is sorted by end first and kept so by
    // mem_matesw_batch_post_ert; afterwards mem_sort_dedup_patch if the LAST anchor consumed an alignment, else a score sort
    if (A.use_ert && !A.no_rescue) list_sort(pool, ord, n, srt, 0);
    for (int j = 0; j < na && !need_full; ++j) {
        const int64_t s = A.aoff[r] + j;
        const bwams_alnreg_t a = A.regs[A.reg_off[r] + A.anchor[s]];
        bool skip[4];
        for (int k = 0; k < 4; ++k) skip[k] = A.pes[k].failed != 0;
        for (int i = 0; i < n; ++i) {
            int64_t dist;
            const int d = infer_dir(l_pac, a.rb, pool[ord[i]].rb, &dist);
            if (dist >= A.pes[d].low && dist <= A.pes[d].high) skip[d] = true;
        }
        last_cnt = 0;
        if (skip[0] && skip[1] && skip[2] && skip[3]) continue;
        int cnt = 0;
        for (int k = 0; k < 4; ++k) {
            if (skip[k]) continue;
            int64_t rb, re;
            if (rescue_window(A, a, k, l_ms, &rb, &re)) {
                const int t = A.task[4 * s + k];
                if (t < 0) { need_full = true; break; }      // the reference aligns on the spot: second pass
                const int32_t *al = A.aln + (int64_t)t * 7;  // score, te, qe, score2, te2, tb, qb
                const int score = al[0], te = al[1], qe = al[2], score2 = al[3], tb = al[5], qb = al[6];
                if (score >= A.opt.min_seed_len && qb >= 0) {
                    const bool is_rev = (k >> 1) != (k & 1);
                    bwams_alnreg_t b;
                    memset(&b, 0, sizeof b);
                    b.rid = a.rid;
                    b.n_comp_is_alt = (int32_t)((uint32_t)is_alt(a) << 30);
                    b.qb = is_rev ? l_ms - (qe + 1) : qb;
                    b.qe = is_rev ? l_ms - qb : qe + 1;
                    b.rb = is_rev ? (l_pac << 1) - (rb + te + 1) : rb + tb;
                    b.re = is_rev ? (l_pac << 1) - (rb + tb) : rb + te + 1;
                    b.score = score;
                    b.csub = score2;
                    b.secondary = -1;
                    b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
                    pool[n_pool] = b;
                    int i;
                    if (!A.use_ert) {
                        for (i = 0; i < n; ++i)
                            if (pool[ord[i]].score < b.score) break;
                        for (int q = n; q > i; --q) ord[q] = ord[q - 1];
                        ord[i] = n_pool;
                    } else {                                  // mem_matesw_batch_post_ert: by end position
                        bool resort = false;
                        for (i = 0; i < n; ++i) {
                            if (pool[ord[i]].re == b.re) { resort = true; break; }
                            if (pool[ord[i]].re > b.re) break;
                        }
                        if (resort) {                         // "let the scores decide", then sort by end again
                            list_sort(pool, ord, n, srt, 1);
                            for (i = 0; i < n; ++i)
                                if (pool[ord[i]].score < b.score) break;
                        }
                        for (int q = n; q > i; --q) ord[q] = ord[q - 1];
                        ord[i] = n_pool;
                        if (resort) list_sort(pool, ord, n + 1, srt, 0);
                    }
                    ++n; ++n_pool;
                }
                ++cnt;
            }
            if (cnt) n = A.use_ert ? list_dedup(A, pool, ord, n) : list_sort_dedup(A, pool, ord, n, srt);
        }
        n_sw += cnt;
        last_cnt = cnt;
    }
    if (A.use_ert && !A.no_rescue && !need_full) {
        if (last_cnt) n = list_sort_dedup(A, pool, ord, n, srt);
        else list_sort(pool, ord, n, srt, 1);
    }
    if (need_full) {
        if (A.pass == 0) { A.full[m] = 1; atomicAdd(&A.ctr->pair_full, 1ull); }
        else atomicAdd(&A.ctr->pair_fail, 1ull);             // cannot happen: every valid window was planned
        A.n_fin[m] = 0; A.n_pri[m] = 0; A.n_sw[m] = 0;
        return;
    }
    A.n_fin[m] = n;
    A.n_sw[m] = n_sw;
}

constexpr int kPostLight = 16;       // pool slots (regions + room for rescued ones) a single lane handles
constexpr int kPostLds = 1024;       // pool slots a wavefront keeps in LDS (a power of two: the bitonic sort pads to one)
constexpr int kPostRankMax = 96;     // longer lists are sorted by the bitonic network, shorter ones by rank

__global__ __launch_bounds__(64) void pair_post_kernel(PairArgs A) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= A.nseq) return;
    if (A.pass == 1 && !A.full[m]) return;                   // second pass: only the flagged reads
    // (the ERT variant of the procedure exists in its one-lane form only)
    if (!A.use_ert && A.ooff[m + 1] - A.ooff[m] > kPostLight) { A.heavy[atomicAdd(&A.ctr->pair_heavy, 1ull)] = (int32_t)m; return; }
    post_read_seq(A, m);
}

// Long lists: a wavefront per read.  What the procedure reads of a region (rid, rb, re, qb, qe, score) lives in LDS,
// indexed by pool slot; the list is an LDS index array.  The loops over the list run across the lanes (orientation
// tests by ballot, insertion point and shift, compactions, the identical-hit pass); the two sorts of
// mem_sort_dedup_patch are rank sorts — every lane counts, from LDS, the records that sort before its own — which
// give ksort.h's order whenever no two keys are equal; if a key repeats (identical hits) lane 0 runs the
// operation-exact introsort instead.  Only the pairwise redundancy pass is sequential (lane 0, LDS).  The pool in
// HBM is written once per region (the copy, a rescued region, the final n_comp); lanes exchange data through LDS only.
__global__ __launch_bounds__(64) void pair_post_wave_kernel(PairArgs A) {
    __shared__ int64_t l_rb[kPostLds], l_re[kPostLds], l_k64[kPostLds];
    __shared__ int32_t l_qb[kPostLds], l_qe[kPostLds], l_sc[kPostLds], l_rid[kPostLds], l_ncia[kPostLds];
    __shared__ int32_t l_ord[kPostLds], l_tmp[kPostLds], l_ks[kPostLds], l_kq[kPostLds];
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t n_heavy = (int64_t)A.ctr->pair_heavy;
    const int64_t l_pac = A.bns.l_pac;
    const int gap = A.opt.max_chain_gap;
    for (;;) {
        const int64_t tk = (int64_t)wave_ticket(&A.ctr->pair_ticket, 1ull);
        if (tk >= n_heavy) break;
        const int64_t m = A.heavy[tk];
        const int64_t o0 = A.ooff[m];
        __syncthreads();
        if (A.ooff[m + 1] - o0 > kPostLds) {                 // beyond the LDS arrays: one lane, through HBM
            if (lane == 0) post_read_seq(A, m);
            continue;
        }
        bwams_alnreg_t *pool = A.pool + o0;
        SortRec *srt = reinterpret_cast<SortRec *>(A.srt) + o0;
        const int64_t m0 = A.reg_off[m];
        int n = (int)(A.reg_off[m + 1] - m0), n_pool = n;
        {   // copy the read's regions into its pool (flg = 0) and their keys into LDS
            const uint4 *src = reinterpret_cast<const uint4 *>(A.regs + m0);
            uint4 *dst = reinterpret_cast<uint4 *>(pool);
            for (int i = lane; i < n * 7; i += 64) {
                uint4 v = src[i];
                if (i % 7 == 6) v.z = 0;                     // flg (bytes 104..107)
                dst[i] = v;
            }
            for (int i = lane; i < n; i += 64) {
                const bwams_alnreg_t *p = A.regs + m0 + i;
                l_rb[i] = p->rb; l_re[i] = p->re; l_qb[i] = p->qb; l_qe[i] = p->qe; l_sc[i] = p->score; l_rid[i] = p->rid;
                l_ncia[i] = p->n_comp_is_alt;
                l_ord[i] = i;
            }
        }
        __syncthreads();
        const int64_t r = m ^ 1;
        const int l_ms = (int)(A.cum[m + 1] - A.cum[m]);
        const int na = A.na[r];
        int n_sw = 0;
        bool need_full = false;

#ifdef BWAMS_PAIRDBG
        const unsigned long long tk_read0 = wall_clock64();
        unsigned long long d_sorts = 0, d_ties = 0, d_tie_t = 0, d_sort_t = 0;
#endif
        // a sort of the list by rank; by_score = 0: key re, 1: (score desc, rb, qb)
        auto sort_list = [&](int by_score) {
#ifdef BWAMS_PAIRDBG
            const unsigned long long tk_s0 = wall_clock64();
#endif
            for (int i = lane; i < n; i += 64) {
                const int sl = l_ord[i];
                l_k64[i] = by_score ? l_rb[sl] : l_re[sl];
                l_ks[i] = by_score ? l_sc[sl] : 0;
                l_kq[i] = by_score ? l_qb[sl] : 0;
            }
            __syncthreads();
            bool tie = false;
            if (n > kPostRankMax) {
                // a bitonic network over the next power of two (pads sort behind everything), keys and the slot index moving
                // together through LDS: n log^2 n / 128 compare-exchanges per lane instead of n^2 / 64 comparisons.  One
                // read of the bench's paired-end chunk reaches 1000 regions and is rescued two dozen times, two sorts
                // each: 17.6 ms of an 18.4 ms launch with the rank sort (profiles/r03_notes.md 91).
                int P = 128;
                while (P < n) P <<= 1;
                for (int i = lane; i < P; i += 64) {
                    if (i < n) l_tmp[i] = l_ord[i];
                    else { l_k64[i] = INT64_MAX; l_ks[i] = INT32_MIN; l_kq[i] = INT32_MAX; l_tmp[i] = -1; }
                }
                __syncthreads();
                for (int k = 2; k <= P; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int t = lane; t < (P >> 1); t += 64) {
                            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                            const int64_t ka = l_k64[i], kb = l_k64[l];
                            const int sa = l_ks[i], sb = l_ks[l], qa = l_kq[i], qb = l_kq[l];
                            const bool b_lt_a = sb > sa || (sb == sa && (kb < ka || (kb == ka && qb < qa)));
                            const bool a_lt_b = sa > sb || (sa == sb && (ka < kb || (ka == kb && qa < qb)));
                            const bool up = (i & k) == 0;
                            if (up ? b_lt_a : a_lt_b) {
                                l_k64[i] = kb; l_k64[l] = ka; l_ks[i] = sb; l_ks[l] = sa; l_kq[i] = qb; l_kq[l] = qa;
                                const int ta = l_tmp[i]; l_tmp[i] = l_tmp[l]; l_tmp[l] = ta;
                            }
                        }
                        __syncthreads();
                    }
                }
                for (int ib = 0; ib < n; ib += 64) {
                    const int i = ib + lane;
                    const bool eq = i >= 1 && i < n && l_k64[i] == l_k64[i - 1] && l_ks[i] == l_ks[i - 1] && l_kq[i] == l_kq[i - 1];
                    tie = tie || (__ballot(eq) != 0);
                }
                if (tie) {                                   // the fallback below reads the keys in the list's order
                    __syncthreads();
                    for (int i = lane; i < n; i += 64) {
                        const int sl = l_ord[i];
                        l_k64[i] = by_score ? l_rb[sl] : l_re[sl];
                        l_ks[i] = by_score ? l_sc[sl] : 0;
                        l_kq[i] = by_score ? l_qb[sl] : 0;
                    }
                }
            } else
            for (int ib = 0; ib < n; ib += 64) {
                const int i = ib + lane;
                int rank = 0, eq = 0;
                if (i < n) {
                    const int64_t k = l_k64[i];
                    const int ks = l_ks[i], kq = l_kq[i];
                    for (int j = 0; j < n; ++j) {
                        const int64_t kj = l_k64[j];
                        const int sj = l_ks[j], qj = l_kq[j];
                        const bool lt = sj > ks || (sj == ks && (kj < k || (kj == k && qj < kq)));
                        rank += lt ? 1 : 0;
                        eq += (sj == ks && kj == k && qj == kq) ? 1 : 0;
                    }
                    l_tmp[rank] = l_ord[i];                  // ranks collide only when keys repeat (then the fallback runs)
                }
                tie = tie || (__ballot(i < n && eq > 1) != 0);
            }
            __syncthreads();
            if (!tie) {
                for (int i = lane; i < n; i += 64) l_ord[i] = l_tmp[i];
            } else {                                         // equal keys: ksort.h's introsort decides their order
                if (lane == 0) {
                    for (int i = 0; i < n; ++i) {
                        SortRec x; x.k = l_k64[i]; x.s = l_ks[i]; x.q = l_kq[i]; x.idx = l_ord[i]; x.pad_ = 0;
                        srt[i] = x;
                    }
                    sort_records(srt, n, by_score);
                    for (int i = 0; i < n; ++i) l_ord[i] = srt[i].idx;
                }
            }
            __syncthreads();
#ifdef BWAMS_PAIRDBG
            { const unsigned long long dt = wall_clock64() - tk_s0; d_sorts++; d_sort_t += dt; if (tie) { d_ties++; d_tie_t += dt; } }
#endif
        };
        auto compact = [&]() {                               // keep the regions with qe > qb, in order
            int nn = 0;
            for (int ib = 0; ib < n; ib += 64) {
                const int i = ib + lane;
                int sl = 0;
                bool alive = false;
                if (i < n) { sl = l_ord[i]; alive = l_qe[sl] > l_qb[sl]; }
                const unsigned long long mk = __ballot(alive);
                if (alive) l_ord[nn + __popcll(mk & below)] = sl;          // nn + rank <= i: never ahead of a pending read
                nn += __popcll(mk);
            }
            __syncthreads();
            n = nn;
        };

        for (int j = 0; j < na && !need_full; ++j) {
            const int64_t s = A.aoff[r] + j;
            const bwams_alnreg_t *ap = A.regs + A.reg_off[r] + A.anchor[s];
            const int64_t a_rb = ap->rb;
            bool skip[4];
            for (int k = 0; k < 4; ++k) skip[k] = A.pes[k].failed != 0;
            for (int ib = 0; ib < n; ib += 64) {
                const int i = ib + lane;
                int d = -1;
                if (i < n) {
                    int64_t dist;
                    const int dd = infer_dir(l_pac, a_rb, l_rb[l_ord[i]], &dist);
                    if (dist >= A.pes[dd].low && dist <= A.pes[dd].high) d = dd;
                }
                for (int k = 0; k < 4; ++k) skip[k] = skip[k] || (__ballot(d == k) != 0);
            }
            if (skip[0] && skip[1] && skip[2] && skip[3]) continue;
            const bwams_alnreg_t a = *ap;
            int cnt = 0;
            for (int k = 0; k < 4; ++k) {
                if (skip[k]) continue;
                int64_t rb, re;
                if (rescue_window(A, a, k, l_ms, &rb, &re)) {
                    const int t = A.task[4 * s + k];
                    if (t < 0) { need_full = true; break; }
                    const int32_t *al = A.aln + (int64_t)t * 7;
                    const int score = al[0], te = al[1], qe = al[2], score2 = al[3], tb = al[5], qb = al[6];
                    if (score >= A.opt.min_seed_len && qb >= 0) {
                        const bool is_rev = (k >> 1) != (k & 1);
                        const int sl = n_pool;
                        const int b_qb = is_rev ? l_ms - (qe + 1) : qb, b_qe = is_rev ? l_ms - qb : qe + 1;
                        const int64_t b_rb = is_rev ? (l_pac << 1) - (rb + te + 1) : rb + tb;
                        const int64_t b_re = is_rev ? (l_pac << 1) - (rb + tb) : rb + te + 1;
                        if (lane == 0) {
                            bwams_alnreg_t b;
                            memset(&b, 0, sizeof b);
                            b.rid = a.rid;
                            b.n_comp_is_alt = (int32_t)((uint32_t)is_alt(a) << 30);
                            b.qb = b_qb; b.qe = b_qe; b.rb = b_rb; b.re = b_re;
                            b.score = score;
                            b.csub = score2;
                            b.secondary = -1;
                            b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
                            pool[sl] = b;
                            l_rb[sl] = b_rb; l_re[sl] = b_re; l_qb[sl] = b_qb; l_qe[sl] = b_qe; l_sc[sl] = score; l_rid[sl] = a.rid;
                            l_ncia[sl] = b.n_comp_is_alt;
                        }
                        // insertion point: the first region scoring less than b
                        int at = n;
                        for (int ib = 0; ib < n && at == n; ib += 64) {
                            const int i = ib + lane;
                            const unsigned long long mk = __ballot(i < n && l_sc[l_ord[i]] < score);
                            if (mk) at = ib + __ffsll((long long)mk) - 1;
                        }
                        for (int ib = ((n - at) / 64) * 64; ib >= 0; ib -= 64) {     // shift up, highest chunk first
                            const int q = at + 1 + ib + lane;                        // q in (at, n]
                            int v = 0;
                            if (q <= n) v = l_ord[q - 1];
                            __syncthreads();
                            if (q <= n) l_ord[q] = v;
                            __syncthreads();
                        }
                        if (lane == 0) l_ord[at] = sl;
                        ++n; ++n_pool;
                        __syncthreads();
                    }
                    ++cnt;
                }
                if (cnt && n > 1) {                          // mem_sort_dedup_patch(opt, 0, 0, 0, n, a)
                    sort_list(0);
                    for (int i = lane; i < n; i += 64) l_ncia[l_ord[i]] = (l_ncia[l_ord[i]] & ~0x3fffffff) | 1;
                    if (lane == 0) {
                        for (int i = 1; i < n; ++i) {
                            const int p = l_ord[i], pr = l_ord[i - 1];
                            if (l_rid[p] != l_rid[pr] || l_rb[p] >= l_re[pr] + gap) continue;
                            for (int jj = i - 1; jj >= 0; --jj) {
                                const int q = l_ord[jj];
                                if (!(l_rid[p] == l_rid[q] && l_rb[p] < l_re[q] + gap)) break;
                                if (l_qe[q] == l_qb[q]) continue;
                                const int64_t or_ = l_re[q] - l_rb[p];
                                const int64_t oq = l_qb[q] < l_qb[p] ? l_qe[q] - l_qb[p] : l_qe[p] - l_qb[q];
                                const int64_t mr = l_re[q] - l_rb[q] < l_re[p] - l_rb[p] ? l_re[q] - l_rb[q] : l_re[p] - l_rb[p];
                                const int64_t mq = l_qe[q] - l_qb[q] < l_qe[p] - l_qb[p] ? l_qe[q] - l_qb[q] : l_qe[p] - l_qb[p];
                                if ((float)or_ > A.opt.mask_level_redun * (float)mr && (float)oq > A.opt.mask_level_redun * (float)mq) {
                                    if (l_sc[p] < l_sc[q]) { l_qe[p] = l_qb[p]; break; }
                                    else l_qe[q] = l_qb[q];
                                }
                            }
                        }
                    }
                    __syncthreads();
                    compact();
                    sort_list(1);
                    for (int ib = 0; ib < n; ib += 64) {                             // identical hits
                        const int i = ib + lane;
                        if (i >= 1 && i < n) {
                            const int p = l_ord[i], pr = l_ord[i - 1];
                            if (l_sc[p] == l_sc[pr] && l_rb[p] == l_rb[pr] && l_qb[p] == l_qb[pr]) l_qe[p] = l_qb[p];
                        }
                    }
                    __syncthreads();
                    compact();
                }
            }
            n_sw += cnt;
        }
        if (need_full) {
            if (lane == 0) {
                if (A.pass == 0) { A.full[m] = 1; atomicAdd(&A.ctr->pair_full, 1ull); }
                else atomicAdd(&A.ctr->pair_fail, 1ull);
                A.n_fin[m] = 0; A.n_pri[m] = 0; A.n_sw[m] = 0;
            }
            continue;
        }
        for (int i = lane; i < n; i += 64) {
            const int sl = l_ord[i];
            A.ord[o0 + i] = sl;
            pool[sl].n_comp_is_alt = l_ncia[sl];
        }
        if (lane == 0) { A.n_fin[m] = n; A.n_sw[m] = n_sw; }
#ifdef BWAMS_PAIRDBG
        if (lane == 0) {
            const unsigned long long dt = wall_clock64() - tk_read0;
            atomicAdd(&A.ctr->dbg[20], 1ull); atomicAdd(&A.ctr->dbg[21], d_sorts); atomicAdd(&A.ctr->dbg[22], d_ties); atomicAdd(&A.ctr->dbg[23], d_tie_t);
            atomicAdd(&A.ctr->dbg[24], dt); atomicMax(&A.ctr->dbg[25], dt); atomicAdd(&A.ctr->dbg[26], d_sort_t); atomicAdd(&A.ctr->dbg[27], (unsigned long long)n);
            atomicAdd(&A.ctr->dbg[28], (unsigned long long)na); atomicAdd(&A.ctr->dbg[29], (unsigned long long)n_sw);
        }
#endif
    }
}

// ---- mem_mark_primary_se: lane per read for short lists, wavefront per read for long ones -------------------------
constexpr int kMarkLight = 24;       // regions a single lane handles
constexpr int kMarkLdsMax = 2048, kMarkLdsSmall = 256;       // regions a wavefront keeps in LDS (68 B each): the two instances

__global__ __launch_bounds__(64) void pair_mark_kernel(PairArgs A) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= A.nseq) return;
    const int n = A.n_fin[m];
    if (n > kMarkLight) { A.heavy[atomicAdd(&A.ctr->pair_heavy, 1ull)] = (int32_t)m; return; }
    const int64_t o0 = A.ooff[m];
    const int64_t id = A.single_end ? A.id_base + m : (((A.id_base + (m >> 1)) << 1) | (m & 1));      // mem_reg2sam passes n_processed + i, mem_sam_pe id << 1 | end
    A.n_pri[m] = list_mark_primary(A, A.pool + o0, A.ord + o0, n, id, reinterpret_cast<SortRec *>(A.srt) + o0, A.zbuf + o0);
}

// The list is long: its two sorts dominate when one lane runs them through HBM.  Both compare keys that cannot tie,
// so each element's final position is the number of elements that sort before it — counted by the 64 lanes from LDS.
// Only mem_mark_primary_se_core's scan stays sequential (lane 0, over LDS copies of qb / qe / score / is_alt); every
// field is written back to the pool by the lane that owns the element, so lanes never exchange data through HBM.
// Two instances (LO < regions <= CAP: (24, 256] with 17 KB of LDS per wave, (256, 2048] with 139 KB) share the list of
// long lists, each with its own ticket counter; the largest also takes what is beyond its LDS (one lane, through HBM).
// One 1024-entry instance kept three waves per CU busy with mostly short lists and sent the longest (> 1024 regions) to
// a single lane: 11.3 ms for a chunk.
template <int CAP, int LO>
__global__ __launch_bounds__(64) void pair_mark_wave_kernel(PairArgs A, unsigned long long *ticket) {
    extern __shared__ __align__(16) unsigned char lds_mark[];
    uint64_t *l_hash = reinterpret_cast<uint64_t *>(lds_mark);
    int32_t *l_slot = reinterpret_cast<int32_t *>(l_hash + CAP);
    int32_t *l_sc = l_slot + CAP, *l_alt = l_sc + CAP, *l_qb = l_alt + CAP, *l_qe = l_qb + CAP;
    int32_t *l_at = l_qe + CAP;                // sorted position -> original index
    int32_t *l_sub = l_at + CAP, *l_subn = l_sub + CAP, *l_sec = l_subn + CAP, *l_secall = l_sec + CAP, *l_altsc = l_secall + CAP;   // by sorted position
    int32_t *l_pos2 = l_altsc + CAP, *l_at2 = l_pos2 + CAP, *l_z = l_at2 + CAP;
    constexpr int kMarkLds = CAP;
    const int lane = threadIdx.x;
    const int64_t n_heavy = (int64_t)A.ctr->pair_heavy;
    int tmp = A.opt.a + A.opt.b;
    tmp = A.opt.o_del + A.opt.e_del > tmp ? A.opt.o_del + A.opt.e_del : tmp;
    tmp = A.opt.o_ins + A.opt.e_ins > tmp ? A.opt.o_ins + A.opt.e_ins : tmp;
    for (;;) {
        const int64_t t = (int64_t)wave_ticket(ticket, 1ull);
        if (t >= n_heavy) break;
        const int64_t m = A.heavy[t];
        const int n = A.n_fin[m];
        if (n <= LO || (n > CAP && CAP != kMarkLdsMax)) continue;         // the other instance's list
        const int64_t o0 = A.ooff[m];
        bwams_alnreg_t *pool = A.pool + o0;
        int32_t *ord = A.ord + o0;
        const int64_t id = A.single_end ? A.id_base + m : (((A.id_base + (m >> 1)) << 1) | (m & 1));      // mem_reg2sam passes n_processed + i, mem_sam_pe id << 1 | end
        __syncthreads();
        if (n > kMarkLds) {                                  // beyond the LDS arrays: one lane, through HBM
            if (lane == 0) A.n_pri[m] = list_mark_primary(A, pool, ord, n, id, reinterpret_cast<SortRec *>(A.srt) + o0, A.zbuf + o0);
            continue;
        }
        int n_pri = 0;
        for (int ib = 0; ib < n; ib += 64) {
            const int i = ib + lane;
            bool pri = false;
            if (i < n) {
                const int slot = ord[i];
                const bwams_alnreg_t *p = &pool[slot];
                l_slot[i] = slot; l_sc[i] = p->score; l_alt[i] = is_alt(*p); l_qb[i] = p->qb; l_qe[i] = p->qe;
                l_hash[i] = hash_64((uint64_t)(id + i));
                pri = !is_alt(*p);
            }
            n_pri += __popcll(__ballot(pri));
        }
        __syncthreads();
        // order 1: score descending, is_alt ascending, hash ascending (the hashes of a read's regions are distinct: no ties)
        if (n > kPostRankMax) {
            // long lists: a bitonic network (n log^2 n / 128 compare-exchanges per lane instead of n^2 / 64 comparisons) over two
            // 64-bit keys and the original index, in the arrays the marking fills only afterwards
            uint64_t *k1 = reinterpret_cast<uint64_t *>(l_sub), *k2 = reinterpret_cast<uint64_t *>(l_sec);      // l_sub | l_subn, l_sec | l_secall
            int32_t *pay = l_altsc;
            int P = 128;
            while (P < n) P <<= 1;
            for (int i = lane; i < P; i += 64) {
                if (i < n) { k1[i] = ((uint64_t)(uint32_t)(0x7fffffff - l_sc[i]) << 1) | (uint64_t)(l_alt[i] ? 1 : 0); k2[i] = l_hash[i]; pay[i] = i; }
                else { k1[i] = ~0ull; k2[i] = ~0ull; pay[i] = -1; }
            }
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = lane; t < (P >> 1); t += 64) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                        const uint64_t a1 = k1[i], b1 = k1[l], a2 = k2[i], b2 = k2[l];
                        const bool b_lt_a = b1 < a1 || (b1 == a1 && b2 < a2);
                        const bool a_lt_b = a1 < b1 || (a1 == b1 && a2 < b2);
                        if (((i & k) == 0) ? b_lt_a : a_lt_b) {
                            k1[i] = b1; k1[l] = a1; k2[i] = b2; k2[l] = a2;
                            const int ta = pay[i]; pay[i] = pay[l]; pay[l] = ta;
                        }
                    }
                    __syncthreads();
                }
            }
            for (int r = lane; r < n; r += 64) l_at[r] = pay[r];
            __syncthreads();
        } else {
            for (int i = lane; i < n; i += 64) {
                const int sc = l_sc[i], al = l_alt[i];
                const uint64_t h = l_hash[i];
                int r = 0;
                for (int j = 0; j < n; ++j) {
                    const int sj = l_sc[j], aj = l_alt[j];
                    r += (sj > sc || (sj == sc && (aj < al || (aj == al && l_hash[j] < h)))) ? 1 : 0;
                }
                l_at[r] = i;
            }
            __syncthreads();
        }
        for (int i = lane; i < n; i += 64) { l_sub[i] = 0; l_subn[i] = 0; l_sec[i] = -1; l_altsc[i] = 0; }
        __syncthreads();
        // mem_mark_primary_se_core over the sorted list (sequential: each element looks at the primaries found so far)
        auto core = [&](const int32_t *at, int cnt) {
            int zn = 0;
            l_z[zn++] = 0;
            for (int i = 1; i < cnt; ++i) {
                const int oi = at[i];
                int k;
                for (k = 0; k < zn; ++k) {
                    const int j = l_z[k], oj = at[j];
                    const int b_max = l_qb[oj] > l_qb[oi] ? l_qb[oj] : l_qb[oi];
                    const int e_min = l_qe[oj] < l_qe[oi] ? l_qe[oj] : l_qe[oi];
                    if (e_min > b_max) {
                        const int li = l_qe[oi] - l_qb[oi], lj = l_qe[oj] - l_qb[oj];
                        const int min_l = li < lj ? li : lj;
                        if ((float)(e_min - b_max) >= (float)min_l * A.opt.mask_level) {
                            if (l_sub[oj] == 0) l_sub[oj] = l_sc[oi];
                            if (l_sc[oj] - l_sc[oi] <= tmp && (l_alt[oj] || !l_alt[oi])) ++l_subn[oj];
                            break;
                        }
                    }
                }
                if (k == zn) l_z[zn++] = i;
                else l_sec[oi] = l_z[k];
            }
        };
        // l_sub / l_subn / l_sec / l_altsc / l_secall / l_pos2 are indexed by ORIGINAL index from here on
        if (lane == 0) core(l_at, n);
        __syncthreads();
        for (int r = lane; r < n; r += 64) {                 // rank in the first round, alt_sc
            const int oi = l_at[r];
            l_secall[oi] = r;
            const int sec = l_sec[oi];
            if (!l_alt[oi] && sec >= 0 && l_alt[l_at[sec]]) l_altsc[oi] = l_sc[l_at[sec]];
        }
        __syncthreads();
        const int32_t *fin_at = l_at;
        if (n_pri < n) {
            if (n_pri > 0) {
                // order 2 (is_alt ascending, score descending, hash ascending) = a stable partition of order 1 by is_alt
                int c_pri = 0, c_alt = n_pri;                // next free position of either half (wave-uniform)
                const unsigned long long below_m = (1ull << lane) - 1ull;
                for (int rb = 0; rb < n; rb += 64) {
                    const int r = rb + lane;
                    const bool in = r < n;
                    const int oi = in ? l_at[r] : 0;
                    const bool al = in && l_alt[oi];
                    const unsigned long long m_alt = __ballot(al), m_pri = __ballot(in && !al);
                    if (in) {
                        const int c = al ? c_alt + __popcll(m_alt & below_m) : c_pri + __popcll(m_pri & below_m);
                        l_pos2[r] = c;
                        l_at2[c] = oi;
                    }
                    c_alt += __popcll(m_alt);
                    c_pri += __popcll(m_pri);
                }
                fin_at = l_at2;
            } else {
                for (int r = lane; r < n; r += 64) l_pos2[r] = r;
            }
            __syncthreads();
            for (int i = lane; i < n; i += 64) {             // i: original index
                if (l_sec[i] >= 0) {
                    l_secall[i] = l_pos2[l_sec[i]];          // z[a[i].secondary], where z maps first-round rank -> position now
                    if (l_alt[i]) l_sec[i] = 0x7fffffff;
                } else l_secall[i] = -1;
            }
            __syncthreads();
            if (n_pri > 0) {
                for (int r = lane; r < n_pri; r += 64) { l_sub[fin_at[r]] = 0; l_sec[fin_at[r]] = -1; }
                __syncthreads();
                if (lane == 0) core(fin_at, n_pri);
                __syncthreads();
            }
        } else {
            for (int i = lane; i < n; i += 64) l_secall[i] = l_sec[i];
            __syncthreads();
        }
        for (int f = lane; f < n; f += 64) {                 // write back, final order
            const int oi = fin_at[f];
            const int slot = l_slot[oi];
            bwams_alnreg_t *p = &pool[slot];
            p->sub = l_sub[oi]; p->alt_sc = l_altsc[oi]; p->secondary = l_sec[oi]; p->secondary_all = l_secall[oi];
            p->hash = l_hash[oi]; p->sub_n += l_subn[oi];
            ord[f] = slot;
        }
        if (lane == 0) A.n_pri[m] = n_pri;
    }
}

__global__ void pair_widen_kernel(PairArgs A, int64_t *wide) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.nseq) return;
    wide[r] = r < A.nseq ? A.n_fin[r] : 0;
}
__global__ void pair_gather_kernel(PairArgs A, const int64_t *out_off, bwams_alnreg_t *out) {
    const int64_t r = blockIdx.x;
    const int64_t o0 = A.ooff[r], d0 = out_off[r];
    const int n = A.n_fin[r];
    const uint4 *src = reinterpret_cast<const uint4 *>(A.pool);
    uint4 *dst = reinterpret_cast<uint4 *>(out);
    for (int i = threadIdx.x; i < n * 7; i += blockDim.x) {       // 112 B = 7 x 16 B
        const int e = i / 7, w = i - e * 7;
        dst[(d0 + e) * 7 + w] = src[(o0 + A.ord[o0 + e]) * 7 + w];
    }
}

// ---- mem_reorder_primary5 (bwamem.cpp:2009-2031): lane per read, on the gathered list ------------------------------------
// Of the primary, non-ALT regions scoring >= T the one that starts leftmost in the read (first of equals) changes places
// with region 0; secondary / secondary_all references to either follow.  Runs between the marking and mem_pair, as in
// mem_sam_pe (bwamem_pair.cpp:1060-1063) and worker_sam (bwamem.cpp:1840).
__global__ __launch_bounds__(64) void pair_reorder5_kernel(PairArgs A, const int64_t *out_off, bwams_alnreg_t *out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    bwams_alnreg_t *a = out + out_off[r];
    const int n = (int)(out_off[r + 1] - out_off[r]), T = A.primary5_T;
    int n_pri = 0, left_st = 0x7fffffff, left_k = -1;
    for (int k = 0; k < n; ++k) {
        const int32_t sec = a[k].secondary, sc = a[k].score, qb = a[k].qb;
        if (sec >= 0 || is_alt(a[k]) || sc < T) continue;
        ++n_pri;
        if (qb < left_st) left_st = qb, left_k = k;
    }
    if (n_pri <= 1 || left_k == 0) return;
    const bwams_alnreg_t t = a[0];
    a[0] = a[left_k];
    a[left_k] = t;
    for (int k = 1; k < n; ++k) {
        int32_t s1 = a[k].secondary, s2 = a[k].secondary_all;
        if (s1 == 0) s1 = left_k; else if (s1 == left_k) s1 = 0;
        if (s2 == 0) s2 = left_k; else if (s2 == left_k) s2 = 0;
        a[k].secondary = s1; a[k].secondary_all = s2;
    }
}

// ---- mem_pair: lane per pair ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pair_pair_kernel(PairArgs A, const int64_t *out_off, const bwams_alnreg_t *out, bwams_pair_t *res) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (A.nseq >> 1)) return;
    const int64_t l_pac = A.bns.l_pac;
    bwams_pair_t R;
    R.score = R.sub = R.n_sub = 0;
    R.z[0] = R.z[1] = -1;
    R.n_pri[0] = A.n_pri[2 * p]; R.n_pri[1] = A.n_pri[2 * p + 1];
    R.n_matesw = A.n_sw[2 * p] + A.n_sw[2 * p + 1];
    if (!A.no_pairing && R.n_pri[0] && R.n_pri[1]) {
        // v: x = rid << 32 | forward position within the sequence; y = score << 32 | i << 2 | strand << 1 | end.
        // As a sort record: k = x, (s, q) = the two halves of y.  y holds a unique (i, end): no ties.
        SortRec *v = reinterpret_cast<SortRec *>(A.srt) + A.ooff[2 * p];        // the two ends' strips are adjacent
        int vn = 0;
        for (int e = 0; e < 2; ++e)
            for (int i = 0; i < R.n_pri[e]; ++i) {
                const bwams_alnreg_t *g = &out[out_off[2 * p + e] + i];
                int64_t x = g->rb < l_pac ? g->rb : (l_pac << 1) - 1 - g->rb;
                x = (int64_t)((uint64_t)g->rid << 32 | (uint64_t)(x - A.bns.contigs[g->rid].offset));
                SortRec t; t.k = x; t.s = g->score; t.q = i << 2 | (g->rb >= l_pac) << 1 | e; t.idx = 0; t.pad_ = 0;
                v[vn++] = t;
            }
        sort_records(v, vn, 4);
        int tmp = A.opt.a + A.opt.b;
        tmp = tmp > A.opt.o_del + A.opt.e_del ? tmp : A.opt.o_del + A.opt.e_del;
        tmp = tmp > A.opt.o_ins + A.opt.e_ins ? tmp : A.opt.o_ins + A.opt.e_ins;
        const int id = (int)(A.id_base + p);
        // u is never stored: its order is total ((q, hash) then (k, i)), so its last two elements are a running
        // top-2 and n_sub a second sweep
        uint64_t b1x = 0, b1y = 0, b2x = 0, b2y = 0;
        long long un = 0;
        int sub = 0;
        for (int sweep = 0; sweep < 2; ++sweep) {
            int y[4] = {-1, -1, -1, -1};
            long long cnt = 0;
            for (int i = 0; i < vn; ++i) {
                for (int r = 0; r < 2; ++r) {
                    const int dir = r << 1 | (v[i].q >> 1 & 1);
                    if (A.pes[dir].failed) continue;
                    const int which = r << 1 | ((v[i].q & 1) ^ 1);
                    if (y[which] < 0) continue;
                    for (int k = y[which]; k >= 0; --k) {
                        if ((v[k].q & 3) != which) continue;
                        const int64_t dist = v[i].k - v[k].k;
                        if (dist > A.pes[dir].high) break;
                        if (dist < A.pes[dir].low) continue;
                        const double ns = ((double)dist - A.pes[dir].avg) / A.pes[dir].std;
                        int q = (int)((double)((int64_t)v[i].s + (int64_t)v[k].s) + .721 * log(2. * erfc(fabs(ns) * 0.70710678118654752440)) * A.opt.a + .499);
                        if (q < 0) q = 0;
                        if (sweep == 0) {
                            const uint64_t uy = (uint64_t)k << 32 | (uint64_t)i;
                            const uint64_t ux = (uint64_t)q << 32 | (hash_64(uy ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffu);
                            if (un == 0 || ux > b1x || (ux == b1x && uy > b1y)) { b2x = b1x; b2y = b1y; b1x = ux; b1y = uy; }
                            else if (un == 1 || ux > b2x || (ux == b2x && uy > b2y)) { b2x = ux; b2y = uy; }
                            ++un;
                        } else if (sub - q <= tmp) ++cnt;
                    }
                }
                y[v[i].q & 3] = i;
            }
            if (sweep == 0) {
                if (un == 0) break;
                sub = un > 1 ? (int)(b2x >> 32) : 0;
                if (un == 1) break;
            } else R.n_sub = (int)(cnt - 1);                 // every element but the best one (whose q >= sub)
        }
        if (un) {
            const int i = (int)(uint32_t)b1y, k = (int)(b1y >> 32);
            R.z[v[i].q & 1] = v[i].q >> 2;
            R.z[v[k].q & 1] = v[k].q >> 2;
            R.score = (int)(b1x >> 32);
            R.sub = sub;
        }
    }
    res[p] = R;
}

}  // namespace

static unsigned blocks_of(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

void launch_pair_count(const PairArgs &A, int64_t *wide, hipStream_t st) {
    pair_count_kernel<<<blocks_of(A.nseq + 1, 256), 256, 0, st>>>(A, wide);
}
void launch_pair_cap(const PairArgs &A, int64_t *wide, hipStream_t st) {
    pair_cap_kernel<<<blocks_of(A.nseq + 1, 256), 256, 0, st>>>(A, wide);
}
void launch_pair_slots(const PairArgs &A, hipStream_t st) {
    if (A.nseq > 0) pair_slots_kernel<<<blocks_of(A.nseq, 256), 256, 0, st>>>(A);
}
void launch_pair_plan(const PairArgs &A, int64_t *wide, hipStream_t st) {
    pair_plan_kernel<<<blocks_of(A.n_slots > 0 ? A.n_slots : 1, 64), 64, 0, st>>>(A, wide);
}
void launch_pair_build(const PairArgs &A, const int64_t *offs, bwams_seqpair_t *pairs, uint8_t *tref, uint8_t *tqer, int cu_count,
                       hipStream_t st) {
    if (A.n_slots <= 0) return;
    int64_t blocks = A.n_slots;                              // 4 entries per slot, 4 waves per block
    if (blocks > (int64_t)cu_count * 16) blocks = (int64_t)cu_count * 16;
    pair_build_kernel<<<(unsigned)blocks, 256, 0, st>>>(A, offs, pairs, tref, tqer);
}
void launch_pair_post(const PairArgs &A, int cu_count, hipStream_t st) {
    if (A.nseq <= 0) return;
    pair_post_kernel<<<blocks_of(A.nseq, 64), 64, 0, st>>>(A);
    pair_post_wave_kernel<<<(unsigned)(cu_count * 2), 64, 0, st>>>(A);
}
void launch_pair_mark(const PairArgs &A, int cu_count, hipStream_t st) {
    if (A.nseq <= 0) return;
    pair_mark_kernel<<<blocks_of(A.nseq, 64), 64, 0, st>>>(A);
    // per launch: the attribute belongs to the current device (a failure here surfaces as the launch error the caller checks)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pair_mark_wave_kernel<kMarkLdsMax, kMarkLdsSmall>), hipFuncAttributeMaxDynamicSharedMemorySize, 68 * kMarkLdsMax);
    pair_mark_wave_kernel<kMarkLdsMax, kMarkLdsSmall><<<(unsigned)cu_count, 64, 68 * kMarkLdsMax, st>>>(A, &A.ctr->pair_ticket);
    pair_mark_wave_kernel<kMarkLdsSmall, kMarkLight><<<(unsigned)(cu_count * 8), 64, 68 * kMarkLdsSmall, st>>>(A, &A.ctr->pair_ticket2);
}
void launch_pair_widen(const PairArgs &A, int64_t *wide, hipStream_t st) {
    pair_widen_kernel<<<blocks_of(A.nseq + 1, 256), 256, 0, st>>>(A, wide);
}
void launch_pair_gather(const PairArgs &A, const int64_t *out_off, bwams_alnreg_t *out, hipStream_t st) {
    if (A.nseq > 0) pair_gather_kernel<<<(unsigned)A.nseq, 64, 0, st>>>(A, out_off, out);
}
void launch_pair_reorder5(const PairArgs &A, const int64_t *out_off, bwams_alnreg_t *out, hipStream_t st) {
    if (A.nseq > 0 && A.primary5_T >= 0) pair_reorder5_kernel<<<blocks_of(A.nseq, 64), 64, 0, st>>>(A, out_off, out);
}
void launch_pair_pair(const PairArgs &A, const int64_t *out_off, const bwams_alnreg_t *out, bwams_pair_t *res, hipStream_t st) {
    if (A.nseq > 1) pair_pair_kernel<<<blocks_of(A.nseq >> 1, 64), 64, 0, st>>>(A, out_off, out, res);
}

}  // namespace bwams
