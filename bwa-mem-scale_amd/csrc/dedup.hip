// dedup.hip — the tail of mem_kernel2_core on the device (/root/reference/src/bwamem.cpp:1446-1481):
// purged regions dropped, mem_sort_dedup_patch (:314-375) with mem_patch_reg (:199-250) and the score-only
// global alignment it calls (bwa_gen_cigar2 -> ksw_global2, bwa.cpp:380-428, ksw.cpp:558-649), the ALT mark.
//
// Sequential per read (sort by end, pairwise redundancy / patch tests against the regions just upstream, sort
// by score, drop identical hits), a handful of regions per read: one lane per read for reads with few regions,
// one wavefront per read for the others (sort records in LDS; the sorts, the pairwise pass — 64 upstream regions at a
// time, events in the serial order — and the patch alignment all run across the lanes).  Both sorts are ksort.h's
// introsort — unstable, so reproduced operation by operation (see chain.hip) on 24-byte sort records whenever two keys
// compare equal.  The banded global alignment of a patch candidate: in the lane tier row by row on the lane ((h, e) row
// in a per-lane HBM strip: rare), in the wave tier 64 columns per step with the row in LDS (global_score_wave).
#include "common.h"
#include "chain_kernels.h"
#include "wave_ops.h"
#include "region_sort.h"

namespace bwams {
namespace {

constexpr int kLightN = 16;          // regions per read handled by a single lane (32: the lane tier was the stage's longest launch, 4.5 -> 3.9 ms)
constexpr int kSmallN = 128, kMidN = 512;      // regions per read of the wave tier's smaller instances
constexpr int kLdsN = 2048;          // sort records a wavefront keeps in LDS (largest instance of the wave tier)
constexpr int MINUS_INF = -0x40000000;
constexpr int kEhLdsLen = 1000;      // reads up to this length align their patch candidates with the row in LDS (9 bytes per base)
__host__ __device__ constexpr size_t dd_lds_bytes(int cap, int max_read_len) {
    return (size_t)60 * cap + (max_read_len <= kEhLdsLen ? (size_t)(max_read_len + 2) * 8 + (((size_t)max_read_len + 2 + 15) & ~(size_t)15) : 0);
}

// ksw_global2 without backtrack; query[j] = qseq[qs * j], target[i] = tseq[ts * i] (ts = qs = -1 on the reverse strand,
// where bwa_gen_cigar2 reverses both sequences); eh: qlen + 1 cells of this lane's strip
__device__ int global_score(const bwams_mem_opt_t &o, int qlen, const uint8_t *qseq, int qs, int tlen, const uint8_t *tseq, int ts,
                            int w, int2 *eh) {
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    eh[0] = make_int2(0, MINUS_INF);
    int j;
    for (j = 1; j <= qlen && j <= w; ++j) eh[j] = make_int2(-(o.o_ins + o.e_ins * j), MINUS_INF);
    for (; j <= qlen; ++j) eh[j] = make_int2(MINUS_INF, MINUS_INF);
    for (int i = 0; i < tlen; ++i) {
        int f = MINUS_INF;
        const int8_t *mrow = &o.mat[tseq[(int64_t)ts * i] * 5];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        int h1 = beg == 0 ? -(o.o_del + o.e_del * (i + 1)) : MINUS_INF;
        // eight cells at a time: the row lives in HBM and a lane waits a full round trip per load, so the
        // loads of a block are issued together
        for (j = beg; j < end; j += 8) {
            int2 blk[8];
            int sc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int jj = j + t < end ? j + t : end - 1;
                blk[t] = eh[jj];
                sc[t] = mrow[qseq[(int64_t)qs * jj]];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (j + t < end) {
                    int m = blk[t].x, e = blk[t].y;
                    m += sc[t];
                    int h = m >= e ? m : e;
                    h = h >= f ? h : f;
                    int tt = m - oe_del;
                    e -= o.e_del;
                    e = e > tt ? e : tt;
                    eh[j + t] = make_int2(h1, e);
                    h1 = h;
                    tt = m - oe_ins;
                    f -= o.e_ins;
                    f = f > tt ? f : tt;
                }
            }
        }
        eh[end] = make_int2(h1, MINUS_INF);
    }
    return eh[qlen].x;
}

// The same alignment with the whole wavefront on a row (all 64 lanes call this with equal arguments): 64 columns per step, the (h, e) row
// and the query in LDS, the row's target base handed round from a register (one load per 64 rows).  Within a row E and the diagonal term
// come from the row above, and F — max over the columns to the left of (M - gap open) minus the extensions in between — is a prefix
// maximum (ksw_global2 feeds F from M only, ksw.cpp:607-618), so the cells of a row are independent but for that scan: the scores are
// the serial loop's, cell for cell.  Every patch candidate of a read with hundreds of overlapping regions goes through here; on one
// lane with the row in HBM such a read held its wavefront for 0.4 s.
__device__ __forceinline__ int dd_incl_max(int v) {
    asm volatile("s_nop 4\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ int global_score_wave(const bwams_mem_opt_t &o, int qlen, const uint8_t *qseq, int qs, int tlen, const uint8_t *tseq, int ts,
                                 int w, int2 *eh, uint8_t *qbuf, int lane) {
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    constexpr int kNeg = -0x30000000;                       // below every reachable score, above MINUS_INF - (what a row subtracts)
    for (int j = lane; j <= qlen; j += 64) {
        eh[j] = j == 0 ? make_int2(0, MINUS_INF) : j <= w ? make_int2(-(o.o_ins + o.e_ins * j), MINUS_INF) : make_int2(MINUS_INF, MINUS_INF);
        if (j < qlen) qbuf[j] = qseq[(int64_t)qs * j];
    }
    __syncthreads();
    // the pair is the wave's: lengths and band in scalar registers (scalar row and chunk loops); cross-lane moves are DPP (the
    // neighbour) and v_readlane (lane 63, the row's end, the row's target base), the matrix row is five bytes of one scalar
    // — the changes that took mem_reg2aln's wave kernel from 167 to 60 instructions per row chunk (profiles/r03_notes.md 88)
    qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); w = __builtin_amdgcn_readfirstlane(w);
    int tv = 0;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) tv = i + lane < tlen ? (int)tseq[(int64_t)ts * (i + lane)] : 4;
        const int tb = __builtin_amdgcn_readlane(tv, i & 63);
        const int8_t *mrow = &o.mat[tb * 5];
        const uint64_t mpk = (uint64_t)(uint8_t)mrow[0] | (uint64_t)(uint8_t)mrow[1] << 8 | (uint64_t)(uint8_t)mrow[2] << 16 |
                             (uint64_t)(uint8_t)mrow[3] << 24 | (uint64_t)(uint8_t)mrow[4] << 32;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1_first = beg == 0 ? -(o.o_del + o.e_del * (i + 1)) : MINUS_INF;
        int f_carry = MINUS_INF, h_carry = h1_first, h_end = h1_first;
        for (int c0 = beg; c0 < end; c0 += 64) {
            const int j = c0 + lane;
            const bool act = j < end;
            const int jj = act ? j : end - 1;
            const int2 p = eh[jj];
            int qb = qbuf[jj];
            qb = qb > 4 ? 4 : qb;
            const int m = p.x + (int)(int8_t)(uint8_t)(mpk >> (qb << 3));
            int e = p.y;
            const int t_ins = m - oe_ins;
            const int g = act ? t_ins + j * o.e_ins : kNeg;
            const int P = dd_incl_max(g);
            const int Pex = lane_shr1(P, kNeg);            // lane 0: nothing to its left in this chunk
            const int fc = f_carry - lane * o.e_ins;       // what the gap open before this chunk has become
            const int fp = Pex - (j - 1) * o.e_ins;
            const int f = fc > fp ? fc : fp;
            int h = m >= e ? m : e;
            h = h >= f ? h : f;
            const int t = m - oe_del;
            e -= o.e_del;
            e = e > t ? e : t;
            const int fn = f - o.e_ins;
            const int hl = lane_shr1(h, h_carry);          // lane 0 takes the previous chunk's last h
            if (act) eh[j] = make_int2(hl, e);
            const int fnext = fn > t_ins ? fn : t_ins;
            f_carry = __builtin_amdgcn_readlane(fnext, 63);
            h_carry = __builtin_amdgcn_readlane(h, 63);
            const int last = end - 1 - c0;                 // the row's last column, if it lies in this chunk
            if (last < 64) h_end = __builtin_amdgcn_readlane(h, last);
        }
        if (lane == 0) eh[end] = make_int2(h_end, MINUS_INF);
        __syncthreads();
    }
    return eh[qlen].x;
}

// The same with the whole row in REGISTERS: lane l keeps the cells j = NC l .. NC l + NC - 1 of the (h, e) row and their query bases
// (NC = 1 .. 4: queries up to 255 bases), so a row is one pass — no LDS round trip per 64 columns, one prefix maximum per row instead of
// one per chunk.  Within the lane F runs cell to cell; between lanes it is the prefix maximum of (M - gap open + e_ins * column), as above.
// A read in a satellite array takes some two hundred patch candidates one after the other (each changes what the next one sees only
// if it succeeds): 28 ms for one read of the grch38_like genome with the row in LDS.
// need: the smallest score mem_patch_reg accepts for this pair (INT_MIN: none).  Every four rows the best any path can still reach — a
// cell's H plus a match for every base of the shorter remainder, minus a gap extension for every base of the difference — is compared with it: a candidate that cannot pass any more ends there
// (its score only enters the acceptance test, bwamem.cpp:236-241; most candidates of a read in a satellite array fail, at half the rows).
template <int NC>
__device__ int global_score_wave_reg(const bwams_mem_opt_t &o, int qlen, const uint8_t *qseq, int qs, int tlen, const uint8_t *tseq, int ts,
                                     int w, int lane, int need) {
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins, e_del = o.e_del, e_ins = o.e_ins;
    constexpr int kNeg = -0x30000000;
    qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); w = __builtin_amdgcn_readfirstlane(w);
    int h[NC], e[NC], qb[NC], je[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = lane * NC + c;
        h[c] = j == 0 ? 0 : (j <= w && j <= qlen) ? -(o.o_ins + e_ins * j) : MINUS_INF;
        e[c] = MINUS_INF;
        const int b = j < qlen ? (int)qseq[(int64_t)qs * j] : 4;
        qb[c] = (b > 4 ? 4 : b) << 3;
        je[c] = j * e_ins;
    }
    uint64_t mp[5];                                        // the matrix rows, five scores a word (scalar registers: the row loop loads nothing but the target)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int8_t *mrow = &o.mat[t * 5];
        mp[t] = (uint64_t)(uint8_t)mrow[0] | (uint64_t)(uint8_t)mrow[1] << 8 | (uint64_t)(uint8_t)mrow[2] << 16 |
                (uint64_t)(uint8_t)mrow[3] << 24 | (uint64_t)(uint8_t)mrow[4] << 32;
    }
    int tv = 0;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) tv = i + lane < tlen ? (int)tseq[(int64_t)ts * (i + lane)] : 4;
        int tb = __builtin_amdgcn_readlane(tv, i & 63);
        tb = tb > 4 ? 4 : tb;
        const uint64_t mpk = tb == 0 ? mp[0] : tb == 1 ? mp[1] : tb == 2 ? mp[2] : tb == 3 ? mp[3] : mp[4];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1_first = beg == 0 ? -(o.o_del + e_del * (i + 1)) : MINUS_INF;
        int m[NC], ti[NC], hv[NC];
        int loc = kNeg;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane * NC + c;
            m[c] = h[c] + (int)(int8_t)(uint8_t)(mpk >> qb[c]);
            ti[c] = m[c] - oe_ins + je[c];                  // (M - gap open) + e_ins * column: what the prefix maximum runs over
            if (j >= beg && j < end) loc = loc > ti[c] ? loc : ti[c];
        }
        int g = lane_shr1(dd_incl_max(loc), kNeg);         // the cells of the lanes to the left
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane * NC + c;
            const int f = g - je[c] + e_ins;
            int hh = m[c] >= e[c] ? m[c] : e[c];
            hv[c] = hh >= f ? hh : f;
            if (j >= beg && j < end) g = g > ti[c] ? g : ti[c];
        }
        if ((i & 3) == 3 && need > MINUS_INF) {
            const int rows_left = tlen - 1 - i;
            int ub = kNeg;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = lane * NC + c;
                const int cols_left = qlen - 1 - j;
                // ... and the remainders' difference is gap bases, an extension each at least (the path may be inside a gap already)
                const int u = hv[c] + (rows_left < cols_left ? o.a * rows_left - e_ins * (cols_left - rows_left)
                                                             : o.a * cols_left - e_del * (rows_left - cols_left));
                if (j >= beg && j < end) ub = ub > u ? ub : u;
            }
            if (__builtin_amdgcn_readlane(dd_incl_max(ub), 63) < need) return MINUS_INF;
        }
        const int hv_left0 = lane_shr1(hv[NC - 1], 0);     // H of the cell left of this lane's first one
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane * NC + c;
            const int left = c == 0 ? hv_left0 : hv[c - 1];
            if (j >= beg && j <= end) h[c] = j == beg ? h1_first : left;
            if (j >= beg && j < end) {
                const int t = m[c] - oe_del, ee = e[c] - e_del;
                e[c] = ee > t ? ee : t;
            } else if (j == end) e[c] = MINUS_INF;
        }
    }
    int last = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) if (qlen % NC == c) last = h[c];
    return __builtin_amdgcn_readlane(last, qlen / NC);
}

// mem_patch_reg's tests on the two regions' coordinates alone (bwamem.cpp:205-217), a upstream of b: false = it returns 0 before any alignment
__device__ __forceinline__ bool patch_geom(const bwams_mem_opt_t &opt, int64_t l_pac, int64_t a_rb, int64_t a_re, int a_qb, int a_qe,
                                           int64_t b_rb, int64_t b_re, int b_qb, int b_qe) {
    if (a_rb < l_pac && b_rb >= l_pac) return false;
    if (a_qb >= b_qb || a_qe >= b_qe || a_re >= b_re) return false;
    int w = (int)((a_re - b_rb) - (a_qe - b_qb));
    w = w > 0 ? w : -w;
    double r = (double)(a_re - b_rb) / (double)(b_re - a_rb) - (double)(a_qe - b_qb) / (double)(b_qe - a_qb);
    r = r > 0. ? r : -r;
    if (a_re < b_rb || a_qe < b_qb) {
        if (w > (opt.w << 1) || r >= (double)0.05f) return false;
    } else if (w > (opt.w << 2) || r >= (double)(0.05f * 2)) return false;
    return true;
}

// mem_patch_reg (bwamem.cpp:199-250): score of the merged alignment, or 0
// WAVE: called by all 64 lanes with equal arguments, eh and qbuf in LDS
template <bool WAVE>
__device__ int patch_reg(const DedupArgs &A, const uint8_t *query, const bwams_alnreg_t &a, const bwams_alnreg_t &b, int *w_out, int2 *eh,
                         uint8_t *qbuf = nullptr, int lane = 0) {
    const bwams_mem_opt_t &opt = A.opt;
    const int64_t l_pac = A.bns.l_pac;
    if (!patch_geom(opt, l_pac, a.rb, a.re, a.qb, a.qe, b.rb, b.re, b.qb, b.qe)) return 0;
    int w = (int)((a.re - b.rb) - (a.qe - b.qb));
    w = w > 0 ? w : -w;
    w += a.w + b.w;
    w = w < (opt.w << 2) ? w : (opt.w << 2);
    // bwa_gen_cigar2 (score only)
    const int l_query = b.qe - a.qb;
    const int64_t rb = a.rb, re = b.re;
    int score = 0;
    if (!(l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac))) {
        const int64_t rlen = re - rb;
        const bool rev = rb >= l_pac;
        const uint8_t *qseq = rev ? query + a.qb + l_query - 1 : query + a.qb;
        const uint8_t *tseq = rev ? A.ref + rb + rlen - 1 : A.ref + rb;
        const int st = rev ? -1 : 1;
        if (l_query == rlen && w == 0) {
            for (int i = 0; i < l_query; ++i) score += opt.mat[tseq[(int64_t)st * i] * 5 + qseq[(int64_t)st * i]];
        } else if (WAVE) {
            int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
            int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
            int max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            const int dl = (int)(rlen - l_query) < 0 ? -(int)(rlen - l_query) : (int)(rlen - l_query);
            int ww = (max_gap + dl + 1) >> 1;
            ww = ww < w ? ww : w;
            const int min_w = dl + 3;
            ww = ww > min_w ? ww : min_w;
            // what the pair must reach (the acceptance test below): lets the alignment stop once it cannot
            const int q_s0 = (int)((double)(b.qe - a.qb) / (double)((b.qe - b.qb) + (a.qe - a.qb)) * (double)(b.score + a.score) + .499);
            const int r_s0 = (int)((double)(b.re - a.rb) / (double)((b.re - b.rb) + (a.re - a.rb)) * (double)(b.score + a.score) + .499);
            const int mx0 = q_s0 > r_s0 ? q_s0 : r_s0;
            int need = MINUS_INF;
            if (mx0 > 0) {
                need = (int)((double)0.90f * (double)mx0);
                while ((double)need / (double)mx0 < (double)0.90f) ++need;     // the smallest score the test accepts
            }
            if (l_query < 64) score = global_score_wave_reg<1>(opt, l_query, qseq, st, (int)rlen, tseq, st, ww, lane, need);
            else if (l_query < 128) score = global_score_wave_reg<2>(opt, l_query, qseq, st, (int)rlen, tseq, st, ww, lane, need);
            else if (l_query < 192) score = global_score_wave_reg<3>(opt, l_query, qseq, st, (int)rlen, tseq, st, ww, lane, need);
            else if (l_query < 256) score = global_score_wave_reg<4>(opt, l_query, qseq, st, (int)rlen, tseq, st, ww, lane, need);
            else score = global_score_wave(opt, l_query, qseq, st, (int)rlen, tseq, st, ww, eh, qbuf, lane);
        } else {
            int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
            int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
            int max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            const int dl = (int)(rlen - l_query) < 0 ? -(int)(rlen - l_query) : (int)(rlen - l_query);
            int ww = (max_gap + dl + 1) >> 1;
            ww = ww < w ? ww : w;
            const int min_w = dl + 3;
            ww = ww > min_w ? ww : min_w;
            score = global_score(opt, l_query, qseq, st, (int)rlen, tseq, st, ww, eh);
        }
    }
    const int q_s = (int)((double)(b.qe - a.qb) / (double)((b.qe - b.qb) + (a.qe - a.qb)) * (double)(b.score + a.score) + .499);
    const int r_s = (int)((double)(b.re - a.rb) / (double)((b.re - b.rb) + (a.re - a.rb)) * (double)(b.score + a.score) + .499);
    if ((double)score / (double)(q_s > r_s ? q_s : r_s) < (double)0.90f) return 0;
    *w_out = w;
    return score;
}

// The whole per-read procedure, run by one lane.  srt: room for the read's sort records (HBM strip or LDS).
__device__ int dedup_read(const DedupArgs &A, int64_t r, SortRec *srt, int2 *eh) {
    const int64_t reg0 = A.seed_off[r];
    const int av_n = (int)(A.seed_off[r + 1] - reg0);
    bwams_alnreg_t *a = A.regs + reg0;
    int32_t *ord = A.ord + reg0;
    const uint8_t *query = A.enc + A.cum[r];
    int n = 0;
    for (int i = 0; i < av_n; ++i)                         // bwamem.cpp:1446-1456
        if (a[i].qe > a[i].qb) ord[n++] = i;
    if (n > 1) {
        for (int i = 0; i < n; ++i) { SortRec x; x.k = a[ord[i]].re; x.s = 0; x.q = 0; x.idx = ord[i]; x.pad_ = 0; srt[i] = x; }
        sort_records(srt, n, 0);
        for (int i = 0; i < n; ++i) { ord[i] = srt[i].idx; a[ord[i]].n_comp_is_alt = 1; }
        for (int i = 1; i < n; ++i) {
            bwams_alnreg_t *p = &a[ord[i]];
            const bwams_alnreg_t *pr = &a[ord[i - 1]];
            if (p->rid != pr->rid || p->rb >= pr->re + A.opt.max_chain_gap) continue;
            for (int j = i - 1; j >= 0; --j) {
                bwams_alnreg_t *q = &a[ord[j]];
                if (!(p->rid == q->rid && p->rb < q->re + A.opt.max_chain_gap)) break;
                if (q->qe == q->qb) continue;
                const int64_t or_ = q->re - p->rb;
                const int64_t oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
                const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
                const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
                int score, w;
                if ((float)or_ > A.opt.mask_level_redun * (float)mr && (float)oq > A.opt.mask_level_redun * (float)mq) {
                    if (p->score < q->score) { p->qe = p->qb; break; }
                    else q->qe = q->qb;
                } else if (q->rb < p->rb && (score = patch_reg<false>(A, query, *q, *p, &w, eh)) > 0) {
                    p->n_comp_is_alt = (p->n_comp_is_alt + q->n_comp_is_alt + 1) & 0x3fffffff;
                    p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                    p->sub = p->sub > q->sub ? p->sub : q->sub;
                    p->csub = p->csub > q->csub ? p->csub : q->csub;
                    p->qb = q->qb; p->rb = q->rb;
                    p->truesc = p->score = score;
                    p->w = w;
                    q->qb = q->qe;
                }
            }
        }
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (a[ord[i]].qe > a[ord[i]].qb) ord[m++] = ord[i];
        n = m;
        for (int i = 0; i < n; ++i) {
            const bwams_alnreg_t *p = &a[ord[i]];
            SortRec x; x.k = p->rb; x.s = p->score; x.q = p->qb; x.idx = ord[i]; x.pad_ = 0;
            srt[i] = x;
        }
        sort_records(srt, n, 1);
        for (int i = 0; i < n; ++i) ord[i] = srt[i].idx;
        for (int i = 1; i < n; ++i) {
            bwams_alnreg_t *p = &a[ord[i]];
            const bwams_alnreg_t *pr = &a[ord[i - 1]];
            if (p->score == pr->score && p->rb == pr->rb && p->qb == pr->qb) p->qe = p->qb;
        }
        m = n ? 1 : 0;
        for (int i = 1; i < n; ++i)
            if (a[ord[i]].qe > a[ord[i]].qb) ord[m++] = ord[i];
        n = m;
    }
    for (int i = 0; i < n; ++i) {                          // bwamem.cpp:1470-1481
        bwams_alnreg_t *p = &a[ord[i]];
        if (p->rid >= 0 && A.bns.contigs[p->rid].is_alt) p->n_comp_is_alt = (p->n_comp_is_alt & 0x3fffffff) | (1 << 30);
    }
    return n;
}

// lane per read: reads left with at most one region are finished here; the others are listed for the
// lane tier (few regions) or the wave tier (many)
__global__ void dedup_triage_kernel(DedupArgs A) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    const int64_t reg0 = A.seed_off[r];
    const int av_n = (int)(A.seed_off[r + 1] - reg0);
    bwams_alnreg_t *a = A.regs + reg0;
    int n = 0, last = 0;
    for (int i = 0; i < av_n; ++i)
        if (a[i].qe > a[i].qb) { last = i; ++n; }
    if (n <= 1) {                                  // mem_sort_dedup_patch returns at once (n_comp stays 0)
        if (n == 1) {
            A.ord[reg0] = last;
            if (a[last].rid >= 0 && A.bns.contigs[a[last].rid].is_alt) a[last].n_comp_is_alt = (a[last].n_comp_is_alt & 0x3fffffff) | (1 << 30);
        }
        A.n_out[r] = n;
        return;
    }
    if (av_n > kLightN) A.heavy[atomicAdd(A.n_heavy_ctr, 1ull)] = (int32_t)r;
    else A.light[atomicAdd(A.n_light_ctr, 1ull)] = (int32_t)r;
}

// lane tier: one lane per listed read (each lane owns a strip for the global alignment)
__global__ __launch_bounds__(64) void dedup_kernel(DedupArgs A, int64_t n_lanes) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_lanes) return;
    int2 *eh = A.eh + g * (int64_t)(A.max_read_len + 2);
    const int64_t n_light = (int64_t)*A.n_light_ctr;
    for (int64_t t = g; t < n_light; t += n_lanes) {
        const int64_t r = A.light[t];
        A.n_out[r] = dedup_read(A, r, reinterpret_cast<SortRec *>(A.srt) + A.seed_off[r], eh);
    }
}

// One wavefront per read with many regions.  Maps, compactions and sorts run across the lanes; the pairwise pass
// (sequential by nature) evaluates 64 upstream regions at a time against the current one from LDS copies of the fields
// its tests read and takes the events one by one (see the kernel).
// Fields that lane 0 rewrites during the pairwise pass are re-read by the other lanes afterwards: those loads go
// past the vector L1 (which may still hold the line from the first pass) with agent-scope atomic loads.
__device__ __forceinline__ void load_qbqe(const bwams_alnreg_t *p, int &qb, int &qe) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&p->qb), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    qb = (int)(uint32_t)v; qe = (int)(uint32_t)(v >> 32);
}
// slots of the regions with qe > qb, in order; ord lives in LDS
__device__ __forceinline__ int wave_compact_alive(const bwams_alnreg_t *a, int32_t *ord, int n_in, bool first, int lane) {
    int n = 0;
    for (int ib = 0; ib < n_in; ib += 64) {
        const int i = ib + lane;
        int slot = 0;
        bool alive = false;
        if (i < n_in) {
            slot = first ? i : ord[i];
            int qb, qe;
            load_qbqe(&a[slot], qb, qe);
            alive = qe > qb;
        }
        const unsigned long long m = __ballot(alive);
        if (alive) ord[n + __popcll(m & ((1ull << lane) - 1ull))] = slot;     // n + rank <= i: never ahead of a pending read
        n += __popcll(m);
    }
    return n;
}

// Three instances share the list of heavy reads, each with its own ticket counter and LDS budget (60 B per region):
// LO < regions <= CAP for (32, 128] (7.7 KB per wave, ~20 waves per CU), (128, 512] (30 KB, five per CU) and
// (512, 2048] (120 KB, one per CU; 355 reads per million on the bench workload).  One 1024-region instance left two waves
// per CU for every heavy read, and the four reads beyond it sorted 1100 records through HBM on a single lane: 14.9 ms.
template <int CAP, int LO>
__global__ __launch_bounds__(64) void dedup_wave_kernel(DedupArgs A, int64_t n_waves, int64_t eh_base, unsigned long long *ticket) {
    extern __shared__ __align__(16) unsigned char lds_dd[];
    SortRec *l_srt = reinterpret_cast<SortRec *>(lds_dd), *l_srt2 = l_srt + CAP;
    // behind the 60 CAP bytes, when the launch gave the room (reads of up to kEhLdsLen bases): the (h, e) row and the query of a patch alignment
    int2 *lds_eh = A.max_read_len <= kEhLdsLen ? reinterpret_cast<int2 *>(lds_dd + (size_t)60 * CAP) : nullptr;
    uint8_t *lds_q = reinterpret_cast<uint8_t *>(lds_eh + (A.max_read_len + 2));
    int4 *l_x = reinterpret_cast<int4 *>(l_srt2);          // during the pairwise pass (the sorts' second buffer is free then): {qb, qe, score, -}
    int64_t *l_rb = reinterpret_cast<int64_t *>(l_srt2 + CAP);
    int32_t *l_ord = reinterpret_cast<int32_t *>(l_rb + CAP);
    const int lane = threadIdx.x;
    int2 *eh = A.eh + (eh_base + blockIdx.x) * (int64_t)(A.max_read_len + 2);
    const int64_t n_heavy = (int64_t)*A.n_heavy_ctr;
    for (;;) {
        const int64_t t = (int64_t)wave_ticket(ticket, 1ull);
        if (t >= n_heavy) break;
        const int64_t r = A.heavy[t];
        const int64_t reg0 = A.seed_off[r];
        const int av_n = (int)(A.seed_off[r + 1] - reg0);
        if (av_n <= LO || (av_n > CAP && CAP != kLdsN)) continue;       // another instance's read (the largest takes what is beyond, too)
        bwams_alnreg_t *a = A.regs + reg0;
        int32_t *ord = A.ord + reg0;
        const uint8_t *query = A.enc + A.cum[r];
        __syncthreads();
        if (av_n > kLdsN || A.force_seq == 1) {                              // beyond the LDS budget: the one-lane form
            if (lane == 0) A.n_out[r] = dedup_read(A, r, reinterpret_cast<SortRec *>(A.srt) + reg0, eh);
            continue;
        }
        const bool prof = CAP == kLdsN && A.dbg != nullptr;
        unsigned long long n_al = 0, t_al = 0, n_it = 0;       // patch alignments of the read, their time, scan trips (diagnostics)
        unsigned long long tk0 = prof ? wall_clock64() : 0ull, tk1 = tk0, tk2 = tk0, tk3 = tk0, tk4 = tk0, tk5 = tk0;
        int n = wave_compact_alive(a, l_ord, av_n, true, lane);         // bwamem.cpp:1446-1456
        const int n_alive0 = n;
        __syncthreads();
        if (n > 1) {
            for (int i = lane; i < n; i += 64) {
                const bwams_alnreg_t *p = &a[l_ord[i]];
                SortRec x; x.k = p->re; x.s = p->rid; x.q = 0; x.idx = l_ord[i]; x.pad_ = 0;
                l_srt[i] = x;
            }
            __syncthreads();
            if (prof) tk1 = wall_clock64();
            wave_sort_records(l_srt, l_srt2, n, 0, lane, false, 0, CAP);
            if (prof) tk2 = wall_clock64();
            for (int i = lane; i < n; i += 64) {
                const int slot = l_srt[i].idx;
                l_ord[i] = slot;
                l_rb[i] = a[slot].rb;
                a[slot].n_comp_is_alt = 1;
            }
            __threadfence_block();
            __syncthreads();
            // The pairwise pass.  Sequential by nature (a merge changes p, a deletion hides q from every later p), but nearly every test
            // ends in "nothing to do": the wave evaluates 64 upstream regions at a time against the current p from LDS copies of the
            // fields the tests read (re, rid: the sort records; rb; qb, qe, score), and only the events — the end of the scan, a redundant
            // pair, a pair that passes mem_patch_reg's coordinate tests — are taken one by one, in the order the serial loop meets them.
            // The records in HBM are written through at every change (patch_reg and the passes behind this one read them).
            for (int i = lane; i < n; i += 64) {
                const bwams_alnreg_t *p = &a[l_ord[i]];
                l_x[i] = make_int4(p->qb, p->qe, p->score, 0);
            }
            __syncthreads();
            const int64_t gap = A.opt.max_chain_gap, l_pac = A.bns.l_pac;
            const float mlr = A.opt.mask_level_redun;
            for (int i = 1; i < n; ++i) {
                const int rid_i = l_srt[i].s;
                if (rid_i != l_srt[i - 1].s || l_rb[i] >= l_srt[i - 1].k + gap) continue;
                int64_t p_rb = l_rb[i];
                const int64_t p_re = l_srt[i].k;
                int4 px = l_x[i];
                bool done = false;
                int jtop = i - 1;
                while (jtop >= 0 && !done) {
                    ++n_it;
                    const int j = jtop - lane;
                    bool end = j < 0, red = false, pat = false;
                    if (!end) {
                        const int64_t q_re = l_srt[j].k;
                        end = !(rid_i == l_srt[j].s && p_rb < q_re + gap);
                        if (!end) {
                            const int4 qx = l_x[j];
                            const int64_t q_rb = l_rb[j];
                            if (qx.y != qx.x) {
                                const int64_t or_ = q_re - p_rb;
                                const int64_t oq = qx.x < px.x ? qx.y - px.x : px.y - qx.x;
                                const int64_t mr = q_re - q_rb < p_re - p_rb ? q_re - q_rb : p_re - p_rb;
                                const int64_t mq = qx.y - qx.x < px.y - px.x ? qx.y - qx.x : px.y - px.x;
                                red = (float)or_ > mlr * (float)mr && (float)oq > mlr * (float)mq;
                                pat = !red && q_rb < p_rb && patch_geom(A.opt, l_pac, q_rb, q_re, qx.x, qx.y, p_rb, p_re, px.x, px.y);
                            }
                        }
                    }
                    const unsigned long long m_end = __ballot(end), m_red = __ballot(red);
                    unsigned long long ev = m_end | m_red | __ballot(pat);
                    int consumed = 64;
                    while (ev) {
                        const int l = __builtin_ctzll(ev);
                        ev &= ev - 1;
                        if ((m_end >> l) & 1) { done = true; break; }
                        const int jj = jtop - l;
                        bwams_alnreg_t *pp = &a[l_ord[i]], *qq = &a[l_ord[jj]];
                        if ((m_red >> l) & 1) {
                            if (px.z < l_x[jj].z) {
                                px.y = px.x;
                                if (lane == 0) { pp->qe = px.x; l_x[i] = px; }
                                done = true;
                                break;
                            }
                            if (lane == 0) { const int qb = l_x[jj].x; qq->qe = qb; l_x[jj].y = qb; }
                            continue;
                        }
                        int score = 0, w = 0;
                        const unsigned long long ta0 = prof ? wall_clock64() : 0ull;
                        if (lds_eh) {
                            const bwams_alnreg_t qa = *qq, pa = *pp;       // written through by lane 0 only, read back behind a barrier
                            score = patch_reg<true>(A, query, qa, pa, &w, lds_eh, lds_q, lane);
                        } else if (lane == 0) score = patch_reg<false>(A, query, *qq, *pp, &w, eh);
                        if (prof) { t_al += wall_clock64() - ta0; ++n_al; }
                        score = __builtin_amdgcn_readfirstlane(score);
                        w = __builtin_amdgcn_readfirstlane(w);
                        if (score > 0) {
                            __syncthreads();
                            const int4 qx = l_x[jj];
                            p_rb = l_rb[jj];
                            px.x = qx.x; px.z = score;
                            __syncthreads();
                            if (lane == 0) {
                                pp->n_comp_is_alt = (pp->n_comp_is_alt + qq->n_comp_is_alt + 1) & 0x3fffffff;
                                pp->seedcov = pp->seedcov > qq->seedcov ? pp->seedcov : qq->seedcov;
                                pp->sub = pp->sub > qq->sub ? pp->sub : qq->sub;
                                pp->csub = pp->csub > qq->csub ? pp->csub : qq->csub;
                                pp->qb = qq->qb; pp->rb = qq->rb;
                                pp->truesc = pp->score = score;
                                pp->w = w;
                                qq->qb = qq->qe;
                                l_rb[i] = p_rb; l_x[i] = px; l_x[jj].x = qx.y;
                            }
                            consumed = l + 1;                            // p changed: the regions further upstream are tested again
                            break;
                        }
                    }
                    __syncthreads();
                    jtop -= consumed;
                }
            }
            __threadfence_block();
            __syncthreads();
            if (prof) tk3 = wall_clock64();
            n = wave_compact_alive(a, l_ord, n, false, lane);
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const bwams_alnreg_t *p = &a[l_ord[i]];
                SortRec x;
                x.k = (int64_t)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(&p->rb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x.s = __hip_atomic_load(&p->score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int qb, qe;
                load_qbqe(p, qb, qe);
                x.q = qb; x.idx = l_ord[i]; x.pad_ = 0;
                l_srt[i] = x;
            }
            __syncthreads();
            if (prof) tk4 = wall_clock64();
            wave_sort_records(l_srt, l_srt2, n, 1, lane, false, 0, CAP);
            if (prof) tk5 = wall_clock64();
            // identical hits: same (score, rb, qb) as the predecessor in sorted order
            int m = 0;
            for (int ib = 0; ib < n; ib += 64) {
                const int i = ib + lane;
                bool keep = false;
                int slot = 0;
                if (i < n) {
                    const SortRec x = l_srt[i];
                    slot = x.idx;
                    keep = i == 0 || !(x.s == l_srt[i - 1].s && x.k == l_srt[i - 1].k && x.q == l_srt[i - 1].q);
                    if (!keep) a[slot].qe = a[slot].qb;
                }
                const unsigned long long mk = __ballot(keep);
                if (keep) l_ord[m + __popcll(mk & ((1ull << lane) - 1ull))] = slot;
                m += __popcll(mk);
            }
            n = m;
            __threadfence_block();
            __syncthreads();
        }
        for (int i = lane; i < n; i += 64) {                            // bwamem.cpp:1470-1481; final order out to HBM
            bwams_alnreg_t *p = &a[l_ord[i]];
            ord[i] = l_ord[i];
            const int nc = __hip_atomic_load(&p->n_comp_is_alt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p->rid >= 0 && A.bns.contigs[p->rid].is_alt) p->n_comp_is_alt = (nc & 0x3fffffff) | (1 << 30);
        }
        if (lane == 0) A.n_out[r] = n;
        if (prof && lane == 0) {
            const unsigned long long tk6 = wall_clock64();
            atomicAdd(&A.dbg[0], 1ull); atomicAdd(&A.dbg[1], (unsigned long long)av_n); atomicAdd(&A.dbg[2], (unsigned long long)n_alive0);
            atomicAdd(&A.dbg[3], tk1 - tk0); atomicAdd(&A.dbg[4], tk2 - tk1); atomicAdd(&A.dbg[5], tk3 - tk2);
            atomicAdd(&A.dbg[6], tk4 - tk3); atomicAdd(&A.dbg[7], tk5 - tk4); atomicAdd(&A.dbg[8], tk6 - tk5);
            atomicMax(&A.dbg[9], tk2 - tk1); atomicMax(&A.dbg[10], tk3 - tk2); atomicMax(&A.dbg[11], tk5 - tk4);
            if (atomicMax(&A.dbg[12], tk6 - tk0) < tk6 - tk0) { A.dbg[13] = (unsigned long long)r; A.dbg[14] = (unsigned long long)n_alive0; A.dbg[15] = n_al; A.dbg[16] = t_al; A.dbg[17] = n_it; }
            atomicAdd(&A.dbg[18], n_al); atomicAdd(&A.dbg[19], t_al);
        }
    }
}

// the surviving regions, in their final order.  Sixteen lanes per read, a 16-byte seventh of a 112-byte region per lane and step: a
// lane per read copied its regions one after the other, 112 bytes at a time at addresses of its own (1.1 ms per million reads for 0.4 GB)
static_assert(sizeof(bwams_alnreg_t) == 112, "a region is seven 16-byte pieces");
__global__ __launch_bounds__(256) void dedup_gather_kernel(DedupArgs A, const int64_t *__restrict__ out_off, bwams_alnreg_t *out) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (r >= A.nseq) return;
    const int64_t reg0 = A.seed_off[r];
    const int n = A.n_out[r];
    if (n <= 0) return;
    const int64_t o0 = out_off[r];
    const uint4 *src = reinterpret_cast<const uint4 *>(A.regs + reg0);
    uint4 *dst = reinterpret_cast<uint4 *>(out + o0);
    for (int u = sub; u < n * 7; u += 16) {
        const int i = u / 7, piece = u - 7 * i;
        dst[u] = src[(int64_t)A.ord[reg0 + i] * 7 + piece];
    }
}

// ---- mem_pestat, the per-pair part (bwamem_pair.cpp:66-108): lane per pair -> key = dir << 60 | insert size, or ~0 ----
__device__ __forceinline__ int cal_sub(const bwams_mem_opt_t &opt, int n, const bwams_alnreg_t *a) {
    int j;
    for (j = 1; j < n; ++j) {
        const int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
        const int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
        if (e_min > b_max) {
            const int lj = a[j].qe - a[j].qb, l0 = a[0].qe - a[0].qb;
            const int min_l = lj < l0 ? lj : l0;
            if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) break;
        }
    }
    return j < n ? a[j].score : opt.min_seed_len * opt.a;
}
__global__ void pestat_kernel(const bwams_alnreg_t *__restrict__ regs, const int64_t *__restrict__ reg_off, int64_t n_pairs,
                              int64_t l_pac, bwams_mem_opt_t opt, unsigned long long *keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    unsigned long long key = ~0ull;
    const int64_t o0 = reg_off[2 * i], o1 = reg_off[2 * i + 1], o2 = reg_off[2 * i + 2];
    const int n0 = (int)(o1 - o0), n1 = (int)(o2 - o1);
    if (n0 && n1) {
        const bwams_alnreg_t *r0 = regs + o0, *r1 = regs + o1;
        if (!((double)cal_sub(opt, n0, r0) > 0.8 * (double)r0[0].score) && !((double)cal_sub(opt, n1, r1) > 0.8 * (double)r1[0].score) &&
            r0[0].rid == r1[0].rid) {
            const int64_t b1 = r0[0].rb, b2 = r1[0].rb;
            const int s1 = b1 >= l_pac, s2 = b2 >= l_pac;
            const int64_t p2 = s1 == s2 ? b2 : (l_pac << 1) - 1 - b2;
            const int64_t is = p2 > b1 ? p2 - b1 : b1 - p2;
            const int dir = (s1 == s2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
            if (is && is <= opt.max_ins) key = ((unsigned long long)dir << 60) | (unsigned long long)is;
        }
    }
    keys[i] = key;
}

constexpr int kTestN = 1024;
// test hook: one wavefront sorts n records held in LDS, as the wave tier does (mode 0: rank sort, or beyond 96 records the bitonic
// network, with the exact fallback on ties, 1: the operation-exact wave introsort always, 2: lane 0 alone through sort_records on a copy in GLOBAL memory — the
// sequential statement of the same sort; 3 / 4: modes 1 / 2 with a depth budget of 2, so that the comb-sort fallback
// (wave_combsort / r_combsort) sorts nearly everything)
__global__ __launch_bounds__(64) void sort_test_kernel(const SortRec *__restrict__ in, int n, int by_score, int mode, int32_t *__restrict__ order,
                                                       SortRec *__restrict__ scratch) {
    __shared__ SortRec l_a[kTestN], l_t[kTestN];
    const int lane = threadIdx.x;
    if (mode == 2 || mode == 4) {
        for (int i = lane; i < n; i += 64) scratch[i] = in[i];
        __syncthreads();
        if (lane == 0) sort_records(scratch, n, by_score, mode == 4 ? 2 : 0);
        __syncthreads();
        for (int i = lane; i < n; i += 64) order[i] = scratch[i].idx;
        return;
    }
    for (int i = lane; i < n; i += 64) l_a[i] = in[i];
    __syncthreads();
    wave_sort_records(l_a, l_t, n, by_score, lane, mode == 1 || mode == 3, mode == 3 ? 2 : 0, kTestN);
    for (int i = lane; i < n; i += 64) order[i] = l_a[i].idx;
}

}  // namespace

int launch_sort_test(const int64_t *k, const int32_t *s_, const int32_t *q, int n, int by_score, int mode, int32_t *order) {
    if (n < 0 || n > kTestN) return -1;
    SortRec *h = (SortRec *)malloc(sizeof(SortRec) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) { h[i].k = k[i]; h[i].s = s_[i]; h[i].q = q[i]; h[i].idx = i; h[i].pad_ = 0; }
    SortRec *d_in = nullptr, *d_scr = nullptr;
    int32_t *d_ord = nullptr;
    int rc = -1;
    if (dev_malloc(&d_in, sizeof(SortRec) * (size_t)(n + 1)) == hipSuccess && dev_malloc(&d_ord, 4 * (size_t)(n + 1)) == hipSuccess &&
        dev_malloc(&d_scr, sizeof(SortRec) * (size_t)(n + 1)) == hipSuccess &&
        hipMemcpy(d_in, h, sizeof(SortRec) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess) {
        sort_test_kernel<<<1, 64>>>(d_in, n, by_score, mode, d_ord, d_scr);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(order, d_ord, 4 * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    if (d_in) (void)hipFree(d_in);
    if (d_scr) (void)hipFree(d_scr);
    if (d_ord) (void)hipFree(d_ord);
    free(h);
    return rc;
}

void launch_pestat(const bwams_alnreg_t *regs, const int64_t *reg_off, int64_t n_pairs, int64_t l_pac, const bwams_mem_opt_t &opt,
                   unsigned long long *keys, hipStream_t st) {
    if (n_pairs <= 0) return;
    pestat_kernel<<<(unsigned)((n_pairs + 255) / 256), 256, 0, st>>>(regs, reg_off, n_pairs, l_pac, opt, keys);
}

size_t dedup_sortrec_bytes(int64_t n) { return (size_t)(n + 1) * sizeof(SortRec); }

// triage, then the lane tier and the wave tier side by side (they work on disjoint reads)
int launch_dedup(const DedupArgs &A, int64_t n_lanes, int64_t n_waves, int64_t n_waves_small, hipStream_t st, hipStream_t aux,
                 hipStream_t aux2, hipStream_t aux3, hipEvent_t fork, hipEvent_t join, hipEvent_t join2, hipEvent_t join3) {
    if (A.nseq <= 0) return 0;
    dedup_triage_kernel<<<(unsigned)((A.nseq + 255) / 256), 256, 0, st>>>(A);
    if (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(aux, fork, 0) != hipSuccess) return -1;
    // the reads with the most regions first (one wave per CU), the bulk beside them on the auxiliary streams
    // the 120 KB of dynamic LDS of the largest instance need the opt-in on EVERY device a batch runs on (the attribute belongs to the
    // device that is current when it is set); cheap enough to repeat per launch, and checked
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(dedup_wave_kernel<kLdsN, kMidN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dd_lds_bytes(kLdsN, A.max_read_len)) != hipSuccess) return -1;
    dedup_wave_kernel<kLdsN, kMidN><<<(unsigned)n_waves, 64, dd_lds_bytes(kLdsN, A.max_read_len), st>>>(A, n_waves, A.eh_lanes, A.ticket);
    dedup_kernel<<<(unsigned)((n_lanes + 63) / 64), 64, 0, aux>>>(A, n_lanes);
    if (hipEventRecord(join, aux) != hipSuccess || hipStreamWaitEvent(st, join, 0) != hipSuccess) return -1;
    if (hipStreamWaitEvent(aux3, fork, 0) != hipSuccess) return -1;
    dedup_wave_kernel<kMidN, kSmallN><<<(unsigned)n_waves, 64, dd_lds_bytes(kMidN, A.max_read_len), aux3>>>(A, n_waves, A.eh_lanes + n_waves, A.ticket3);
    if (hipEventRecord(join3, aux3) != hipSuccess || hipStreamWaitEvent(st, join3, 0) != hipSuccess) return -1;
    if (hipStreamWaitEvent(aux2, fork, 0) != hipSuccess) return -1;
    dedup_wave_kernel<kSmallN, kLightN><<<(unsigned)n_waves_small, 64, dd_lds_bytes(kSmallN, A.max_read_len), aux2>>>(A, n_waves_small, A.eh_lanes + 2 * n_waves, A.ticket2);
    if (hipEventRecord(join2, aux2) != hipSuccess || hipStreamWaitEvent(st, join2, 0) != hipSuccess) return -1;
    return 0;
}
void launch_dedup_gather(const DedupArgs &A, const int64_t *out_off, bwams_alnreg_t *out, hipStream_t st) {
    if (A.nseq <= 0) return;
    dedup_gather_kernel<<<(unsigned)((A.nseq * 16 + 255) / 256), 256, 0, st>>>(A, out_off, out);
}

}  // namespace bwams
