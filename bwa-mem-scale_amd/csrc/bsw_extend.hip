// bsw_extend.hip — banded Smith-Waterman seed extension for gfx950 (MI355X).
//
// Reference semantics: BandedPairWiseSW::scalarBandedSWA
// (/root/reference/src/bandedSWA.cpp:116-237), i.e. ksw_extend2: affine-gap
// extension from (0,0) with initial score h0, a band that is clipped to |i-j| <= w
// and shrinks to the non-zero part of the previous row, z-drop exit, and the six
// outputs score / qle / tle / gtle / gscore / max_off.
//
// Mapping (DESIGN.md §"BSW kernel"): one extension task per WAVEFRONT, one query
// column per lane (column j belongs to lane j & 63, chunk j >> 6), rows walked in
// order.  A row is fully parallel across its columns because in this recurrence
// the horizontal gap F is opened from M (the diagonal move), never from H:
//     F(i,j+1) = max(F(i,j) - e_ins, max(M(i,j) - o_ins - e_ins, 0))
// so F is a max-plus prefix scan of values known from the previous row, done with
// wavefront shuffles; E and M are column-local.  The row-wise band bookkeeping of
// the scalar code (row maximum and its last column, first/last non-zero column,
// z-drop) becomes wave reductions and ballots, and every decision is taken on
// complete rows exactly as the scalar loop does — an anti-diagonal sweep cannot do
// that, because the band of row i depends on all of row i-1.  Arithmetic is int32
// (scores are < 2^15 for the reference's int16 class; no saturation is relied on).
#include "common.h"
#include "wave_ops.h"

namespace bwams {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kTaskChunk = 8;      // tasks a wave reserves per atomic (one word serves ~90 M tickets/s)

__device__ __forceinline__ int wave_incl_prefix_max(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v = max(v, t);
    }
    return v;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void bsw_kernel(
    bwams_seqpair_t *__restrict__ pairs, int64_t n, const uint8_t *__restrict__ ref,
    const uint8_t *__restrict__ qer, int w0, SwParams prm, int qmax, int qlo, DevCounters *ctr, unsigned long long *head) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per_wave = (((size_t)(qmax + 1) * 8 + (size_t)qmax + 64 + 15) / 16) * 16;
    int2 *eh = reinterpret_cast<int2 *>(lds + wave * per_wave);
    uint8_t *qs = reinterpret_cast<uint8_t *>(eh + (qmax + 1));

    const int o_del = prm.o_del, e_del = prm.e_del, o_ins = prm.o_ins, e_ins = prm.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    unsigned long long cells = 0;

    int64_t pid = 0, pid_end = 0;          // this wave's reserved task range
    while (true) {
        if (pid >= pid_end) {
            pid = (int64_t)wave_ticket(head, (unsigned long long)kTaskChunk);   // out of line: see wave_ops.h
            pid_end = pid + kTaskChunk < n ? pid + kTaskChunk : n;
            if (pid >= n) break;
        }
        const int64_t cur = pid++;

        const bwams_seqpair_t sp = pairs[cur];
        const int qlen = sp.len2, tlen = sp.len1, h0 = sp.h0;
        if (qlen <= qlo) continue;                        // handled by a register-resident variant
        const uint8_t *tq = qer + sp.idq;
        const uint8_t *tr = ref + sp.idr;

        // row -1 of the DP and the query, each column on its owner lane
        for (int j = lane; j <= qlen; j += 64) {
            int h = h0;
            if (j >= 1) {
                h = h0 - oe_ins - (j - 1) * e_ins;
                h = h > 0 ? h : 0;
            }
            eh[j] = make_int2(h, 0);
            if (j < qlen) qs[j] = tq[j];
        }

        // clamp the band to the longest gap the score can pay for (bandedSWA.cpp:147-156)
        int w = w0;
        {
            int max_ins = (int)((double)(qlen * prm.max_sc + prm.end_bonus - o_ins) / e_ins + 1.);
            max_ins = max_ins > 1 ? max_ins : 1;
            w = w < max_ins ? w : max_ins;
            int max_del = (int)((double)(qlen * prm.max_sc + prm.end_bonus - o_del) / e_del + 1.);
            max_del = max_del > 1 ? max_del : 1;
            w = w < max_del ? w : max_del;
        }

        int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;

        int tb_next = tlen > 0 ? tr[0] : 4;
        for (int i = 0; i < tlen; ++i) {
            const int tb = tb_next;
            if (i + 1 < tlen) tb_next = tr[i + 1];           // overlap the next row's load with this row
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            int h1 = 0;
            if (beg == 0) {
                h1 = h0 - (o_del + e_del * (i + 1));
                if (h1 < 0) h1 = 0;
            }
            const int sc0 = prm.mat[tb * 5 + 0], sc1 = prm.mat[tb * 5 + 1], sc2 = prm.mat[tb * 5 + 2],
                      sc3 = prm.mat[tb * 5 + 3], sc4 = prm.mat[tb * 5 + 4];

            int m = 0, mj = -1;
            int first_nz = 1 << 30, last_nz = -1;
            int f_carry = 0, hl_carry = h1, h_last = h1;
            if (beg < end) {
                cells += (unsigned long long)(end - beg);
                const int c_lo = beg >> 6, c_hi = (end - 1) >> 6;
                for (int c = c_lo; c <= c_hi; ++c) {
                    const int jb = c << 6;
                    const int j = jb + lane;
                    const bool act = j >= beg && j < end;
                    const int c0 = jb > beg ? jb : beg;          // column the carries refer to
                    int2 cell = make_int2(0, 0);
                    int qj = 4;
                    if (act) {
                        cell = eh[j];
                        qj = qs[j];
                    }
                    const int S = qj == 0 ? sc0 : qj == 1 ? sc1 : qj == 2 ? sc2 : qj == 3 ? sc3 : sc4;
                    const int M = (act && cell.x) ? cell.x + S : 0;
                    const int e = cell.y;
                    int tj = M - oe_ins;
                    tj = tj > 0 ? tj : 0;
                    const int g = act ? tj + j * e_ins : NEG;
                    const int P = wave_incl_prefix_max(g, lane);
                    int Pex = __shfl_up(P, 1);
                    if (lane == 0) Pex = NEG;
                    int F = f_carry - (j - c0) * e_ins;
                    const int F2 = Pex - (j - 1) * e_ins;
                    F = F > F2 ? F : F2;
                    int h = M > e ? M : e;
                    h = h > F ? h : F;
                    int e2 = M - oe_del;
                    e2 = e2 > 0 ? e2 : 0;
                    const int e1 = e - e_del;
                    e2 = e2 > e1 ? e2 : e1;
                    int hl = __shfl_up(h, 1);
                    if (lane == 0 || j == beg) hl = hl_carry;
                    if (act) eh[j] = make_int2(hl, e2);

                    // row maximum and the last column that attains it
                    const int hm = act ? h : -1;
                    const int cm = wave_max(hm);
                    if (cm >= m) {
                        const unsigned long long eq = __ballot(act && h == cm);
                        m = cm;
                        mj = jb + 63 - __clzll((long long)eq);
                    }
                    // first / last column whose stored (h, e) is non-zero
                    const unsigned long long nz = __ballot(act && (hl != 0 || e2 != 0));
                    if (nz) {
                        const int lo = jb + __ffsll((long long)nz) - 1;
                        const int hi = jb + 63 - __clzll((long long)nz);
                        first_nz = first_nz < lo ? first_nz : lo;
                        last_nz = hi;
                    }
                    // carries into the next chunk / the row's last H
                    int fn = F - e_ins;
                    fn = fn > tj ? fn : tj;
                    const int last_lane = (end - 1 < jb + 63 ? end - 1 : jb + 63) - jb;
                    f_carry = __shfl(fn, 63);
                    hl_carry = __shfl(h, 63);
                    h_last = __shfl(h, last_lane);
                }
            }
            const int j_exit = beg < end ? end : beg;
            const int h1f = beg < end ? h_last : h1;
            if ((end & 63) == lane) eh[end] = make_int2(h1f, 0);
            if (j_exit == qlen) {
                max_ie = gscore > h1f ? max_ie : i;
                gscore = gscore > h1f ? gscore : h1f;
            }
            if (m == 0) break;
            if (m > mx) {
                mx = m; max_i = i; max_j = mj;
                int d = mj - i;
                d = d < 0 ? -d : d;
                max_off = max_off > d ? max_off : d;
            } else if (prm.zdrop > 0) {
                if (i - max_i > mj - max_j) {
                    if (mx - m - ((i - max_i) - (mj - max_j)) * e_del > prm.zdrop) break;
                } else {
                    if (mx - m - ((mj - max_j) - (i - max_i)) * e_ins > prm.zdrop) break;
                }
            }
            // next row's band: skip leading zeros, cut trailing zeros (bandedSWA.cpp:217-221)
            const int nbeg = first_nz < end ? first_nz : end;
            int jj;
            if (h1f != 0) jj = end;
            else if (last_nz >= nbeg) jj = last_nz;
            else jj = nbeg - 1;
            beg = nbeg;
            end = jj + 2 < qlen ? jj + 2 : qlen;
        }

        if (lane == 0) {
            bwams_seqpair_t *o = &pairs[cur];
            o->score = mx;
            o->qle = max_j + 1;
            o->tle = max_i + 1;
            o->gtle = max_ie + 1;
            o->gscore = gscore;
            o->max_off = max_off;
        }
    }
    if (lane == 0 && cells) atomicAdd(&ctr->bsw_cells, cells);
}

// ---- register-resident variant -----------------------------------------------------------
// Same algorithm as bsw_kernel, for queries of at most 64 * NCH bases: each column's (h, e)
// lives in registers of its owner lane, cross-lane traffic is DPP only (row_shr / row_bcast /
// wave_shr: no LDS, no ds_bpermute), the target row is broadcast with v_readlane from a
// 64-row register slab that is prefetched one slab ahead.
template <int NCH>
__global__ __launch_bounds__(kWavesPerBlock * 64) void bsw_kernel_reg(
    bwams_seqpair_t *__restrict__ pairs, int64_t n, const uint8_t *__restrict__ ref,
    const uint8_t *__restrict__ qer, int w0, SwParams prm, int qlo, DevCounters *ctr, unsigned long long *head) {
    const int lane = threadIdx.x & 63;
    const int o_del = prm.o_del, e_del = prm.e_del, o_ins = prm.o_ins, e_ins = prm.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    unsigned long long cells = 0;

    int64_t pid = 0, pid_end = 0;          // this wave's reserved task range
    while (true) {
        if (pid >= pid_end) {
            pid = (int64_t)wave_ticket(head, (unsigned long long)kTaskChunk);   // out of line: see wave_ops.h
            pid_end = pid + kTaskChunk < n ? pid + kTaskChunk : n;
            if (pid >= n) break;
        }
        const int64_t cur = pid++;
        const int qlen = __builtin_amdgcn_readfirstlane(pairs[cur].len2);
        if (qlen <= qlo || qlen > 64 * NCH) continue;           // another variant's task
        const int tlen = __builtin_amdgcn_readfirstlane(pairs[cur].len1);
        const int h0 = __builtin_amdgcn_readfirstlane(pairs[cur].h0);
        const uint8_t *tq = qer + __builtin_amdgcn_readfirstlane(pairs[cur].idq);
        const uint8_t *tr = ref + __builtin_amdgcn_readfirstlane(pairs[cur].idr);

        // query profile: the scores of column j against target bases 0..3 packed as four int8 (one v_bfe_i32 per cell
        // instead of a select tree), against N separately
        int H[NCH], E[NCH], JE[NCH], PK[NCH], P4[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = c * 64 + lane;
            int h = 0;
            if (j == 0) h = h0;
            else if (j <= qlen) { h = h0 - oe_ins - (j - 1) * e_ins; h = h > 0 ? h : 0; }
            H[c] = h;
            E[c] = 0;
            JE[c] = j * e_ins;
            const int qj = j < qlen ? tq[j] : 4;
            PK[c] = (int)(((uint32_t)(uint8_t)prm.mat[0 * 5 + qj]) | ((uint32_t)(uint8_t)prm.mat[1 * 5 + qj] << 8) |
                          ((uint32_t)(uint8_t)prm.mat[2 * 5 + qj] << 16) | ((uint32_t)(uint8_t)prm.mat[3 * 5 + qj] << 24));
            P4[c] = prm.mat[4 * 5 + qj];
        }
        int w = w0;
        {
            int max_ins = (int)((double)(qlen * prm.max_sc + prm.end_bonus - o_ins) / e_ins + 1.);
            max_ins = max_ins > 1 ? max_ins : 1;
            w = w < max_ins ? w : max_ins;
            int max_del = (int)((double)(qlen * prm.max_sc + prm.end_bonus - o_del) / e_del + 1.);
            max_del = max_del > 1 ? max_del : 1;
            w = w < max_del ? w : max_del;
        }
        int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        bool done = false;
        int tslab_next = lane < tlen ? tr[lane] : 4;
        for (int rb = 0; rb < tlen && !done; rb += 64) {
            const int tslab = tslab_next;
            if (rb + 64 < tlen) tslab_next = (rb + 64 + lane < tlen) ? tr[rb + 64 + lane] : 4;
            const int rlim = tlen - rb < 64 ? tlen - rb : 64;
            for (int ri = 0; ri < rlim; ++ri) {
                const int i = rb + ri;
                const int tb = __builtin_amdgcn_readlane(tslab, ri);
                if (beg < i - w) beg = i - w;
                if (end > i + w + 1) end = i + w + 1;
                if (end > qlen) end = qlen;
                int h1 = 0;
                if (beg == 0) {
                    h1 = h0 - (o_del + e_del * (i + 1));
                    if (h1 < 0) h1 = 0;
                }
                int m = 0, mj = -1;
                int first_nz = 1 << 30, last_nz = -1;
                int h_last = h1;
                if (beg < end) {
                    cells += (unsigned long long)(end - beg);
                    const int c_lo = beg >> 6, c_hi = (end - 1) >> 6;
                    // phase A — per chunk, independent of the other chunks: diagonal move, gap-open
                    // source and its prefix maximum (the scans of different chunks overlap in the pipeline)
                    int Mv[NCH], Tj[NCH], Pm[NCH];
                    bool Act[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        Mv[c] = 0; Tj[c] = 0; Pm[c] = NEG; Act[c] = false;
                        if (c < c_lo || c > c_hi) continue;
                        const int j = (c << 6) + lane;
                        const bool act = j >= beg && j < end;
                        const int S = tb < 4 ? __builtin_amdgcn_sbfe(PK[c], (unsigned)(tb << 3), 8u) : P4[c];
                        const int hd = H[c];
                        const int M = (act && hd) ? hd + S : 0;
                        int tj = M - oe_ins;
                        tj = tj > 0 ? tj : 0;
                        Mv[c] = M; Tj[c] = tj; Act[c] = act;
                        Pm[c] = scan_max(act ? tj + JE[c] : NEG);
                    }
                    // phase B — stitch the chunks: F from all columns to the left, H, E, shifted H
                    int carry_src = NEG, carry_h = 0, hmax_lane = -1;
                    int Hc[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        Hc[c] = 0;
                        if (c < c_lo || c > c_hi) continue;
                        const int j = (c << 6) + lane;
                        int Pex = lane_shr1(Pm[c], carry_src);
                        Pex = Pex > carry_src ? Pex : carry_src;
                        int F = Pex - (JE[c] - e_ins);
                        F = F > 0 ? F : 0;                                  // F(beg) = 0, F >= 0 everywhere
                        const int e = E[c];
                        int h = Mv[c] > e ? Mv[c] : e;
                        h = h > F ? h : F;
                        int e2 = Mv[c] - oe_del;
                        e2 = e2 > 0 ? e2 : 0;
                        const int e1 = e - e_del;
                        e2 = e2 > e1 ? e2 : e1;
                        int hl = lane_shr1(h, carry_h);
                        if (j == beg) hl = h1;
                        if (Act[c]) { H[c] = hl; E[c] = e2; }
                        Hc[c] = h;
                        hmax_lane = max(hmax_lane, Act[c] ? h : -1);
                        const unsigned long long nz = __ballot(Act[c] && (hl != 0 || e2 != 0));
                        if (nz) {
                            const int lo = (c << 6) + __ffsll((long long)nz) - 1;
                            first_nz = first_nz < lo ? first_nz : lo;
                            last_nz = (c << 6) + 63 - __clzll((long long)nz);
                        }
                        carry_h = __builtin_amdgcn_readlane(h, 63);
                        const int cs = __builtin_amdgcn_readlane(Pm[c], 63);
                        carry_src = carry_src > cs ? carry_src : cs;
                        if (c == c_hi) h_last = __builtin_amdgcn_readlane(h, (end - 1) & 63);
                    }
                    // row maximum over all chunks (one scan) and the last column attaining it
                    m = __builtin_amdgcn_readlane(scan_max(hmax_lane), 63);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (c < c_lo || c > c_hi) continue;
                        const unsigned long long eq = __ballot(Act[c] && Hc[c] == m);
                        if (eq) mj = (c << 6) + 63 - __clzll((long long)eq);
                    }
                }
                const int j_exit = beg < end ? end : beg;
                const int h1f = beg < end ? h_last : h1;
                // eh[end] = {h1f, 0}: column `end` is a real cell only while end < 64 * NCH
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if ((end >> 6) == c && (end & 63) == lane) { H[c] = h1f; E[c] = 0; }
                if (j_exit == qlen) {
                    max_ie = gscore > h1f ? max_ie : i;
                    gscore = gscore > h1f ? gscore : h1f;
                }
                if (m == 0) { done = true; break; }
                if (m > mx) {
                    mx = m; max_i = i; max_j = mj;
                    int d = mj - i;
                    d = d < 0 ? -d : d;
                    max_off = max_off > d ? max_off : d;
                } else if (prm.zdrop > 0) {
                    if (i - max_i > mj - max_j) {
                        if (mx - m - ((i - max_i) - (mj - max_j)) * e_del > prm.zdrop) { done = true; break; }
                    } else {
                        if (mx - m - ((mj - max_j) - (i - max_i)) * e_ins > prm.zdrop) { done = true; break; }
                    }
                }
                const int nbeg = first_nz < end ? first_nz : end;
                int jj;
                if (h1f != 0) jj = end;
                else if (last_nz >= nbeg) jj = last_nz;
                else jj = nbeg - 1;
                beg = nbeg;
                end = jj + 2 < qlen ? jj + 2 : qlen;
            }
        }
        if (lane == 0) {
            bwams_seqpair_t *o = &pairs[cur];
            o->score = mx;
            o->qle = max_j + 1;
            o->tle = max_i + 1;
            o->gtle = max_ie + 1;
            o->gscore = gscore;
            o->max_off = max_off;
        }
    }
    if (lane == 0 && cells) atomicAdd(&ctr->bsw_cells, cells);
}

__global__ void bsw_reset_kernel(DevCounters *ctr) {
    for (int i = 0; i < 4; ++i) ctr->bsw_head[i] = 0;      // bsw_cells accumulates until the caller clears it
}

}  // namespace

// One kernel per query-length class (register-resident for 1..64, 65..128, 129..192 bases, LDS-resident
// beyond), each with its own ticket counter.  With auxiliary streams the classes run concurrently, so
// the tail of one overlaps the body of the next.
int launch_bsw(bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, const uint8_t *qer, int w, const SwParams &prm, int qmax,
               DevCounters *ctr, int cu_count, hipStream_t st, hipStream_t *aux, hipEvent_t fork, hipEvent_t *join) {
    bsw_reset_kernel<<<1, 1, 0, st>>>(ctr);
    if (n <= 0) return 0;
    int64_t blocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t maxb = (int64_t)cu_count * 8;
    if (blocks > maxb) blocks = maxb;
    const int n_class = qmax > 192 ? 4 : qmax > 128 ? 3 : qmax > 64 ? 2 : 1;
    hipStream_t q[4] = {st, st, st, st};
    if (aux && n_class > 1) {
        if (hipEventRecord(fork, st) != hipSuccess) return -1;
        for (int c = 1; c < n_class; ++c) {
            q[c] = aux[c - 1];
            if (hipStreamWaitEvent(q[c], fork, 0) != hipSuccess) return -1;
        }
    }
    // the longest queries first: their tasks are the most expensive
    if (n_class > 3) {
        // (h, e) row + query of one task per wave in LDS: fewer waves per block for very long queries
        const size_t per_wave = (((size_t)(qmax + 1) * 8 + (size_t)qmax + 64 + 15) / 16) * 16;
        int waves = (int)((size_t)160 * 1024 / per_wave);
        if (waves < 1) return -2;                  // a query of more than ~18 k bases does not fit a CU's LDS
        waves = waves < kWavesPerBlock ? waves : kWavesPerBlock;
        const size_t lds = per_wave * (size_t)waves;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bsw_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int64_t lblocks = (n + waves - 1) / waves;
        if (lblocks > maxb) lblocks = maxb;
        bsw_kernel<<<(unsigned)lblocks, waves * 64, lds, q[3]>>>(pairs, n, ref, qer, w, prm, qmax, 192, ctr, &ctr->bsw_head[3]);
    }
    if (n_class > 2) bsw_kernel_reg<3><<<(unsigned)blocks, kWavesPerBlock * 64, 0, q[2]>>>(pairs, n, ref, qer, w, prm, 128, ctr, &ctr->bsw_head[2]);
    if (n_class > 1) bsw_kernel_reg<2><<<(unsigned)blocks, kWavesPerBlock * 64, 0, q[1]>>>(pairs, n, ref, qer, w, prm, 64, ctr, &ctr->bsw_head[1]);
    bsw_kernel_reg<1><<<(unsigned)blocks, kWavesPerBlock * 64, 0, q[0]>>>(pairs, n, ref, qer, w, prm, -1, ctr, &ctr->bsw_head[0]);
    if (aux && n_class > 1)
        for (int c = 1; c < n_class; ++c) {
            if (hipEventRecord(join[c - 1], q[c]) != hipSuccess) return -1;
            if (hipStreamWaitEvent(st, join[c - 1], 0) != hipSuccess) return -1;
        }
    return 0;
}

size_t bsw_lds_bytes(int qmax) {
    const size_t per_wave = (((size_t)(qmax + 1) * 8 + (size_t)qmax + 64 + 15) / 16) * 16;
    return per_wave * kWavesPerBlock;
}

}  // namespace bwams
