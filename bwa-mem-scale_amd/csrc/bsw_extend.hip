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
    const uint8_t *__restrict__ qer, int w0, SwParams prm, int qmax, int qlo, DevCounters *ctr, unsigned long long *head,
    const int32_t *__restrict__ list, const unsigned long long *n_list) {
    extern __shared__ __align__(16) unsigned char lds[];
    n = (int64_t)*n_list;                              // this kernel walks the list of tasks left to the one-task-per-wave kernels
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
        const int64_t cur = list[pid++];

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

// ---- eight tasks per wavefront -------------------------------------------------------------
// bsw_kernel above spends one wavefront on one task, with the row's band bookkeeping (beg, end, row maximum and its
// column, z-drop, ...) in scalar registers: measured, a row of <= 128 cells cost ~110 vector but ~250 scalar
// instructions, and the one scalar unit of a CU serves four SIMDs (profiles/r01_notes.md 26-28) — the scalar pipe,
// not the vector one, bounded the one-task-per-wave kernels.  In bsw_qwin_kernel a task owns kBswLpt = 8 lanes (half a
// DPP row), so
//   * eight tasks share a wavefront and every cross-lane step stays inside a DPP row (quad_perm scans + one ds_swizzle for
//     the max-plus prefix of F, row_shr for the H shift, quad_perm / row_half_mirror butterflies for the four per-row
//     reductions);
//   * the bookkeeping lives in vector registers, replicated over the task's lanes: no scalar work per row at all;
//   * a task slot that finishes takes the next task from the wave's reservation while the others carry on.
// Tasks are binned by query length first (bsw_classify_kernel) — the class fixes the LDS per task — one launch per
// class, concurrently.  Results are bit-identical to scalarBandedSWA: the recurrences and every per-row decision are
// those of bsw_kernel above.  Measured on the bench workload (5.9 M tasks, 13.4 G cells per step): one task per wave
// 89 ms; 16 lanes x all columns in registers 83; 16 x 3-column window 80; 8 x 4 66; 8 x 3 71; 8 x 6 72; 4 x 8 73 (LDS
// and registers cost occupancy); one task per LANE with its row in LDS 170 (the row caps a CU at four waves).
constexpr int kQuadCpl[5] = {2, 4, 6, 9, 12};                 // a class holds queries of <= 16 * kQuadCpl - 1 bases (16 * kQuadCpl columns of LDS)
constexpr int kBswLpt = 8, kBswWin = 4;                       // lanes per task, window columns per lane (bsw_qwin_kernel)
constexpr int kNumBswClass = 6;                              // five quad classes + the rest (one task per wave, LDS)
constexpr int kQuadChunk = 16;                               // tasks a wave reserves per atomic

__host__ __device__ __forceinline__ int bsw_class_of(int qlen, int h0, int max_sc) {
    // a column's row state is one LDS word (H and E in 14 bits each, the query base on top): needs scores < 2^14
    const long long top = (long long)h0 + (long long)qlen * max_sc;       // no score of the task can exceed it
    if (qlen > 16 * kQuadCpl[4] - 1 || top >= (1 << 14) || h0 < 0) return 5;
    return qlen <= 16 * kQuadCpl[0] - 1 ? 0 : qlen <= 16 * kQuadCpl[1] - 1 ? 1 : qlen <= 16 * kQuadCpl[2] - 1 ? 2
           : qlen <= 16 * kQuadCpl[3] - 1 ? 3 : 4;
}

// list[c * n + k] = k-th task of class c (any order: results go back by task index).  Appends are aggregated per block of
// 1024 tasks: waves count into LDS, one global atomic per class and block (one per class and WAVE was 270 k atomics on six
// addresses per launch: 1.9 ms of a kernel that reads 160 MB).
__global__ __launch_bounds__(1024) void bsw_classify_kernel(const bwams_seqpair_t *__restrict__ pairs, int64_t n, int max_sc, int32_t *__restrict__ list,
                                                            unsigned long long *cnt) {
    __shared__ unsigned int l_cnt[kNumBswClass];
    __shared__ unsigned long long l_base[kNumBswClass];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < kNumBswClass) l_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int cls = t < n ? bsw_class_of(pairs[t].len2, pairs[t].h0, max_sc) : -1;
    unsigned int my_off = 0;                       // offset of this lane's task inside the block's share of its class
#pragma unroll
    for (int c = 0; c < kNumBswClass; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            unsigned int wbase = 0;
            if (lane == leader) wbase = atomicAdd(&l_cnt[c], (unsigned int)__popcll(m));
            wbase = (unsigned int)__shfl((int)wbase, leader);
            if (cls == c) my_off = wbase + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumBswClass && l_cnt[threadIdx.x]) l_base[threadIdx.x] = atomicAdd(&cnt[threadIdx.x], (unsigned long long)l_cnt[threadIdx.x]);
    __syncthreads();
    if (cls >= 0) list[(int64_t)cls * n + (int64_t)l_base[cls] + my_off] = (int32_t)t;
}

// row-local (16-lane) DPP helpers.  A VALU write followed by a DPP read of the same register needs two wait states.
__device__ __forceinline__ int row_scan_max(int v) {            // inclusive prefix max over the lanes of a DPP row
    asm("s_nop 4\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int row_all_max(int v) {             // maximum over the 16 lanes of a DPP row, in every lane
    asm("s_nop 4\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int row_shr1(int v, int fill) { return dppi<0x111, 0xF, 0xF>(fill, v); }   // lane g gets lane g - 1's v; lane 0 the fill

// ---- row state in LDS, a register window that follows the band ----------------------------------------------------
// A kernel whose cost follows the query length computes mostly dead columns: the band of a typical extension is
// narrow: ~23 live columns per row on the bench workload, because it shrinks to the non-zero span of the previous row.
// Here the row state eh[] of a task lives in LDS (one word per column: H | E << 14 | the query base << 28), and each row loads just the
// live columns [beg, end) into a window of kWin columns per lane (LPT * kWin per task: 32 as launched), computes them (F as a prefix
// maximum: a lane's own columns sequentially, then four DPP steps over the lanes), and stores them back; a band wider than the window takes further passes with the prefix maximum and
// the last H carried over.  Columns outside the band keep their stale values in LDS, as scalarBandedSWA's eh[] does.
// cross-lane steps over a task's LPT lanes (16 = one DPP row, 8 = half a row)
template <int LPT> __device__ __forceinline__ int grp_scan_max(int v, int g);
template <> __device__ __forceinline__ __attribute__((unused)) int grp_scan_max<16>(int v, int) { return row_scan_max(v); }
template <> __device__ __forceinline__ __attribute__((unused)) int grp_scan_max<8>(int v, int g) {
    // inclusive scan inside each quad (two quad_perm steps), then lanes 4..7 take the first quad's total (lane 3 of the
    // half row, fetched with ds_swizzle: source = (lane & 0b11000) | 0b00011)
    asm("s_nop 4\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    const int t = __builtin_amdgcn_ds_swizzle(v, 0x78);
    return (g & 4) ? (v > t ? v : t) : v;
}
template <int LPT> __device__ __forceinline__ int grp_all_max(int v);
template <> __device__ __forceinline__ __attribute__((unused)) int grp_all_max<16>(int v) { return row_all_max(v); }
template <> __device__ __forceinline__ __attribute__((unused)) int grp_all_max<8>(int v) {
    asm("s_nop 4\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
// lane g of the group gets lane g - 1's v, lane 0 the fill
template <int LPT> __device__ __forceinline__ int grp_shr1(int v, int fill, int g) {
    const int t = row_shr1(v, fill);
    return (LPT < 16 && g == 0) ? fill : t;
}


// __any() goes through a 0 / 1 VGPR and a compare; the ballot itself is a scalar AND with exec
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

template <int LPT, int kWin>
__global__ __launch_bounds__(kWavesPerBlock * 64) void bsw_qwin_kernel(
    bwams_seqpair_t *__restrict__ pairs, const int32_t *__restrict__ list, const unsigned long long *n_list_p,
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ qer, int w0, SwParams prm, DevCounters *ctr, unsigned long long *head,
    int cols) {
    extern __shared__ uint32_t qwin_lds[];                         // [wave][task slot][cols]: H (14 bits) | E << 14 | query base << 28
    constexpr int TPW = 64 / LPT;                                  // tasks per wavefront
    const int lane = threadIdx.x & 63, g = lane & (LPT - 1), q = lane / LPT;
    uint32_t *const row_eh = qwin_lds + (size_t)(((threadIdx.x >> 6) * TPW + q) * cols);
    const int o_del = prm.o_del, e_del = prm.e_del, o_ins = prm.o_ins, e_ins = prm.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    // score rows of the matrix, one per target base: {four scores packed to bytes, the score against N}; a lane picks its task's row
    // with one LDS read per DP row instead of a chain of selects
    __shared__ uint2 pk_tab[5];
    if (threadIdx.x < 5) {
        const int t = threadIdx.x;
        pk_tab[t] = make_uint2(((uint32_t)(uint8_t)prm.mat[t * 5 + 0]) | ((uint32_t)(uint8_t)prm.mat[t * 5 + 1] << 8) |
                               ((uint32_t)(uint8_t)prm.mat[t * 5 + 2] << 16) | ((uint32_t)(uint8_t)prm.mat[t * 5 + 3] << 24),
                               (uint32_t)(int)prm.mat[t * 5 + 4]);
    }
    __syncthreads();
    const int64_t n_list = (int64_t)*n_list_p;
    const unsigned long long kLeaders = LPT == 16 ? 0x0001000100010001ull : 0x0101010101010101ull;   // lane 0 of every task slot
    int64_t pid = 0, pid_end = 0;
    bool exhausted = false;
    bool alive = false;
    int cur = 0, qlen = 0, tlen = 0, h0 = 0, w = 0, i = 0, beg = 0, end = 0;
    int mx = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, tb_next = 4;
    const uint8_t *tr = ref;
    unsigned long long cells = 0;

    for (;;) {
        const unsigned long long need_m = __ballot(!alive);
        if (need_m) {
            if (pid >= pid_end && !exhausted) {
                pid = (int64_t)wave_ticket(head, (unsigned long long)kQuadChunk);
                pid_end = pid + kQuadChunk < n_list ? pid + kQuadChunk : n_list;
                if (pid >= n_list) { exhausted = true; pid_end = pid; }
            }
            const int avail = (int)(pid_end - pid);
            const int nq = __popcll(need_m & kLeaders);
            const int rank = __popcll(need_m & kLeaders & ((1ull << (q * LPT)) - 1ull));
            if (!alive && rank < avail) {
                cur = list[pid + rank];
                const bwams_seqpair_t sp = pairs[cur];
                qlen = sp.len2; tlen = sp.len1; h0 = sp.h0;
                const uint8_t *tq = qer + sp.idq;
                tr = ref + sp.idr;
                for (int c = g; c <= qlen; c += LPT) {            // row -1 of the DP and the query, LPT columns at a time
                    int h = h0;
                    if (c >= 1) { h = h0 - oe_ins - (c - 1) * e_ins; h = h > 0 ? h : 0; }
                    uint32_t qb = c < qlen ? (uint32_t)tq[c] : 4u;
                    qb = qb > 4u ? 4u : qb;
                    row_eh[c] = (uint32_t)h | (qb << 28);
                }
                w = w0;
                {
                    int max_ins = (int)((double)(qlen * prm.max_sc + prm.end_bonus - o_ins) / e_ins + 1.);
                    max_ins = max_ins > 1 ? max_ins : 1;
                    w = w < max_ins ? w : max_ins;
                    int max_del = (int)((double)(qlen * prm.max_sc + prm.end_bonus - o_del) / e_del + 1.);
                    max_del = max_del > 1 ? max_del : 1;
                    w = w < max_del ? w : max_del;
                }
                mx = h0; max_i = -1; max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
                beg = 0; end = qlen; i = 0;
                alive = tlen > 0;
                if (!alive && g == 0) {
                    bwams_seqpair_t *o = &pairs[cur];
                    o->score = mx; o->qle = 0; o->tle = 0; o->gtle = 0; o->gscore = -1; o->max_off = 0;
                }
                tb_next = alive ? (int)tr[0] : 4;
            }
            pid += nq < avail ? nq : avail;
            if (exhausted && !wave_any(alive)) break;
        }

        // ---- one row of every live task
        int tb = tb_next;
        tb = tb > 4 ? 4 : tb;
        if (alive && i + 1 < tlen) tb_next = tr[i + 1];
        const uint2 pp = pk_tab[tb];
        const int pkt = (int)pp.x, pnt = (int)pp.y;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1 = 0;
        if (beg == 0) {
            h1 = h0 - (o_del + e_del * (i + 1));
            h1 = h1 < 0 ? 0 : h1;
        }
        const bool row = alive && beg < end;
        int key = -1, first_nz = 1 << 20, last_nz = -1, hlast = -1;
        int c_max = NEG, c_h = h1;                                  // carried into a further pass: prefix maximum, last H
        for (int base = beg; wave_any(row && base < end); base += LPT * kWin) {
            const bool in = row && base < end;
            const int jb = base + g * kWin;
            int Mv[kWin], Pl[kWin], Ev[kWin];
            uint32_t Qb[kWin];
            int run = NEG;
#pragma unroll
            for (int c = 0; c < kWin; ++c) {
                const int j = jb + c;
                const bool act = in && j < end;
                uint32_t wd = 0u;
                if (act) wd = row_eh[j];
                const int hd = (int)(wd & 0x3fffu), e = (int)((wd >> 14) & 0x3fffu);
                const uint32_t qb = wd >> 28;
                // byte qb of {pnt : pkt} (v_perm_b32: selector 0-3 = bytes of pkt, 4 = the low byte of pnt), sign-extended
                const int S = __builtin_amdgcn_sbfe((int)__builtin_amdgcn_perm((uint32_t)pnt, (uint32_t)pkt, qb), 0u, 8u);
                const int M = hd ? hd + S : 0;                      // a column outside [beg, end) read wd = 0: M = 0 there
                int tj = M - oe_ins;
                tj = tj > 0 ? tj : 0;
                // columns >= end lie to the right of every live one and nothing of theirs is stored or carried (a further pass
                // exists only when all of this pass's columns are live): their term may enter the running maximum
                const int x = tj + j * e_ins;
                run = run > x ? run : x;
                Mv[c] = M; Pl[c] = run; Ev[c] = e; Qb[c] = wd;
            }
            const int scan = grp_scan_max<LPT>(run, g);
            int Lex = grp_shr1<LPT>(scan, NEG, g);
            Lex = Lex > c_max ? Lex : c_max;
            int Hh[kWin], E2[kWin];
#pragma unroll
            for (int c = 0; c < kWin; ++c) {
                const int j = jb + c;
                int Pex = c ? Pl[c - 1] : NEG;
                Pex = Pex > Lex ? Pex : Lex;
                int F = Pex - (j - 1) * e_ins;
                F = F > 0 ? F : 0;
                const int e = Ev[c];
                int h = Mv[c] > e ? Mv[c] : e;
                h = h > F ? h : F;
                int e2 = Mv[c] - oe_del;
                e2 = e2 > 0 ? e2 : 0;
                const int e1 = e - e_del;
                e2 = e2 > e1 ? e2 : e1;
                Hh[c] = h; E2[c] = e2;
            }
            const int h_in = grp_shr1<LPT>(Hh[kWin - 1], c_h, g);
#pragma unroll
            for (int c = 0; c < kWin; ++c) {
                const int j = jb + c;
                const bool act = in && j < end;
                // column beg is lane 0's first column of the first pass, where h_in = c_h = h1 already: no select for it
                const int hl = c ? Hh[c - 1] : h_in;
                if (act) {
                    const uint32_t he = (uint32_t)hl | ((uint32_t)E2[c] << 14);
                    row_eh[j] = (he & 0x0fffffffu) | (Qb[c] & ~0x0fffffffu);      // v_bfi_b32: the query base stays where it is
                    const int k = (Hh[c] << 8) | j;
                    key = key > k ? key : k;
                    if (hl != 0 || E2[c] != 0) { first_nz = first_nz < j ? first_nz : j; last_nz = j; }
                }
            }
            {                                                       // H of column end - 1, once per pass
                const int cl = end - 1 - jb;
                if (in && cl >= 0 && cl < kWin) {
                    int hv = Hh[0];
#pragma unroll
                    for (int c = 1; c < kWin; ++c) hv = cl == c ? Hh[c] : hv;
                    hlast = hv;
                }
            }
            if (wave_any(row && base + LPT * kWin < end)) {            // a further pass: carry the prefix maximum and the last column's H
                const int pm = grp_all_max<LPT>(scan);
                c_max = c_max > pm ? c_max : pm;
                c_h = grp_all_max<LPT>(g == LPT - 1 ? Hh[kWin - 1] : -1);
            }
        }
        key = grp_all_max<LPT>(key);
        first_nz = -grp_all_max<LPT>(-first_nz);
        last_nz = grp_all_max<LPT>(last_nz);
        hlast = grp_all_max<LPT>(hlast);
        const int m = row ? key >> 8 : 0, mj = row ? key & 0xff : -1;
        const int h1f = row ? hlast : h1;
        if (alive) {
            if (g == 0) {                                           // eh[end] = {h1f, 0}
                row_eh[end] = (uint32_t)h1f | (row_eh[end] & 0xf0000000u);
            }
            if (row) cells += (unsigned long long)(g == 0 ? end - beg : 0);
            const int j_exit = row ? end : beg;
            if (j_exit == qlen) {
                max_ie = gscore > h1f ? max_ie : i;
                gscore = gscore > h1f ? gscore : h1f;
            }
            bool fin = m == 0;
            if (!fin) {
                if (m > mx) {
                    mx = m; max_i = i; max_j = mj;
                    int d = mj - i;
                    d = d < 0 ? -d : d;
                    max_off = max_off > d ? max_off : d;
                } else if (prm.zdrop > 0) {
                    if (i - max_i > mj - max_j) fin = mx - m - ((i - max_i) - (mj - max_j)) * e_del > prm.zdrop;
                    else fin = mx - m - ((mj - max_j) - (i - max_i)) * e_ins > prm.zdrop;
                }
            }
            if (!fin) {
                const int nbeg = first_nz < end ? first_nz : end;
                int jj;
                if (h1f != 0) jj = end;
                else if (last_nz >= nbeg) jj = last_nz;
                else jj = nbeg - 1;
                beg = nbeg;
                end = jj + 2 < qlen ? jj + 2 : qlen;
                ++i;
                fin = i >= tlen;
            }
            if (fin) {
                if (g == 0) {
                    bwams_seqpair_t *o = &pairs[cur];
                    o->score = mx; o->qle = max_j + 1; o->tle = max_i + 1; o->gtle = max_ie + 1; o->gscore = gscore; o->max_off = max_off;
                }
                alive = false;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) cells += ((unsigned long long)__shfl_down((unsigned)(cells >> 32), o) << 32) | (unsigned)__shfl_down((unsigned)cells, o);
    if (lane == 0 && cells) atomicAdd(&ctr->bsw_cells, cells);
}


__global__ void bsw_reset_kernel(DevCounters *ctr) {
    for (int i = 0; i < 4; ++i) ctr->bsw_head[i] = 0;      // bsw_cells accumulates until the caller clears it
    for (int i = 0; i < kNumBswClass; ++i) ctr->bsw_cls_cnt[i] = ctr->bsw_cls_head[i] = 0;
}

}  // namespace

// Tasks are binned (bsw_classify_kernel) by query length into five eight-tasks-per-wave launches and, for what is left
// (queries beyond 191 bases, scores beyond 2^14), the one-task-per-wave LDS kernel.  Every launch has its own ticket counter; with auxiliary
// streams they run concurrently, the classes of the longest queries first.  `list` holds kNumBswClass * n task indices.
int launch_bsw(bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, const uint8_t *qer, int w, const SwParams &prm, int qmax,
               DevCounters *ctr, int cu_count, hipStream_t st, int32_t *list, hipStream_t *aux, hipEvent_t fork, hipEvent_t *join) {
    bsw_reset_kernel<<<1, 1, 0, st>>>(ctr);
    if (n <= 0) return 0;
    bsw_classify_kernel<<<(unsigned)((n + 1023) / 1024), 1024, 0, st>>>(pairs, n, prm.max_sc, list, ctr->bsw_cls_cnt);
    int64_t blocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t maxb = (int64_t)cu_count * 8;
    if (blocks > maxb) blocks = maxb;
    hipStream_t q[5] = {st, st, st, st, st};
    const int n_aux = aux ? 4 : 0;
    if (n_aux) {
        if (hipEventRecord(fork, st) != hipSuccess) return -1;
        for (int c = 0; c < n_aux; ++c) {
            q[c + 1] = aux[c];
            if (hipStreamWaitEvent(q[c + 1], fork, 0) != hipSuccess) return -1;
        }
    }
    const unsigned B = (unsigned)blocks, T = kWavesPerBlock * 64;
    unsigned long long *cnt = ctr->bsw_cls_cnt, *hd = ctr->bsw_cls_head;
    // per launch: the attribute belongs to the current device (a batch on a second GPU of the process needs it too)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(bsw_qwin_kernel<kBswLpt, kBswWin>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)kWavesPerBlock * (64 / kBswLpt) * 192 * 4)) != hipSuccess) return -1;
    // every class on a stream of its own, the classes of the longest queries first.  (With two classes per stream and the
    // one-task-per-wave kernel — usually without a single task, but 2048 blocks that wait for a free CU slot — in front of one
    // of them, the kernel trace showed two class launches starting 14 ms late.)
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 192 * 4, q[4]>>>(pairs, list + 4 * n, cnt + 4, ref, qer, w, prm, ctr, hd + 4, 192);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 144 * 4, q[3]>>>(pairs, list + 3 * n, cnt + 3, ref, qer, w, prm, ctr, hd + 3, 144);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 96 * 4, q[2]>>>(pairs, list + 2 * n, cnt + 2, ref, qer, w, prm, ctr, hd + 2, 96);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 64 * 4, q[1]>>>(pairs, list + 1 * n, cnt + 1, ref, qer, w, prm, ctr, hd + 1, 64);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 32 * 4, q[0]>>>(pairs, list + 0 * n, cnt + 0, ref, qer, w, prm, ctr, hd + 0, 32);
    {
        // what neither packed form can take: (h, e) row + query of one task per wave in LDS, fewer waves per block for very long queries.
        // Last on the main stream, persistent waves on a small grid: the class is usually empty.
        const size_t per_wave = (((size_t)(qmax + 1) * 8 + (size_t)qmax + 64 + 15) / 16) * 16;
        int waves = (int)((size_t)160 * 1024 / per_wave);
        if (waves < 1) return -2;                  // a query of more than ~18 k bases does not fit a CU's LDS
        waves = waves < kWavesPerBlock ? waves : kWavesPerBlock;
        const size_t lds = per_wave * (size_t)waves;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bsw_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int64_t lblocks = (n + waves - 1) / waves;
        const int64_t lcap = qmax > 16 * kQuadCpl[4] - 1 ? maxb : (int64_t)cu_count * 2;     // long queries: this IS the main class
        if (lblocks > lcap) lblocks = lcap;
        bsw_kernel<<<(unsigned)lblocks, waves * 64, lds, q[0]>>>(pairs, n, ref, qer, w, prm, qmax, -1, ctr, &ctr->bsw_head[3], list + 5 * n, cnt + 5);
    }
    if (n_aux)
        for (int c = 0; c < n_aux; ++c) {
            if (hipEventRecord(join[c], q[c + 1]) != hipSuccess) return -1;
            if (hipStreamWaitEvent(st, join[c], 0) != hipSuccess) return -1;
        }
    return 0;
}
size_t bsw_list_bytes(int64_t n) { return (size_t)kNumBswClass * (size_t)(n > 0 ? n : 1) * sizeof(int32_t); }

size_t bsw_lds_bytes(int qmax) {
    const size_t per_wave = (((size_t)(qmax + 1) * 8 + (size_t)qmax + 64 + 15) / 16) * 16;
    return per_wave * kWavesPerBlock;
}

}  // namespace bwams
