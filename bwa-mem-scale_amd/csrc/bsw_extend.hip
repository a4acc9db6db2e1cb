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
    const int32_t *__restrict__ list, const unsigned long long *n_list, const int64_t *__restrict__ src, int dir) {
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
        // src != nullptr: in place — the task's sequences start at src[2 id] (query) / src[2 id + 1] (target) and run in direction dir
        const uint8_t *tq = qer + (src ? src[2 * (int64_t)sp.id] : (int64_t)sp.idq);
        const uint8_t *tr = ref + (src ? src[2 * (int64_t)sp.id + 1] : (int64_t)sp.idr);

        // row -1 of the DP and the query, each column on its owner lane
        for (int j = lane; j <= qlen; j += 64) {
            int h = h0;
            if (j >= 1) {
                h = h0 - oe_ins - (j - 1) * e_ins;
                h = h > 0 ? h : 0;
            }
            eh[j] = make_int2(h, 0);
            if (j < qlen) qs[j] = tq[j * dir];
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
            if (i + 1 < tlen) tb_next = tr[(i + 1) * dir];           // overlap the next row's load with this row
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
    int cols, const int64_t *__restrict__ src, int dir) {
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
                const uint8_t *tq = qer + (src ? src[2 * (int64_t)sp.id] : (int64_t)sp.idq);
                tr = ref + (src ? src[2 * (int64_t)sp.id + 1] : (int64_t)sp.idr);
                for (int c = g; c <= qlen; c += LPT) {            // row -1 of the DP and the query, LPT columns at a time
                    int h = h0;
                    if (c >= 1) { h = h0 - oe_ins - (c - 1) * e_ins; h = h > 0 ? h : 0; }
                    uint32_t qb = c < qlen ? (uint32_t)tq[c * dir] : 4u;
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
        if (alive && i + 1 < tlen) tb_next = tr[(i + 1) * dir];
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


// ---- sixteen tasks per wavefront, packed 16-bit columns ---------------------------------------------------------------
// bsw_qwin_kernel spends ≈ 290 vector instructions on a row of eight tasks: 177 in the four-columns-per-lane pass, 114 in the
// row's bookkeeping, all of it 32-bit — although every score of these classes is below 2^14 and the packed 16-bit forms
// (v_pk_add / sub / max / min / mul_lo) issue at the same rate (tools/ubench_valu.hip).  Here a task owns FOUR lanes (one quad:
// every cross-lane step is a quad_perm DPP) and a lane owns EIGHT columns as four packed registers, so a wavefront carries sixteen
// tasks: the bookkeeping is shared by twice as many tasks and the pass handles two columns per instruction, with the register
// footprint of the 8 x 4 form (the 32-bit 4 x 8 form had lost to it on occupancy, note 32).
//   * Row state in LDS per task: three arrays of column PAIRS — H (H(i, j-1) as scalarBandedSWA's eh[].h), E, and the v_perm
//     selector that picks the two columns' scores out of the target base's score row — read and written 16 bytes per lane.
//   * The window of a pass starts at a multiple of 8 columns, so a lane's eight columns are one aligned 16-byte access; the
//     columns of the window outside [beg, end) are masked (their state reads as zero and is written back unchanged).
//   * F is the same max-plus prefix scan: inside the lane over its eight columns (packed, the carry broadcast with v_perm),
//     over the four lanes with two quad_perm steps.
//   * The row maximum and the last column attaining it travel as one 32-bit key (H << 8 | j) per column through v_max3.
// Recurrences and per-row decisions are those of bsw_qwin_kernel; results are bit-identical to scalarBandedSWA.
// Eligible: scores below 2^14 (the class condition), every score against N equal to -1 (the selector's 0xff bytes), gap
// extension costs small enough for j * e_ins to stay inside 16 bits.  Other scoring schemes keep the 32-bit kernel.
namespace pk {
__device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t sub(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t max(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t max0(uint32_t a) { uint32_t d; asm("v_pk_max_i16 %0, %1, 0" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t minu(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t hi2(uint32_t a) { return __builtin_amdgcn_perm(a, a, 0x03020302u); }      // the high half in both halves
__device__ __forceinline__ uint32_t lo2(uint32_t a) { return __builtin_amdgcn_perm(a, a, 0x01000100u); }      // the low half in both halves
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }   // v_bfi_b32
template <int CTRL> __device__ __forceinline__ int qdpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int quad_all_max(int v) {
    int t = qdpp<0xB1>(v);                       // quad_perm [1,0,3,2]
    v = v > t ? v : t;
    t = qdpp<0x4E>(v);                           // quad_perm [2,3,0,1]
    return v > t ? v : t;
}
}  // namespace pk

constexpr int kPkNeg = -16384;
#ifndef BWAMS_PK_WAVES
#define BWAMS_PK_WAVES 2
#endif
constexpr int kPkWaves = BWAMS_PK_WAVES;      // wavefronts per workgroup: LDS is handed out per workgroup, small ones pack a CU better

__host__ __device__ __forceinline__ bool bsw_pk_eligible(const SwParams &prm) {
    for (int i = 0; i < 5; ++i)
        if (prm.mat[i * 5 + 4] != -1 || prm.mat[4 * 5 + i] != -1) return false;
    return prm.e_ins >= 0 && prm.e_ins <= 64 && prm.e_del >= 0 && prm.e_del <= 64 && prm.o_ins >= 0 && prm.o_ins + prm.e_ins < 8000 &&
           prm.o_del >= 0 && prm.o_del + prm.e_del < 8000;
}

__global__ __launch_bounds__(kPkWaves * 64) void bsw_pk_kernel(
    bwams_seqpair_t *__restrict__ pairs, const int32_t *__restrict__ list, const unsigned long long *n_list_p,
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ qer, int w0, SwParams prm, DevCounters *ctr, unsigned long long *head,
    int cols, const int64_t *__restrict__ src, int dir) {
    // 16 words of score rows, then [wave][task slot][H pairs | E pairs | selector bytes]: 4 + 4 + 2 bytes per pair.  Everything is carved out
    // of the dynamic region: a static __shared__ array in front of it would shift its base off 16 bytes (16-byte DS accesses
    // off their alignment are replayed at 64 cycles each)
    extern __shared__ __align__(16) uint32_t pk_lds[];
    constexpr int LPT = 4, TPW = 16;
    const int lane = threadIdx.x & 63, g = lane & (LPT - 1), q = lane / LPT;
    const int P = cols >> 1;
    // per task 10 P + 24 bytes (P = cols / 2 pairs, a multiple of 4): H and E each P + 2 words, the selector bytes P / 2 + 1 words, and one
    // word to keep the next task 8-byte aligned.  The two spare words let a lane whose eight columns begin at cols - 4 read and write
    // back (masked) past the last pair without touching a neighbour
    uint32_t *const t_hp = pk_lds + 16 + (size_t)((threadIdx.x >> 6) * TPW + q) * (size_t)((5 * P) / 2 + 6);
    uint32_t *const t_ep = t_hp + P + 2;
    uint16_t *const t_sel = reinterpret_cast<uint16_t *>(t_ep + P + 2);   // per pair the selectors' low bytes {2 qb0, 2 qb1} (0x0d for N)
    uint16_t *const t_hp16 = reinterpret_cast<uint16_t *>(t_hp), *const t_ep16 = reinterpret_cast<uint16_t *>(t_ep);
    const int o_del = prm.o_del, e_del = prm.e_del, o_ins = prm.o_ins, e_ins = prm.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    // the score row of a target base as four 16-bit scores {A, C | G, T}; every score against N is -1 = the selector's 0xff bytes
    uint2 *const pk_tab16 = reinterpret_cast<uint2 *>(pk_lds);
    if (threadIdx.x < 5) {
        const int t = threadIdx.x;
        pk_tab16[t] = make_uint2((uint32_t)(uint16_t)(int16_t)prm.mat[t * 5 + 0] | ((uint32_t)(uint16_t)(int16_t)prm.mat[t * 5 + 1] << 16),
                                 (uint32_t)(uint16_t)(int16_t)prm.mat[t * 5 + 2] | ((uint32_t)(uint16_t)(int16_t)prm.mat[t * 5 + 3] << 16));
    }
    __syncthreads();
    const uint32_t ONE = 0x00010001u, TWO = 0x00020002u, C257 = 0x01010101u;
    const uint32_t OEI = (uint32_t)oe_ins * 0x10001u, OED = (uint32_t)oe_del * 0x10001u, EDX = (uint32_t)e_del * 0x10001u;
    const uint32_t E1X = (uint32_t)e_ins * 0x10001u, E2X = (uint32_t)(2 * e_ins) * 0x10001u;
    const uint32_t NEGP = (uint32_t)(uint16_t)(int16_t)kPkNeg * 0x10001u;
    const int64_t n_list = (int64_t)*n_list_p;
    const unsigned long long kLeaders = 0x1111111111111111ull;    // lane 0 of every task slot
    int64_t pid = 0, pid_end = 0;
    bool exhausted = false;
    bool alive = false;
    int cur = 0, qlen = 0, tlen = 0, h0 = 0, w = 0, i = 0, beg = 0, end = 0;
    int mx = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, tb_next = 4;
    const uint8_t *tr = ref;
    unsigned long long cells = 0;
    // the row in progress (a row may take several iterations: one per window of 32 columns)
    bool start_row = true, row = false;
    int base = 0, h1 = 0, key = -1, hlast = -1, c_max = kPkNeg, c_h = 0;
    uint32_t nzl = 0u, nzf = 0u;
    uint2 tt = make_uint2(0u, 0u);

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
                const uint8_t *tq = qer + (src ? src[2 * (int64_t)sp.id] : (int64_t)sp.idq);
                tr = ref + (src ? src[2 * (int64_t)sp.id + 1] : (int64_t)sp.idr);
                start_row = true;
                for (int p = g; p <= (qlen >> 1); p += LPT) {      // row -1 of the DP and the query, a pair of columns per lane and step
                    uint32_t hh = 0, ss = 0;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int c = 2 * p + k;
                        int h = 0;
                        if (c <= qlen) { h = h0; if (c >= 1) { h = h0 - oe_ins - (c - 1) * e_ins; h = h > 0 ? h : 0; } }
                        uint32_t qb = c < qlen ? (uint32_t)tq[c * dir] : 4u;
                        const uint32_t sel = qb < 4u ? 2u * qb : 0x0du;
                        hh |= (uint32_t)h << (16 * k);
                        ss |= sel << (8 * k);
                    }
                    t_hp[p] = hh; t_ep[p] = 0u; t_sel[p] = (uint16_t)ss;
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

        // ---- one PASS (a window of 32 columns) of the current row of every live task.  The tasks of a wavefront do not wait for
        // each other at the end of a row: a task whose row needs a second window takes it in the next iteration while the others
        // are already on their next rows, so an iteration costs the same whether one task or all sixteen have a wide band.
        if (alive && start_row) {                                     // row prologue
            int tb = tb_next;
            tb = tb > 4 ? 4 : tb;
            if (i + 1 < tlen) tb_next = tr[(i + 1) * dir];
            tt = pk_tab16[tb];
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            h1 = 0;
            if (beg == 0) {
                h1 = h0 - (o_del + e_del * (i + 1));
                h1 = h1 < 0 ? 0 : h1;
            }
            row = beg < end;
            key = -1; hlast = -1; nzl = 0u; nzf = 0u;
            c_max = kPkNeg; c_h = 0;
            base = beg & ~3;
            start_row = false;
        }
        {
            const bool in = alive && row;
            const int jb = base + g * 8;
            const bool inr = in && jb < cols;
            // the lane's live columns as a bit mask, then as four pair masks
            int lo = beg - jb, hi = end - jb;
            lo = lo < 0 ? 0 : (lo > 8 ? 8 : lo);
            hi = hi < 0 ? 0 : (hi > 8 ? 8 : hi);
            const uint32_t m8 = inr ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
            // the window starts at a multiple of FOUR columns (a band of the typical 23 columns then fits one window almost always;
            // at a multiple of eight a fifth of the rows needed a second one): the lane's eight columns are 8-byte aligned in the H and E
            // arrays, 4-byte aligned in the selector bytes.  Lanes without a live column read too (whatever lies there: it is masked)
            const int pj = inr ? (jb >> 1) : 0;
            const uint2 hpa = *reinterpret_cast<const uint2 *>(t_hp + pj), hpb = *reinterpret_cast<const uint2 *>(t_hp + pj + 2);
            const uint2 epa = *reinterpret_cast<const uint2 *>(t_ep + pj), epb = *reinterpret_cast<const uint2 *>(t_ep + pj + 2);
            uint2 sl2;
            sl2.x = *reinterpret_cast<const uint32_t *>(t_sel + pj);
            sl2.y = *reinterpret_cast<const uint32_t *>(t_sel + pj + 2);
            const uint32_t hpv[4] = {hpa.x, hpa.y, hpb.x, hpb.y}, epv[4] = {epa.x, epa.y, epb.x, epb.y};
            // a pair's selector {2 qb0, 2 qb0 + 1, 2 qb1, 2 qb1 + 1} from its two stored bytes (N: 0x0d, 0x0e: both give 0xff)
            const uint32_t slv[4] = {__builtin_amdgcn_perm(sl2.x, sl2.x, 0x01010000u) + 0x01000100u, __builtin_amdgcn_perm(sl2.x, sl2.x, 0x03030202u) + 0x01000100u,
                                     __builtin_amdgcn_perm(sl2.y, sl2.y, 0x01010000u) + 0x01000100u, __builtin_amdgcn_perm(sl2.y, sl2.y, 0x03030202u) + 0x01000100u};
            uint32_t AM[4], M[4], E[4], PX[4], JE[4], JP[4];
            JE[0] = (uint32_t)(jb * e_ins) * 0x10001u + ((uint32_t)e_ins << 16);
            JP[0] = (uint32_t)jb * 0x10001u + 0x00020001u;            // (j + 1, j + 2)
#pragma unroll
            for (int c = 1; c < 4; ++c) { JE[c] = pk::add(JE[c - 1], E2X); JP[c] = pk::add(JP[c - 1], TWO); }
            // Stage by stage over the four pairs, not pair by pair: dependent packed (VOP3P) instructions need a wait state between them
            // (the compiler fills it with s_nop 0: 34 per window in the pair-by-pair order), and the four pairs are independent.
            uint32_t Hm[4], T1[4], T2[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t blo = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * c, 1), bhi = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * c + 1, 1);
                AM[c] = pk::bfi(0xffffu, blo, bhi);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { Hm[c] = hpv[c] & AM[c]; E[c] = epv[c] & AM[c]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) T1[c] = pk::add(Hm[c], __builtin_amdgcn_perm(tt.y, tt.x, slv[c]));
#pragma unroll
            for (int c = 0; c < 4; ++c) T2[c] = pk::minu(Hm[c], ONE);
#pragma unroll
            for (int c = 0; c < 4; ++c) M[c] = pk::mul(T1[c], T2[c]);                                   // M = H ? H + S : 0
#pragma unroll
            for (int c = 0; c < 4; ++c) T1[c] = pk::sub(M[c], OEI);
#pragma unroll
            for (int c = 0; c < 4; ++c) T1[c] = pk::max0(T1[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) T1[c] = pk::add(T1[c], JE[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) PX[c] = pk::max(T1[c], T1[c] << 16);                            // the pair's inclusive prefix (x >= 0)
#pragma unroll
            for (int c = 1; c < 4; ++c) PX[c] = pk::max(PX[c], pk::hi2(PX[c - 1]));
            // over the four lanes of the task
            int scan = (int)(PX[3] >> 16);
            {
                int t = pk::qdpp<0x90>(scan);                          // quad_perm [0,0,1,2]
                scan = scan > t ? scan : t;
                t = pk::qdpp<0x44>(scan);                              // quad_perm [0,1,0,1]
                scan = scan > t ? scan : t;
            }
            int Lex = pk::qdpp<0x90>(scan);
            Lex = g == 0 ? kPkNeg : Lex;
            Lex = Lex > c_max ? Lex : c_max;
            const uint32_t Lexp = pk::lo2((uint32_t)Lex);
            uint32_t Hn[4], E2[4], F4[4], ME[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) F4[c] = __builtin_amdgcn_alignbit(PX[c], c ? PX[c - 1] : NEGP, 16);      // {prefix up to column j - 1}
#pragma unroll
            for (int c = 0; c < 4; ++c) { F4[c] = pk::max(F4[c], Lexp); E2[c] = pk::sub(M[c], OED); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { F4[c] = pk::sub(F4[c], JE[c]); T1[c] = pk::sub(E[c], EDX); ME[c] = pk::max(M[c], E[c]); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { F4[c] = pk::add(F4[c], E1X); E2[c] = pk::max(E2[c], T1[c]); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { F4[c] = pk::max0(F4[c]); E2[c] = pk::max0(E2[c]); }                    // F = max(Pex - (j - 1) e_ins, 0)
#pragma unroll
            for (int c = 0; c < 4; ++c) Hn[c] = pk::max(ME[c], F4[c]);
            // what is stored: H(i, j - 1) beside E(i + 1, j); the column to the left of the lane's first comes from the lane before
            const int h3 = (int)(Hn[3] >> 16);
            int hin = pk::qdpp<0x90>(h3);
            hin = g == 0 ? c_h : hin;
            uint32_t HL[4];
            HL[0] = __builtin_amdgcn_alignbit(Hn[0], (uint32_t)hin << 16, 16);
#pragma unroll
            for (int c = 1; c < 4; ++c) HL[c] = __builtin_amdgcn_alignbit(Hn[c], Hn[c - 1], 16);
            if (m8) {
                *reinterpret_cast<uint2 *>(t_hp + pj) = make_uint2(pk::bfi(AM[0], HL[0], hpv[0]), pk::bfi(AM[1], HL[1], hpv[1]));
                *reinterpret_cast<uint2 *>(t_hp + pj + 2) = make_uint2(pk::bfi(AM[2], HL[2], hpv[2]), pk::bfi(AM[3], HL[3], hpv[3]));
                *reinterpret_cast<uint2 *>(t_ep + pj) = make_uint2(pk::bfi(AM[0], E2[0], epv[0]), pk::bfi(AM[1], E2[1], epv[1]));
                *reinterpret_cast<uint2 *>(t_ep + pj + 2) = make_uint2(pk::bfi(AM[2], E2[2], epv[2]), pk::bfi(AM[3], E2[3], epv[3]));
            }
            // row maximum with the last column attaining it; first / last column with a non-zero stored cell
            int lkey = -1;
            // key = H << 8 | (j + 1), one v_perm per column: the bytes {index, H low, H high, 0}.  A dead column contributes
            // (0 << 8 | j + 1) < 256: below every live key of a row whose maximum is positive, and the column of the maximum is
            // not read when the maximum is 0
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t ha = Hn[c] & AM[c];
                const int k0 = (int)__builtin_amdgcn_perm(ha, JP[c], 0x0c050400u), k1 = (int)__builtin_amdgcn_perm(ha, JP[c], 0x0c070602u);
                const int kk = k0 > k1 ? k0 : k1;
                lkey = lkey > kk ? lkey : kk;
            }
            {
                uint32_t NF[4], NA[4], NB[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) NF[c] = (HL[c] | E2[c]) & AM[c];
#pragma unroll
                for (int c = 0; c < 4; ++c) { NF[c] = pk::minu(NF[c], ONE); NB[c] = pk::sub(C257, JP[c]); }
#pragma unroll
                for (int c = 0; c < 4; ++c) { NA[c] = pk::mul(NF[c], JP[c]); NB[c] = pk::mul(NF[c], NB[c]); }
                NA[0] = pk::max(NA[0], NA[1]); NA[2] = pk::max(NA[2], NA[3]); NB[0] = pk::max(NB[0], NB[1]); NB[2] = pk::max(NB[2], NB[3]);
                NA[0] = pk::max(NA[0], NA[2]); NB[0] = pk::max(NB[0], NB[2]);
                nzl = pk::max(nzl, NA[0]);
                nzf = pk::max(nzf, NB[0]);
            }
            lkey = m8 ? lkey : -1;                                  // a lane without a live column contributes nothing
            key = key > lkey ? key : lkey;
            {                                                       // H of column end - 1, once per row
                const int oe = hi - 1;
                if (m8 && jb + hi == end) {
                    uint32_t hv = Hn[0];
                    hv = (oe >> 1) == 1 ? Hn[1] : hv;
                    hv = (oe >> 1) == 2 ? Hn[2] : hv;
                    hv = (oe >> 1) == 3 ? Hn[3] : hv;
                    hlast = (int)((hv >> ((oe & 1) << 4)) & 0xffffu);
                }
            }
            if (in && base == (beg & ~3) && h1 != 0 && g == 0) t_hp16[beg] = (uint16_t)h1;      // eh[beg].h = h1: H(i, beg - 1) of a row that starts at column 0
            // a further window of this row: carry the prefix maximum and the last column's H into the next iteration
            const int pm = pk::qdpp<0xFF>(scan);                       // quad_perm [3,3,3,3]: the task's inclusive total
            const int ch = pk::qdpp<0xFF>(h3);
            const bool more = in && base + LPT * 8 < end;
            if (more) {
                c_max = c_max > pm ? c_max : pm;
                c_h = ch;
                base += LPT * 8;
            }
            // ---- the end of a row
            const int rkey = pk::quad_all_max(key), rhl = pk::quad_all_max(hlast);
            const int a_ = (int)(nzl & 0xffffu), b_ = (int)(nzl >> 16), c_ = (int)(nzf & 0xffffu), d_ = (int)(nzf >> 16);
            int last_nz = pk::quad_all_max(a_ > b_ ? a_ : b_) - 1;      // -1: none
            int first_nz = 256 - pk::quad_all_max(c_ > d_ ? c_ : d_);   // 256: none (beyond every column)
            if (alive && !more) {
                if (row && h1 != 0) {                                    // column beg stores h1 (see above)
                    first_nz = first_nz < beg ? first_nz : beg;
                    last_nz = last_nz > beg ? last_nz : beg;
                }
                const int m = row ? rkey >> 8 : 0, mj = row ? (rkey & 0xff) - 1 : -1;
                const int h1f = row ? rhl : h1;
                if (g == 0) { t_hp16[end] = (uint16_t)h1f; t_ep16[end] = 0; }   // eh[end] = {h1f, 0}
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
                start_row = true;
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
               DevCounters *ctr, int cu_count, hipStream_t st, int32_t *list, hipStream_t *aux, hipEvent_t fork, hipEvent_t *join,
               const int64_t *src, int dir) {
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
    const bool pk_env = knobs().bsw_pk != 0;       // A-B knob: the 32-bit eight-task kernel
    if (pk_env && bsw_pk_eligible(prm)) {
        auto pk_lds = [](int cols) { return (size_t)64 + (size_t)kPkWaves * 16 * ((size_t)(cols / 2) * 10 + 24); };
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(bsw_pk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk_lds(192)) != hipSuccess) return -1;
        static const int kCols[5] = {32, 64, 96, 144, 192};
        const unsigned Bp = (unsigned)((int64_t)B * kWavesPerBlock / kPkWaves);
        for (int c = 4; c >= 0; --c)
            bsw_pk_kernel<<<Bp, kPkWaves * 64, pk_lds(kCols[c]), q[c]>>>(pairs, list + (int64_t)c * n, cnt + c, ref, qer, w, prm, ctr, hd + c, kCols[c], src, dir);
    } else {
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 192 * 4, q[4]>>>(pairs, list + 4 * n, cnt + 4, ref, qer, w, prm, ctr, hd + 4, 192, src, dir);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 144 * 4, q[3]>>>(pairs, list + 3 * n, cnt + 3, ref, qer, w, prm, ctr, hd + 3, 144, src, dir);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 96 * 4, q[2]>>>(pairs, list + 2 * n, cnt + 2, ref, qer, w, prm, ctr, hd + 2, 96, src, dir);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 64 * 4, q[1]>>>(pairs, list + 1 * n, cnt + 1, ref, qer, w, prm, ctr, hd + 1, 64, src, dir);
    bsw_qwin_kernel<kBswLpt, kBswWin><<<B, T, (size_t)kWavesPerBlock * (64 / kBswLpt) * 32 * 4, q[0]>>>(pairs, list + 0 * n, cnt + 0, ref, qer, w, prm, ctr, hd + 0, 32, src, dir);
    }
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
        bsw_kernel<<<(unsigned)lblocks, waves * 64, lds, q[0]>>>(pairs, n, ref, qer, w, prm, qmax, -1, ctr, &ctr->bsw_head[3], list + 5 * n, cnt + 5, src, dir);
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
