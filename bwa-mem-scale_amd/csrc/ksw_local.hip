// ksw_local.hip — local Smith-Waterman of mate rescue for gfx950 (MI355X).
//
// Reference semantics: ksw_align2 over ksw_u8 / ksw_i16 (/root/reference/src/ksw.cpp:347-381,
// :111-232, :234-338), as called for every rescue candidate by mem_matesw
// (/root/reference/src/bwamem_pair.cpp:214-217) and in batches by mem_sam_pe_batch (:880-979):
// score / te / qe of the best local alignment, score2 / te2 of the best row maximum outside
// the neighbourhood of te, and — with KSW_XSTART — tb / qb from a second pass over the
// reversed prefixes that stops at the first score.
//
// The reference is a striped SSE2 kernel; what of its layout is observable (padding of the
// query to a multiple of 16 or 8 columns that score 0, row maxima over the padded row, the
// byte kernel's 255 stop) is reproduced on a row-parallel DP — see oracle/ksw_oracle.c for the
// argument, which is pinned to the reference object.  Mapping: one task per wavefront, one
// padded query column per lane and chunk (registers), rows in order; the horizontal gap is a
// max-plus prefix scan of the F-free cell values (DPP), the row maximum one more scan; the
// run-merged list of row maxima lives in LDS.  int32 arithmetic.  VALU-bound, no MFMA.
#include "common.h"
#include "wave_ops.h"

namespace bwams {
namespace {

constexpr int kKswWaves = 4;
constexpr int KSW_XBYTE = 0x10000, KSW_XSTOP = 0x20000, KSW_XSUBO = 0x40000, KSW_XSTART = 0x80000;

struct KswOut {
    int score, te, qe, score2, te2, tb, qb;
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One pass of ksw_u8 / ksw_i16.  The query is q[qoff + qsign * j], j < qlen; the target is
// t[tflip - i] for i <= tflip (reversed prefix) and t[i] otherwise (tflip = -1: plain).
template <int NCH>
__device__ void ksw_pass(const SwParams &prm, int size, int qlen, const uint8_t *__restrict__ q, int qoff, int qsign,
                         int tlen, const uint8_t *__restrict__ t, int tflip, int xtra, uint32_t *__restrict__ blist,
                         int bcap, KswOut &r) {
    const int lane = threadIdx.x & 63;
    const int p = size == 1 ? 16 : 8;
    const int P = ((qlen + p - 1) / p) * p;
    const int e_del = prm.e_del, e_ins = prm.e_ins;
    const int oe_del = prm.o_del + e_del, oe_ins = prm.o_ins + e_ins;
    int mn = 127;
    for (int a = 0; a < 25; ++a) mn = mn < prm.mat[a] ? mn : prm.mat[a];
    const int shift = -mn;
    const int minsc = (xtra & KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;
    const int endsc = (xtra & KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;

    int Hd[NCH], E[NCH], Hm[NCH], JE[NCH], P0[NCH], P1[NCH], P2[NCH], P3[NCH], P4[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = c * 64 + lane;
        Hd[c] = 0; E[c] = 0; Hm[c] = 0;
        JE[c] = j * e_ins;
        P0[c] = P1[c] = P2[c] = P3[c] = P4[c] = 0;         // pad columns score 0
        if (j < qlen) {
            const int qj = q[qoff + qsign * j];
            P0[c] = prm.mat[0 * 5 + qj]; P1[c] = prm.mat[1 * 5 + qj]; P2[c] = prm.mat[2 * 5 + qj];
            P3[c] = prm.mat[3 * 5 + qj]; P4[c] = prm.mat[4 * 5 + qj];
        }
    }
    const int nch = (P + 63) >> 6;                         // chunks in use (wave-uniform)
    int gmax = 0, te = -1, n_b = 0, b_last_i = -2, b_last_v = 0;
    bool stop = false;

    auto tload = [&](int i) -> int {
        if (i >= tlen) return 4;
        return i <= tflip ? t[tflip - i] : t[i];
    };
    int slab_next = tload(lane);
    for (int rb = 0; rb < tlen && !stop; rb += 64) {
        const int slab = slab_next;
        if (rb + 64 < tlen) slab_next = tload(rb + 64 + lane);
        const int rlim = tlen - rb < 64 ? tlen - rb : 64;
        for (int ri = 0; ri < rlim; ++ri) {
            const int i = rb + ri;
            const int tb = __builtin_amdgcn_readlane(slab, ri);
            int carry_src = NEG;             // max over earlier chunks of (Hnf + j * e_ins)
            int carry_h = 0;                 // H(i, j-1) entering the chunk (for the next row's diagonal)
            int hmax_lane = 0;
            int Hcur[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                Hcur[c] = 0;
                if (c >= nch) continue;
                const int j = c * 64 + lane;
                const bool act = j < P;
                const int S = tb == 0 ? P0[c] : tb == 1 ? P1[c] : tb == 2 ? P2[c] : tb == 3 ? P3[c] : P4[c];
                int hd = Hd[c] + S;
                hd = hd > 0 ? hd : 0;
                const int e = E[c];
                const int hnf = hd > e ? hd : e;
                const int src = act ? hnf + JE[c] : NEG;
                const int Pm = scan_max(src);
                int Pex = lane_shr1(Pm, carry_src);
                Pex = Pex > carry_src ? Pex : carry_src;
                int f = Pex - oe_ins - (JE[c] - e_ins);
                f = f > 0 ? f : 0;
                const int h = act ? (hnf > f ? hnf : f) : 0;
                Hcur[c] = h;
                hmax_lane = hmax_lane > h ? hmax_lane : h;
                int e1 = e - e_del;
                int e2 = h - oe_del;
                e1 = e1 > e2 ? e1 : e2;
                E[c] = e1 > 0 ? e1 : 0;
                Hd[c] = lane_shr1(h, carry_h);            // H(i, j-1): next row's diagonal input
                carry_h = __builtin_amdgcn_readlane(h, 63);
                const int cs = __builtin_amdgcn_readlane(Pm, 63);
                carry_src = carry_src > cs ? carry_src : cs;
            }
            const int imax = __builtin_amdgcn_readlane(scan_max(hmax_lane), 63);
            // run-merged list of row maxima >= minsc (ksw.cpp:194-202)
            if (imax >= minsc) {
                if (n_b == 0 || b_last_i + 1 != i) {
                    if (n_b < bcap && lane == 0) blist[n_b] = ((uint32_t)imax << 16) | (uint32_t)i;
                    n_b++;
                    b_last_i = i; b_last_v = imax;
                } else if (b_last_v < imax) {
                    if (n_b - 1 < bcap && lane == 0) blist[n_b - 1] = ((uint32_t)imax << 16) | (uint32_t)i;
                    b_last_i = i; b_last_v = imax;
                }
                // else: a row directly after the entry's recorded row that does not improve it leaves
                // the entry (and its recorded row) unchanged, exactly as the reference's test does
            }
            if (imax > gmax) {
                gmax = imax; te = i;
#pragma unroll
                for (int c = 0; c < NCH; ++c) Hm[c] = Hcur[c];
                if ((size == 1 && gmax + shift >= 255) || gmax >= endsc) { stop = true; break; }
            }
        }
    }
    r.score = (size == 1 && gmax + shift >= 255) ? 255 : gmax;
    r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1; r.tb = -1; r.qb = -1;
    if (!(size == 1 && r.score == 255)) {
        // qe: smallest column attaining the maximum of the saved row
        int ml = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) ml = ml > Hm[c] ? ml : Hm[c];
        const int mval = __builtin_amdgcn_readlane(scan_max(ml), 63);
        int qe = -1;
#pragma unroll
        for (int c = NCH - 1; c >= 0; --c) {
            const unsigned long long eq = __ballot(c < nch && c * 64 + lane < P && Hm[c] == mval);
            if (eq) qe = c * 64 + __ffsll((long long)eq) - 1;
        }
        r.qe = qe;
        if (n_b > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int mx = prm.max_sc;
            const int d = (r.score + mx - 1) / mx;
            const int low = te - d, high = te + d;
            const int nn = n_b < bcap ? n_b : bcap;
            // best (value, then earliest entry) among entries outside [low, high]
            int best = -1;                   // key = value << 16 | (0xffff - entry index)
            for (int k = lane; k < nn; k += 64) {
                const uint32_t v = blist[k];
                const int ei = (int)(v & 0xffff), ev = (int)(v >> 16);
                if (ei < low || ei > high) {
                    const int key = (ev << 15) | (0x7fff - (k & 0x7fff));
                    best = best > key ? best : key;
                }
            }
            best = __builtin_amdgcn_readlane(scan_max(best), 63);
            if (best >= 0) {
                const int k = 0x7fff - (best & 0x7fff);
                const uint32_t v = blist[k];
                r.score2 = (int)(v >> 16);
                r.te2 = (int)(v & 0xffff);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(kKswWaves * 64) void ksw_kernel(const bwams_seqpair_t *__restrict__ pairs, int64_t n,
                                                             const uint8_t *__restrict__ ref,
                                                             const uint8_t *__restrict__ qer, SwParams prm, int plo,
                                                             int bcap, KswOut *__restrict__ out, DevCounters *ctr) {
    extern __shared__ uint32_t ksw_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *blist = ksw_lds + (size_t)wave * bcap;
    int64_t pid = 0, pid_end = 0;
    while (true) {
        if (pid >= pid_end) {
            pid = (int64_t)wave_ticket(&ctr->work_head, 4ull);   // out of line: see wave_ops.h
            pid_end = pid + 4 < n ? pid + 4 : n;
            if (pid >= n) break;
        }
        const int64_t cur = pid++;
        const int qlen = uni(pairs[cur].len2);
        const int xtra = uni(pairs[cur].h0);
        const int size = (xtra & KSW_XBYTE) ? 1 : 2;
        const int p = size == 1 ? 16 : 8;
        const int P = ((qlen + p - 1) / p) * p;
        if (P <= plo || P > 64 * NCH) continue;            // another variant's task
        const int tlen = uni(pairs[cur].len1);
        const uint8_t *tq = qer + uni(pairs[cur].idq);
        const uint8_t *tr = ref + uni(pairs[cur].idr);
        KswOut r;
        ksw_pass<NCH>(prm, size, qlen, tq, 0, 1, tlen, tr, -1, xtra, blist, bcap, r);
        if (r.qe >= 0 && !((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff)))) {
            KswOut rr;
            ksw_pass<NCH>(prm, size, r.qe + 1, tq, r.qe, -1, tlen, tr, r.te, KSW_XSTOP | r.score, blist, bcap, rr);
            if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
        }
        if (lane == 0) out[cur] = r;
    }
}

__global__ void ksw_reset_kernel(DevCounters *ctr) { ctr->work_head = 0; }

}  // namespace

// out: n records of 7 int32 (kswr_t layout).  Returns 0, or -2 when the row-maxima lists of a block (tmax / 2 + 2
// entries per wave) do not fit one CU's LDS: targets up to kKswMaxTarget bases.
int launch_ksw(const bwams_seqpair_t *pairs, int64_t n, const uint8_t *ref, const uint8_t *qer, const SwParams &prm,
               int pmax, int tmax, void *out, DevCounters *ctr, int cu_count, hipStream_t st) {
    if (n <= 0) return 0;
    int bcap = tmax / 2 + 2;                               // runs of row maxima are separated by at least one row
    if (bcap < 64) bcap = 64;
    const size_t lds = (size_t)kKswWaves * bcap * 4;
    if (lds > 160 * 1024) return -2;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ksw_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ksw_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int64_t blocks = (n + kKswWaves - 1) / kKswWaves;
    const int64_t maxb = (int64_t)cu_count * 8;
    if (blocks > maxb) blocks = maxb;
    KswOut *o = reinterpret_cast<KswOut *>(out);
    ksw_reset_kernel<<<1, 1, 0, st>>>(ctr);
    ksw_kernel<3><<<(unsigned)blocks, kKswWaves * 64, lds, st>>>(pairs, n, ref, qer, prm, -1, bcap, o, ctr);
    if (pmax > 192) {
        ksw_reset_kernel<<<1, 1, 0, st>>>(ctr);
        ksw_kernel<8><<<(unsigned)blocks, kKswWaves * 64, lds, st>>>(pairs, n, ref, qer, prm, 192, bcap, o, ctr);
    }
    return 0;
}

}  // namespace bwams
