// reg2aln.hip — the SAM-side alignment step on the device: CIGAR, NM, MD, position of every final region.
//
// Replaces, for a whole chunk,
//   mem_reg2aln        /root/reference/src/bwamem.cpp:2533-2628   (band inference, retry loop, clip / squeeze, position)
//   bwa_gen_cigar2     /root/reference/src/bwa.cpp:380-467        (gap-free shortcut, band choice, NM and MD)
//   ksw_global2        /root/reference/src/ksw.cpp:558-668        (banded global alignment with traceback)
// (mem_approx_mapq_se, :1983-2008, is a dozen double operations per region with log(): bwams_reg2aln_fetch evaluates it on the
// host with the C library's log; the SAM text stage evaluates it on the device from a host-filled log table
// (sam_text.hip: sam_mapq_kernel) with equal results — both so that the last bit is the reference's.)
//
// Mapping.  Regions are independent.  Most need no dynamic programming at all (equal lengths and an inferred band of 0:
// the reference's "no gap" shortcut) — aln_simple_kernel finishes those, one lane per region, and lists the rest.
// aln_dp_kernel runs the banded global alignment for the listed regions, one lane per region, dense waves: the (h, e) row
// of ksw_global2 only ever holds 2w + 2 live columns (the band slides by one column per row and eh[end] is rewritten
// every row), so it lives in an LDS ring of the band's size class (32 / 128 columns per lane); wider bands fall back
// to a row in HBM scratch.  The direction matrix z goes to HBM packed four cells to a word; the traceback reads it back.
// CIGAR and MD are written to per-region scratch and compacted into the flat pools afterwards.
#include "common.h"
#include "chain_kernels.h"
#include "wave_ops.h"

namespace bwams {
namespace {

constexpr int kMinusInf = -0x40000000;

__device__ __forceinline__ int infer_bw(int l1, int l2, int score, int a, int q, int r) {
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    const int d = l1 > l2 ? l1 - l2 : l2 - l1;
    if (w < d) w = d;
    return w;
}

__device__ __forceinline__ int64_t read_of_region(const int64_t *__restrict__ off, int64_t nseq, int64_t k) {
    int64_t lo = 0, hi = nseq;                       // largest r with off[r] <= k
    while (lo + 1 < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}

// bns_pos2rid (bntseq.cpp:397-411)
__device__ __forceinline__ int pos2rid_f(const DevBns &b, int64_t pos_f) {
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

struct Seqs {                                        // the two sequences as bwa_gen_cigar2 aligns them
    const uint8_t *q, *r;                            // query segment, reference segment (text[rb, re))
    int lq, lr;
    bool rev;                                        // rb >= l_pac: both are read backwards
    __device__ __forceinline__ int qa(int i) const { return rev ? q[lq - 1 - i] : q[i]; }
    __device__ __forceinline__ int ra(int i) const { return rev ? r[lr - 1 - i] : r[i]; }
};

__device__ __forceinline__ int put_num(char *s, int l, int x) {          // kputw
    char buf[12];
    int n = 0;
    if (x == 0) buf[n++] = '0';
    while (x > 0) { buf[n++] = (char)('0' + x % 10); x /= 10; }
    while (n > 0) s[l++] = buf[--n];
    return l;
}

// NM and MD from a CIGAR (bwa.cpp:430-459)
// QA / RA: the aligned query / reference base at an index (the sequences in HBM, or the copies a wave kernel holds in LDS)
template <class QA, class RA>
__device__ int nm_md_of(const Seqs &S, QA qa, RA ra, const uint32_t *cigar, int n_cigar, char *md, int *md_len) {
    int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
    const char *int2base = S.rev ? "TGCAN" : "ACGTN";
    for (int k = 0; k < n_cigar; ++k) {
        const int op = (int)(cigar[k] & 0xf), len = (int)(cigar[k] >> 4);
        if (op == 0) {
            for (int i = 0; i < len; ++i) {
                const int rb = ra(y + i);
                if (qa(x + i) != rb) {
                    l = put_num(md, l, u);
                    md[l++] = int2base[rb > 4 ? 4 : rb];
                    ++n_mm; u = 0;
                } else ++u;
            }
            x += len; y += len;
        } else if (op == 2) {
            if (k > 0 && k < n_cigar - 1) {
                l = put_num(md, l, u); md[l++] = '^';
                for (int i = 0; i < len; ++i) { const int rb = ra(y + i); md[l++] = int2base[rb > 4 ? 4 : rb]; }
                u = 0; n_gap += len;
            }
            y += len;
        } else if (op == 1) { x += len; n_gap += len; }
    }
    l = put_num(md, l, u); md[l++] = 0;
    *md_len = l;
    return n_mm + n_gap;
}
__device__ int nm_md(const Seqs &S, const uint32_t *cigar, int n_cigar, char *md, int *md_len) {
    return nm_md_of(S, [&](int i) { return S.qa(i); }, [&](int i) { return S.ra(i); }, cigar, n_cigar, md, md_len);
}

// the tail of mem_reg2aln (bwamem.cpp:2570-2626): squeeze, clip, position
__device__ void finish_record(const RegAlnArgs &A, int64_t k, const bwams_alnreg_t &ar, int l_query, uint32_t *cigar, int n_cigar,
                              int NM, int md_len) {
    const int64_t l_pac = A.bns.l_pac;
    const int64_t p0 = ar.rb < l_pac ? ar.rb : ar.re - 1;
    const int is_rev = p0 >= l_pac;
    int64_t pos = is_rev ? (l_pac << 1) - 1 - p0 : p0;
    if (n_cigar > 0) {
        if ((cigar[0] & 0xf) == 2) {
            pos += cigar[0] >> 4;
            --n_cigar;
            for (int i = 0; i < n_cigar; ++i) cigar[i] = cigar[i + 1];
        } else if ((cigar[n_cigar - 1] & 0xf) == 2) --n_cigar;
    }
    if (ar.qb != 0 || ar.qe != l_query) {
        const int clip5 = is_rev ? l_query - ar.qe : ar.qb, clip3 = is_rev ? ar.qb : l_query - ar.qe;
        if (clip5) {
            for (int i = n_cigar; i > 0; --i) cigar[i] = cigar[i - 1];
            cigar[0] = (uint32_t)clip5 << 4 | 3;
            ++n_cigar;
        }
        if (clip3) cigar[n_cigar++] = (uint32_t)clip3 << 4 | 3;
    }
    bwams_aln_t a;
    a.rid = pos2rid_f(A.bns, pos);
    a.pos = pos - (a.rid >= 0 ? A.bns.contigs[a.rid].offset : 0);
    a.flag = ar.secondary >= 0 ? 0x100 : 0;
    a.is_rev = is_rev;
    a.is_alt = (int32_t)(((uint32_t)ar.n_comp_is_alt >> 30) & 1u);
    a.mapq = 0;                                       // filled by bwams_reg2aln_fetch (host) / sam_mapq_kernel (device)
    a.NM = NM;
    a.n_cigar = n_cigar; a.md_len = md_len;
    a.cigar_off = 0; a.md_off = 0;
    a.score = ar.score; a.sub = ar.sub > ar.csub ? ar.sub : ar.csub; a.alt_sc = ar.alt_sc;
    a.pad_ = 0;
    A.rec[k] = a;
}

// scratch of region k: cigar words, then MD bytes, then (DP regions) the z words and the HBM (h, e) row
__device__ __forceinline__ uint32_t *scr_cigar(const RegAlnArgs &A, int64_t k) { return reinterpret_cast<uint32_t *>(A.scr + A.scr_off[k]); }
__device__ __forceinline__ char *scr_md(const RegAlnArgs &A, int64_t k, int lq, int lr) {
    return reinterpret_cast<char *>(A.scr + A.scr_off[k] + (size_t)(lq + lr + 4) * 4);
}
__host__ __device__ __forceinline__ size_t md_cap(int lr) { return ((size_t)3 * lr + 16 + 15) & ~(size_t)15; }

// ---- pass 1: scratch need of every region ---------------------------------------------------------------
__global__ void aln_plan_kernel(RegAlnArgs A) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_regs) return;
    const bwams_alnreg_t ar = A.regs[k];
    const int lq = ar.qe - ar.qb;
    const int64_t lr64 = ar.re - ar.rb;
    const int64_t l_pac = A.bns.l_pac;
    const bool skip = A.only && !A.only[k];            // not needed by the SAM text: the record stays an unmapped one
    const bool bad = skip || ar.rb < 0 || ar.re < 0 || lq <= 0 || lr64 <= 0 || (ar.rb < l_pac && ar.re > l_pac) || ar.re > 2 * l_pac || lr64 > (1 << 20);
    int64_t need = 64;
    int cls = -1;                                     // -1: no DP; 0 / 1: LDS ring of 32 / 128 columns; 2: HBM row
    if (!bad) {
        const int lr = (int)lr64;
        need = (int64_t)(lq + lr + 4) * 4 + (int64_t)md_cap(lr);
        int tmp = infer_bw(lq, lr, ar.truesc, A.opt.a, A.opt.o_del, A.opt.e_del);
        int w2 = infer_bw(lq, lr, ar.truesc, A.opt.a, A.opt.o_ins, A.opt.e_ins);
        w2 = w2 > tmp ? w2 : tmp;
        if (w2 > A.opt.w) w2 = w2 < ar.w ? w2 : ar.w;
        if (!(lq == lr && w2 == 0)) {
            // the widest band the retry loop can reach: w2, 2 w2, 4 w2 (capped at 4 opt.w), never below |lr - lq| + 3
            const int d = lr > lq ? lr - lq : lq - lr;
            int64_t wmax = (int64_t)w2 << 2;
            wmax = wmax < ((int64_t)A.opt.w << 2) ? wmax : ((int64_t)A.opt.w << 2);
            wmax = wmax > d + 3 ? wmax : d + 3;
            const int64_t n_col = lq < 2 * wmax + 1 ? lq : 2 * wmax + 1;
            // the class follows the band of the FIRST try (bwa.cpp:414-423); a retry that outgrows the ring — rare: the
            // global score fell short of the local one — sends the region to the HBM-row launch, which runs last
            {
                const int8_t m0 = A.opt.mat[0];
                const int max_ins = (int)((double)(((lq + 1) >> 1) * m0 - A.opt.o_ins) / A.opt.e_ins + 1.);
                const int max_del = (int)((double)(((lq + 1) >> 1) * m0 - A.opt.o_del) / A.opt.e_del + 1.);
                int max_gap = max_ins > max_del ? max_ins : max_del;
                max_gap = max_gap > 1 ? max_gap : 1;
                int w1 = (max_gap + d + 1) >> 1;
                const int w2c = w2 < A.opt.w << 2 ? w2 : A.opt.w << 2;
                w1 = w1 < w2c ? w1 : w2c;
                w1 = w1 > d + 3 ? w1 : d + 3;
                cls = 2 * w1 + 2 <= 32 ? 0 : 2 * w1 + 2 <= 128 ? 1 : 2;
            }
            need += ((n_col + 3) / 4) * 4 * (int64_t)lr;                 // z: (n_col + 3) / 4 words per row, for the widest band
            need += (int64_t)(lq + 1) * 8;                               // the HBM (h, e) row, should the region end up there
        }
    }
    A.need[k] = (need + 15) & ~(int64_t)15;
    A.cls[k] = bad ? -2 : cls;
}

// ---- pass 2: regions without DP; the others are listed by class ----------------------------------------------
__global__ void aln_simple_kernel(RegAlnArgs A) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_regs) return;
    const int cls = A.cls[k];
    const bwams_alnreg_t ar = A.regs[k];
    const int64_t r = read_of_region(A.reg_off, A.nseq, k);
    const int l_query = (int)(A.cum[r + 1] - A.cum[r]);
    if (cls == -2) {                                  // mem_reg2aln's unmapped record (or a region bwa_gen_cigar2 rejects)
        bwams_aln_t a;
        memset(&a, 0, sizeof a);
        a.rid = -1; a.pos = -1; a.flag = 0x4;
        A.rec[k] = a;
        return;
    }
    if (cls >= 0) {
        A.list[(int64_t)cls * A.n_regs + (int64_t)atomicAdd(&A.n_list[cls], 1ull)] = (int32_t)k;
        return;
    }
    Seqs S;
    S.lq = ar.qe - ar.qb; S.lr = (int)(ar.re - ar.rb);
    S.q = A.enc + A.cum[r] + ar.qb; S.r = A.ref + ar.rb; S.rev = ar.rb >= A.bns.l_pac;
    uint32_t *cigar = scr_cigar(A, k);
    char *md = scr_md(A, k, S.lq, S.lr);
    cigar[0] = (uint32_t)S.lq << 4 | 0;
    int md_len = 0;
    const int NM = nm_md(S, cigar, 1, md, &md_len);
    finish_record(A, k, ar, l_query, cigar, 1, NM, md_len);
}

// ---- pass 3: banded global alignment with traceback, lane per region --------------------------------------
// The (h, e) row: RING columns per lane in LDS (column j at slot j & (RING - 1), word layout [slot][lane] so that any
// per-lane slot is conflict-free), or — RING == 0 — the whole row in HBM scratch.
template <int RING>
struct EhRow {
    int2 *base;                                        // LDS: this lane's column 0; HBM: eh[0]
    __device__ __forceinline__ int2 get(int j) const { return RING ? base[(j & (RING - 1)) * 64] : base[j]; }
    __device__ __forceinline__ void put(int j, int2 v) const { if (RING) base[(j & (RING - 1)) * 64] = v; else base[j] = v; }
};

template <int RING>
__device__ int global2_cigar(const RegAlnArgs &A, const Seqs &S, int w, const EhRow<RING> &eh, uint32_t *z, uint32_t *cigar, int *n_cigar_) {
    const bwams_mem_opt_t &o = A.opt;
    const int qlen = S.lq, tlen = S.lr;
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const int zw = (n_col + 3) >> 2;                   // z words per row
    // first row: only columns 0 .. min(qlen, w + 1) can be read before they are rewritten
    eh.put(0, make_int2(0, kMinusInf));
    for (int j = 1; j <= qlen && j <= w + 1; ++j)
        eh.put(j, j <= w ? make_int2(-(o.o_ins + o.e_ins * j), kMinusInf) : make_int2(kMinusInf, kMinusInf));
    for (int i = 0; i < tlen; ++i) {
        int f = kMinusInf, h1;
        const int tb = S.ra(i);
        const int8_t *mrow = &o.mat[(tb > 4 ? 4 : tb) * 5];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o.o_del + o.e_del * (i + 1)) : kMinusInf;
        uint32_t *zi = z + (size_t)i * zw;
        uint32_t pack = 0;
        for (int j = beg; j < end; ++j) {
            const int2 p = eh.get(j);
            int m = p.x, e = p.y, h, t;
            uint32_t d;
            const int qb = S.qa(j);
            m += mrow[qb > 4 ? 4 : qb];
            d = m >= e ? 0u : 1u;
            h = m >= e ? m : e;
            d = h >= f ? d : 2u;
            h = h >= f ? h : f;
            t = m - oe_del;
            e -= o.e_del;
            d |= e > t ? 1u << 2 : 0u;
            e = e > t ? e : t;
            eh.put(j, make_int2(h1, e));
            h1 = h;
            t = m - oe_ins;
            f -= o.e_ins;
            d |= f > t ? 2u << 4 : 0u;
            f = f > t ? f : t;
            const int c = j - beg;
            pack |= d << ((c & 3) * 8);
            if ((c & 3) == 3) { zi[c >> 2] = pack; pack = 0; }
        }
        if ((end - beg) & 3) zi[(end - beg) >> 2] = pack;
        eh.put(end, make_int2(h1, kMinusInf));
    }
    const int score = eh.get(qlen).x;
    int n = 0, which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int run_op = -1, run_len = 0;                        // the operation being extended stays in registers: a step reads z only
    auto push = [&](int op, int len) {
        if (op != run_op) {
            if (run_op >= 0) cigar[n++] = (uint32_t)run_len << 4 | (uint32_t)run_op;
            run_op = op; run_len = len;
        } else run_len += len;
    };
    const uint8_t *zb = reinterpret_cast<const uint8_t *>(z);
    while (i >= 0 && k >= 0) {
        const int c = k - (i > w ? i - w : 0);
        const uint32_t d = zb[(size_t)i * zw * 4 + c];
        which = (int)(d >> (which << 1)) & 3;
        if (which == 0) { push(0, 1); --i; --k; }
        else if (which == 1) { push(2, 1); --i; }
        else { push(1, 1); --k; }
    }
    if (i >= 0) push(2, i + 1);
    if (k >= 0) push(1, k + 1);
    if (run_op >= 0) cigar[n++] = (uint32_t)run_len << 4 | (uint32_t)run_op;
    for (int a = 0; a < n >> 1; ++a) { const uint32_t t = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = t; }
    *n_cigar_ = n;
    return score;
}

template <int RING>
__global__ __launch_bounds__(64) void aln_dp_kernel(RegAlnArgs A, int cls) {
    extern __shared__ int2 eh_lds[];                   // [RING][64]
    const int lane = threadIdx.x;
    const int64_t n = (int64_t)A.n_list[cls];
    const int32_t *list = A.list + (int64_t)cls * A.n_regs;
    for (int64_t t = (int64_t)blockIdx.x * 64 + lane; t < n; t += (int64_t)gridDim.x * 64) {
        const int64_t k = list[t];
        const bwams_alnreg_t ar = A.regs[k];
        const int64_t r = read_of_region(A.reg_off, A.nseq, k);
        const int l_query = (int)(A.cum[r + 1] - A.cum[r]);
        Seqs S;
        S.lq = ar.qe - ar.qb; S.lr = (int)(ar.re - ar.rb);
        S.q = A.enc + A.cum[r] + ar.qb; S.r = A.ref + ar.rb; S.rev = ar.rb >= A.bns.l_pac;
        uint32_t *cigar = scr_cigar(A, k);
        char *md = scr_md(A, k, S.lq, S.lr);
        uint32_t *z = reinterpret_cast<uint32_t *>(md + md_cap(S.lr));
        EhRow<RING> eh;
        if (RING) eh.base = eh_lds + lane;
        else {
            const int d = S.lr > S.lq ? S.lr - S.lq : S.lq - S.lr;
            int tmp = infer_bw(S.lq, S.lr, ar.truesc, A.opt.a, A.opt.o_del, A.opt.e_del);
            int w0 = infer_bw(S.lq, S.lr, ar.truesc, A.opt.a, A.opt.o_ins, A.opt.e_ins);
            w0 = w0 > tmp ? w0 : tmp;
            if (w0 > A.opt.w) w0 = w0 < ar.w ? w0 : ar.w;
            int64_t wmax = (int64_t)w0 << 2;
            wmax = wmax < ((int64_t)A.opt.w << 2) ? wmax : ((int64_t)A.opt.w << 2);
            wmax = wmax > d + 3 ? wmax : d + 3;
            const int64_t n_col = S.lq < 2 * wmax + 1 ? S.lq : 2 * wmax + 1;
            eh.base = reinterpret_cast<int2 *>(reinterpret_cast<char *>(z) + ((n_col + 3) / 4) * 4 * (size_t)S.lr);
        }
        // mem_reg2aln's loop (bwamem.cpp:2558-2568)
        int tmp = infer_bw(S.lq, S.lr, ar.truesc, A.opt.a, A.opt.o_del, A.opt.e_del);
        int w2 = infer_bw(S.lq, S.lr, ar.truesc, A.opt.a, A.opt.o_ins, A.opt.e_ins);
        w2 = w2 > tmp ? w2 : tmp;
        if (w2 > A.opt.w) w2 = w2 < ar.w ? w2 : ar.w;
        int it = 0, last_sc = -(1 << 30), score = 0, n_cigar = 0;
        bool requeue = false;
        do {
            w2 = w2 < A.opt.w << 2 ? w2 : A.opt.w << 2;
            // bwa_gen_cigar2's band (bwa.cpp:414-423); the gap-free shortcut cannot apply to a listed region's first try,
            // and w2 only grows
            const int8_t m0 = A.opt.mat[0];
            const int max_ins = (int)((double)(((S.lq + 1) >> 1) * m0 - A.opt.o_ins) / A.opt.e_ins + 1.);
            const int max_del = (int)((double)(((S.lq + 1) >> 1) * m0 - A.opt.o_del) / A.opt.e_del + 1.);
            int max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            const int d = S.lr > S.lq ? S.lr - S.lq : S.lq - S.lr;
            int w = (max_gap + d + 1) >> 1;
            w = w < w2 ? w : w2;
            w = w > d + 3 ? w : d + 3;
            if (RING && 2 * w + 2 > RING) { requeue = true; break; }
            score = global2_cigar<RING>(A, S, w, eh, z, cigar, &n_cigar);
            if (score == last_sc || w2 == A.opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
        } while (++it < 3 && score < ar.truesc - A.opt.a);
        if (requeue) {                                   // redone from the start by the HBM-row launch
            A.list[2 * A.n_regs + (int64_t)atomicAdd(&A.n_list[2], 1ull)] = (int32_t)k;
            continue;
        }
        int md_len = 0;
        const int NM = nm_md(S, cigar, n_cigar, md, &md_len);
        finish_record(A, k, ar, l_query, cigar, n_cigar, NM, md_len);
    }
}


// ---- pass 3b: the same alignment, a wavefront per region -----------------------------------------------------------------
// Bands beyond the 32-column ring: lane per region kept 64 KB of LDS per wave for the 128-column ring (two waves per CU) and
// took 131 ms for the ~0.3 M such regions of a million reads — regions at repeat copies whose score deficit alone infers a band of
// ~100 although the optimal alignment has no gap.  Here a wave owns a region and walks the rows with one band column per lane:
// like the extension's recurrence, ksw_global2 opens the horizontal gap from the diagonal move m, never from h
// (f' = max(f - e_ins, m - o_ins - e_ins), ksw.cpp:619-623), so f along a row is a max-plus prefix scan of values known from the
// previous row; h, e, and the direction byte are column-local.  The (h, e) row is in LDS whole (8 B per query column), the
// direction matrix goes to HBM a byte per cell as before, the traceback runs once, on lane 0, after the retry loop has settled.
constexpr int kWaveCols = 512;                        // query columns a wave's LDS row holds (4 KB)
constexpr int kWaveTgt = 1024;                        // target bases staged in LDS (longer targets are read from HBM row by row)
constexpr int kNegScan = -2000000000;

// inclusive prefix maximum over the wavefront: four DPP steps inside each row of 16 lanes, then the row totals carried across with
// row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).  A lane without a source keeps its value.  (Six
// ds_bpermute shuffles, each waiting on the one before, were most of a row step's latency.)
__device__ __forceinline__ int wave_incl_max(int v, int) {
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

// the DP of ksw_global2 (ksw.cpp:588-637) for band w; returns the score, leaves the direction bytes in z
__device__ int global2_dp_wave(const RegAlnArgs &A, const Seqs &S, int w, int2 *eh, const uint8_t *qs, const uint8_t *ts, uint32_t *z, int lane) {
    const bwams_mem_opt_t &o = A.opt;
    // the region is the wave's: lengths and band in scalar registers, so that the row and chunk loops are scalar loops
    const int qlen = __builtin_amdgcn_readfirstlane(S.lq), tlen = __builtin_amdgcn_readfirstlane(S.lr);
    w = __builtin_amdgcn_readfirstlane(w);
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const int zw = (n_col + 3) >> 2;
    for (int j = lane; j <= qlen && j <= w + 1; j += 64)
        eh[j] = j == 0 ? make_int2(0, kMinusInf) : j <= w ? make_int2(-(o.o_ins + o.e_ins * j), kMinusInf) : make_int2(kMinusInf, kMinusInf);
    __syncthreads();
    for (int i = 0; i < tlen; ++i) {
        const int tb = __builtin_amdgcn_readfirstlane(ts ? (int)ts[i] : S.ra(i));
        const int8_t *mrow = &o.mat[(tb > 4 ? 4 : tb) * 5];
        // the row of the scoring matrix as five bytes of one scalar: a lane's score is a shift and a sign extension
        const uint64_t mpk = (uint64_t)(uint8_t)mrow[0] | (uint64_t)(uint8_t)mrow[1] << 8 | (uint64_t)(uint8_t)mrow[2] << 16 |
                             (uint64_t)(uint8_t)mrow[3] << 24 | (uint64_t)(uint8_t)mrow[4] << 32;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1_first = beg == 0 ? -(o.o_del + o.e_del * (i + 1)) : kMinusInf;
        uint8_t *zi = reinterpret_cast<uint8_t *>(z + (size_t)i * zw);
        int f_carry = kMinusInf, h_carry = h1_first, h_end = h1_first;
        for (int c0 = beg; c0 < end; c0 += 64) {
            // branch-free but for the two stores; lanes behind the row's end read its last column and are masked out of the scan.
            // Cross-lane moves are DPP (the neighbour) and v_readlane (lane 63, the row's last column): no LDS permutes.
            const int j = c0 + lane;
            const bool act = j < end;
            const int jj = act ? j : end - 1;
            const int2 p = eh[jj];
            int qb = qs[jj];
            qb = qb > 4 ? 4 : qb;
            const int m = p.x + (int)(int8_t)(uint8_t)(mpk >> (qb << 3));
            int e = p.y;
            const int t_ins = m - oe_ins;
            const int g = act ? t_ins + j * o.e_ins : kNegScan;
            const int P = wave_incl_max(g, lane);
            const int Pex = lane_shr1(P, kNegScan);                        // lane 0: nothing to its left in this chunk
            const int fc = f_carry - lane * o.e_ins;                       // what the gap open before this chunk has become
            const int fp = Pex - (j - 1) * o.e_ins;
            const int f = fc > fp ? fc : fp;
            uint32_t d = m >= e ? 0u : 1u;
            int h = m >= e ? m : e;
            d = h >= f ? d : 2u;
            h = h >= f ? h : f;
            const int t = m - oe_del;
            e -= o.e_del;
            d |= e > t ? 1u << 2 : 0u;
            e = e > t ? e : t;
            const int fn = f - o.e_ins;
            d |= fn > t_ins ? 2u << 4 : 0u;
            const int hl = lane_shr1(h, h_carry);                          // lane 0 takes the previous chunk's last h
            if (act) {
                eh[j] = make_int2(hl, e);
                zi[j - beg] = (uint8_t)d;
            }
            const int fnext = fn > t_ins ? fn : t_ins;
            f_carry = __builtin_amdgcn_readlane(fnext, 63);
            h_carry = __builtin_amdgcn_readlane(h, 63);
            const int last = end - 1 - c0;                                 // the row's last column, if it lies in this chunk
            if (last < 64) h_end = __builtin_amdgcn_readlane(h, last);
        }
        if (lane == 0) eh[end] = make_int2(h_end, kMinusInf);
        __syncthreads();
    }
    return eh[qlen].x;
}

// The traceback of ksw_global2 (ksw.cpp:639-664) over the direction bytes the last DP left: lane 0 walks, all lanes fetch.  A step of the walk reads one direction byte that the
// DP left in HBM scratch, and the next byte's address depends on it: 270 dependent loads of ~0.5 us each were a third of this
// kernel's time per region (profiles/r03_notes.md 88).  The path only moves up and to the left, so from (i, k) it stays inside
// the kTbWin x kTbWin cells above and to the left of it for at least kTbWin steps: the wave copies that window into LDS with
// one round trip (lane -> row l >> 1, 16 bytes of it), lane 0 walks until it leaves the window, and so on.
constexpr int kTbWin = 32;
__device__ void global2_traceback_wave(const Seqs &S, int w, const uint32_t *z, uint32_t *cigar, int *n_cigar_, uint8_t *zl, int lane) {
    const int qlen = S.lq, tlen = S.lr;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    const int zw = (n_col + 3) >> 2;
    const uint8_t *zb = reinterpret_cast<const uint8_t *>(z);
    int n = 0, which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int run_op = -1, run_len = 0;
    auto push = [&](int op, int len) {
        if (op != run_op) {
            if (run_op >= 0) cigar[n++] = (uint32_t)run_len << 4 | (uint32_t)run_op;
            run_op = op; run_len = len;
        } else run_len += len;
    };
    while (i >= 0 && k >= 0) {                           // i, k are wave-uniform here
        const int i0 = i, k0 = k;
        {
            const int ii = i0 - (lane >> 1);
            const int kb = k0 - kTbWin + 1 + (lane & 1) * 16;
            uint8_t *dst = zl + (lane >> 1) * kTbWin + (lane & 1) * 16;
            if (ii >= 0) {
                const int beg = ii > w ? ii - w : 0;
                const int end = ii + w + 1 < qlen ? ii + w + 1 : qlen;       // the row holds columns [beg, end)
                const uint8_t *row = zb + (size_t)ii * zw * 4;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int kk = kb + t;
                    dst[t] = (kk >= beg && kk < end) ? row[kk - beg] : (uint8_t)0;
                }
            }
        }
        __syncthreads();
        if (lane == 0) {
            while (i >= 0 && k >= 0 && i > i0 - kTbWin && k > k0 - kTbWin) {
                const uint32_t d = zl[(i0 - i) * kTbWin + (k - (k0 - kTbWin + 1))];
                which = (int)(d >> (which << 1)) & 3;
                if (which == 0) { push(0, 1); --i; --k; }
                else if (which == 1) { push(2, 1); --i; }
                else { push(1, 1); --k; }
            }
        }
        i = __builtin_amdgcn_readfirstlane(i);
        k = __builtin_amdgcn_readfirstlane(k);
        __syncthreads();
    }
    if (lane == 0) {
        if (i >= 0) push(2, i + 1);
        if (k >= 0) push(1, k + 1);
        if (run_op >= 0) cigar[n++] = (uint32_t)run_len << 4 | (uint32_t)run_op;
        for (int a = 0; a < n >> 1; ++a) { const uint32_t t = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = t; }
        *n_cigar_ = n;
    }
}

__global__ __launch_bounds__(64) void aln_dp_wave_kernel(RegAlnArgs A, int cls, unsigned long long *ticket) {
    __shared__ int2 eh[kWaveCols + 2];
    __shared__ uint8_t qs[kWaveCols + 64], ts_[kWaveTgt + 64];    // the two sequences in the order the DP reads them: a row step waits on no HBM load
    __shared__ uint8_t zl[kTbWin * kTbWin];                       // the traceback's window of direction bytes
    __shared__ int n_cig_s;
    const int lane = threadIdx.x;
    const int64_t n = (int64_t)A.n_list[cls];
    const int32_t *list = A.list + (int64_t)cls * A.n_regs;
    for (;;) {
        const int64_t t = (int64_t)wave_ticket(ticket, 1ull);
        if (t >= n) break;
        const int64_t k = list[t];
        const bwams_alnreg_t ar = A.regs[k];
        const int64_t r = read_of_region(A.reg_off, A.nseq, k);
        const int l_query = (int)(A.cum[r + 1] - A.cum[r]);
        Seqs S;
        S.lq = ar.qe - ar.qb; S.lr = (int)(ar.re - ar.rb);
        S.q = A.enc + A.cum[r] + ar.qb; S.r = A.ref + ar.rb; S.rev = ar.rb >= A.bns.l_pac;
        if (S.lq + 1 > kWaveCols) {                      // a query beyond the LDS row: the lane-per-region HBM launch, which runs last
            if (lane == 0) A.list[3 * A.n_regs + (int64_t)atomicAdd(&A.n_list[3], 1ull)] = (int32_t)k;
            continue;
        }
#ifdef BWAMS_ALNDBG
        const unsigned long long tk0 = wall_clock64();
        unsigned long long n_dp = 0, sum_w = 0;
#endif
        uint32_t *cigar = scr_cigar(A, k);
        char *md = scr_md(A, k, S.lq, S.lr);
        uint32_t *z = reinterpret_cast<uint32_t *>(md + md_cap(S.lr));
        for (int j = lane; j < S.lq; j += 64) qs[j] = (uint8_t)S.qa(j);
        const uint8_t *ts = S.lr <= kWaveTgt ? ts_ : nullptr;
        if (ts) for (int i = lane; i < S.lr; i += 64) ts_[i] = (uint8_t)S.ra(i);
        __syncthreads();
        // mem_reg2aln's loop (bwamem.cpp:2558-2568)
        int tmp = infer_bw(S.lq, S.lr, ar.truesc, A.opt.a, A.opt.o_del, A.opt.e_del);
        int w2 = infer_bw(S.lq, S.lr, ar.truesc, A.opt.a, A.opt.o_ins, A.opt.e_ins);
        w2 = w2 > tmp ? w2 : tmp;
        if (w2 > A.opt.w) w2 = w2 < ar.w ? w2 : ar.w;
#ifdef BWAMS_ALNDBG
        unsigned long long tk1 = 0;
#endif
        int it = 0, last_sc = -(1 << 30), score = 0, w = 0;
        do {
            w2 = w2 < A.opt.w << 2 ? w2 : A.opt.w << 2;
            const int8_t m0 = A.opt.mat[0];
            const int max_ins = (int)((double)(((S.lq + 1) >> 1) * m0 - A.opt.o_ins) / A.opt.e_ins + 1.);
            const int max_del = (int)((double)(((S.lq + 1) >> 1) * m0 - A.opt.o_del) / A.opt.e_del + 1.);
            int max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            const int d = S.lr > S.lq ? S.lr - S.lq : S.lq - S.lr;
            w = (max_gap + d + 1) >> 1;
            w = w < w2 ? w : w2;
            w = w > d + 3 ? w : d + 3;
#ifdef BWAMS_ALNDBG
            if (it == 0) tk1 = wall_clock64();
            n_dp++; sum_w += (unsigned long long)w;
#endif
            score = global2_dp_wave(A, S, w, eh, qs, ts, z, lane);
            if (score == last_sc || w2 == A.opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
        } while (++it < 3 && score < ar.truesc - A.opt.a);
        __syncthreads();                                 // the direction bytes of every lane are visible to lane 0
#ifdef BWAMS_ALNDBG
        const unsigned long long tk2 = wall_clock64();
#endif
        global2_traceback_wave(S, w, z, cigar, &n_cig_s, zl, lane);
        if (lane == 0) {
            int n_cigar = n_cig_s, md_len = 0;
#ifdef BWAMS_ALNDBG
            const unsigned long long tk3 = wall_clock64();
#endif
            const int NM = ts ? nm_md_of(S, [&](int i_) { return (int)qs[i_]; }, [&](int i_) { return (int)ts_[i_]; }, cigar, n_cigar, md, &md_len)
                              : nm_md_of(S, [&](int i_) { return (int)qs[i_]; }, [&](int i_) { return S.ra(i_); }, cigar, n_cigar, md, &md_len);
            finish_record(A, k, ar, l_query, cigar, n_cigar, NM, md_len);
#ifdef BWAMS_ALNDBG
            const unsigned long long tk4 = wall_clock64();
            atomicAdd(&A.n_list[8], 1ull); atomicAdd(&A.n_list[9], tk1 - tk0); atomicAdd(&A.n_list[10], tk2 - tk1);
            atomicAdd(&A.n_list[11], tk3 - tk2); atomicAdd(&A.n_list[12], tk4 - tk3); atomicAdd(&A.n_list[13], n_dp);
            atomicAdd(&A.n_list[14], sum_w); atomicAdd(&A.n_list[15], (unsigned long long)S.lr); atomicMax(&A.n_list[16], tk4 - tk0);
#endif
        }
        __syncthreads();
    }
}

// ---- pass 4: compaction into the flat pools -------------------------------------------------------------
__global__ void aln_sizes_kernel(RegAlnArgs A, int64_t *wide) {             // wide[0 .. n] = n_cigar, wide[n + 1 .. 2n + 1] = md_len
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > A.n_regs) return;
    wide[k] = k < A.n_regs ? A.rec[k].n_cigar : 0;
    wide[A.n_regs + 1 + k] = k < A.n_regs ? A.rec[k].md_len : 0;
}
__global__ void aln_gather_kernel(RegAlnArgs A, const int64_t *offs, uint32_t *cig_out, char *md_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_regs) return;
    bwams_aln_t a = A.rec[k];
    a.cigar_off = offs[k];
    a.md_off = offs[A.n_regs + 1 + k];
    A.rec[k] = a;
    if (A.cls[k] == -2) return;
    const bwams_alnreg_t ar = A.regs[k];
    const uint32_t *cigar = scr_cigar(A, k);
    const char *md = scr_md(A, k, ar.qe - ar.qb, (int)(ar.re - ar.rb));
    for (int i = 0; i < a.n_cigar; ++i) cig_out[a.cigar_off + i] = cigar[i];
    for (int i = 0; i < a.md_len; ++i) md_out[a.md_off + i] = md[i];
}

}  // namespace

void launch_aln_plan(const RegAlnArgs &A, hipStream_t st) {
    if (A.n_regs > 0) aln_plan_kernel<<<(unsigned)((A.n_regs + 255) / 256), 256, 0, st>>>(A);
}
void launch_aln_run(const RegAlnArgs &A, int cu_count, hipStream_t st) {
    if (A.n_regs <= 0) return;
    aln_simple_kernel<<<(unsigned)((A.n_regs + 255) / 256), 256, 0, st>>>(A);
    aln_dp_kernel<32><<<(unsigned)(cu_count * 8), 64, 32 * 64 * sizeof(int2), st>>>(A, 0);
    // wider bands (class 1) and the regions whose retry outgrew the 32-column ring (class 2, listed by the launch above): a wave
    // per region; queries beyond its LDS row go on to the lane-per-region launch with the row in HBM (class 3)
    aln_dp_wave_kernel<<<(unsigned)(cu_count * 24), 64, 0, st>>>(A, 1, A.n_list + 4);
    aln_dp_wave_kernel<<<(unsigned)(cu_count * 4), 64, 0, st>>>(A, 2, A.n_list + 5);
    aln_dp_kernel<0><<<(unsigned)(cu_count * 4), 64, 0, st>>>(A, 3);
}
void launch_aln_sizes(const RegAlnArgs &A, int64_t *wide, hipStream_t st) {
    aln_sizes_kernel<<<(unsigned)((A.n_regs + 1 + 255) / 256), 256, 0, st>>>(A, wide);
}
void launch_aln_gather(const RegAlnArgs &A, const int64_t *offs, uint32_t *cig, char *md, hipStream_t st) {
    if (A.n_regs > 0) aln_gather_kernel<<<(unsigned)((A.n_regs + 255) / 256), 256, 0, st>>>(A, offs, cig, md);
}

}  // namespace bwams
