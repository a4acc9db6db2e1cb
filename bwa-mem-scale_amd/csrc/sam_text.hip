// sam_text.hip — the single-end SAM text of a chunk on the device.
//
// Replaces, per read, what worker_sam's single-end branch (/root/reference/src/bwamem.cpp:1836-1844) runs after
// mem_mark_primary_se:
//   mem_reg2sam        /root/reference/src/bwamem.cpp:2091-2150      which regions become records, supplementary flag, mapq cap
//   mem_gen_alt        /root/reference/src/bwamem_extra.cpp:123-187  the XA strings (get_pri_idx)
//   mem_aln2sam        /root/reference/src/bwamem.cpp:2380-2531      the record itself (m = NULL, V17 build)
//   mem_approx_mapq_se /root/reference/src/bwamem.cpp:1983-2008      (mem_reg2aln's mapping quality)
// mem_reg2aln itself — CIGAR, NM, MD, position — ran before (reg2aln.hip); this stage reads its records and pools.
//
// Mapping.  Reads are independent and a read's text is a few hundred bytes assembled from many small pieces, so a lane owns a
// read and runs the reference's control flow twice: once counting bytes (sam_text_kernel<false>), an exclusive scan gives
// every read its place in one flat buffer, once writing (sam_text_kernel<true>).  Nothing is kept between the passes; the
// list of selected regions, the XA string of a region and the SA entries are recomputed where the reference looks them up
// (all O(regions of the read)).  HBM-bound in principle (~0.5 KB read + ~0.35 KB written per read), in practice bound by
// the byte-granular stores of a lane; it is a fraction of a millisecond per 10^5 reads either way.
//
// Mapping quality: mem_approx_mapq_se is a dozen double operations and three logarithms of small integers (the alignment
// length, seedcov, sub_n + 1).  The logarithms come from a table the host fills with the C library's log(), the arithmetic
// runs with floating-point contraction off, so the device value is the host's bit for bit (tests compare it against
// bwams_reg2aln_fetch, which computes it on the host).
#include <limits.h>
#include "common.h"
#include "chain_kernels.h"

namespace bwams {
namespace {

struct Writer {
    char *p;                 // nullptr: count only
    int64_t n;
    __device__ __forceinline__ void c(char ch) { if (p) p[n] = ch; ++n; }
    __device__ __forceinline__ void s(const char *q, int64_t len) {
        if (p) for (int64_t i = 0; i < len; ++i) p[n + i] = q[i];
        n += len;
    }
    __device__ __forceinline__ void z(const char *q) { while (*q) c(*q++); }
    __device__ void num(long long v) {                   // kputw / kputl (kstring.h:92-141)
        char buf[24];
        int l = 0;
        if (v == 0) { c('0'); return; }
        unsigned long long x = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        for (; x > 0; x /= 10) buf[l++] = (char)('0' + x % 10);
        if (v < 0) buf[l++] = '-';
        while (l > 0) c(buf[--l]);
    }
    // printf("%.3f") of a double: the exact binary value rounded to three decimals, ties to even (what glibc prints in the
    // default rounding mode)
    __device__ void f3(double x) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(x);
        if (bits >> 63) { c('-'); bits &= ~(1ull << 63); }
        const int ex = (int)(bits >> 52);
        unsigned long long m = bits & ((1ull << 52) - 1);
        int e;                                            // x = m * 2^e
        if (ex == 0) e = -1074; else { m |= 1ull << 52; e = ex - 1075; }
        unsigned long long N;
        if (e >= 0) N = (m << (e > 10 ? 10 : e)) * 1000ull;      // (not reached by score / alt_sc: quotients below 2^31)
        else {
            const unsigned long long P = m * 1000ull;    // < 2^63
            const int sh = -e;
            if (sh >= 64) N = 0;
            else {
                N = P >> sh;
                const unsigned long long rem = P & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
                if (rem > half || (rem == half && (N & 1ull))) ++N;
            }
        }
        num((long long)(N / 1000ull));
        c('.');
        const int fr = (int)(N % 1000ull);
        c((char)('0' + fr / 100)); c((char)('0' + fr / 10 % 10)); c((char)('0' + fr % 10));
    }
};

__device__ __forceinline__ bool reg_is_alt(const bwams_alnreg_t &p) { return (((uint32_t)p.n_comp_is_alt >> 30) & 3u) != 0; }

// mem_approx_mapq_se (bwamem.cpp:1983-2008); log(i) = lt[i]
__device__ int approx_mapq_se_dev(const bwams_mem_opt_t &opt, const bwams_alnreg_t &a, const double *__restrict__ lt, int lt_n,
                                  double coef_fac, unsigned long long *bad) {
#pragma clang fp contract(off)
    int mapq, l, sub = a.sub ? a.sub : opt.min_seed_len * opt.a;
    double identity;
    const int coef_len = opt.mapq_coef_len;
    sub = a.csub > sub ? a.csub : sub;
    if (sub >= a.score) return 0;
    l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
    if (l < 0 || l >= lt_n || a.seedcov < 0 || a.seedcov >= lt_n || a.sub_n + 1 >= lt_n) { atomicAdd(bad, 1ull); return 0; }
    identity = 1. - (double)(l * opt.a - a.score) / (opt.a + opt.b) / l;
    if (a.score == 0) mapq = 0;
    else if (coef_len > 0) {
        double tmp = l < coef_len ? 1. : coef_fac / lt[l];
        tmp *= identity * identity;
        mapq = (int)(6.02 * (a.score - sub) / opt.a * tmp * tmp + .499);
    } else {
        mapq = (int)(30.0 * (1. - (double)sub / a.score) * lt[a.seedcov] + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (a.sub_n > 0) mapq -= (int)(4.343 * lt[a.sub_n + 1] + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - a.frac_rep) + .499);
    return mapq;
}

__global__ void sam_mapq_kernel(SamArgs A) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_regs) return;
    const bwams_alnreg_t a = A.regs[k];
    A.mapq[k] = (A.rec[k].rid >= 0 && a.secondary < 0) ? approx_mapq_se_dev(A.opt, a, A.logtab, A.logtab_n, A.coef_fac, A.bad) : 0;
}

// the mate's record as mem_aln2sam reads it (rid, pos, strand, CIGAR, is_alt)
struct Mate {
    bool present;                // m != NULL
    int64_t pos;
    int32_t rid, is_rev, is_alt, n_cigar;
    int64_t cigar_off;
};

// one read's view
struct Read {
    const bwams_alnreg_t *a;     // its regions
    const bwams_aln_t *rec;      // their mem_reg2aln records
    const int32_t *mapq;
    int n;
    const uint8_t *seq;
    const char *qual;            // or nullptr
    int l_seq;
    const char *name; int l_name;
    const char *comment; int l_comment;      // l_comment 0: none
    // paired-end, paired branch (bwamem_pair.cpp:725-748): the region edits of mem_sam_pe as a view
    int sw_k, sw_z;              // secondary_all switch: regions pointing at sw_k (and sw_k itself) point at sw_z, sw_z at nothing; -1: none
    __device__ __forceinline__ int sec_all(int j) const {
        const int v = a[j].secondary_all;
        if (sw_k < 0) return v;
        if (j == sw_z) return -1;
        return (v == sw_k || j == sw_k) ? sw_z : v;
    }
};

// mem_reg2sam's filter (bwamem.cpp:2108-2116)
__device__ __forceinline__ bool selected(const SamArgs &A, const Read &R, int k) {
    const bwams_alnreg_t &p = R.a[k];
    if (p.score < A.sopt.T) return false;
    if (p.secondary >= 0 && (reg_is_alt(p) || !(A.sopt.flag & BWAMS_MEM_F_ALL))) return false;
    if (p.secondary >= 0 && p.secondary < INT_MAX && p.score < R.a[p.secondary].score * A.opt.drop_ratio) return false;
    return true;
}

// get_pri_idx (bwamem_extra.cpp:123-128)
__device__ __forceinline__ int pri_idx(const SamArgs &A, const Read &R, int i) {
    const int k = R.sec_all(i);
    if (k >= 0 && R.a[i].score >= R.a[k].score * (double)A.sopt.XA_drop_ratio) return k;
    return -1;
}

__device__ __forceinline__ const char *ctg_name(const SamArgs &A, int rid) { return A.ctg_names + A.ctg_off[rid]; }

// the mapping quality of the which-th selected record (bwamem.cpp:2123-2128)
__device__ __forceinline__ int capped_mapq(const SamArgs &A, const Read &R, int k, int which, int mapq0) {
    const int mq = R.mapq[k];
    if (!(A.sopt.flag & BWAMS_MEM_F_KEEP_SUPP_MAPQ) && which && !reg_is_alt(R.a[k]) && mq > mapq0) return mapq0;
    return mq;
}

__device__ void put_cigar(const SamArgs &A, Writer &W, const bwams_aln_t &t, const char *ops, int hard_rule, int which) {
    const uint32_t *cg = A.cig + t.cigar_off;
    for (int i = 0; i < t.n_cigar; ++i) {
        int c = (int)(cg[i] & 0xf);
        if (hard_rule && !(A.sopt.flag & BWAMS_MEM_F_SOFTCLIP) && !t.is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
        W.num((long long)(cg[i] >> 4));
        W.c(ops[c]);
    }
}

// the XA string of region r as mem_gen_alt builds it; returns whether anything was written
__device__ bool put_xa(const SamArgs &A, const Read &R, Writer &W, int r, bool with_tag) {
    int cnt = 0;
    bool has_alt = false;
    for (int i = 0; i < R.n; ++i)
        if (pri_idx(A, R, i) == r) { ++cnt; has_alt |= reg_is_alt(R.a[i]); }
    if (cnt == 0) return false;
    if (cnt > A.sopt.max_XA_hits_alt || (!has_alt && cnt > A.sopt.max_XA_hits)) return false;
    if (with_tag) W.s("\tXA:Z:", 6);
    for (int i = 0; i < R.n; ++i) {
        if (pri_idx(A, R, i) != r) continue;
        const bwams_aln_t &t = R.rec[i];
        W.z(ctg_name(A, t.rid));
        W.c(','); W.c("+-"[t.is_rev]); W.num(t.pos + 1);
        W.c(',');
        put_cigar(A, W, t, "MIDSHN", 0, 0);
        W.c(','); W.num(t.NM);
        W.c(';');
    }
    return true;
}

__device__ __forceinline__ int get_rlen(const SamArgs &A, int64_t cigar_off, int n_cigar) {        // bwamem.cpp:2639-2648
    int l = 0;
    for (int k = 0; k < n_cigar; ++k) {
        const uint32_t c = A.cig[cigar_off + k];
        if ((c & 0xf) == 0 || (c & 0xf) == 2) l += (int)(c >> 4);
    }
    return l;
}

// what the caller of mem_aln2sam decided about a record
struct RecInfo {
    int k;              // region (index within the read) or -1: the unaligned record
    int which;          // ordinal in the list handed to mem_aln2sam
    int flag, mapq, sub;
    int n_list;         // the list for the SA tag: 0 = the mem_reg2sam selection (recomputed), else list[0 .. n_list)
    int list[2];
    int list_mapq[2];
    bool xa;            // the record carries its region's XA string
};

// the XR tag of MEM_F_REF_HDR (bwamem.cpp:2522-2529, :2218-2225): the sequence's annotation, TABs as blanks
__device__ __forceinline__ void put_xr(const SamArgs &A, Writer &W, int rid) {
    if (!(A.sopt.flag & BWAMS_MEM_F_REF_HDR) || rid < 0) return;
    const char *a = A.ctg_annos + A.ctg_anno_off[rid];
    if (!a[0]) return;
    W.s("\tXR:Z:", 6);
    for (; *a; ++a) W.c(*a == '\t' ? ' ' : *a);
}

// mem_aln2sam (bwamem.cpp:2393-2531)
__device__ void put_record(const SamArgs &A, const Read &R, Writer &W, const RecInfo &I, const Mate &M_, int mapq0) {
    bwams_aln_t t;
    const int k = I.k, which = I.which;
    int flag = I.flag, mapq = I.mapq, sub = I.sub;
    if (k >= 0) t = R.rec[k];
    else {                                                // mem_reg2aln(ar = 0): everything zero but rid, pos, flag
        t.pos = -1; t.rid = -1; t.flag = 0x4; t.is_rev = t.is_alt = t.mapq = t.NM = t.n_cigar = t.md_len = 0;
        t.cigar_off = t.md_off = 0; t.score = t.sub = t.alt_sc = 0; t.pad_ = 0;
    }
    Mate m = M_;
    flag |= m.present ? 0x1 : 0;
    flag |= t.rid < 0 ? 0x4 : 0;
    flag |= m.present && m.rid < 0 ? 0x8 : 0;
    if (t.rid < 0 && m.present && m.rid >= 0) { t.rid = m.rid; t.pos = m.pos; t.is_rev = m.is_rev; t.n_cigar = 0; }
    if (m.present && m.rid < 0 && t.rid >= 0) { m.rid = t.rid; m.pos = t.pos; m.is_rev = t.is_rev; m.n_cigar = 0; }
    flag |= t.is_rev ? 0x10 : 0;
    flag |= m.present && m.is_rev ? 0x20 : 0;
    W.s(R.name, R.l_name); W.c('\t');
    W.num((flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)); W.c('\t');
    if (t.rid >= 0) {
        W.z(ctg_name(A, t.rid)); W.c('\t');
        W.num(t.pos + 1); W.c('\t');
        W.num(mapq); W.c('\t');
        if (t.n_cigar) put_cigar(A, W, t, "MIDSH", 1, which);
        else W.c('*');
    } else W.s("*\t0\t0\t*", 7);
    W.c('\t');
    if (m.present && m.rid >= 0) {
        if (t.rid == m.rid) W.c('='); else W.z(ctg_name(A, m.rid));
        W.c('\t');
        W.num(m.pos + 1); W.c('\t');
        if (t.rid == m.rid) {
            const int64_t p0 = t.pos + (t.is_rev ? get_rlen(A, t.cigar_off, t.n_cigar) - 1 : 0);
            const int64_t p1 = m.pos + (m.is_rev ? get_rlen(A, m.cigar_off, m.n_cigar) - 1 : 0);
            if (m.n_cigar == 0 || t.n_cigar == 0) W.c('0');
            else W.num(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
        } else W.c('0');
    } else W.s("*\t0\t0", 5);
    W.c('\t');
    if (flag & 0x100) {
        W.s("*\t*", 3);
    } else {
        int qb = 0, qe = R.l_seq;
        if (t.n_cigar && which && !(A.sopt.flag & BWAMS_MEM_F_SOFTCLIP) && !t.is_alt) {
            const uint32_t c0 = A.cig[t.cigar_off], c1 = A.cig[t.cigar_off + t.n_cigar - 1];
            const int lead = ((c0 & 0xf) == 4 || (c0 & 0xf) == 3) ? (int)(c0 >> 4) : 0;
            const int tail = ((c1 & 0xf) == 4 || (c1 & 0xf) == 3) ? (int)(c1 >> 4) : 0;
            if (!t.is_rev) { qb += lead; qe -= tail; } else { qe -= lead; qb += tail; }
        }
        if (!t.is_rev) {
            for (int i = qb; i < qe; ++i) W.c("ACGTN"[R.seq[i]]);
            W.c('\t');
            if (R.qual) W.s(R.qual + qb, qe > qb ? qe - qb : 0); else W.c('*');
        } else {
            for (int i = qe - 1; i >= qb; --i) W.c("TGCAN"[R.seq[i]]);
            W.c('\t');
            if (R.qual) { for (int i = qe - 1; i >= qb; --i) W.c(R.qual[i]); } else W.c('*');
        }
    }
    if (t.n_cigar) {
        W.s("\tNM:i:", 6); W.num(t.NM);
        W.s("\tMD:Z:", 6); W.s(A.md + t.md_off, t.md_len > 0 ? t.md_len - 1 : 0);
    }
    if (m.present && m.n_cigar) {                          // MC:Z (V17): the mate's CIGAR under this record's clipping rule
        W.s("\tMC:Z:", 6);
        bwams_aln_t mt;
        mt.cigar_off = m.cigar_off; mt.n_cigar = m.n_cigar; mt.is_alt = m.is_alt;
        put_cigar(A, W, mt, "MIDSH", 1, which);
    }
    if (t.score >= 0) { W.s("\tAS:i:", 6); W.num(t.score); }
    if (sub >= 0) { W.s("\tXS:i:", 6); W.num(sub); }
    if (A.sopt.rg_id[0]) { W.s("\tRG:Z:", 6); W.z(A.sopt.rg_id); }
    if (!(flag & 0x100)) {
        if (I.n_list == 0) {                              // the list is mem_reg2sam's selection
            bool other = false;
            if (k >= 0)
                for (int i = 0; i < R.n && !other; ++i)
                    other = i != k && selected(A, R, i) && !(R.rec[i].flag & 0x100);
            if (other) {
                W.s("\tSA:Z:", 6);
                int wi = 0;
                for (int i = 0; i < R.n; ++i) {
                    if (!selected(A, R, i)) continue;
                    const int w_i = wi++;
                    const bwams_aln_t &r = R.rec[i];
                    if (i == k || (r.flag & 0x100)) continue;
                    W.z(ctg_name(A, r.rid)); W.c(',');
                    W.num(r.pos + 1); W.c(',');
                    W.c("+-"[r.is_rev]); W.c(',');
                    put_cigar(A, W, r, "MIDSH", 0, 0);
                    W.c(','); W.num(capped_mapq(A, R, i, w_i, mapq0));
                    W.c(','); W.num(r.NM);
                    W.c(';');
                }
            }
        } else if (I.n_list > 1) {                        // mem_sam_pe's list of at most two records, none of them 0x100
            W.s("\tSA:Z:", 6);
            for (int i = 0; i < I.n_list; ++i) {
                if (i == which) continue;
                const bwams_aln_t &r = R.rec[I.list[i]];
                W.z(ctg_name(A, r.rid)); W.c(',');
                W.num(r.pos + 1); W.c(',');
                W.c("+-"[r.is_rev]); W.c(',');
                put_cigar(A, W, r, "MIDSH", 0, 0);
                W.c(','); W.num(I.list_mapq[i]);
                W.c(','); W.num(r.NM);
                W.c(';');
            }
        }
        if (t.alt_sc > 0) { W.s("\tpa:f:", 6); W.f3((double)t.score / t.alt_sc); }
    }
    if (k >= 0 && I.xa) put_xa(A, R, W, k, true);
    if (R.l_comment) { W.c('\t'); W.s(R.comment, R.l_comment); }
    put_xr(A, W, t.rid);
    W.c('\n');
}

// mem_reg2sam (bwamem.cpp:2091-2150) with extra_flag and the mate record m
__device__ void put_reg2sam(const SamArgs &A, const Read &R, Writer &W, int extra_flag, const Mate &m) {
    int first = -1;
    for (int k = 0; k < R.n && first < 0; ++k)
        if (selected(A, R, k)) first = k;
    RecInfo I;
    I.n_list = 0; I.list[0] = I.list[1] = 0; I.list_mapq[0] = I.list_mapq[1] = 0;
    I.xa = !(A.sopt.flag & BWAMS_MEM_F_ALL);
    if (first < 0) {
        I.k = -1; I.which = 0; I.flag = 0x4 | extra_flag; I.mapq = 0; I.sub = 0;
        put_record(A, R, W, I, m, 0);
        return;
    }
    const int mapq0 = R.mapq[first];
    int which = 0;
    for (int k = first; k < R.n; ++k) {
        if (!selected(A, R, k)) continue;
        I.k = k; I.which = which;
        I.flag = R.rec[k].flag | extra_flag;
        if (which && R.a[k].secondary < 0) I.flag |= (A.sopt.flag & BWAMS_MEM_F_NO_MULTI) ? 0x10000 : 0x800;
        I.mapq = capped_mapq(A, R, k, which, mapq0);
        I.sub = R.a[k].secondary >= 0 ? -1 : R.rec[k].sub;
        put_record(A, R, W, I, m, mapq0);
        ++which;
    }
}

__device__ __forceinline__ void load_read_regs(const SamArgs &A, int64_t r, Read &R) {      // what the marking needs: the regions alone
    const int64_t o = A.reg_off[r];
    R.a = A.regs + o; R.rec = nullptr; R.mapq = nullptr; R.n = (int)(A.reg_off[r + 1] - o);
    R.seq = nullptr; R.l_seq = 0; R.qual = nullptr; R.name = nullptr; R.l_name = 0; R.comment = nullptr; R.l_comment = 0;
    R.sw_k = R.sw_z = -1;
}
__device__ __forceinline__ void load_read(const SamArgs &A, int64_t r, Read &R) {
    const int64_t o = A.reg_off[r];
    R.a = A.regs + o; R.rec = A.rec + o; R.mapq = A.mapq + o; R.n = (int)(A.reg_off[r + 1] - o);
    R.seq = A.enc + A.cum[r]; R.l_seq = (int)(A.cum[r + 1] - A.cum[r]);
    R.qual = (A.quals && R.l_seq > 0) ? A.quals + A.cum[r] : nullptr;      // kseq2bseq1: an empty quality string is no quality string
    R.name = A.names + A.name_off[r]; R.l_name = (int)(A.name_off[r + 1] - A.name_off[r]);
    R.comment = A.comments ? A.comments + A.comment_off[r] : nullptr;
    R.l_comment = A.comments ? (int)(A.comment_off[r + 1] - A.comment_off[r]) : 0;
    R.sw_k = R.sw_z = -1;
}

__device__ __forceinline__ Mate mate_of(const Read &R, int k) {          // mem_reg2aln of region k (or of nothing) as a mate record
    Mate m;
    m.present = true;
    if (k >= 0) {
        const bwams_aln_t &t = R.rec[k];
        m.pos = t.pos; m.rid = t.rid; m.is_rev = t.is_rev; m.is_alt = t.is_alt; m.n_cigar = t.n_cigar; m.cigar_off = t.cigar_off;
    } else { m.pos = -1; m.rid = -1; m.is_rev = 0; m.is_alt = 0; m.n_cigar = 0; m.cigar_off = 0; }
    return m;
}

// mem_infer_dir (bwamem_pair.cpp:57-65)
__device__ __forceinline__ int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist) {
    const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
    const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

__device__ __forceinline__ int raw_mapq(int diff, int a) {               // bwamem_pair.cpp:432
#pragma clang fp contract(off)
    return (int)(6.02 * diff / a + .499);
}

// What mem_sam_pe decides for a pair before any text is written (bwamem_pair.cpp:686-748): which branch, the two regions,
// their mapping qualities, the region edits.  Both ends' lanes evaluate it; it reads both ends' regions.
struct PairPlan {
    bool paired;                 // the paired branch (else no_pairing)
    int z[2], q_se[2];
    int sub[2];                  // the chosen regions' sub after the edit (mem_reg2aln's a.sub = max(sub, csub))
    bool was_secondary[2];       // secondary >= 0 before the edit (it becomes -2: no 0x100)
    int sw_k[2];                 // secondary_all switch per end (-1: none)
    int extra_flag;
};

__device__ void plan_pair(const SamArgs &A, const Read R[2], const bwams_pair_t &pr, PairPlan &P) {
#pragma clang fp contract(off)
    const int o = pr.score;
    P.paired = false; P.extra_flag = 1;
    P.z[0] = pr.z[0]; P.z[1] = pr.z[1];
    P.sw_k[0] = P.sw_k[1] = -1;
    if (!(pr.n_pri[0] && pr.n_pri[1] && o > 0)) return;
    for (int i = 0; i < 2; ++i) {
        int j;
        for (j = 1; j < pr.n_pri[i]; ++j)
            if (R[i].a[j].secondary < 0 && R[i].a[j].score >= A.sopt.T) break;
        if (j < pr.n_pri[i]) return;                       // is_multi: no_pairing
    }
    const int score_un = R[0].a[0].score + R[1].a[0].score - A.opt.pen_unpaired;
    int subo = pr.sub > score_un ? pr.sub : score_un;
    int q_pe = raw_mapq(o - subo, A.opt.a);
    if (pr.n_sub > 0) {
        if (pr.n_sub + 1 >= A.logtab_n) { atomicAdd(A.bad, 1ull); return; }
        q_pe -= (int)(4.343 * A.logtab[pr.n_sub + 1] + .499);
    }
    if (q_pe < 0) q_pe = 0;
    if (q_pe > 60) q_pe = 60;
    q_pe = (int)(q_pe * (1. - .5 * (R[0].a[0].frac_rep + R[1].a[0].frac_rep)) + .499);
    P.paired = true;
    if (o > score_un) {
        for (int i = 0; i < 2; ++i) {
            bwams_alnreg_t c = R[i].a[P.z[i]];
            P.was_secondary[i] = c.secondary >= 0;
            if (c.secondary >= 0) { c.sub = R[i].a[c.secondary].score; c.secondary = -2; }
            P.sub[i] = c.sub > c.csub ? c.sub : c.csub;
            int q = approx_mapq_se_dev(A.opt, c, A.logtab, A.logtab_n, A.coef_fac, A.bad);
            q = q > q_pe ? q : q_pe < q + 40 ? q_pe : q + 40;
            const int cap = raw_mapq(c.score - c.csub, A.opt.a);
            P.q_se[i] = q < cap ? q : cap;
        }
        P.extra_flag |= 2;
    } else {
        P.z[0] = P.z[1] = 0;
        for (int i = 0; i < 2; ++i) {
            const bwams_alnreg_t &c = R[i].a[0];
            P.was_secondary[i] = c.secondary >= 0;
            P.sub[i] = c.sub > c.csub ? c.sub : c.csub;
            P.q_se[i] = approx_mapq_se_dev(A.opt, c, A.logtab, A.logtab_n, A.coef_fac, A.bad);
        }
    }
    for (int i = 0; i < 2; ++i) {
        const int k = R[i].a[P.z[i]].secondary_all;
        if (k >= 0 && k < pr.n_pri[i]) P.sw_k[i] = k;
    }
}

// mem_perfect2sam_cont + mem_aln2sam_perfect (bwamem.cpp:2280-2325, :2153-2227): the records of a read the EMF resolved, from the
// regions mem_perfect2reg left for it (one per location of get_perfect_locations after perfect_dedup_patch, in that order)
__device__ void put_perfect(const SamArgs &A, const Read &R, Writer &W, const bwams_alnreg_t *regs, int n) {
    int n_out = 0;
    const bool all = (A.sopt.flag & BWAMS_MEM_F_ALL) != 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !(n_out == 0 || all)) break;
        for (int k = 0; k < n; ++k) {
            const bwams_alnreg_t &r = regs[k];
            if ((int)reg_is_alt(r) != pass) continue;
            const bool is_rev = r.rb >= A.bns_l_pac, secondary = n_out > 0;
            int64_t pos = is_rev ? (A.bns_l_pac << 1) - r.re : r.rb;
            if (R.l_seq != A.er_seed_len && is_rev) pos -= R.l_seq - A.er_seed_len;
            pos -= A.contigs[r.rid].offset;
            const int flag = (is_rev ? 0x10 : 0) | (secondary ? 0x100 : 0);
            W.s(R.name, R.l_name); W.c('\t');
            W.num(flag); W.c('\t');
            W.z(ctg_name(A, r.rid)); W.c('\t');
            W.num(pos + 1); W.c('\t');
            W.num(60); W.c('\t');
            W.num(R.l_seq); W.c('M');
            W.c('\t');
            W.s("*\t0\t0", 5);
            W.c('\t');
            if (secondary) W.s("*\t*", 3);
            else if (!is_rev) {
                for (int i = 0; i < R.l_seq; ++i) W.c("ACGTN"[R.seq[i]]);
                W.c('\t');
                if (R.qual) W.s(R.qual, R.l_seq); else W.c('*');
            } else {
                for (int i = R.l_seq - 1; i >= 0; --i) W.c("TGCAN"[R.seq[i]]);
                W.c('\t');
                if (R.qual) { for (int i = R.l_seq - 1; i >= 0; --i) W.c(R.qual[i]); } else W.c('*');
            }
            W.s("\tNM:i:", 6); W.num(0);
            W.s("\tMD:Z:", 6); W.num(R.l_seq);
            W.s("\tAS:i:", 6); W.num((long long)R.l_seq * A.opt.a);
            if (!secondary) { W.s("\tXS:i:", 6); W.num((k == 0 && n > 1) ? (long long)R.l_seq * A.opt.a : 0); }
            if (A.sopt.rg_id[0]) { W.s("\tRG:Z:", 6); W.z(A.sopt.rg_id); }
            if (R.l_comment) { W.c('\t'); W.s(R.comment, R.l_comment); }
            put_xr(A, W, r.rid);
            W.c('\n');
            ++n_out;
            if (!all) break;
        }
    }
}

// ---- which regions the text needs -----------------------------------------------------------------------------------------
// The reference calls mem_reg2aln only for what it prints: the records mem_reg2sam selects, the members of the XA strings of the
// printed records (mem_gen_alt skips a primary with too many candidates), and in mem_sam_pe the paired regions, the ALT hit and the
// mate's record.  All of it follows from the regions, the pairing result and the options alone — no alignment is consulted — so the
// set is computed before mem_reg2aln runs and the other regions (two thirds of them on the bench chunk) are never aligned.  The
// marking walks the same control flow as the text kernels below.
__device__ void need_xa(const SamArgs &A, const Read &R, uint8_t *need, int r) {
    int cnt = 0;
    bool has_alt = false;
    for (int i = 0; i < R.n; ++i)
        if (pri_idx(A, R, i) == r) { ++cnt; has_alt |= reg_is_alt(R.a[i]); }
    if (cnt == 0 || cnt > A.sopt.max_XA_hits_alt || (!has_alt && cnt > A.sopt.max_XA_hits)) return;
    for (int i = 0; i < R.n; ++i)
        if (pri_idx(A, R, i) == r) need[i] = 1;
}
__device__ void need_reg2sam(const SamArgs &A, const Read &R, uint8_t *need) {
    for (int k = 0; k < R.n; ++k)
        if (selected(A, R, k)) {
            need[k] = 1;
            if (!(A.sopt.flag & BWAMS_MEM_F_ALL)) need_xa(A, R, need, k);
        }
}

__global__ void sam_need_kernel(SamArgs A, uint8_t *need) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.nseq; r += (int64_t)gridDim.x * blockDim.x) {
        if (!A.pairs) {
            Read R;
            load_read_regs(A, r, R);
            need_reg2sam(A, R, need + A.reg_off[r]);
            continue;
        }
        if (r & 1) continue;                               // the lane of the pair's first read marks both ends
        Read R[2];
        load_read_regs(A, r, R[0]);
        load_read_regs(A, r + 1, R[1]);
        const bwams_pair_t pr = A.pairs[r >> 1];
        bool paired = pr.n_pri[0] && pr.n_pri[1] && pr.score > 0;
        for (int i = 0; i < 2 && paired; ++i)
            for (int j = 1; j < pr.n_pri[i]; ++j)
                if (R[i].a[j].secondary < 0 && R[i].a[j].score >= A.sopt.T) { paired = false; break; }
        if (paired) {
            const int score_un = R[0].a[0].score + R[1].a[0].score - A.opt.pen_unpaired;
            const bool pe = pr.score > score_un;
            for (int i = 0; i < 2; ++i) {
                uint8_t *nd = need + A.reg_off[r + i];
                const int z = pe ? pr.z[i] : 0;
                nd[z] = 1;
                const int k = R[i].a[z].secondary_all;
                R[i].sw_k = (k >= 0 && k < pr.n_pri[i]) ? k : -1;
                R[i].sw_z = z;
                if (pr.n_pri[i] < R[i].n) {
                    const bwams_alnreg_t &p = R[i].a[pr.n_pri[i]];
                    if (!(p.score < A.sopt.T || p.secondary >= 0 || !reg_is_alt(p))) {
                        nd[pr.n_pri[i]] = 1;
                        if (!(A.sopt.flag & BWAMS_MEM_F_ALL)) need_xa(A, R[i], nd, pr.n_pri[i]);
                    }
                }
                if (!(A.sopt.flag & BWAMS_MEM_F_ALL)) need_xa(A, R[i], nd, z);
            }
        } else {
            for (int i = 0; i < 2; ++i) {
                uint8_t *nd = need + A.reg_off[r + i];
                if (R[i].n) {
                    if (R[i].a[0].score >= A.sopt.T) nd[0] = 1;
                    else if (pr.n_pri[i] < R[i].n && R[i].a[pr.n_pri[i]].score >= A.sopt.T) nd[pr.n_pri[i]] = 1;
                }
                need_reg2sam(A, R[i], nd);
            }
        }
    }
}

template <bool EMIT>
__global__ void sam_text_kernel(SamArgs A) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.nseq; r += (int64_t)gridDim.x * blockDim.x) {
        Read R;
        load_read(A, r, R);
        Writer W;
        W.p = EMIT ? A.out + A.out_off[r] : nullptr;
        W.n = 0;
        Mate none;
        none.present = false; none.pos = -1; none.rid = -1; none.is_rev = none.is_alt = none.n_cigar = 0; none.cigar_off = 0;
        const int n_er = A.er_regs ? (int)(A.er_off[r + 1] - A.er_off[r]) : 0;
        if (n_er > 0) put_perfect(A, R, W, A.er_regs + A.er_off[r], n_er);      // worker_sam: perfect.exist -> mem_perfect2sam_cont, continue
        else put_reg2sam(A, R, W, 0, none);
        if (!EMIT) A.len[r] = W.n;
    }
}

// mem_sam_pe from mem_pair's result on (bwamem_pair.cpp:686-833); a lane per READ, both lanes of a pair take the pair's decisions
template <bool EMIT>
__global__ void sam_text_pe_kernel(SamArgs A) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.nseq; r += (int64_t)gridDim.x * blockDim.x) {
        const int me = (int)(r & 1);
        Read R[2];
        load_read(A, r - me, R[0]);
        load_read(A, r - me + 1, R[1]);
        const bwams_pair_t pr = A.pairs[r >> 1];
        if (!EMIT && me == 0) {                            // mem_sam_pe: "paired reads have different names" is fatal (bwamem_pair.cpp:797)
            bool same = R[0].l_name == R[1].l_name;
            for (int i = 0; same && i < R[0].l_name; ++i) same = R[0].name[i] == R[1].name[i];
            if (!same) atomicAdd(A.bad + 2, 1ull);
        }
        PairPlan P;
        plan_pair(A, R, pr, P);
        Writer W;
        W.p = EMIT ? A.out + A.out_off[r] : nullptr;
        W.n = 0;
        if (P.paired) {
            R[0].sw_k = P.sw_k[0]; R[0].sw_z = P.z[0];
            R[1].sw_k = P.sw_k[1]; R[1].sw_z = P.z[1];
            const Read &S = R[me];
            const int i = me;
            RecInfo I;
            I.n_list = 1; I.list[0] = P.z[i]; I.list_mapq[0] = P.q_se[i]; I.list[1] = 0; I.list_mapq[1] = 0;
            bool alt_hit = false;
            if (pr.n_pri[i] < S.n) {                       // the read has ALT hits
                const bwams_alnreg_t &p = S.a[pr.n_pri[i]];
                alt_hit = !(p.score < A.sopt.T || p.secondary >= 0 || !reg_is_alt(p));
            }
            if (alt_hit) { I.n_list = 2; I.list[1] = pr.n_pri[i]; I.list_mapq[1] = S.rec[pr.n_pri[i]].rid >= 0 ? S.mapq[pr.n_pri[i]] : 0; }
            I.xa = !(A.sopt.flag & BWAMS_MEM_F_ALL);
            const Mate m = mate_of(R[!me], P.z[!me]);
            // h[i]
            I.k = P.z[i]; I.which = 0;
            I.flag = (S.rec[P.z[i]].flag & ~0x100) | (0x40 << i) | P.extra_flag;      // the edited region is never secondary >= 0
            I.mapq = P.q_se[i]; I.sub = P.sub[i];
            put_record(A, S, W, I, m, 0);
            if (alt_hit) {                                 // g[i]
                const int k = pr.n_pri[i];
                I.k = k; I.which = 1;
                I.flag = S.rec[k].flag | 0x800 | (0x40 << i) | P.extra_flag;
                I.mapq = S.mapq[k]; I.sub = S.rec[k].sub;
                put_record(A, S, W, I, m, 0);
            }
        } else {
            int extra_flag = 1;
            int which[2];
            for (int i = 0; i < 2; ++i) {
                which[i] = -1;
                if (R[i].n) {
                    if (R[i].a[0].score >= A.sopt.T) which[i] = 0;
                    else if (pr.n_pri[i] < R[i].n && R[i].a[pr.n_pri[i]].score >= A.sopt.T) which[i] = pr.n_pri[i];
                }
            }
            const int rid0 = which[0] >= 0 ? R[0].rec[which[0]].rid : -1, rid1 = which[1] >= 0 ? R[1].rec[which[1]].rid : -1;
            if (!(A.sopt.flag & BWAMS_MEM_F_NOPAIRING) && rid0 == rid1 && rid0 >= 0) {      // bwamem_pair.cpp:1176
                int64_t dist;
                const int d = infer_dir(A.bns_l_pac, R[0].a[0].rb, R[1].a[0].rb, &dist);
                if (!A.pes[d].failed && dist >= A.pes[d].low && dist <= A.pes[d].high) extra_flag |= 2;
            }
            const Mate m = mate_of(R[!me], which[!me]);
            put_reg2sam(A, R[me], W, (me ? 0x81 : 0x41) | extra_flag, m);
        }
        if (!EMIT) A.len[r] = W.n;
    }
}

}  // namespace

void launch_sam_need(const SamArgs &A, uint8_t *need, int cu_count, hipStream_t st) {
    if (A.nseq <= 0) return;
    int64_t blocks = (A.nseq + 63) / 64;
    const int64_t cap = (int64_t)cu_count * 32;
    if (blocks > cap) blocks = cap;
    sam_need_kernel<<<(unsigned)blocks, 64, 0, st>>>(A, need);
}
void launch_sam_mapq(const SamArgs &A, hipStream_t st) {
    if (A.n_regs > 0) sam_mapq_kernel<<<(unsigned)((A.n_regs + 255) / 256), 256, 0, st>>>(A);
}
void launch_sam_text(const SamArgs &A, bool emit, int cu_count, hipStream_t st) {
    if (A.nseq <= 0) return;
    int64_t blocks = (A.nseq + 63) / 64;
    const int64_t cap = (int64_t)cu_count * 32;
    if (blocks > cap) blocks = cap;
    if (A.pairs) {
        if (emit) sam_text_pe_kernel<true><<<(unsigned)blocks, 64, 0, st>>>(A);
        else sam_text_pe_kernel<false><<<(unsigned)blocks, 64, 0, st>>>(A);
    } else if (emit) sam_text_kernel<true><<<(unsigned)blocks, 64, 0, st>>>(A);
    else sam_text_kernel<false><<<(unsigned)blocks, 64, 0, st>>>(A);
}

}  // namespace bwams
