// emf_regs.hip — the regions of the reads the exact-match filter resolved.
//
// mem_perfect2reg with get_perfect_locations and perfect_dedup_patch
// (/root/reference/src/perfect_map.cpp:659-869): an exactly matching read has one location, or the
// list of locations its seed entry carries (forward and reverse-complement occurrences; for reads
// longer than the table's L each listed location is verified on the tail); locations closer than
// 95 % of a read length on the same strand of the same sequence are collapsed; every survivor
// becomes a full-length mem_alnreg_t (score = l_seq * a).  One lane per read; a read's scratch is
// sized by the length of its list (count -> scan -> fill + collapse -> scan -> emit).
#include "common.h"
#include "chain_kernels.h"

namespace bwams {
namespace {

struct AlnP { int64_t loc, pos; int32_t rid; int32_t rev_alt; };        // mem_aln_perfect_t: is_rev bit 0, is_alt bit 1

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
__device__ __forceinline__ bool tail_matches(const DevEmf &t, uint32_t loc, const uint8_t *seed, bool is_rev, int len) {
    const int L = t.seed_len;
    len -= L;
    if (!is_rev) {
        if (loc + (uint32_t)len >= t.seq_len) return false;
        for (int i = 0; i < len; ++i)
            if (t.ref[loc + L + i] != seed[L + i]) return false;
        return true;
    }
    if (loc < (uint32_t)len) return false;
    for (int i = 0; i < len; ++i)
        if (t.ref[loc - len + i] != 3 - seed[L + len - 1 - i]) return false;
    return true;
}
__device__ __forceinline__ void list_of(const DevEmf &t, uint32_t multi, uint32_t &nfw, uint32_t &nrc, uint32_t &base) {
    const uint32_t first = t.loc_table[multi];
    const bool many = (first & 0x80000000u) != 0;
    const uint32_t st = many ? (first & 0x7fffffffu) : multi;
    if (!many) { nfw = (t.loc_table[st] >> 16) & 0xffff; nrc = t.loc_table[st] & 0xffff; base = st + 1; }
    else { nfw = t.loc_table[st]; nrc = t.loc_table[st + 1]; base = st + 2; }
}

// locations an entry can expand to (__get_num_location, perfect.h:148-161)
__global__ void emfregs_count_kernel(EmfRegArgs A, int64_t *wide) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > A.nseq) return;
    int64_t m = 0;
    if (r < A.nseq && (A.code[r] == 3 || A.code[r] == 4)) {
        const uint32_t multi = A.perfect[2 * r] >> 2;
        m = 1;
        if (multi) { uint32_t nfw, nrc, base; list_of(A.t, multi, nfw, nrc, base); m = 1 + (int64_t)nfw + (int64_t)nrc; }
    }
    wide[r] = m;
}

__device__ __forceinline__ void init_aln(const EmfRegArgs &A, AlnP *a, int64_t pos, int len, bool is_rev) {
    a->loc = pos;
    a->rid = pos2rid(A.bns, pos);
    if (len != A.t.seed_len && is_rev) pos = pos - (len - A.t.seed_len);
    a->pos = pos - A.bns.contigs[a->rid].offset;
    a->rev_alt = (is_rev ? 1 : 0) | (A.bns.contigs[a->rid].is_alt != 0 ? 2 : 0);
}
__device__ __forceinline__ int init_multi(const EmfRegArgs &A, AlnP *av, int n, uint32_t num, uint32_t base, const uint8_t *seq, int l_seq,
                                          bool is_rev, uint32_t matched) {
    for (uint32_t i = 0; i < num; ++i) {
        const uint32_t loc = A.t.loc_table[base + (is_rev ? num - 1 - i : i)];
        if (loc == matched) continue;
        if (A.t.seed_len == l_seq || tail_matches(A.t, loc, seq, is_rev, l_seq)) init_aln(A, &av[n++], (int64_t)loc, l_seq, is_rev);
    }
    return n;
}

// get_perfect_locations + perfect_dedup_patch, in the read's scratch slice
__global__ void emfregs_fill_kernel(EmfRegArgs A, const int64_t *__restrict__ scr_off, int32_t *n_final, uint8_t *first_is_rev) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    n_final[r] = 0;
    first_is_rev[r] = 0;
    if (!(A.code[r] == 3 || A.code[r] == 4)) return;
    AlnP *av = reinterpret_cast<AlnP *>(A.scratch) + scr_off[r];
    const uint32_t flags = A.perfect[2 * r], location = A.perfect[2 * r + 1];
    const bool rc_matched = (flags & 2u) != 0;
    const uint32_t multi = flags >> 2;
    const uint8_t *seq = A.enc + A.cum[r];
    const int l_seq = (int)(A.cum[r + 1] - A.cum[r]);
    int n = 0;
    if (!multi) init_aln(A, &av[n++], (int64_t)location, l_seq, rc_matched);
    else {
        uint32_t nfw, nrc, base;
        list_of(A.t, multi, nfw, nrc, base);
        if (!rc_matched) {
            init_aln(A, &av[n++], (int64_t)location, l_seq, false);
            n = init_multi(A, av, n, nfw, base, seq, l_seq, false, location);
            n = init_multi(A, av, n, nrc, base + nfw, seq, l_seq, true, location);
        } else {
            n = init_multi(A, av, n, nrc, base + nfw, seq, l_seq, false, location);
            init_aln(A, &av[n++], (int64_t)location, l_seq, true);
            n = init_multi(A, av, n, nfw, base, seq, l_seq, true, location);
        }
    }
    if (n > 1) {
        for (int i = 1; i < n; ++i) {
            const AlnP p = av[i];
            const AlnP pr = av[i - 1];
            if (p.rid != pr.rid || (p.rev_alt & 1) != (pr.rev_alt & 1) || p.pos >= pr.pos + l_seq + A.opt.max_chain_gap) continue;
            for (int j = i - 1; j >= 0; --j) {
                AlnP *q = &av[j];
                if (!(p.rid == q->rid && (p.rev_alt & 1) == (q->rev_alt & 1) && p.pos < q->pos + l_seq + A.opt.max_chain_gap)) break;
                if (q->rid < 0) continue;          // (an excluded entry has rid = -1 and ends the scan just above, as in the reference)
                if ((float)(q->pos + l_seq - p.pos) > A.opt.mask_level_redun * (float)l_seq) q->rid = -1;
            }
        }
        int k = 0;
        for (int i = 0; i < n; ++i)
            if (av[i].rid >= 0) { if (k != i) av[k] = av[i]; ++k; }
        n = k;
    }
    n_final[r] = n;
    first_is_rev[r] = (uint8_t)(av[0].rev_alt & 1);
}

__global__ void emfregs_widen_kernel(const int32_t *a, int64_t n, int64_t *wide) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n) return;
    wide[g] = g < n ? (int64_t)a[g] : 0;
}

// mem_perfect2reg (perfect_map.cpp:817-867)
__global__ void emfregs_emit_kernel(EmfRegArgs A, const int64_t *__restrict__ scr_off, const int32_t *__restrict__ n_final,
                                    const int64_t *__restrict__ out_off, bwams_alnreg_t *out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nseq) return;
    const int n = n_final[r];
    const AlnP *av = reinterpret_cast<const AlnP *>(A.scratch) + scr_off[r];
    const int l_seq = (int)(A.cum[r + 1] - A.cum[r]);
    for (int i = 0; i < n; ++i) {
        const AlnP p = av[i];
        bwams_alnreg_t x;
        if (!(p.rev_alt & 1)) { x.rb = p.loc; x.re = p.loc + l_seq; }
        else { x.rb = (A.bns.l_pac << 1) - (p.loc + l_seq); x.re = (A.bns.l_pac << 1) - p.loc; }
        x.qb = 0; x.qe = l_seq; x.rid = p.rid; x.pad0_ = 0; x.chain = 0;
        x.score = x.truesc = l_seq * A.opt.a;
        x.sub = x.alt_sc = x.csub = x.sub_n = 0;
        x.w = A.opt.w; x.seedcov = 0; x.secondary = x.secondary_all = 0;
        x.seedlen0 = l_seq;
        x.n_comp_is_alt = 1 | ((p.rev_alt & 2) ? (1 << 30) : 0);
        x.frac_rep = 0.f; x.pad1_ = 0; x.hash = 0; x.flg = 0; x.pad2_ = 0;
        out[out_off[r] + i] = x;
    }
}

// worker_sam's paired-end branch gives a resolved end its regions (mem_perfect2reg) before mem_sam_pe: the two region lists as one
__global__ void emfregs_merge_count_kernel(const int64_t *__restrict__ off_a, const int64_t *__restrict__ off_b, int64_t nseq, int64_t *wide) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nseq) return;
    wide[r] = r < nseq ? (off_a[r + 1] - off_a[r]) + (off_b[r + 1] - off_b[r]) : 0;
}
__global__ void emfregs_merge_kernel(const bwams_alnreg_t *__restrict__ a, const int64_t *__restrict__ off_a, const bwams_alnreg_t *__restrict__ bb,
                                     const int64_t *__restrict__ off_b, int64_t nseq, const int64_t *__restrict__ off_o, bwams_alnreg_t *out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nseq) return;
    int64_t o = off_o[r];
    for (int64_t i = off_a[r]; i < off_a[r + 1]; ++i) out[o++] = a[i];
    for (int64_t i = off_b[r]; i < off_b[r + 1]; ++i) out[o++] = bb[i];
}

}  // namespace

void launch_emfregs_merge_count(const int64_t *off_a, const int64_t *off_b, int64_t nseq, int64_t *wide, hipStream_t st) {
    emfregs_merge_count_kernel<<<(unsigned)((nseq + 256) / 256), 256, 0, st>>>(off_a, off_b, nseq, wide);
}
void launch_emfregs_merge(const bwams_alnreg_t *a, const int64_t *off_a, const bwams_alnreg_t *b, const int64_t *off_b, int64_t nseq,
                          const int64_t *off_o, bwams_alnreg_t *out, hipStream_t st) {
    if (nseq > 0) emfregs_merge_kernel<<<(unsigned)((nseq + 255) / 256), 256, 0, st>>>(a, off_a, b, off_b, nseq, off_o, out);
}

size_t emfregs_scratch_bytes(int64_t n) { return (size_t)(n + 1) * sizeof(AlnP); }

void launch_emfregs_count(const EmfRegArgs &A, int64_t *wide, hipStream_t st) {
    emfregs_count_kernel<<<(unsigned)((A.nseq + 256) / 256), 256, 0, st>>>(A, wide);
}
void launch_emfregs_fill(const EmfRegArgs &A, const int64_t *scr_off, int32_t *n_final, uint8_t *first_is_rev, int64_t *wide,
                         hipStream_t st) {
    if (A.nseq > 0) emfregs_fill_kernel<<<(unsigned)((A.nseq + 255) / 256), 256, 0, st>>>(A, scr_off, n_final, first_is_rev);
    emfregs_widen_kernel<<<(unsigned)((A.nseq + 256) / 256), 256, 0, st>>>(n_final, A.nseq, wide);
}
void launch_emfregs_emit(const EmfRegArgs &A, const int64_t *scr_off, const int32_t *n_final, const int64_t *out_off,
                         bwams_alnreg_t *out, hipStream_t st) {
    if (A.nseq > 0) emfregs_emit_kernel<<<(unsigned)((A.nseq + 255) / 256), 256, 0, st>>>(A, scr_off, n_final, out_off, out);
}

}  // namespace bwams
