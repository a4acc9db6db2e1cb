// api_chain.hip — C-ABI entry points of the chaining and chain-to-alignment stages
// (include/bwams.h): launch sequences over chain.hip, ext_aln.hip and bsw_extend.hip on the
// batch's stream.  No CPU fallback: every entry point runs HIP kernels or returns an error.
#include <cmath>
#include <map>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <utility>
#include <vector>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include "chain_kernels.h"

namespace bwams {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = bytes + bytes / 8 + 4096;
        hipError_t e = dev_malloc(&p, cap);
        if (e != hipSuccess) { cap = 0; return e; }
        return e;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ChainState {
    // chaining scratch (per SA hit)
    DevBuf s_next, s_ql, crec, flt, f_rec, f_first, f_kept, f_sel, nodes;
    // per read
    DevBuf n_kept, n_kept_seeds, n_chn, heavy, redo, read_base, frac, wide, chain_off, slice, okeys, okeys2, ovals, ovals2;
    // results
    DevBuf chains, seeds, seeds2;
    DevBuf sw_qb, sw_rb, sw_read, sw_newn, sw_res;      // mem_flt_chained_seeds (long reads)
    DevBuf dd_regs, dd_ord, dd_srt, dd_eh, dd_nout, dd_wide, dd_off, dd_out, dd_light;   // mem_sort_dedup_patch
    DevBuf pe_keys, pe_keys2;                           // mem_pestat
    // mate rescue + mem_mark_primary_se + mem_pair
    DevBuf pr_na, pr_wide, pr_offs, pr_anchor, pr_slot, pr_task, pr_trb, pr_tl1, pr_twide, pr_toffs, pr_pairs, pr_tref, pr_tqer,
           pr_aln, pr_pool, pr_ord, pr_srt, pr_z, pr_nfin, pr_npri, pr_nsw, pr_full, pr_owide, pr_ooff, pr_out, pr_res;
    DevBuf et_mems, et_moff, et_hits, et_hoff, et_smem, et_cnt, et_off, et_coord, et_srt;      // ERT mode input translation
    int64_t pr_total = 0, pr_tasks = 0, pr_redone = 0;
    bool pair_done = false, pr_single = false;
    DevBuf er_wide, er_off, er_scr, er_n, er_rev, er_out, er_ooff, mg_wide, mg_off, mg_out;   // mem_perfect2reg (+ its merge into the final regions)
    int64_t er_total = 0, er_nseq = 0;
    bool er_done = false;
    DevBuf al_need, al_cls, al_off, al_scr, al_list, al_rec, al_wide, al_offs, al_cig, al_md, al_cnt, al_only;   // mem_reg2aln
    int64_t al_n = 0, al_ncig = 0, al_nmd = 0;
    int al_source = 0;
    bool al_done = false;
    DevBuf sm_names, sm_noff, sm_qual, sm_comm, sm_coff, sm_mapq, sm_len, sm_off, sm_out, sm_logtab, sm_bad;   // SAM text
    int64_t sm_bytes = 0, sm_nseq = 0, sm_nregs = 0;
    int64_t sm_merged_n = -1;            // >= 0: sm_out / sm_off hold that many reads' text merged from two runs (bwams_process_chunk_smart)
    bool sm_up = false, sm_has_qual = false, sm_has_comm = false, sm_done = false, sm_log_ok = false;
    int64_t n_final = 0;
    bool dedup_done = false;
    int64_t n_chains = 0, n_seeds = 0, nseq = 0, n_chain_redo = 0;
    bool chain_done = false;
    // extension
    DevBuf regs, srt, rmax, cnt, ewide, eoffs, state, kreg, cur, lim;
    DevBuf lpairs, lref, lqer, rpairs, rref, rqer, retry;
    DevBuf lsrc, rsrc;           // in-place extension (bwams_extend_run): per task the start offsets {query, target} instead of copied bytes
    bool tasks_inplace = false;
    int64_t n_left = 0, n_right = 0, lref_b = 0, lqer_b = 0, rref_b = 0, rqer_b = 0;
    int64_t n_retry_left = 0, n_retry_right = 0, n_rounds = 0;
    bool built = false, ext_done = false;
    bwams_mem_opt_t opt{};
    hipEvent_t ev[16] = {};       // 0-1 chain, 2-3 plan+build, 4-5 left, 6-7 right, 8-9 selection (first round each), 10-11 all rounds
    bool ev_ok = false;
    hipStream_t aux[7] = {};      // the chaining tiers run concurrently (the device's shared set: aux_acquire)
    int aux_device = -1;
    hipEvent_t fork = nullptr, join[7] = {};
};

// The auxiliary streams are ONE set per device, shared by its batches (reference-counted).  A batch of its own set made 9 streams per
// batch; the runtime maps streams onto GPU_MAX_HW_QUEUES (8) hardware queues and a queue completes its packets in order, so with
// three batches on a device (two chunks in flight + the caller's) a slot's copies landed behind another slot's kernels or not,
// depending on the order in which the process had created its streams (bench.py: 4.5 .. 5.6 Mreads/s streaming in a process that had
// created other batches before, 7.3 .. 7.6 in a fresh one).  Sharing is safe: every use is fork event -> launches -> join event, and
// a stream is a total order.
struct AuxSet { hipStream_t q[7] = {}; int refs = 0; };
static std::mutex g_aux_mu;
static std::map<int, AuxSet> g_aux;
static int aux_acquire(int device, hipStream_t *out) {
    std::lock_guard<std::mutex> g(g_aux_mu);
    AuxSet &a = g_aux[device];
    if (a.refs == 0)
        for (auto &q : a.q) BWAMS_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    ++a.refs;
    for (int i = 0; i < 7; ++i) out[i] = a.q[i];
    return BWAMS_OK;
}
static void aux_release(int device) {
    std::lock_guard<std::mutex> g(g_aux_mu);
    auto it = g_aux.find(device);
    if (it == g_aux.end()) return;
    if (--it->second.refs == 0) {
        for (auto &q : it->second.q) if (q) (void)hipStreamDestroy(q);
        g_aux.erase(it);
    }
}

void chain_state_free(ChainState *s) {
    if (!s) return;
    DevBuf *all[] = {&s->lsrc, &s->rsrc, &s->s_next, &s->s_ql, &s->crec, &s->flt, &s->f_rec, &s->f_first, &s->f_kept, &s->f_sel,
                     &s->nodes, &s->n_kept, &s->n_kept_seeds, &s->n_chn, &s->heavy, &s->redo, &s->slice, &s->okeys, &s->okeys2, &s->ovals, &s->ovals2, &s->read_base, &s->frac, &s->wide,
                     &s->chain_off, &s->chains, &s->seeds, &s->seeds2, &s->sw_qb, &s->sw_rb, &s->sw_read, &s->sw_newn, &s->sw_res, &s->dd_regs, &s->dd_ord, &s->dd_srt, &s->dd_eh,
                     &s->dd_nout, &s->dd_wide, &s->dd_off, &s->dd_out, &s->dd_light, &s->pe_keys, &s->pe_keys2, &s->pr_na, &s->pr_wide, &s->pr_offs, &s->pr_anchor, &s->pr_slot, &s->pr_task, &s->pr_trb, &s->pr_tl1, &s->pr_twide, &s->pr_toffs,
                     &s->pr_pairs, &s->pr_tref, &s->pr_tqer, &s->pr_aln, &s->pr_pool, &s->pr_ord, &s->pr_srt, &s->pr_z, &s->pr_nfin, &s->pr_npri, &s->pr_nsw, &s->pr_full, &s->pr_owide,
                     &s->pr_ooff, &s->pr_out, &s->pr_res, &s->et_mems, &s->et_moff, &s->et_hits, &s->et_hoff, &s->et_smem, &s->et_cnt, &s->et_off, &s->et_coord, &s->et_srt, &s->er_wide, &s->er_off, &s->er_scr, &s->er_n, &s->er_rev, &s->er_out, &s->er_ooff, &s->mg_wide, &s->mg_off, &s->mg_out, &s->al_need, &s->al_cls, &s->al_off, &s->al_scr, &s->al_list, &s->al_rec, &s->al_wide, &s->al_offs, &s->al_cig, &s->al_md, &s->al_cnt, &s->al_only, &s->sm_names, &s->sm_noff, &s->sm_qual, &s->sm_comm, &s->sm_coff, &s->sm_mapq, &s->sm_len, &s->sm_off, &s->sm_out, &s->sm_logtab, &s->sm_bad, &s->regs, &s->srt, &s->rmax, &s->cnt, &s->state, &s->kreg, &s->cur, &s->lim,
                     &s->ewide, &s->eoffs, &s->lpairs, &s->lref, &s->lqer, &s->rpairs, &s->rref, &s->rqer, &s->retry};
    for (DevBuf *d : all)
        if (d->p) (void)hipFree(d->p);
    if (s->ev_ok) {
        for (auto &e : s->ev) (void)hipEventDestroy(e);
        for (auto &e : s->join) (void)hipEventDestroy(e);
        (void)hipEventDestroy(s->fork);
        if (s->aux_device >= 0) aux_release(s->aux_device);
    }
    delete s;
}

namespace {

__global__ void widen2_kernel(const int32_t *a, const int32_t *b, int64_t n, int64_t *wide) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * (n + 1)) return;
    const int64_t row = g / (n + 1), i = g - row * (n + 1);
    wide[g] = i < n ? (int64_t)(row ? b[i] : a[i]) : 0;
}

int scan_rows(bwams_batch *b, const int64_t *in, int64_t *out, int rows, int64_t n1) {
    for (int r = 0; r < rows; ++r) {
        size_t tb = 0;
        BWAMS_HIP(rocprim::exclusive_scan(nullptr, tb, in + r * n1, out + r * n1, (int64_t)0, (size_t)n1,
                                          rocprim::plus<int64_t>(), b->stream));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(b->stream));
            if (b->d_tmp) (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(rocprim::exclusive_scan(b->d_tmp, tb, in + r * n1, out + r * n1, (int64_t)0, (size_t)n1,
                                          rocprim::plus<int64_t>(), b->stream));
    }
    return BWAMS_OK;
}

int get_state(bwams_batch *b, ChainState **out) {
    if (!b->chain) {
        b->chain = new ChainState();
        for (auto &e : b->chain->ev) BWAMS_HIP(hipEventCreate(&e));
        for (auto &e : b->chain->join) BWAMS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        BWAMS_HIP(hipEventCreateWithFlags(&b->chain->fork, hipEventDisableTiming));
        b->chain->ev_ok = true;
        if (int rc = aux_acquire(b->idx->device, b->chain->aux)) return rc;
        b->chain->aux_device = b->idx->device;
    }
    *out = b->chain;
    return BWAMS_OK;
}

int check_opt(const bwams_mem_opt_t *o, const char *who) {
    if (!o || o->e_del <= 0 || o->e_ins <= 0 || o->max_occ <= 0 || o->w < 0) {
        set_last_error(std::string(who) + ": null options, non-positive gap extension penalty or max_occ");
        return BWAMS_ERR_ARG;
    }
    return BWAMS_OK;
}

int dev_bns(bwams_index *ix, DevBns *out) {
    const int64_t l_pac = (ix->fmi.ref_seq_len - 1) / 2;
    if (!ix->d_contigs) {                       // default: one sequence spanning the whole text
        bwams_contig_t c;
        c.offset = 0; c.len = (int32_t)l_pac; c.is_alt = 0;
        if (l_pac > 0x7fffffffLL) {
            set_last_error("the index holds more than 2^31 bases: call bwams_index_set_contigs with the real sequences");
            return BWAMS_ERR_ARG;
        }
        BWAMS_HIP(dev_malloc(&ix->d_contigs, sizeof c));
        BWAMS_HIP(hipMemcpy(ix->d_contigs, &c, sizeof c, hipMemcpyHostToDevice));
        ix->n_seqs = 1;
    }
    out->contigs = reinterpret_cast<const bwams_contig_t *>(ix->d_contigs);
    out->n_seqs = ix->n_seqs;
    out->l_pac = l_pac;
    return BWAMS_OK;
}

void sw_params(const bwams_mem_opt_t &o, int end_bonus, SwParams *prm) {
    prm->o_del = o.o_del; prm->e_del = o.e_del; prm->o_ins = o.o_ins; prm->e_ins = o.e_ins;
    prm->zdrop = o.zdrop; prm->end_bonus = end_bonus;
    int mx = 0;
    for (int i = 0; i < 25; ++i) {
        prm->mat[i] = o.mat[i];
        mx = mx > o.mat[i] ? mx : o.mat[i];
    }
    prm->max_sc = mx;
}

}  // namespace
}  // namespace bwams

using namespace bwams;

extern "C" {

int bwams_index_set_contigs(bwams_index_t *ix, const bwams_contig_t *contigs, int32_t n_seqs) {
    if (!ix || !contigs || n_seqs <= 0) return BWAMS_ERR_ARG;
    const int64_t l_pac = (ix->fmi.ref_seq_len - 1) / 2;
    int64_t at = 0;
    for (int32_t i = 0; i < n_seqs; ++i) {
        if (contigs[i].offset != at || contigs[i].len <= 0) {
            set_last_error("bwams_index_set_contigs: sequences must tile [0, l_pac) in order");
            return BWAMS_ERR_ARG;
        }
        at += contigs[i].len;
    }
    if (at != l_pac) {
        set_last_error("bwams_index_set_contigs: sequence lengths do not add up to l_pac");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(ix->device));
    if (ix->d_contigs) (void)hipFree(ix->d_contigs);
    ix->d_contigs = nullptr;
    BWAMS_HIP(dev_malloc(&ix->d_contigs, (size_t)n_seqs * sizeof(bwams_contig_t)));
    BWAMS_HIP(hipMemcpy(ix->d_contigs, contigs, (size_t)n_seqs * sizeof(bwams_contig_t), hipMemcpyHostToDevice));
    ix->n_seqs = n_seqs;
    return BWAMS_OK;
}

// mem_flt_chained_seeds for the chunk's long reads (seed_sw.hip): re-score short seeds with the local-SW
// kernel, drop the weak ones, re-pack the seed array
static int flt_chained_seeds(bwams_batch *b, ChainState *s, const bwams_mem_opt_t *opt, const DevBns &bns) {
    if (!b->idx->d_ref) {
        set_last_error("bwams_chain_run: reads of ~1100 bases and more need the .0123 reference (mem_flt_chained_seeds)");
        return BWAMS_ERR_ARG;
    }
    int mx = -128, mn = 127;
    for (int i = 0; i < 25; ++i) { mx = mx > opt->mat[i] ? mx : opt->mat[i]; mn = mn < opt->mat[i] ? mn : opt->mat[i]; }
    if (mx <= 0 || (opt->o_ins + opt->e_ins) + (opt->o_del + opt->e_del) <= mx - mn) {
        set_last_error("bwams_chain_run: the local SW of mem_flt_chained_seeds needs max(mat) > 0 and oe_ins + oe_del > max(mat) - min(mat)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    hipStream_t st = b->stream;
    const int64_t N = s->n_seeds, N1 = N + 1, C = s->n_chains, n1 = s->nseq + 1;
    BWAMS_HIP(s->cnt.ensure((size_t)N1 * 6 * 4));
    BWAMS_HIP(s->ewide.ensure((size_t)(N1 > C + 1 ? N1 : C + 1) * 6 * 8));
    BWAMS_HIP(s->eoffs.ensure((size_t)(N1 > C + 1 ? N1 : C + 1) * 6 * 8));
    BWAMS_HIP(s->sw_qb.ensure((size_t)N1 * 4)); BWAMS_HIP(s->sw_rb.ensure((size_t)N1 * 8));
    BWAMS_HIP(s->sw_read.ensure((size_t)N1 * 4)); BWAMS_HIP(s->sw_newn.ensure((size_t)(C + 1) * 4));
    BWAMS_HIP(s->seeds2.ensure((size_t)N1 * sizeof(bwams_chain_seed_t)));
    SeedSwArgs W;
    W.chains = s->chains.as<bwams_chain_t>(); W.n_chains = C;
    W.seeds = s->seeds.as<bwams_chain_seed_t>(); W.n_seeds = N;
    W.enc = b->d_enc; W.cum = b->d_cum; W.nseq = s->nseq; W.ref = b->idx->fmi.ref; W.bns = bns; W.opt = *opt;
    W.cnt = s->cnt.as<int32_t>(); W.win_qb = s->sw_qb.as<int32_t>(); W.win_rb = s->sw_rb.as<int64_t>();
    W.seed_read = s->sw_read.as<int32_t>();
    launch_seedsw_plan(W, s->ewide.as<int64_t>(), st);
    int rc = scan_rows(b, s->ewide.as<int64_t>(), s->eoffs.as<int64_t>(), 3, N1);
    if (rc) return rc;
    int64_t tot[3];
    for (int r = 0; r < 3; ++r)
        BWAMS_HIP(hipMemcpyAsync(&tot[r], s->eoffs.as<int64_t>() + r * N1 + N, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    if (tot[1] >= ((int64_t)1 << 31) || tot[2] >= ((int64_t)1 << 31)) {
        set_last_error("bwams_chain_run: seed re-scoring buffers exceed the 31-bit offsets of SeqPair; use smaller chunks");
        return BWAMS_ERR_CAPACITY;
    }
    BWAMS_HIP(s->lpairs.ensure((size_t)(tot[0] + 1) * sizeof(bwams_seqpair_t)));
    BWAMS_HIP(s->lqer.ensure((size_t)tot[1] + 64)); BWAMS_HIP(s->lref.ensure((size_t)tot[2] + 64));
    BWAMS_HIP(s->sw_res.ensure((size_t)(tot[0] + 1) * sizeof(bwams_kswr_t)));
    if (tot[0] > 0) {
        launch_seedsw_build(W, s->eoffs.as<int64_t>(), s->lpairs.as<bwams_seqpair_t>(), s->lref.as<uint8_t>(), s->lqer.as<uint8_t>(),
                            b->cu_count, st);
        SwParams prm;
        sw_params(*opt, 0, &prm);
        (void)launch_ksw(s->lpairs.as<bwams_seqpair_t>(), tot[0], s->lref.as<uint8_t>(), s->lqer.as<uint8_t>(), prm, 208, 200,
                         s->sw_res.p, b->d_ctr, b->cu_count, st);
    }
    // keep / drop, new chain lengths, packed offsets (eoffs row 0 still holds the task index of each seed)
    int64_t *cw = s->ewide.as<int64_t>();                // reused: C + 1 entries
    int64_t *coff = s->eoffs.as<int64_t>() + 3 * N1;     // behind the three rows in use
    launch_seedsw_apply(W, s->eoffs.as<int64_t>(), s->sw_res.as<bwams_kswr_t>(), s->sw_newn.as<int32_t>(), cw, st);
    if ((rc = scan_rows(b, cw, coff, 1, C + 1))) return rc;
    int64_t new_total = 0;
    BWAMS_HIP(hipMemcpyAsync(&new_total, coff + C, 8, hipMemcpyDeviceToHost, st));
    launch_seedsw_repack(W, s->sw_newn.as<int32_t>(), coff, s->seeds2.as<bwams_chain_seed_t>(), s->chain_off.as<int64_t>(),
                         s->chain_off.as<int64_t>() + n1, st);
    BWAMS_HIP(hipStreamSynchronize(st));
    std::swap(s->seeds, s->seeds2);
    s->n_seeds = new_total;
    return BWAMS_OK;
}

// the seeds of a chunk as the chaining kernels read them: SMEM-like records in (rid, m, n) order and, per record,
// the reference positions to chain (already strided to at most max_occ)
struct SeedView {
    const bwams_smem_t *smem;
    int64_t n_smem;
    const int64_t *sa_off, *sa_coord;
    int64_t n_sa;
    bool one_smem_quirk;        // mem_chain_seeds' `pos < num_smem - 1` (bwamem.cpp:819); mem_chain_new has no such guard
};
static int chain_common(bwams_batch *b, const bwams_mem_opt_t *opt, const SeedView &sv, int64_t *n_chains, int64_t *n_seeds);

int bwams_chain_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_chains, int64_t *n_seeds) {
    if (!b || !b->seed_done || !b->with_sa) {
        set_last_error("bwams_chain_run: run bwams_seed_run(with_sa = 1) first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_chain_run");
    if (rc) return rc;
    if ((rc = bwams_seed_counts(b, nullptr, nullptr))) return rc;     // sizes of the seed stage (and its overflow check)
    SeedView sv;
    sv.smem = b->d_sorted; sv.n_smem = b->n_smem; sv.sa_off = b->d_sa_off; sv.sa_coord = b->d_sa_coord; sv.n_sa = b->n_sa;
    sv.one_smem_quirk = true;
    return chain_common(b, opt, sv, n_chains, n_seeds);
}

int bwams_chain_run_ert(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_ert_mem_t *mems, const int64_t *mem_off,
                        const uint64_t *hits, const int64_t *hit_off, int64_t *n_chains, int64_t *n_seeds) {
    if (!b || b->nseq <= 0 || !mem_off || !hit_off) {
        set_last_error("bwams_chain_run_ert: upload the reads first (bwams_seed_upload) and pass the MEM / hit offsets");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_chain_run_ert");
    if (rc) return rc;
    const int64_t nseq = b->nseq, n1 = nseq + 1;
    const int64_t n_mems = mem_off[nseq], n_hits = hit_off[nseq];
    if (mem_off[0] != 0 || hit_off[0] != 0 || n_mems < 0 || n_hits < 0 || (n_mems && !mems) || (n_hits && !hits)) return BWAMS_ERR_ARG;
    for (int64_t r = 0; r < nseq; ++r) {
        if (mem_off[r + 1] < mem_off[r] || hit_off[r + 1] < hit_off[r]) { set_last_error("bwams_chain_run_ert: offsets must be non-decreasing"); return BWAMS_ERR_ARG; }
        const int64_t nh = hit_off[r + 1] - hit_off[r];
        for (int64_t i = mem_off[r]; i < mem_off[r + 1]; ++i) {
            const bwams_ert_mem_t &m = mems[i];
            if (m.start < 0 || m.end <= m.start || m.hitcount < 0 || m.hitbeg < 0 || (int64_t)m.hitbeg + m.hitcount > nh) {
                set_last_error("bwams_chain_run_ert: a MEM with an empty span or a hit slice outside its read's hit array");
                return BWAMS_ERR_ARG;
            }
        }
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    ChainState *s;
    if ((rc = get_state(b, &s))) return rc;
    hipStream_t st = b->stream;
    BWAMS_HIP(s->et_mems.ensure((size_t)(n_mems + 1) * sizeof(bwams_ert_mem_t)));
    BWAMS_HIP(s->et_moff.ensure((size_t)n1 * 8)); BWAMS_HIP(s->et_hoff.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->et_hits.ensure((size_t)(n_hits + 1) * 8));
    BWAMS_HIP(s->et_smem.ensure((size_t)(n_mems + 1) * sizeof(bwams_smem_t)));
    BWAMS_HIP(s->et_cnt.ensure((size_t)(n_mems + 1) * 8)); BWAMS_HIP(s->et_off.ensure((size_t)(n_mems + 1) * 8));
    BWAMS_HIP(s->et_srt.ensure((size_t)(n_mems + 1) * 24));
    if (n_mems) BWAMS_HIP(hipMemcpyAsync(s->et_mems.p, mems, (size_t)n_mems * sizeof(bwams_ert_mem_t), hipMemcpyHostToDevice, st));
    if (n_hits) BWAMS_HIP(hipMemcpyAsync(s->et_hits.p, hits, (size_t)n_hits * 8, hipMemcpyHostToDevice, st));
    BWAMS_HIP(hipMemcpyAsync(s->et_moff.p, mem_off, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    BWAMS_HIP(hipMemcpyAsync(s->et_hoff.p, hit_off, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    ErtArgs E;
    E.mems = s->et_mems.as<bwams_ert_mem_t>(); E.mem_off = s->et_moff.as<int64_t>();
    E.hits = s->et_hits.as<uint64_t>(); E.hit_off = s->et_hoff.as<int64_t>();
    E.nseq = nseq; E.n_mems = n_mems; E.l_pac = (b->idx->fmi.ref_seq_len - 1) / 2;
    E.max_occ = opt->max_occ; E.pad_ = 0;
    E.smem_out = s->et_smem.as<bwams_smem_t>(); E.cnt = s->et_cnt.as<int64_t>(); E.srt = s->et_srt.p;
    launch_ert_sort(E, st);
    if ((rc = scan_rows(b, s->et_cnt.as<int64_t>(), s->et_off.as<int64_t>(), 1, n_mems + 1))) return rc;
    int64_t n_sa = 0;
    BWAMS_HIP(hipMemcpyAsync(&n_sa, s->et_off.as<int64_t>() + n_mems, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(s->et_coord.ensure((size_t)(n_sa + 1) * 8));
    launch_ert_pick(E, s->et_off.as<int64_t>(), s->et_coord.as<int64_t>(), st);
    SeedView sv;
    sv.smem = s->et_smem.as<bwams_smem_t>(); sv.n_smem = n_mems; sv.sa_off = s->et_off.as<int64_t>();
    sv.sa_coord = s->et_coord.as<int64_t>(); sv.n_sa = n_sa; sv.one_smem_quirk = false;
    return chain_common(b, opt, sv, n_chains, n_seeds);
}

static int chain_common(bwams_batch *b, const bwams_mem_opt_t *opt, const SeedView &sv, int64_t *n_chains, int64_t *n_seeds) {
    int rc;
    if (b->max_read_len >= 32768) {
        set_last_error("bwams_chain_run: reads of 32768 bases or more are not supported (16-bit query coordinates)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    ChainState *s;
    if ((rc = get_state(b, &s))) return rc;
    s->chain_done = s->built = s->ext_done = s->dedup_done = s->pair_done = false;
    hipStream_t st = b->stream;
    const int64_t nseq = b->nseq, n_sa = sv.n_sa, n1 = nseq + 1;
    const size_t ns = (size_t)(n_sa > 0 ? n_sa : 1);
    BWAMS_HIP(s->s_next.ensure(ns * 4));  BWAMS_HIP(s->s_ql.ensure(ns * 8));
    BWAMS_HIP(s->crec.ensure(chain_rec_bytes(n_sa)));
    BWAMS_HIP(s->flt.ensure(ns * 8));     BWAMS_HIP(s->f_rec.ensure(ns * 16));
    BWAMS_HIP(s->f_first.ensure(ns * 4)); BWAMS_HIP(s->f_kept.ensure(ns * 4)); BWAMS_HIP(s->f_sel.ensure(ns * 4));
    BWAMS_HIP(s->n_chn.ensure((size_t)n1 * 4));       BWAMS_HIP(s->heavy.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->redo.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->slice.ensure((size_t)n1 * 16));
    BWAMS_HIP(s->okeys.ensure((size_t)n1 * 4));       BWAMS_HIP(s->okeys2.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->ovals.ensure((size_t)n1 * 4));       BWAMS_HIP(s->ovals2.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->nodes.ensure(chain_node_bytes(n_sa, nseq)));
    BWAMS_HIP(s->n_kept.ensure((size_t)n1 * 4));      BWAMS_HIP(s->n_kept_seeds.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->read_base.ensure((size_t)n1 * 8));   BWAMS_HIP(s->frac.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->wide.ensure((size_t)n1 * 16));       BWAMS_HIP(s->chain_off.ensure((size_t)n1 * 16));

    ChainArgs A;
    A.smem = sv.smem; A.n_smem = sv.n_smem; A.sa_off = sv.sa_off; A.sa_coord = sv.sa_coord;
    A.cum = b->d_cum; A.nseq = nseq;
    if ((rc = dev_bns(b->idx, &A.bns))) return rc;
    A.opt = *opt;
    A.s_next = s->s_next.as<int32_t>(); A.s_ql = s->s_ql.as<int2>();
    A.crec = s->crec.p;
    A.flt = s->flt.as<uint2>(); A.f_rec = s->f_rec.as<uint4>(); A.f_first = s->f_first.as<int32_t>();
    A.f_kept = s->f_kept.as<int32_t>(); A.f_sel = s->f_sel.as<int32_t>(); A.nodes = s->nodes.p;
    A.n_kept = s->n_kept.as<int32_t>(); A.n_kept_seeds = s->n_kept_seeds.as<int32_t>();
    A.n_chn = s->n_chn.as<int32_t>(); A.heavy = s->heavy.as<int32_t>(); A.redo = s->redo.as<int32_t>();
    A.slice = s->slice.as<int64_t>(); A.order = s->ovals2.as<uint32_t>();
    A.read_base = s->read_base.as<int64_t>(); A.frac_rep = s->frac.as<float>();
    A.ctr = b->d_ctr;
    A.seed_batch = knobs().chain_batch;

    BWAMS_HIP(hipEventRecord(s->ev[0], st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->chain_redo, 0, 2 * sizeof(unsigned long long), st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->chain_overflow, 0, 29 * sizeof(unsigned long long), st));   // overflow, longread, n_heavy, chain_class[10], chain_ticket[10], heavy_tickets[6]
    // mem_chain_seeds' loop guard `pos < num_smem - 1` (bwamem.cpp:819) makes a work item with exactly
    // one SMEM produce no chain at all
    if ((sv.one_smem_quirk ? sv.n_smem <= 1 : sv.n_smem <= 0) || n_sa == 0) {
        BWAMS_HIP(hipMemsetAsync(s->n_kept.p, 0, (size_t)n1 * 4, st));
        BWAMS_HIP(hipMemsetAsync(s->n_kept_seeds.p, 0, (size_t)n1 * 4, st));
    } else {
        // reads of similar seed count share a wave: sort read ids by descending count
        launch_chain_count(A, s->okeys.as<uint32_t>(), s->ovals.as<uint32_t>(), st);
        size_t tb = 0;
        BWAMS_HIP(rocprim::radix_sort_pairs_desc(nullptr, tb, s->okeys.as<uint32_t>(), s->okeys2.as<uint32_t>(),
                                                 s->ovals.as<uint32_t>(), s->ovals2.as<uint32_t>(), (size_t)nseq, 0, 32, st));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(st));
            if (b->d_tmp) (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(rocprim::radix_sort_pairs_desc(b->d_tmp, tb, s->okeys.as<uint32_t>(), s->okeys2.as<uint32_t>(),
                                                 s->ovals.as<uint32_t>(), s->ovals2.as<uint32_t>(), (size_t)nseq, 0, 32, st));
        if (launch_chain(A, s->okeys.as<uint32_t>(), b->cu_count, st, s->aux, s->fork, s->join)) {
            set_last_error("bwams_chain_run: stream fork/join failed");
            return BWAMS_ERR_DEVICE;
        }
    }
    int64_t tot[2] = {0, 0};
    if (nseq > 0) {
        const int64_t g = 2 * n1;
        widen2_kernel<<<(unsigned)((g + 255) / 256), 256, 0, st>>>(A.n_kept, A.n_kept_seeds, nseq, s->wide.as<int64_t>());
        if ((rc = scan_rows(b, s->wide.as<int64_t>(), s->chain_off.as<int64_t>(), 2, n1))) return rc;
        BWAMS_HIP(hipMemcpyAsync(&tot[0], s->chain_off.as<int64_t>() + nseq, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipMemcpyAsync(&tot[1], s->chain_off.as<int64_t>() + n1 + nseq, 8, hipMemcpyDeviceToHost, st));
    } else {
        BWAMS_HIP(hipMemsetAsync(s->chain_off.p, 0, (size_t)n1 * 16, st));       // no reads: both offset rows are {0}
    }
    BWAMS_HIP(hipMemcpyAsync(b->h_ctr, b->d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    {
        const bool vb = knobs().verbose != 0;
        if (vb) {
            const unsigned long long *d = b->h_ctr->dbg;
            fprintf(stderr, "[bwams_chain_run] filter wave tier: reads by chains <=32 %llu <=64 %llu <=128 %llu <=256 %llu <=512 %llu <=960 %llu more %llu; "
                            "Mcycles: sequential(HBM) %.1f sort %.1f filter %.1f; chains %llu selected %llu; longest read: sort %.2f filter %.2f Mcycles, most chains %llu, most selected %llu\n",
                    d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7] / 1e6, d[8] / 1e6, d[9] / 1e6, d[11], d[10], d[12] / 1e6, d[13] / 1e6, d[14], d[15]);
#ifdef BWAMS_CHAINDBG
            static const char *cn[8] = {"XL", "L", "L2", "L1", "M2", "M", "M1", "S"};
            fprintf(stderr, "[bwams_chain_run] chaining wave tier, per class: reads / mean us / longest us / wave-ms:");
            for (int c = 0; c < 8; ++c) fprintf(stderr, "  %s %llu / %.0f / %.0f / %.1f", cn[c], d[32 + 3 * c], d[32 + 3 * c] ? d[33 + 3 * c] * 1e-2 / d[32 + 3 * c] : 0.0, d[34 + 3 * c] * 1e-2, d[33 + 3 * c] * 1e-5);
            fprintf(stderr, "\n");
            fprintf(stderr, "[bwams_chain_run] wave tier phases, G cycles: preamble %.2f chaining %.2f weights+copy %.2f (sort %.2f filter %.2f: all reads); reads %llu seeds %llu; passes %llu settling %llu seeds (%llu new chains), %llu seeds one by one; pass parts, G cycles: batch prologue %.2f search %.2f record+test %.2f settle %.2f commit %.2f one-by-one %.2f\n",
                    d[56] / 1e9, d[57] / 1e9, d[58] / 1e9, d[8] / 1e9, d[9] / 1e9, d[59], d[60], d[61], d[63], d[64], d[62], d[65] / 1e9, d[66] / 1e9, d[67] / 1e9, d[68] / 1e9, d[69] / 1e9, d[70] / 1e9);
#endif
        }
    }
    if (b->h_ctr->chain_overflow) {
        set_last_error("bwams_chain_run: internal B-tree node region exhausted");
        return BWAMS_ERR_CAPACITY;
    }
    const bool has_long = b->h_ctr->chain_longread != 0;
    s->n_chain_redo = (int64_t)b->h_ctr->chain_redo;
    s->n_chains = tot[0]; s->n_seeds = tot[1]; s->nseq = nseq;
    BWAMS_HIP(s->chains.ensure((size_t)(tot[0] + 1) * sizeof(bwams_chain_t)));
    BWAMS_HIP(s->seeds.ensure((size_t)(tot[1] + 1) * sizeof(bwams_chain_seed_t)));
    if (tot[0] > 0)
        launch_chain_emit(A, s->chain_off.as<int64_t>(), s->chain_off.as<int64_t>() + n1, s->chains.as<bwams_chain_t>(),
                          s->seeds.as<bwams_chain_seed_t>(), st);
    if (has_long && tot[0] > 0 && (rc = flt_chained_seeds(b, s, opt, A.bns))) return rc;
    BWAMS_HIP(hipEventRecord(s->ev[1], st));
    BWAMS_HIP(hipGetLastError());
    s->chain_done = true;
    s->opt = *opt;
    if (n_chains) *n_chains = s->n_chains;
    if (n_seeds) *n_seeds = s->n_seeds;
    return BWAMS_OK;
}

int bwams_chain_fetch(bwams_batch_t *b, bwams_chain_t *chains, int64_t chain_cap, bwams_chain_seed_t *seeds,
                      int64_t seed_cap, int64_t *chain_off) {
    if (!b || !b->chain || !b->chain->chain_done) {
        set_last_error("bwams_chain_fetch: no chains on the device");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (s->n_chains > chain_cap || s->n_seeds > seed_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (s->n_chains) BWAMS_HIP(hipMemcpyAsync(chains, s->chains.p, (size_t)s->n_chains * sizeof(bwams_chain_t), hipMemcpyDeviceToHost, st));
    if (s->n_seeds) BWAMS_HIP(hipMemcpyAsync(seeds, s->seeds.p, (size_t)s->n_seeds * sizeof(bwams_chain_seed_t), hipMemcpyDeviceToHost, st));
    if (chain_off) BWAMS_HIP(hipMemcpyAsync(chain_off, s->chain_off.p, (size_t)(s->nseq + 1) * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

int bwams_chain_upload(bwams_batch_t *b, const bwams_chain_t *chains, int64_t n_chains, const bwams_chain_seed_t *seeds,
                       int64_t n_seeds, const int64_t *chain_off) {
    if (!b || !chain_off || n_chains < 0 || n_seeds < 0 || (n_chains && (!chains || !seeds))) return BWAMS_ERR_ARG;
    if (b->nseq <= 0) {
        set_last_error("bwams_chain_upload: upload the reads first (bwams_seed_upload)");
        return BWAMS_ERR_ARG;
    }
    const int64_t nseq = b->nseq, n1 = nseq + 1;
    if (chain_off[0] != 0 || chain_off[nseq] != n_chains) return BWAMS_ERR_ARG;
    // seeds must be laid out chain after chain, chains read after read (what bwams_chain_fetch returns)
    int64_t at = 0;
    std::string bad;
    int64_t *soff = new int64_t[(size_t)n1];
    for (int64_t r = 0; r < nseq && bad.empty(); ++r) {
        soff[r] = at;
        if (chain_off[r + 1] < chain_off[r]) bad = "chain_off must be non-decreasing";
        for (int64_t j = chain_off[r]; j < chain_off[r + 1] && bad.empty(); ++j) {
            if (chains[j].seqid != r) bad = "chain.seqid does not match chain_off";
            else if (chains[j].seed_off != at || chains[j].n < 0) bad = "chain.seed_off must enumerate the seed array in order";
            at += chains[j].n;
        }
    }
    soff[nseq] = at;
    if (bad.empty() && at != n_seeds) bad = "seed counts do not add up";
    if (!bad.empty()) {
        delete[] soff;
        set_last_error("bwams_chain_upload: " + bad);
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    ChainState *s;
    int rc = get_state(b, &s);
    if (rc) { delete[] soff; return rc; }
    s->chain_done = s->built = s->ext_done = s->dedup_done = s->pair_done = false;
    hipStream_t st = b->stream;
    BWAMS_HIP(s->chain_off.ensure((size_t)n1 * 16));
    BWAMS_HIP(s->chains.ensure((size_t)(n_chains + 1) * sizeof(bwams_chain_t)));
    BWAMS_HIP(s->seeds.ensure((size_t)(n_seeds + 1) * sizeof(bwams_chain_seed_t)));
    BWAMS_HIP(hipMemcpyAsync(s->chain_off.p, chain_off, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    BWAMS_HIP(hipMemcpyAsync(s->chain_off.as<int64_t>() + n1, soff, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    if (n_chains) BWAMS_HIP(hipMemcpyAsync(s->chains.p, chains, (size_t)n_chains * sizeof(bwams_chain_t), hipMemcpyHostToDevice, st));
    if (n_seeds) BWAMS_HIP(hipMemcpyAsync(s->seeds.p, seeds, (size_t)n_seeds * sizeof(bwams_chain_seed_t), hipMemcpyHostToDevice, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    delete[] soff;
    s->n_chains = n_chains; s->n_seeds = n_seeds; s->nseq = nseq;
    s->chain_done = true;
    return BWAMS_OK;
}

/* ------------------------------------------------------ chain -> alignment regions ---- */

static int ext_args(bwams_batch *b, ChainState *s, const bwams_mem_opt_t *opt, ExtArgs *A) {
    const int64_t n1 = s->nseq + 1;
    A->chains = s->chains.as<bwams_chain_t>(); A->n_chains = s->n_chains;
    A->seeds = s->seeds.as<bwams_chain_seed_t>(); A->n_seeds = s->n_seeds;
    A->chain_off = s->chain_off.as<int64_t>(); A->seed_off = s->chain_off.as<int64_t>() + n1;
    A->enc = b->d_enc; A->cum = b->d_cum; A->nseq = s->nseq;
    A->ref = b->idx->fmi.ref;
    int rc = dev_bns(b->idx, &A->bns);
    if (rc) return rc;
    A->opt = *opt;
    A->regs = s->regs.as<bwams_alnreg_t>(); A->srt = s->srt.as<uint32_t>(); A->rmax = s->rmax.as<int64_t>();
    A->cnt = s->cnt.as<int32_t>(); A->ctr = b->d_ctr;
    A->state = s->state.as<int32_t>(); A->kreg = s->kreg.p;
    A->cur = s->cur.as<int32_t>(); A->lim = s->lim.as<int32_t>();
    A->sel_heavy = s->heavy.as<int32_t>(); A->n_sel_heavy = &b->d_ctr->sel_heavy; A->sel_ticket = &b->d_ctr->sel_ticket;
    return BWAMS_OK;
}

// allocate the per-seed arrays and run the plan kernel (windows, seed order, regions, task sizes)
static int ext_plan(bwams_batch *b, ChainState *s, const bwams_mem_opt_t *opt, int extend_all, ExtArgs *A) {
    const int64_t N1 = s->n_seeds + 1, n1 = s->nseq + 1;
    BWAMS_HIP(s->regs.ensure((size_t)N1 * sizeof(bwams_alnreg_t)));
    BWAMS_HIP(s->srt.ensure((size_t)N1 * 4));
    BWAMS_HIP(s->rmax.ensure((size_t)(s->n_chains + 1) * 16));
    BWAMS_HIP(s->cnt.ensure((size_t)N1 * 6 * 4));
    BWAMS_HIP(s->ewide.ensure((size_t)N1 * 6 * 8));
    BWAMS_HIP(s->eoffs.ensure((size_t)N1 * 6 * 8));
    BWAMS_HIP(s->state.ensure((size_t)N1 * 4));
    BWAMS_HIP(s->kreg.ensure((size_t)N1 * 32));
    BWAMS_HIP(s->cur.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->lim.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->heavy.ensure((size_t)n1 * 4));
    int rc = ext_args(b, s, opt, A);
    if (rc) return rc;
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->sel_heavy, 0, 4 * sizeof(unsigned long long), b->stream));
    launch_ext_heavy_list(*A, b->stream);
    BWAMS_HIP(hipMemsetAsync(s->cur.p, 0, (size_t)n1 * 4, b->stream));
    BWAMS_HIP(hipMemsetAsync(s->lim.p, 0, (size_t)n1 * 4, b->stream));
    launch_ext_plan(*A, extend_all, b->stream);
    return BWAMS_OK;
}

// build the task lists of the seeds requested this round
static int ext_build_round(bwams_batch *b, ChainState *s, const ExtArgs &A, int64_t tot[6], bool inplace) {
    hipStream_t st = b->stream;
    const int64_t N = s->n_seeds, N1 = N + 1;
    launch_ext_widen(A, s->ewide.as<int64_t>(), st);
    int rc = scan_rows(b, s->ewide.as<int64_t>(), s->eoffs.as<int64_t>(), 6, N1);
    if (rc) return rc;
    for (int r = 0; r < 6; ++r)
        BWAMS_HIP(hipMemcpyAsync(&tot[r], s->eoffs.as<int64_t>() + r * N1 + N, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    for (int r : {1, 2, 4, 5})
        if (!inplace && tot[r] >= ((int64_t)1 << 31)) {
            set_last_error("extension task buffers exceed the 31-bit offsets of SeqPair; use smaller chunks");
            return BWAMS_ERR_CAPACITY;
        }
    s->n_left = tot[0]; s->lqer_b = tot[1]; s->lref_b = tot[2];
    s->n_right = tot[3]; s->rqer_b = tot[4]; s->rref_b = tot[5];
    BWAMS_HIP(s->lpairs.ensure((size_t)(tot[0] + 1) * sizeof(bwams_seqpair_t)));
    BWAMS_HIP(s->rpairs.ensure((size_t)(tot[3] + 1) * sizeof(bwams_seqpair_t)));
    const int64_t mx = tot[0] > tot[3] ? tot[0] : tot[3];
    BWAMS_HIP(s->retry.ensure((size_t)(mx + 1) * sizeof(bwams_seqpair_t)));
    s->tasks_inplace = inplace;
    if (inplace) {               // no bytes are copied: 16 bytes of offsets per task
        BWAMS_HIP(s->lsrc.ensure((size_t)(tot[0] + 1) * 16)); BWAMS_HIP(s->rsrc.ensure((size_t)(tot[3] + 1) * 16));
    } else {
        BWAMS_HIP(s->lqer.ensure((size_t)tot[1] + 64)); BWAMS_HIP(s->lref.ensure((size_t)tot[2] + 64));
        BWAMS_HIP(s->rqer.ensure((size_t)tot[4] + 64)); BWAMS_HIP(s->rref.ensure((size_t)tot[5] + 64));
    }
    if (tot[0] + tot[3] > 0)
        launch_ext_build(A, s->eoffs.as<int64_t>(), s->lpairs.as<bwams_seqpair_t>(), s->lref.as<uint8_t>(), s->lqer.as<uint8_t>(),
                         s->rpairs.as<bwams_seqpair_t>(), s->rref.as<uint8_t>(), s->rqer.as<uint8_t>(),
                         inplace ? s->lsrc.as<int64_t>() : nullptr, inplace ? s->rsrc.as<int64_t>() : nullptr, b->cu_count, st);
    return BWAMS_OK;
}

int bwams_extend_build(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_left, int64_t *n_right) {
    if (!b || !b->chain || !b->chain->chain_done) {
        set_last_error("bwams_extend_build: run bwams_chain_run (or bwams_chain_upload) first");
        return BWAMS_ERR_ARG;
    }
    if (!b->idx->d_ref) {
        set_last_error("bwams_extend_build: the index was opened without its .0123 reference");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_extend_build");
    if (rc) return rc;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    ChainState *s = b->chain;
    s->built = s->ext_done = s->dedup_done = s->pair_done = false;
    hipStream_t st = b->stream;
    ExtArgs A;
    BWAMS_HIP(hipEventRecord(s->ev[2], st));
    if ((rc = ext_plan(b, s, opt, 1, &A))) return rc;           // every seed, as the reference builds them
    int64_t tot[6];
    if ((rc = ext_build_round(b, s, A, tot, false))) return rc;
    BWAMS_HIP(hipEventRecord(s->ev[3], st));
    BWAMS_HIP(hipGetLastError());
    s->built = true;
    s->opt = *opt;
    if (n_left) *n_left = tot[0];
    if (n_right) *n_right = tot[3];
    return BWAMS_OK;
}

// one side: extend at w, settle, re-run the unsettled tasks at 2w (MAX_BAND_TRY = 2, bwamem.cpp:79)
static int run_side(bwams_batch *b, ChainState *s, const ExtArgs &A, int right, int64_t *n_retry_out) {
    hipStream_t st = b->stream;
    bwams_seqpair_t *pairs = right ? s->rpairs.as<bwams_seqpair_t>() : s->lpairs.as<bwams_seqpair_t>();
    // in place: the sequences are read where they lie (the chunk's base codes, the resident .0123 text), backwards on the left side
    const bool ip = s->tasks_inplace;
    const uint8_t *ref = ip ? A.ref : (right ? s->rref.as<uint8_t>() : s->lref.as<uint8_t>());
    const uint8_t *qer = ip ? A.enc : (right ? s->rqer.as<uint8_t>() : s->lqer.as<uint8_t>());
    const int64_t *src = ip ? (right ? s->rsrc.as<int64_t>() : s->lsrc.as<int64_t>()) : nullptr;
    const int dir = ip && !right ? -1 : 1;
    const int64_t n = right ? s->n_right : s->n_left;
    SwParams prm;
    sw_params(A.opt, right ? A.opt.pen_clip3 : A.opt.pen_clip5, &prm);
    const int qmax = b->max_read_len > 1 ? b->max_read_len : 1;
    if (n == 0) return BWAMS_OK;
    unsigned long long *d_nretry = &b->d_ctr->n_retry;
    BWAMS_HIP(hipMemsetAsync(d_nretry, 0, sizeof(unsigned long long), st));
    if (int erc = bsw_list_ensure(b, n)) return erc;
    if (int lrc = launch_bsw(pairs, n, ref, qer, A.opt.w, prm, qmax, b->d_ctr, b->cu_count, st, b->d_bsw_list, s->aux, s->fork, s->join, src, dir)) {
        set_last_error(lrc == -2 ? "banded SW: a query longer than ~18000 bases does not fit the LDS kernel" : "banded SW: stream fork/join failed");
        return lrc == -2 ? BWAMS_ERR_UNSUPPORTED : BWAMS_ERR_DEVICE;
    }
    launch_ext_post(A, right, pairs, n, A.opt.w, 0, s->retry.as<bwams_seqpair_t>(), d_nretry, st);
    unsigned long long nr = 0;
    BWAMS_HIP(hipMemcpyAsync(&nr, d_nretry, sizeof nr, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    if (nr) {
        if (launch_bsw(s->retry.as<bwams_seqpair_t>(), (int64_t)nr, ref, qer, A.opt.w << 1, prm, qmax, b->d_ctr, b->cu_count, st, b->d_bsw_list, s->aux, s->fork, s->join, src, dir)) return BWAMS_ERR_DEVICE;
        launch_ext_post(A, right, s->retry.as<bwams_seqpair_t>(), (int64_t)nr, A.opt.w << 1, 1, nullptr, d_nretry, st);
    }
    *n_retry_out += (int64_t)nr;
    return BWAMS_OK;
}

// Rounds of (build the requested tasks, extend left, extend right, select).  Round 0 extends the first
// seed visited of every chain; a later round extends the seeds the selection found it must keep but
// that had not been extended yet.  After kMaxRounds everything still undecided is extended at once.
int bwams_extend_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_regs) {
    if (!b || !b->chain || !b->chain->chain_done) {
        set_last_error("bwams_extend_run: run bwams_chain_run (or bwams_chain_upload) first");
        return BWAMS_ERR_ARG;
    }
    if (!b->idx->d_ref) {
        set_last_error("bwams_extend_run: the index was opened without its .0123 reference");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_extend_run");
    if (rc) return rc;
    int kMaxRounds = 6;
    if (knobs().ext_max_rounds > 0) kMaxRounds = knobs().ext_max_rounds;      // test knob: force the extend-the-rest fallback
    const bool adaptive_off = knobs().ext_all_rounds != 0;                    // test knob: never cut the rounds short
    const bool inplace_on = knobs().ext_inplace != 0;                          // A-B knob: 0 = copy the tasks' bytes into flat buffers
    ChainState *s = b->chain;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    s->built = s->ext_done = s->dedup_done = s->pair_done = false;
    ExtArgs A;
    BWAMS_HIP(hipEventRecord(s->ev[10], st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->bsw_cells, 0, sizeof(unsigned long long), st));      // DP cells of this run, all rounds
    BWAMS_HIP(hipEventRecord(s->ev[2], st));
    if ((rc = ext_plan(b, s, opt, opt->extend_all != 0, &A))) return rc;
    int64_t tot_left = 0, tot_right = 0;
    s->n_retry_left = s->n_retry_right = 0;
    int round = 0;
    for (;; ++round) {
        int64_t tot[6];
        if ((rc = ext_build_round(b, s, A, tot, inplace_on))) return rc;
        if (round == 0) { BWAMS_HIP(hipEventRecord(s->ev[3], st)); BWAMS_HIP(hipEventRecord(s->ev[4], st)); }
        tot_left += tot[0]; tot_right += tot[3];
        if ((rc = run_side(b, s, A, 0, &s->n_retry_left))) return rc;
        if (round == 0) { BWAMS_HIP(hipEventRecord(s->ev[5], st)); BWAMS_HIP(hipEventRecord(s->ev[6], st)); }
        launch_ext_right_h0(A, s->rpairs.as<bwams_seqpair_t>(), s->n_right, st);
        if ((rc = run_side(b, s, A, 1, &s->n_retry_right))) return rc;
        if (round == 0) { BWAMS_HIP(hipEventRecord(s->ev[7], st)); BWAMS_HIP(hipEventRecord(s->ev[8], st)); }
        BWAMS_HIP(hipMemsetAsync(&b->d_ctr->n_req, 0, sizeof(unsigned long long), st));
        BWAMS_HIP(hipMemsetAsync(&b->d_ctr->sel_ticket, 0, 3 * sizeof(unsigned long long), st));
        BWAMS_HIP(hipMemsetAsync(&b->d_ctr->n_rest, 0, sizeof(unsigned long long), st));
        const bool vb_sel = knobs().verbose != 0;
        const bool verbose_sel = vb_sel;
        if (verbose_sel) BWAMS_HIP(hipMemsetAsync(b->d_ctr->dbg, 0, sizeof b->d_ctr->dbg, st));
        if (s->n_seeds && launch_ext_select(A, b->cu_count, st, s->aux, s->fork, s->join)) {
            set_last_error("bwams_extend_run: stream fork/join failed");
            return BWAMS_ERR_DEVICE;
        }
        if (verbose_sel) {           // filled only by a build of ext_aln.hip with -DBWAMS_SELDBG
            unsigned long long d[16];
            BWAMS_HIP(hipMemcpyAsync(d, b->d_ctr->dbg, sizeof d, hipMemcpyDeviceToHost, st));
            BWAMS_HIP(hipStreamSynchronize(st));
            if (d[0])
                fprintf(stderr, "[bwams_extend_run] selection walk, round %d: %llu reads, Mticks total %.2f fetch %.2f scan %.2f keep-anyway %.2f; %llu slots, %llu chunks, "
                                "%llu keep-anyway calls; longest read: %.3f Mticks (fetch %.3f scan %.3f keep %.3f), %llu slots %llu chunks %llu calls, %llu regions\n",
                        round, d[0], d[1] / 1e6, d[2] / 1e6, d[3] / 1e6, d[4] / 1e6, d[5], d[6], d[7], d[8] / 1e6, d[9] / 1e6, d[10] / 1e6, d[11] / 1e6, d[12], d[13], d[14], d[15]);
        }
        if (round == 0) BWAMS_HIP(hipEventRecord(s->ev[9], st));
        unsigned long long n_req = 0, n_rest = 0;
        BWAMS_HIP(hipMemcpyAsync(&n_req, &b->d_ctr->n_req, sizeof n_req, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipMemcpyAsync(&n_rest, &b->d_ctr->n_rest, sizeof n_rest, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        if (n_req == 0) break;
        // A round costs about as much as ~10^5 extensions whatever it holds (launches, the selection's walk of the heaviest
        // reads).  When the seeds still undecided — at most n_rest, the slots behind this round's requests — are few beside what
        // has been extended already, extending them all now (as the reference does with every seed) is cheaper than the rounds
        // that would sort out which of them are dead.
        const bool few_left = !adaptive_off && (int64_t)(n_req + n_rest) * 32 < tot_left + tot_right;
        if (round + 1 >= kMaxRounds || few_left) launch_ext_request_rest(A, st);
    }
    BWAMS_HIP(hipEventRecord(s->ev[11], st));
    BWAMS_HIP(hipGetLastError());
    s->n_rounds = round + 1;
    s->n_left = tot_left; s->n_right = tot_right;
    s->ext_done = true;
    s->opt = *opt;
    if (n_regs) *n_regs = s->n_seeds;
    return BWAMS_OK;
}

int bwams_extend_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off, int32_t *seed_aln) {
    if (!b || !b->chain || !(b->chain->ext_done || b->chain->built)) {
        set_last_error("bwams_extend_fetch: no regions on the device");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (s->n_seeds > reg_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (s->n_seeds) BWAMS_HIP(hipMemcpyAsync(regs, s->regs.p, (size_t)s->n_seeds * sizeof(bwams_alnreg_t), hipMemcpyDeviceToHost, st));
    if (reg_off) BWAMS_HIP(hipMemcpyAsync(reg_off, s->chain_off.as<int64_t>() + (s->nseq + 1), (size_t)(s->nseq + 1) * 8, hipMemcpyDeviceToHost, st));
    if (seed_aln && s->n_seeds)
        BWAMS_HIP(hipMemcpy2DAsync(seed_aln, 4, reinterpret_cast<const char *>(s->seeds.p) + offsetof(bwams_chain_seed_t, aln),
                                   sizeof(bwams_chain_seed_t), 4, (size_t)s->n_seeds, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

/* ---------------------------------------------- the tail of mem_kernel2_core ---- */

int bwams_dedup_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int64_t *n_regs) {
    if (!b || !b->chain || !b->chain->ext_done) {
        set_last_error("bwams_dedup_run: run bwams_extend_run first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_dedup_run");
    if (rc) return rc;
    ChainState *s = b->chain;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    s->dedup_done = s->pair_done = false;
    const int64_t N = s->n_seeds, n1 = s->nseq + 1;
    const int64_t L = b->max_read_len > 1 ? b->max_read_len : 1;
    // strips for the global alignment: as many lanes as 1 GiB of (h, e) rows allows, at most 64 Ki
    int64_t n_lanes = ((int64_t)1 << 30) / ((L + 2) * 8);
    n_lanes = n_lanes > 65536 ? 65536 : n_lanes < 64 ? 64 : n_lanes;
    const int64_t n_waves = (int64_t)b->cu_count * 4, n_waves_small = (int64_t)b->cu_count * 16;
    BWAMS_HIP(s->dd_regs.ensure((size_t)(N + 1) * sizeof(bwams_alnreg_t)));
    BWAMS_HIP(s->dd_out.ensure((size_t)(N + 1) * sizeof(bwams_alnreg_t)));
    BWAMS_HIP(s->dd_ord.ensure((size_t)(N + 1) * 4));
    BWAMS_HIP(s->dd_srt.ensure(dedup_sortrec_bytes(N)));
    BWAMS_HIP(s->dd_eh.ensure((size_t)(n_lanes + 2 * n_waves + n_waves_small) * (size_t)(L + 2) * 8));
    BWAMS_HIP(s->dd_nout.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->dd_wide.ensure((size_t)n1 * 16));
    BWAMS_HIP(s->dd_off.ensure((size_t)n1 * 8));
    DedupArgs D;
    D.regs = s->dd_regs.as<bwams_alnreg_t>();
    D.seed_off = s->chain_off.as<int64_t>() + n1;
    D.enc = b->d_enc; D.cum = b->d_cum; D.nseq = s->nseq; D.ref = b->idx->fmi.ref;
    if ((rc = dev_bns(b->idx, &D.bns))) return rc;
    D.opt = *opt;
    D.ord = s->dd_ord.as<int32_t>(); D.srt = s->dd_srt.p; D.eh = s->dd_eh.as<int2>(); D.eh_lanes = n_lanes;
    D.max_read_len = (int32_t)L; D.n_out = s->dd_nout.as<int32_t>();
    BWAMS_HIP(s->heavy.ensure((size_t)n1 * 4));
    D.force_seq = knobs().dedup_seq;      // 1: every read through the one-lane form (tests)
    BWAMS_HIP(s->dd_light.ensure((size_t)n1 * 4));
    D.heavy = s->heavy.as<int32_t>(); D.light = s->dd_light.as<int32_t>();
    D.n_heavy_ctr = &b->d_ctr->dedup_heavy; D.ticket = &b->d_ctr->dedup_ticket; D.n_light_ctr = &b->d_ctr->dedup_light;
    D.ticket2 = &b->d_ctr->dedup_ticket2; D.ticket3 = &b->d_ctr->dedup_ticket3;
    const bool vb_dd = knobs().verbose != 0;
    const bool verbose_dd = vb_dd;
    D.dbg = verbose_dd ? b->d_ctr->dbg : nullptr;
    if (verbose_dd) BWAMS_HIP(hipMemsetAsync(b->d_ctr->dbg, 0, sizeof b->d_ctr->dbg, st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->dedup_heavy, 0, 3 * sizeof(unsigned long long), st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->dedup_ticket2, 0, 2 * sizeof(unsigned long long), st));
    BWAMS_HIP(hipEventRecord(s->ev[12], st));
    // work on a copy: bwams_extend_fetch stays valid
    if (N) BWAMS_HIP(hipMemcpyAsync(D.regs, s->regs.p, (size_t)N * sizeof(bwams_alnreg_t), hipMemcpyDeviceToDevice, st));
    BWAMS_HIP(hipMemsetAsync(D.n_out, 0, (size_t)n1 * 4, st));
    if (launch_dedup(D, n_lanes, n_waves, n_waves_small, st, s->aux[0], s->aux[1], s->aux[2], s->fork, s->join[0], s->join[1], s->join[2])) {
        set_last_error("bwams_dedup_run: stream fork/join failed");
        return BWAMS_ERR_DEVICE;
    }
    int64_t total = 0;
    if (s->nseq > 0) {
        widen2_kernel<<<(unsigned)((2 * n1 + 255) / 256), 256, 0, st>>>(D.n_out, D.n_out, s->nseq, s->dd_wide.as<int64_t>());
        if ((rc = scan_rows(b, s->dd_wide.as<int64_t>(), s->dd_off.as<int64_t>(), 1, n1))) return rc;
        launch_dedup_gather(D, s->dd_off.as<int64_t>(), s->dd_out.as<bwams_alnreg_t>(), st);
        BWAMS_HIP(hipMemcpyAsync(&total, s->dd_off.as<int64_t>() + s->nseq, 8, hipMemcpyDeviceToHost, st));
    } else {
        BWAMS_HIP(hipMemsetAsync(s->dd_off.p, 0, 8, st));          // an empty chunk: reg_off = {0}
    }
    BWAMS_HIP(hipEventRecord(s->ev[13], st));
    if (verbose_dd) BWAMS_HIP(hipMemcpyAsync(b->h_ctr->dbg, b->d_ctr->dbg, sizeof b->d_ctr->dbg, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(hipGetLastError());
    if (verbose_dd) {
        const unsigned long long *d = b->h_ctr->dbg;
        fprintf(stderr, "[bwams_dedup_run] largest wave instance: %llu reads, %llu slots, %llu alive; Mcycles: load %.1f sort(end) %.1f pairs %.1f reload %.1f sort(score) %.1f store %.1f; "
                        "longest read: sort(end) %.2f pairs %.2f sort(score) %.2f, whole %.2f (read %llu, %llu regions; %llu patch alignments in %.2f, %llu scan trips); all reads: %llu patch alignments in %.1f\n",
                d[0], d[1], d[2], d[3] / 1e6, d[4] / 1e6, d[5] / 1e6, d[6] / 1e6, d[7] / 1e6, d[8] / 1e6, d[9] / 1e6, d[10] / 1e6, d[11] / 1e6, d[12] / 1e6, d[13], d[14],
                d[15], d[16] / 1e6, d[17], d[18], d[19] / 1e6);
    }
    s->n_final = total;
    s->dedup_done = true;
    if (n_regs) *n_regs = total;
    return BWAMS_OK;
}

int bwams_dedup_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off) {
    if (!b || !b->chain || !b->chain->dedup_done) {
        set_last_error("bwams_dedup_fetch: run bwams_dedup_run first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (s->n_final > reg_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (s->n_final) BWAMS_HIP(hipMemcpyAsync(regs, s->dd_out.p, (size_t)s->n_final * sizeof(bwams_alnreg_t), hipMemcpyDeviceToHost, st));
    if (reg_off) BWAMS_HIP(hipMemcpyAsync(reg_off, s->dd_off.p, (size_t)(s->nseq + 1) * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

/* ------------------------------------------------- mate rescue, mem_mark_primary_se, mem_pair ---- */

static int pair_run_impl(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_pestat_t pes[4], int64_t id_base, int32_t flags,
                         int primary5_T, int no_pairing, int64_t *n_regs, int64_t *n_tasks);

int bwams_pair_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_pestat_t pes[4], int64_t id_base, int32_t flags,
                   int64_t *n_regs, int64_t *n_tasks) {
    return pair_run_impl(b, opt, pes, id_base, flags, -1, 0, n_regs, n_tasks);
}

int bwams_pair_run_sam(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sam_opt, const bwams_pestat_t pes[4],
                       int64_t id_base, int32_t flags, int64_t *n_regs, int64_t *n_tasks) {
    if (!sam_opt) return pair_run_impl(b, opt, pes, id_base, flags, -1, 0, n_regs, n_tasks);
    if (sam_opt->T < 0) {
        set_last_error("bwams_pair_run_sam: T must not be negative");
        return BWAMS_ERR_ARG;
    }
    if (sam_opt->flag & BWAMS_MEM_F_NO_RESCUE) flags |= BWAMS_PAIR_NO_RESCUE;
    return pair_run_impl(b, opt, pes, id_base, flags, (sam_opt->flag & BWAMS_MEM_F_PRIMARY5) ? sam_opt->T : -1,
                         (sam_opt->flag & BWAMS_MEM_F_NOPAIRING) != 0, n_regs, n_tasks);
}

static int pair_run_impl(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_pestat_t pes[4], int64_t id_base, int32_t flags,
                         int primary5_T, int no_pairing, int64_t *n_regs, int64_t *n_tasks) {
    const int single_end = (flags & BWAMS_PAIR_SINGLE_END) != 0;
    const int no_rescue = (flags & BWAMS_PAIR_NO_RESCUE) || single_end, use_ert = (flags & BWAMS_PAIR_USE_ERT) != 0;
    static const bwams_pestat_t no_pes[4] = {{0, 0, 1, 0, 0., 0.}, {0, 0, 1, 0, 0., 0.}, {0, 0, 1, 0, 0., 0.}, {0, 0, 1, 0, 0., 0.}};
    if (single_end && !pes) pes = no_pes;
    if (!b || !b->chain || !b->chain->dedup_done) {
        set_last_error("bwams_pair_run: run bwams_dedup_run first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_pair_run");
    if (rc) return rc;
    ChainState *s = b->chain;
    if (!pes || (!single_end && (s->nseq & 1))) {
        set_last_error("bwams_pair_run: needs the insert-size statistics and an even number of reads (ends of pair p at 2p, 2p + 1)");
        return BWAMS_ERR_ARG;
    }
    int tmax = 1;
    for (int k = 0; k < 4; ++k)
        if (!pes[k].failed && pes[k].high - pes[k].low + b->max_read_len > tmax) tmax = pes[k].high - pes[k].low + b->max_read_len;
    if (!no_rescue && (b->max_read_len > 512 || tmax > kKswMaxTarget)) {
        set_last_error("bwams_pair_run: mate rescue needs reads of at most 512 bases and windows (high - low + read length) of at most 20000");
        return BWAMS_ERR_UNSUPPORTED;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    s->pair_done = false;
    const int64_t nseq = s->nseq, n1 = nseq + 1;
    BWAMS_HIP(s->pr_na.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->pr_wide.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->pr_offs.ensure((size_t)n1 * 16));
    BWAMS_HIP(s->pr_nfin.ensure((size_t)n1 * 4)); BWAMS_HIP(s->pr_npri.ensure((size_t)n1 * 4)); BWAMS_HIP(s->pr_nsw.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->pr_full.ensure((size_t)n1));
    BWAMS_HIP(s->pr_owide.ensure((size_t)n1 * 8)); BWAMS_HIP(s->pr_ooff.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->pr_res.ensure((size_t)(nseq / 2 + 1) * sizeof(bwams_pair_t)));
    PairArgs A;
    A.regs = s->dd_out.as<bwams_alnreg_t>(); A.reg_off = s->dd_off.as<int64_t>();
    A.enc = b->d_enc; A.cum = b->d_cum; A.nseq = nseq; A.ref = b->idx->fmi.ref;
    if ((rc = dev_bns(b->idx, &A.bns))) return rc;
    A.opt = *opt;
    for (int k = 0; k < 4; ++k) A.pes[k] = pes[k];
    A.id_base = id_base; A.no_rescue = no_rescue ? 1 : 0; A.pass = 0;
    A.drop_plan = knobs().pair_drop_plan;                 // test knob: exercise the second pass
    A.use_ert = use_ert ? 1 : 0;
    A.single_end = single_end;
    A.no_pairing = no_pairing; A.primary5_T = primary5_T;
    A.na = s->pr_na.as<int32_t>();
    int64_t *aoff = s->pr_offs.as<int64_t>(), *ooff = aoff + n1;
    A.aoff = aoff; A.ooff = ooff;
    A.n_fin = s->pr_nfin.as<int32_t>(); A.n_pri = s->pr_npri.as<int32_t>(); A.n_sw = s->pr_nsw.as<int32_t>();
    A.full = s->pr_full.as<uint8_t>(); A.ctr = b->d_ctr;
    A.anchor = nullptr; A.slot_read = nullptr; A.n_slots = 0; A.task = nullptr; A.trb = nullptr; A.tl1 = nullptr; A.aln = nullptr;
    A.pool = nullptr; A.ord = nullptr; A.zbuf = nullptr; A.srt = nullptr; A.heavy = nullptr;
    BWAMS_HIP(hipEventRecord(s->ev[14], st));
    BWAMS_HIP(hipMemsetAsync(A.full, 0, (size_t)n1, st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->pair_full, 0, 2 * sizeof(unsigned long long), st));
    // anchors per read, pool capacities
    launch_pair_count(A, s->pr_wide.as<int64_t>(), st);
    if ((rc = scan_rows(b, s->pr_wide.as<int64_t>(), aoff, 1, n1))) return rc;
    launch_pair_cap(A, s->pr_wide.as<int64_t>(), st);
    if ((rc = scan_rows(b, s->pr_wide.as<int64_t>(), ooff, 1, n1))) return rc;
    int64_t n_slots = 0, n_pool = 0;
    BWAMS_HIP(hipMemcpyAsync(&n_slots, aoff + nseq, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipMemcpyAsync(&n_pool, ooff + nseq, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    A.n_slots = n_slots;
    const int64_t E1 = 4 * n_slots + 1;
    BWAMS_HIP(s->pr_anchor.ensure((size_t)(n_slots + 1) * 4)); BWAMS_HIP(s->pr_slot.ensure((size_t)(n_slots + 1) * 4));
    BWAMS_HIP(s->pr_task.ensure((size_t)E1 * 4)); BWAMS_HIP(s->pr_trb.ensure((size_t)E1 * 8)); BWAMS_HIP(s->pr_tl1.ensure((size_t)E1 * 4));
    BWAMS_HIP(s->pr_twide.ensure((size_t)E1 * 24)); BWAMS_HIP(s->pr_toffs.ensure((size_t)E1 * 24));
    BWAMS_HIP(s->pr_pool.ensure((size_t)(n_pool + 1) * sizeof(bwams_alnreg_t)));
    BWAMS_HIP(s->pr_ord.ensure((size_t)(n_pool + 1) * 4)); BWAMS_HIP(s->pr_z.ensure((size_t)(n_pool + 1) * 4));
    BWAMS_HIP(s->pr_srt.ensure((size_t)(n_pool + 1) * 24));
    A.anchor = s->pr_anchor.as<int32_t>(); A.slot_read = s->pr_slot.as<int32_t>();
    A.task = s->pr_task.as<int32_t>(); A.trb = s->pr_trb.as<int64_t>(); A.tl1 = s->pr_tl1.as<int32_t>();
    A.pool = s->pr_pool.as<bwams_alnreg_t>(); A.ord = s->pr_ord.as<int32_t>(); A.zbuf = s->pr_z.as<int32_t>(); A.srt = s->pr_srt.p;
    const int pr_trace = knobs().trace_pair;     // debugging aid: a synchronisation and a line per launch
#define PR_TRACE(msg) do { if (pr_trace) { BWAMS_HIP(hipStreamSynchronize(st)); fprintf(stderr, "[bwams_pair_run] %s\n", msg); } } while (0)
    PR_TRACE("count / cap done");
    launch_pair_slots(A, st);
    PR_TRACE("slots done");
    BWAMS_HIP(s->heavy.ensure((size_t)n1 * 4));
    A.heavy = s->heavy.as<int32_t>();
    SwParams prm;
    sw_params(*opt, 0, &prm);
    s->pr_tasks = 0; s->pr_redone = 0;
    for (int pass = 0; pass < 2; ++pass) {
        A.pass = pass;
        int64_t tot[3] = {0, 0, 0};
        launch_pair_plan(A, s->pr_twide.as<int64_t>(), st);
        PR_TRACE("plan done");
        if ((rc = scan_rows(b, s->pr_twide.as<int64_t>(), s->pr_toffs.as<int64_t>(), 3, E1))) return rc;
        for (int r = 0; r < 3; ++r)
            BWAMS_HIP(hipMemcpyAsync(&tot[r], s->pr_toffs.as<int64_t>() + r * E1 + (E1 - 1), 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        if (tot[1] >= ((int64_t)1 << 31) || tot[2] >= ((int64_t)1 << 31)) {
            set_last_error("bwams_pair_run: rescue windows exceed the 31-bit offsets of SeqPair; use smaller chunks");
            return BWAMS_ERR_CAPACITY;
        }
        BWAMS_HIP(s->pr_pairs.ensure((size_t)(tot[0] + 1) * sizeof(bwams_seqpair_t)));
        BWAMS_HIP(s->pr_tref.ensure((size_t)tot[1] + 64)); BWAMS_HIP(s->pr_tqer.ensure((size_t)tot[2] + 64));
        BWAMS_HIP(s->pr_aln.ensure((size_t)(tot[0] + 1) * 28));
        A.aln = s->pr_aln.as<int32_t>();
        launch_pair_build(A, s->pr_toffs.as<int64_t>(), s->pr_pairs.as<bwams_seqpair_t>(), s->pr_tref.as<uint8_t>(), s->pr_tqer.as<uint8_t>(),
                          b->cu_count, st);
        PR_TRACE("build done");
        if (tot[0] > 0 && launch_ksw(s->pr_pairs.as<bwams_seqpair_t>(), tot[0], s->pr_tref.as<uint8_t>(), s->pr_tqer.as<uint8_t>(), prm,
                                     ((b->max_read_len + 15) / 16) * 16, tmax, s->pr_aln.p, b->d_ctr, b->cu_count, st)) {
            set_last_error("bwams_pair_run: rescue window too long for the local-SW kernel");
            return BWAMS_ERR_UNSUPPORTED;
        }
        PR_TRACE("ksw done");
        BWAMS_HIP(hipMemsetAsync(&b->d_ctr->pair_heavy, 0, 2 * sizeof(unsigned long long), st));
        launch_pair_post(A, b->cu_count, st);
#ifdef BWAMS_PAIRDBG
        if (knobs().verbose) {
            unsigned long long d[80];
            BWAMS_HIP(hipStreamSynchronize(st));
            BWAMS_HIP(hipMemcpy(d, b->d_ctr->dbg, sizeof d, hipMemcpyDeviceToHost));
            fprintf(stderr, "[pair_post_wave] reads %llu (mean %.0f regions at the end, %.1f anchors, %.1f rescues), %.3f ms of a wave per read (longest %.3f ms); sorts %llu = %.1f per read, %.3f ms per read; "
                    "with equal keys %llu, %.3f ms per read in them\n", d[20], d[20] ? (double)d[27] / d[20] : 0.0, d[20] ? (double)d[28] / d[20] : 0.0, d[20] ? (double)d[29] / d[20] : 0.0,
                    d[20] ? d[24] * 1e-5 / d[20] : 0.0, d[25] * 1e-5, d[21], d[20] ? (double)d[21] / d[20] : 0.0, d[20] ? d[26] * 1e-5 / d[20] : 0.0, d[22], d[20] ? d[23] * 1e-5 / d[20] : 0.0);
            BWAMS_HIP(hipMemsetAsync(b->d_ctr->dbg + 20, 0, 10 * sizeof(unsigned long long), st));
        }
#endif
        PR_TRACE("post done");
        s->pr_tasks += tot[0];
        unsigned long long flags[2] = {0, 0};
        BWAMS_HIP(hipMemcpyAsync(flags, &b->d_ctr->pair_full, sizeof flags, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        if (flags[1]) {
            set_last_error("bwams_pair_run: internal error, a rescue alignment was missing in the second pass");
            return BWAMS_ERR_DEVICE;
        }
        if (pass == 0) s->pr_redone = (int64_t)flags[0];
        if (pass == 1 || flags[0] == 0) break;
    }
    // mem_mark_primary_se of every read, regions in final order, then mem_pair
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->pair_heavy, 0, 2 * sizeof(unsigned long long), st));
    BWAMS_HIP(hipMemsetAsync(&b->d_ctr->pair_ticket2, 0, sizeof(unsigned long long), st));
    launch_pair_mark(A, b->cu_count, st);
    PR_TRACE("mark done");
    launch_pair_widen(A, s->pr_owide.as<int64_t>(), st);
    if ((rc = scan_rows(b, s->pr_owide.as<int64_t>(), s->pr_ooff.as<int64_t>(), 1, n1))) return rc;
    int64_t total = 0;
    BWAMS_HIP(hipMemcpyAsync(&total, s->pr_ooff.as<int64_t>() + nseq, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(s->pr_out.ensure((size_t)(total + 1) * sizeof(bwams_alnreg_t)));
    launch_pair_gather(A, s->pr_ooff.as<int64_t>(), s->pr_out.as<bwams_alnreg_t>(), st);
    PR_TRACE("gather done");
    launch_pair_reorder5(A, s->pr_ooff.as<int64_t>(), s->pr_out.as<bwams_alnreg_t>(), st);
    PR_TRACE("reorder5 done");
    if (!single_end) launch_pair_pair(A, s->pr_ooff.as<int64_t>(), s->pr_out.as<bwams_alnreg_t>(), s->pr_res.as<bwams_pair_t>(), st);
    PR_TRACE("pair done");
    BWAMS_HIP(hipEventRecord(s->ev[15], st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(hipGetLastError());
    s->pr_total = total;
    s->pr_single = single_end != 0;
    s->pair_done = true;
    if (n_regs) *n_regs = total;
    if (n_tasks) *n_tasks = s->pr_tasks;
    return BWAMS_OK;
}

int bwams_pair_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off, bwams_pair_t *pairs) {
    if (!b || !b->chain || !b->chain->pair_done) {
        set_last_error("bwams_pair_fetch: run bwams_pair_run first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (s->pr_total > reg_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (regs && s->pr_total) BWAMS_HIP(hipMemcpyAsync(regs, s->pr_out.p, (size_t)s->pr_total * sizeof(bwams_alnreg_t), hipMemcpyDeviceToHost, st));
    if (reg_off) BWAMS_HIP(hipMemcpyAsync(reg_off, s->pr_ooff.p, (size_t)(s->nseq + 1) * 8, hipMemcpyDeviceToHost, st));
    if (pairs && s->nseq > 1 && !s->pr_single) BWAMS_HIP(hipMemcpyAsync(pairs, s->pr_res.p, (size_t)(s->nseq / 2) * sizeof(bwams_pair_t), hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

/* ------------------------------------------------------------- mem_reg2aln ---- */

// mem_approx_mapq_se (bwamem.cpp:1983-2008) on the host: a dozen double operations per region, with the C library's log
static int approx_mapq_se(const bwams_mem_opt_t *opt, const bwams_alnreg_t *a) {
    int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
    double identity;
    const int coef_len = opt->mapq_coef_len;
    const double coef_fac = coef_len > 0 ? log((double)coef_len) : 0.;
    sub = a->csub > sub ? a->csub : sub;
    if (sub >= a->score) return 0;
    l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
    identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
    if (a->score == 0) mapq = 0;
    else if (coef_len > 0) {
        double tmp = l < coef_len ? 1. : coef_fac / log(l);
        tmp *= identity * identity;
        mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
    } else {
        mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - a->frac_rep) + .499);
    return mapq;
}

static int reg2aln_impl(bwams_batch_t *b, const bwams_mem_opt_t *opt, int32_t source, const uint8_t *only, int64_t *n_aln,
                        int64_t *n_cigar_ops, int64_t *md_bytes) {
    if (!b || !b->chain || (source == 0 && !b->chain->dedup_done) || (source == 1 && !b->chain->pair_done) || source < 0 || source > 1) {
        set_last_error("bwams_reg2aln_run: run bwams_dedup_run (source 0) or bwams_pair_run (source 1) first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_reg2aln_run");
    if (rc) return rc;
    if (!b->idx->d_ref) {
        set_last_error("bwams_reg2aln_run: the index was opened without its .0123 reference");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    s->al_done = false;
    const int64_t n = source ? s->pr_total : s->n_final;
    RegAlnArgs A;
    memset(&A, 0, sizeof A);
    A.regs = source ? s->pr_out.as<bwams_alnreg_t>() : s->dd_out.as<bwams_alnreg_t>();
    A.reg_off = source ? s->pr_ooff.as<int64_t>() : s->dd_off.as<int64_t>();
    A.n_regs = n; A.nseq = s->nseq;
    A.enc = b->d_enc; A.cum = b->d_cum; A.ref = b->idx->fmi.ref;
    if ((rc = dev_bns(b->idx, &A.bns))) return rc;
    A.opt = *opt;
    A.only = only;
    const int64_t n1 = n + 1;
    BWAMS_HIP(s->al_need.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->al_cls.ensure((size_t)n1 * 4));
    BWAMS_HIP(s->al_off.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->al_list.ensure((size_t)n1 * 4 * 4));
    BWAMS_HIP(s->al_rec.ensure((size_t)n1 * sizeof(bwams_aln_t)));
    BWAMS_HIP(s->al_wide.ensure((size_t)(2 * n1) * 8));
    BWAMS_HIP(s->al_offs.ensure((size_t)(2 * n1) * 8));
    BWAMS_HIP(s->al_cnt.ensure(256));
    A.need = s->al_need.as<int64_t>(); A.cls = s->al_cls.as<int32_t>(); A.scr_off = s->al_off.as<int64_t>();
    A.list = s->al_list.as<int32_t>(); A.n_list = s->al_cnt.as<unsigned long long>(); A.rec = s->al_rec.as<bwams_aln_t>();
    int64_t tot[2] = {0, 0};
    if (n > 0) {
        BWAMS_HIP(hipMemsetAsync(s->al_cnt.p, 0, 256, st));
        BWAMS_HIP(hipMemsetAsync(s->al_need.as<int64_t>() + n, 0, 8, st));
        launch_aln_plan(A, st);
        if ((rc = scan_rows(b, A.need, s->al_off.as<int64_t>(), 1, n1))) return rc;
        int64_t scr_bytes = 0;
        BWAMS_HIP(hipMemcpyAsync(&scr_bytes, s->al_off.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        BWAMS_HIP(s->al_scr.ensure((size_t)scr_bytes + 64));
        A.scr = s->al_scr.as<uint8_t>();
        launch_aln_run(A, b->cu_count, st);
        launch_aln_sizes(A, s->al_wide.as<int64_t>(), st);
        if ((rc = scan_rows(b, s->al_wide.as<int64_t>(), s->al_offs.as<int64_t>(), 2, n1))) return rc;
        BWAMS_HIP(hipMemcpyAsync(&tot[0], s->al_offs.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipMemcpyAsync(&tot[1], s->al_offs.as<int64_t>() + n1 + n, 8, hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipStreamSynchronize(st));
        BWAMS_HIP(s->al_cig.ensure((size_t)(tot[0] + 1) * 4));
        BWAMS_HIP(s->al_md.ensure((size_t)tot[1] + 16));
        launch_aln_gather(A, s->al_offs.as<int64_t>(), s->al_cig.as<uint32_t>(), s->al_md.as<char>(), st);
        BWAMS_HIP(hipStreamSynchronize(st));
        BWAMS_HIP(hipGetLastError());
#ifdef BWAMS_ALNDBG
        if (knobs().verbose) {
            unsigned long long c[32];
            BWAMS_HIP(hipMemcpy(c, s->al_cnt.p, 256, hipMemcpyDeviceToHost));
            fprintf(stderr, "[reg2aln] regions %lld: class lists %llu / %llu / %llu / %llu; wave kernel: %llu regions, per region setup %.1f us, DP %.1f us (%.2f DPs, mean band %.1f, %.0f rows), "
                    "traceback %.1f us, NM/MD + record %.1f us, slowest region %.1f us\n", (long long)n, c[0], c[1], c[2], c[3], c[8], c[8] ? c[9] * 1e-2 / c[8] : 0.0, c[8] ? c[10] * 1e-2 / c[8] : 0.0,
                    c[8] ? (double)c[13] / c[8] : 0.0, c[13] ? (double)c[14] / c[13] : 0.0, c[8] ? (double)c[15] / c[8] : 0.0, c[8] ? c[11] * 1e-2 / c[8] : 0.0, c[8] ? c[12] * 1e-2 / c[8] : 0.0, c[16] * 1e-2);
        }
#endif
    }
    s->al_n = n; s->al_ncig = tot[0]; s->al_nmd = tot[1]; s->al_source = source; s->al_done = true;
    s->opt = *opt;
    if (n_aln) *n_aln = n;
    if (n_cigar_ops) *n_cigar_ops = tot[0];
    if (md_bytes) *md_bytes = tot[1];
    return BWAMS_OK;
}

int bwams_reg2aln_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, int32_t source, int64_t *n_aln, int64_t *n_cigar_ops,
                      int64_t *md_bytes) {
    return reg2aln_impl(b, opt, source, nullptr, n_aln, n_cigar_ops, md_bytes);
}

int bwams_reg2aln_run_sam(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, const bwams_pestat_t *pes,
                          int64_t *n_aln, int64_t *n_needed, int64_t *n_cigar_ops, int64_t *md_bytes) {
    if (!b || !sopt || !b->chain || !b->chain->pair_done || b->chain->pr_single == (pes != nullptr)) {
        set_last_error("bwams_reg2aln_run_sam: run bwams_pair_run first (BWAMS_PAIR_SINGLE_END and pes = NULL, or the paired-end form and its pes)");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_reg2aln_run_sam");
    if (rc) return rc;
    ChainState *s = b->chain;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    const int64_t n = s->pr_total;
    BWAMS_HIP(s->al_only.ensure((size_t)n + 64));
    BWAMS_HIP(hipMemsetAsync(s->al_only.p, 0, (size_t)n + 1, st));
    SamArgs A;
    memset(&A, 0, sizeof A);
    A.regs = s->pr_out.as<bwams_alnreg_t>(); A.reg_off = s->pr_ooff.as<int64_t>();
    A.n_regs = n; A.nseq = s->nseq;
    A.opt = *opt; A.sopt = *sopt;
    A.pairs = pes ? s->pr_res.as<bwams_pair_t>() : nullptr;
    if (pes) memcpy(A.pes, pes, sizeof A.pes);
    launch_sam_need(A, s->al_only.as<uint8_t>(), b->cu_count, st);
    if (n_needed) {
        // a count for the caller (and the bench): one reduction over the mask
        size_t tb = 0;
        BWAMS_HIP(s->sm_bad.ensure(64));
        int64_t *d_sum = s->sm_bad.as<int64_t>() + 1;
        BWAMS_HIP(rocprim::reduce(nullptr, tb, s->al_only.as<uint8_t>(), d_sum, (int64_t)0, (size_t)(n > 0 ? n : 0), rocprim::plus<int64_t>(), st));
        if (tb > b->tmp_bytes) {
            BWAMS_HIP(hipStreamSynchronize(st));
            if (b->d_tmp) (void)hipFree(b->d_tmp);
            b->d_tmp = nullptr;
            BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
            b->tmp_bytes = tb;
        }
        tb = b->tmp_bytes;
        BWAMS_HIP(rocprim::reduce(b->d_tmp, tb, s->al_only.as<uint8_t>(), d_sum, (int64_t)0, (size_t)(n > 0 ? n : 0), rocprim::plus<int64_t>(), st));
        BWAMS_HIP(hipMemcpyAsync(n_needed, d_sum, 8, hipMemcpyDeviceToHost, st));
    }
    return reg2aln_impl(b, opt, 1, s->al_only.as<uint8_t>(), n_aln, n_cigar_ops, md_bytes);
}

int bwams_reg2aln_fetch(bwams_batch_t *b, bwams_aln_t *aln, int64_t aln_cap, uint32_t *cigar, int64_t cigar_cap, char *md, int64_t md_cap) {
    if (!b || !b->chain || !b->chain->al_done) {
        set_last_error("bwams_reg2aln_fetch: run bwams_reg2aln_run first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (s->al_n > aln_cap || s->al_ncig > cigar_cap || s->al_nmd > md_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    std::vector<bwams_alnreg_t> regs((size_t)s->al_n);
    if (s->al_n) {
        BWAMS_HIP(hipMemcpyAsync(aln, s->al_rec.p, (size_t)s->al_n * sizeof(bwams_aln_t), hipMemcpyDeviceToHost, st));
        BWAMS_HIP(hipMemcpyAsync(regs.data(), s->al_source ? s->pr_out.p : s->dd_out.p, (size_t)s->al_n * sizeof(bwams_alnreg_t),
                                 hipMemcpyDeviceToHost, st));
        if (s->al_ncig) BWAMS_HIP(hipMemcpyAsync(cigar, s->al_cig.p, (size_t)s->al_ncig * 4, hipMemcpyDeviceToHost, st));
        if (s->al_nmd) BWAMS_HIP(hipMemcpyAsync(md, s->al_md.p, (size_t)s->al_nmd, hipMemcpyDeviceToHost, st));
    }
    BWAMS_HIP(hipStreamSynchronize(st));
    for (int64_t k = 0; k < s->al_n; ++k)
        if (aln[k].rid >= 0) aln[k].mapq = regs[(size_t)k].secondary < 0 ? approx_mapq_se(&s->opt, &regs[(size_t)k]) : 0;
    return BWAMS_OK;
}

/* ------------------------------------------------------------ SAM text (single-end) ---- */

int bwams_index_set_contig_names(bwams_index_t *ix, const char *names, const int32_t *name_off) {
    if (!ix || !names || !name_off) return BWAMS_ERR_ARG;
    DevBns bns;
    int rc = dev_bns(ix, &bns);                 // materialises the one-sequence default
    if (rc) return rc;
    const int32_t n = ix->n_seqs;
    for (int32_t i = 0; i < n; ++i)
        if (name_off[i] < 0 || name_off[i + 1] <= name_off[i] || names[name_off[i + 1] - 1] != 0) {
            set_last_error("bwams_index_set_contig_names: names must be NUL-terminated, back to back, name_off[n_seqs + 1] ascending");
            return BWAMS_ERR_ARG;
        }
    BWAMS_HIP(hipSetDevice(ix->device));
    if (ix->d_ctg_names) { (void)hipFree(ix->d_ctg_names); (void)hipFree(ix->d_ctg_off); ix->d_ctg_names = ix->d_ctg_off = nullptr; }
    BWAMS_HIP(dev_malloc(&ix->d_ctg_names, (size_t)name_off[n]));
    BWAMS_HIP(dev_malloc(&ix->d_ctg_off, (size_t)(n + 1) * 4));
    BWAMS_HIP(hipMemcpy(ix->d_ctg_names, names, (size_t)name_off[n], hipMemcpyHostToDevice));
    BWAMS_HIP(hipMemcpy(ix->d_ctg_off, name_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    return BWAMS_OK;
}

int bwams_index_set_contig_annos(bwams_index_t *ix, const char *annos, const int32_t *anno_off) {
    if (!ix || !annos || !anno_off) return BWAMS_ERR_ARG;
    DevBns bns;
    int rc = dev_bns(ix, &bns);
    if (rc) return rc;
    const int32_t n = ix->n_seqs;
    for (int32_t i = 0; i < n; ++i)
        if (anno_off[i] < 0 || anno_off[i + 1] <= anno_off[i] || annos[anno_off[i + 1] - 1] != 0) {
            set_last_error("bwams_index_set_contig_annos: annotations must be NUL-terminated, back to back, anno_off[n_seqs + 1] ascending");
            return BWAMS_ERR_ARG;
        }
    BWAMS_HIP(hipSetDevice(ix->device));
    if (ix->d_ctg_annos) { (void)hipFree(ix->d_ctg_annos); (void)hipFree(ix->d_ctg_anno_off); ix->d_ctg_annos = ix->d_ctg_anno_off = nullptr; }
    BWAMS_HIP(dev_malloc(&ix->d_ctg_annos, (size_t)anno_off[n]));
    BWAMS_HIP(dev_malloc(&ix->d_ctg_anno_off, (size_t)(n + 1) * 4));
    BWAMS_HIP(hipMemcpy(ix->d_ctg_annos, annos, (size_t)anno_off[n], hipMemcpyHostToDevice));
    BWAMS_HIP(hipMemcpy(ix->d_ctg_anno_off, anno_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    return BWAMS_OK;
}

int bwams_sam_upload(bwams_batch_t *b, const char *names, const int64_t *name_off, const char *quals, const char *comments,
                     const int64_t *comment_off) {
    if (!b || !names || !name_off || (comments && !comment_off)) {
        set_last_error("bwams_sam_upload: names and their offsets are required; comments come with offsets");
        return BWAMS_ERR_ARG;
    }
    if (b->nseq <= 0 || !b->d_cum) {
        set_last_error("bwams_sam_upload: upload the reads first (bwams_seed_upload)");
        return BWAMS_ERR_ARG;
    }
    BWAMS_HIP(hipSetDevice(b->idx->device));
    ChainState *s;
    int rc = get_state(b, &s);
    if (rc) return rc;
    hipStream_t st = b->stream;
    const int64_t nseq = b->nseq, n1 = nseq + 1;
    if (name_off[0] != 0 || (comments && comment_off[0] != 0)) {
        set_last_error("bwams_sam_upload: offsets start at 0");
        return BWAMS_ERR_ARG;
    }
    s->sm_up = s->sm_done = false;
    BWAMS_HIP(s->sm_names.ensure((size_t)name_off[nseq] + 16));
    BWAMS_HIP(s->sm_noff.ensure((size_t)n1 * 8));
    BWAMS_HIP(hipMemcpyAsync(s->sm_names.p, names, (size_t)name_off[nseq], hipMemcpyDefault, st));
    BWAMS_HIP(hipMemcpyAsync(s->sm_noff.p, name_off, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    s->sm_has_qual = quals != nullptr;
    if (quals) {
        BWAMS_HIP(s->sm_qual.ensure((size_t)b->nbases + 16));
        BWAMS_HIP(hipMemcpyAsync(s->sm_qual.p, quals, (size_t)b->nbases, hipMemcpyDefault, st));
    }
    s->sm_has_comm = comments != nullptr;
    if (comments) {
        BWAMS_HIP(s->sm_comm.ensure((size_t)comment_off[nseq] + 16));
        BWAMS_HIP(s->sm_coff.ensure((size_t)n1 * 8));
        BWAMS_HIP(hipMemcpyAsync(s->sm_comm.p, comments, (size_t)comment_off[nseq], hipMemcpyDefault, st));
        BWAMS_HIP(hipMemcpyAsync(s->sm_coff.p, comment_off, (size_t)n1 * 8, hipMemcpyHostToDevice, st));
    }
    BWAMS_HIP(hipStreamSynchronize(st));
    s->sm_nseq = nseq;
    s->sm_up = true;
    return BWAMS_OK;
}

static int sam_run_impl(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, const bwams_pestat_t *pes,
                        int64_t *sam_bytes, bwams_emf_t *emf = nullptr) {
    const bool pe = pes != nullptr;
    if (!b || !sopt || !b->chain || !b->chain->al_done || b->chain->al_source != 1 || b->chain->pr_single == pe) {
        set_last_error(pe ? "bwams_sam_run_pe: run bwams_pair_run (paired-end) and bwams_reg2aln_run(source 1) first"
                          : "bwams_sam_run: run bwams_pair_run(BWAMS_PAIR_SINGLE_END) and bwams_reg2aln_run(source 1) first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (pe && (s->nseq & 1)) return BWAMS_ERR_ARG;
    if (!s->sm_up || s->sm_nseq != s->nseq) {
        set_last_error("bwams_sam_run: run bwams_sam_upload for this chunk first");
        return BWAMS_ERR_ARG;
    }
    if (!b->idx->d_ctg_names) {
        set_last_error("bwams_sam_run: the index has no sequence names (bwams_index_set_contig_names)");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_sam_run");
    if (rc) return rc;
    // MEM_F_PRIMARY5 / MEM_F_NO_RESCUE act in bwams_pair_run_sam, before the text; MEM_F_NOPAIRING there and in the proper-pair flag
    if (sopt->flag & ~(BWAMS_MEM_F_ALL | BWAMS_MEM_F_NO_MULTI | BWAMS_MEM_F_SOFTCLIP | BWAMS_MEM_F_KEEP_SUPP_MAPQ | BWAMS_MEM_F_PRIMARY5 |
                       BWAMS_MEM_F_NOPAIRING | BWAMS_MEM_F_NO_RESCUE | BWAMS_MEM_F_REF_HDR)) {
        set_last_error("bwams_sam_run: MEM_F_PE / MEM_F_SMARTPE (the caller's business) and MEM_F_XB are not built");
        return BWAMS_ERR_UNSUPPORTED;
    }
    if ((sopt->flag & BWAMS_MEM_F_REF_HDR) && !b->idx->d_ctg_annos) {
        set_last_error("bwams_sam_run: MEM_F_REF_HDR needs the sequences' annotations (bwams_index_set_contig_annos)");
        return BWAMS_ERR_ARG;
    }
    if (!memchr(sopt->rg_id, 0, sizeof sopt->rg_id)) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    s->sm_done = false; s->sm_merged_n = -1;
    const int64_t nseq = s->nseq, n1 = nseq + 1, n = s->al_n;
    constexpr int kLogN = 1 << 16;
    if (!s->sm_log_ok) {                                   // log(i) with the C library's log, as the reference's host code computes it
        std::vector<double> lt((size_t)kLogN);
        for (int i = 0; i < kLogN; ++i) lt[(size_t)i] = log((double)i);
        BWAMS_HIP(s->sm_logtab.ensure((size_t)kLogN * 8));
        BWAMS_HIP(hipMemcpy(s->sm_logtab.p, lt.data(), (size_t)kLogN * 8, hipMemcpyHostToDevice));
        s->sm_log_ok = true;
    }
    BWAMS_HIP(s->sm_mapq.ensure((size_t)(n + 1) * 4));
    BWAMS_HIP(s->sm_len.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->sm_off.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->sm_bad.ensure(64));
    SamArgs A;
    memset(&A, 0, sizeof A);
    A.regs = s->pr_out.as<bwams_alnreg_t>(); A.reg_off = s->pr_ooff.as<int64_t>();
    A.n_regs = n; A.nseq = nseq;
    A.rec = s->al_rec.as<bwams_aln_t>(); A.cig = s->al_cig.as<uint32_t>(); A.md = s->al_md.as<char>();
    A.enc = b->d_enc; A.cum = b->d_cum;
    A.names = s->sm_names.as<char>(); A.name_off = s->sm_noff.as<int64_t>();
    A.quals = s->sm_has_qual ? s->sm_qual.as<char>() : nullptr;
    A.comments = s->sm_has_comm ? s->sm_comm.as<char>() : nullptr;
    A.comment_off = s->sm_has_comm ? s->sm_coff.as<int64_t>() : nullptr;
    A.ctg_names = reinterpret_cast<const char *>(b->idx->d_ctg_names);
    A.ctg_off = reinterpret_cast<const int32_t *>(b->idx->d_ctg_off);
    A.ctg_annos = reinterpret_cast<const char *>(b->idx->d_ctg_annos);
    A.ctg_anno_off = reinterpret_cast<const int32_t *>(b->idx->d_ctg_anno_off);
    A.opt = *opt; A.sopt = *sopt;
    A.logtab = s->sm_logtab.as<double>(); A.logtab_n = kLogN;
    A.coef_fac = opt->mapq_coef_len > 0 ? log((double)opt->mapq_coef_len) : 0.;
    if (emf) {
        if (!s->er_done || s->er_nseq != nseq) {
            set_last_error("bwams_sam_run_emf: run bwams_emf_run and bwams_emf_regs_run for this chunk first");
            return BWAMS_ERR_ARG;
        }
        A.er_regs = s->er_out.as<bwams_alnreg_t>(); A.er_off = s->er_ooff.as<int64_t>(); A.er_seed_len = emf->t.seed_len;
    }
    {
        DevBns bns_;
        if ((rc = dev_bns(b->idx, &bns_))) return rc;
        A.contigs = bns_.contigs;
    }
    A.pairs = pe ? s->pr_res.as<bwams_pair_t>() : nullptr;
    if (pe) memcpy(A.pes, pes, sizeof A.pes);
    A.bns_l_pac = (b->idx->fmi.ref_seq_len - 1) / 2;
    A.mapq = s->sm_mapq.as<int32_t>(); A.bad = s->sm_bad.as<unsigned long long>();
    A.len = s->sm_len.as<int64_t>(); A.out_off = s->sm_off.as<int64_t>(); A.out = nullptr;
    BWAMS_HIP(hipMemsetAsync(s->sm_bad.p, 0, 24, st));
    BWAMS_HIP(hipMemsetAsync(s->sm_len.as<int64_t>() + nseq, 0, 8, st));
    launch_sam_mapq(A, st);
    launch_sam_text(A, false, b->cu_count, st);
    if ((rc = scan_rows(b, A.len, s->sm_off.as<int64_t>(), 1, n1))) return rc;
    int64_t total = 0;
    unsigned long long bad = 0, bad_names = 0;
    BWAMS_HIP(hipMemcpyAsync(&total, s->sm_off.as<int64_t>() + nseq, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipMemcpyAsync(&bad, s->sm_bad.p, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipMemcpyAsync(&bad_names, s->sm_bad.as<unsigned long long>() + 2, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    if (bad_names) {
        set_last_error("bwams_sam_run_pe: paired reads have different names (" + std::to_string(bad_names) + " pair(s)); the reference stops here");
        return BWAMS_ERR_ARG;
    }
    if (bad) {
        set_last_error("bwams_sam_run: an alignment longer than 65535 bases or more than 65534 competing pairings (mapping quality table)");
        return BWAMS_ERR_UNSUPPORTED;
    }
    BWAMS_HIP(s->sm_out.ensure((size_t)total + 16));
    A.out = s->sm_out.as<char>();
    launch_sam_text(A, true, b->cu_count, st);
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(hipGetLastError());
    s->sm_bytes = total; s->sm_nregs = n; s->sm_done = true;
    if (sam_bytes) *sam_bytes = total;
    return BWAMS_OK;
}

int bwams_sam_run(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, int64_t *sam_bytes) {
    return sam_run_impl(b, opt, sopt, nullptr, sam_bytes);
}

int bwams_sam_run_emf(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, bwams_emf_t *emf, int64_t *sam_bytes) {
    if (!emf) return BWAMS_ERR_ARG;
    return sam_run_impl(b, opt, sopt, nullptr, sam_bytes, emf);
}

int bwams_sam_run_pe(bwams_batch_t *b, const bwams_mem_opt_t *opt, const bwams_sam_opt_t *sopt, const bwams_pestat_t pes[4],
                     int64_t *sam_bytes) {
    if (!pes) return BWAMS_ERR_ARG;
    return sam_run_impl(b, opt, sopt, pes, sam_bytes);
}

int bwams_sam_fetch(bwams_batch_t *b, char *sam, int64_t cap, int64_t *read_off, int32_t *mapq, int64_t mapq_cap) {
    if (!b || !b->chain || !b->chain->sm_done) {
        set_last_error("bwams_sam_fetch: run bwams_sam_run first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if ((sam && s->sm_bytes > cap) || (mapq && s->sm_nregs > mapq_cap)) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (sam && s->sm_bytes) BWAMS_HIP(hipMemcpyAsync(sam, s->sm_out.p, (size_t)s->sm_bytes, hipMemcpyDeviceToHost, st));
    const int64_t n_out = s->sm_merged_n >= 0 ? s->sm_merged_n : s->nseq;
    if (read_off) BWAMS_HIP(hipMemcpyAsync(read_off, s->sm_off.p, (size_t)(n_out + 1) * 8, hipMemcpyDeviceToHost, st));
    if (mapq && s->sm_nregs) BWAMS_HIP(hipMemcpyAsync(mapq, s->sm_mapq.p, (size_t)s->sm_nregs * 4, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

/* ------------------------------------------------------------ mem_process_seqs ---- */

// The outer boundary for one chunk, text to text: what kt_pipeline's step 0 parsing and step 1 (mem_process_seqs, src/bwamem.cpp:1850-1980)
// do between the decompressed FASTQ bytes and seqs[i].sam, as the sequence of the stage calls above.
// mem_process_seqs for a decoded chunk (fq is closed here): the stage calls in worker order
// worker_bwt + worker_aln for the chunk the batch holds (reads and names uploaded): EMF, seeding, chaining, extension, de-duplication
static int process_stage1(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo) {
    int rc;
    int64_t t0 = 0, t1 = 0;
    if (emf) {                                            // kernel 0 of mem_kernel1_core: is_pm[], then mem_perfect2reg for the resolved reads
        if ((rc = bwams_emf_run(b, emf))) return rc;
        if ((rc = bwams_emf_regs_run(b, emf, mo, &t0))) return rc;
    }
    if ((rc = ert ? bwams_seed_run_ert(b, ert, so, 1) : bwams_seed_run(b, so, 1))) return rc;
    if ((rc = bwams_chain_run(b, mo, &t0, &t1))) return rc;
    if ((rc = bwams_extend_run(b, mo, &t0))) return rc;
    return bwams_dedup_run(b, mo, &t0);
}

// worker_sam: primary marking / mate rescue + pairing, mem_reg2aln of what is printed, the SAM text.  pes: the chunk's statistics
// (paired-end; mem_pestat runs between the two stages, over the WHOLE chunk: bwamem.cpp:1881-1891).  id_base: n_processed for
// single-end, n_processed >> 1 for paired-end, plus the reads / pairs of the chunk in front of this batch when the chunk is sharded.
static int process_stage2(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                          int32_t paired, const bwams_pestat_t *pes, int64_t id_base, int32_t flags, int64_t *sam_bytes) {
    int rc;
    int64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (!paired) {
        if ((rc = bwams_pair_run_sam(b, mo, sam_opt, nullptr, id_base, BWAMS_PAIR_SINGLE_END, &t0, &t1))) return rc;
        if ((rc = bwams_reg2aln_run_sam(b, mo, sam_opt, nullptr, &t0, &t1, &t2, &t3))) return rc;
        return emf ? bwams_sam_run_emf(b, mo, sam_opt, emf, sam_bytes) : bwams_sam_run(b, mo, sam_opt, sam_bytes);
    }
    if (emf && (rc = bwams_emf_regs_merge(b, &t0))) return rc;            // worker_sam gives the resolved ends their regions (:1689-1702)
    if ((rc = bwams_pair_run_sam(b, mo, sam_opt, pes, id_base, (flags & BWAMS_PAIR_NO_RESCUE) | (ert ? BWAMS_PAIR_USE_ERT : 0), &t0, &t1))) return rc;
    if ((rc = bwams_reg2aln_run_sam(b, mo, sam_opt, pes, &t0, &t1, &t2, &t3))) return rc;
    return bwams_sam_run_pe(b, mo, sam_opt, pes, sam_bytes);
}

static void process_empty(bwams_batch_t *b, int64_t *sam_bytes) {      // an empty chunk: no reads, no text
    if (b->chain) { b->chain->sm_done = true; b->chain->sm_bytes = 0; b->chain->sm_nregs = 0; b->chain->nseq = 0; b->chain->sm_merged_n = -1; }
    b->nseq = 0;
    if (sam_bytes) *sam_bytes = 0;
}

// mem_process_seqs for the chunk the batch holds
static int process_uploaded(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                            const bwams_sam_opt_t *sam_opt, int32_t paired, const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags,
                            int64_t *sam_bytes) {
    int rc = process_stage1(b, emf, ert, so, mo);
    if (rc) return rc;
    bwams_pestat_t pes[4];
    if (paired) {
        if (pes0) memcpy(pes, pes0, sizeof pes);
        else if ((rc = bwams_pestat(b, mo, pes))) return rc;          // mem_pestat sees the regions of worker_aln only (bwamem.cpp:1881-1891)
    }
    return process_stage2(b, emf, ert, mo, sam_opt, paired, paired ? pes : nullptr, paired ? n_processed >> 1 : n_processed, flags, sam_bytes);
}

// mem_process_seqs for a decoded chunk (fq is closed here): the stage calls in worker order
static int process_decoded(bwams_batch_t *b, bwams_fastq_t *fq, int64_t n, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so,
                           const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt, int32_t paired, const bwams_pestat_t *pes0,
                           int64_t n_processed, int32_t flags, int64_t *sam_bytes) {
    if (n == 0) {
        bwams_fastq_close(fq);
        process_empty(b, sam_bytes);
        return BWAMS_OK;
    }
    const int rc = bwams_fastq_to_batch_opt(fq, b, (flags & BWAMS_CHUNK_COPY_COMMENT) ? 1 : 0);      // process(): comments only with `mem -C`
    bwams_fastq_close(fq);
    if (rc) return rc;
    return process_uploaded(b, emf, ert, so, mo, sam_opt, paired, pes0, n_processed, flags, sam_bytes);
}

// mem_process_seqs for a chunk that arrives the way the reference hands it over — parsed records (bseq1_t: name, comment, seq, qual),
// here as flat arrays: enc_qdb / cum_len as bwams_seed_upload takes them, names / quals / comments as bwams_sam_upload takes them.
int bwams_process_reads(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                        const bwams_sam_opt_t *sam_opt, const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names,
                        const int64_t *name_off, const char *quals, const char *comments, const int64_t *comment_off, int32_t paired,
                        const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags, int64_t *sam_bytes) {
    if (!b || !so || !mo || !sam_opt || n_reads < 0 || (n_reads > 0 && (!enc_qdb || !cum_len || !names || !name_off))) {
        set_last_error("bwams_process_reads: batch, options, reads and names are required");
        return BWAMS_ERR_ARG;
    }
    if (paired && (n_reads & 1)) {
        set_last_error("bwams_process_reads: a paired-end chunk holds an even number of reads (ends interleaved)");
        return BWAMS_ERR_ARG;
    }
    if (n_reads == 0) { process_empty(b, sam_bytes); return BWAMS_OK; }
    int rc = bwams_seed_upload(b, enc_qdb, cum_len, nullptr, (int32_t)n_reads);
    if (rc) return rc;
    if ((rc = bwams_sam_upload(b, names, name_off, quals, comments, comment_off))) return rc;
    return process_uploaded(b, emf, ert, so, mo, sam_opt, paired, pes0, n_processed, flags, sam_bytes);
}

// The same in two halves, for a chunk sharded over several batches (one per GPU): stage 1 up to the regions mem_pestat reads, then —
// after the caller has merged the shards' bwams_pestat_keys with bwams_pestat_from_keys — stage 2 with the chunk's statistics and this
// shard's first read / pair id.  bwams_process_reads == _stage1 + bwams_pestat + _stage2 on one batch.
int bwams_process_reads_stage1(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                               const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names, const int64_t *name_off,
                               const char *quals, const char *comments, const int64_t *comment_off) {
    if (!b || !so || !mo || n_reads < 0 || (n_reads > 0 && (!enc_qdb || !cum_len || !names || !name_off))) {
        set_last_error("bwams_process_reads_stage1: batch, options, reads and names are required");
        return BWAMS_ERR_ARG;
    }
    if (n_reads == 0) { process_empty(b, nullptr); return BWAMS_OK; }
    int rc = bwams_seed_upload(b, enc_qdb, cum_len, nullptr, (int32_t)n_reads);
    if (rc) return rc;
    if ((rc = bwams_sam_upload(b, names, name_off, quals, comments, comment_off))) return rc;
    return process_stage1(b, emf, ert, so, mo);
}

// Stage 1 itself in two halves, so that a pipeline can put the next chunk's reads and names into one batch (PCIe) while another batch
// of the same device computes: _upload = bwams_seed_upload + bwams_sam_upload, _stage1_run = worker_bwt + worker_aln on what it left.
int bwams_process_reads_upload(bwams_batch_t *b, const uint8_t *enc_qdb, const int64_t *cum_len, int64_t n_reads, const char *names,
                               const int64_t *name_off, const char *quals, const char *comments, const int64_t *comment_off) {
    if (!b || n_reads < 0 || (n_reads > 0 && (!enc_qdb || !cum_len || !names || !name_off))) {
        set_last_error("bwams_process_reads_upload: batch, reads and names are required");
        return BWAMS_ERR_ARG;
    }
    if (n_reads == 0) { process_empty(b, nullptr); return BWAMS_OK; }
    int rc = bwams_seed_upload(b, enc_qdb, cum_len, nullptr, (int32_t)n_reads);
    if (rc) return rc;
    return bwams_sam_upload(b, names, name_off, quals, comments, comment_off);
}

int bwams_process_reads_stage1_run(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo) {
    if (!b || !so || !mo) {
        set_last_error("bwams_process_reads_stage1_run: batch and options are required");
        return BWAMS_ERR_ARG;
    }
    if (b->nseq == 0) return BWAMS_OK;
    return process_stage1(b, emf, ert, so, mo);
}

int bwams_batch_device(const bwams_batch_t *b, int32_t *device) {
    if (!b || !b->idx || !device) return BWAMS_ERR_ARG;
    *device = b->idx->device;
    return BWAMS_OK;
}

int bwams_process_reads_stage2(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_mem_opt_t *mo, const bwams_sam_opt_t *sam_opt,
                               int32_t paired, const bwams_pestat_t *pes, int64_t id_base, int32_t flags, int64_t *sam_bytes) {
    if (!b || !mo || !sam_opt || (paired && !pes)) {
        set_last_error("bwams_process_reads_stage2: batch, options and (paired-end) the chunk's statistics are required");
        return BWAMS_ERR_ARG;
    }
    if (b->nseq == 0) { process_empty(b, sam_bytes); return BWAMS_OK; }
    return process_stage2(b, emf, ert, mo, sam_opt, paired, pes, id_base, flags, sam_bytes);
}

// Page-locked host memory for the buffers that cross PCIe every chunk (reads and names up, SAM text down): the copies of
// bwams_seed_upload / bwams_sam_upload / bwams_sam_fetch then run at the link's rate instead of through a pageable bounce buffer.
int bwams_host_alloc(size_t bytes, void **out) {
    if (!out) return BWAMS_ERR_ARG;
    *out = nullptr;
    BWAMS_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return BWAMS_OK;
}
int bwams_host_free(void *p) {
    if (p) BWAMS_HIP(hipHostFree(p));
    return BWAMS_OK;
}

int bwams_process_chunk(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                        const bwams_sam_opt_t *sam_opt, const char *fastq, int64_t n_bytes, int32_t paired, const bwams_pestat_t *pes0,
                        int64_t n_processed, int32_t flags, int64_t *n_reads, int64_t *sam_bytes) {
    if (!b || !so || !mo || !sam_opt || !fastq || n_bytes < 0) {
        set_last_error("bwams_process_chunk: batch, options and text are required");
        return BWAMS_ERR_ARG;
    }
    bwams_fastq_t *fq = nullptr;
    int64_t n = 0, nb = 0;
    int rc = bwams_fastq_decode(b->idx->device, fastq, n_bytes, &fq, &n, &nb);
    if (rc) return rc;
    if (paired && (n & 1)) {
        bwams_fastq_close(fq);
        set_last_error("bwams_process_chunk: a paired-end chunk holds an even number of reads (ends interleaved)");
        return BWAMS_ERR_ARG;
    }
    rc = process_decoded(b, fq, n, emf, ert, so, mo, sam_opt, paired, pes0, n_processed, flags, sam_bytes);
    if (!rc && n_reads) *n_reads = n;
    return rc;
}

// A paired-end chunk read from two files (bseq_read_orig with ks2, src/bwa.cpp:275-318): record k of the first text and record k of the
// second are the ends of pair k.
int bwams_process_chunk2(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                         const bwams_sam_opt_t *sam_opt, const char *fastq1, int64_t n_bytes1, const char *fastq2, int64_t n_bytes2,
                         const bwams_pestat_t *pes0, int64_t n_processed, int32_t flags, int64_t *n_reads, int64_t *sam_bytes) {
    if (!b || !so || !mo || !sam_opt || !fastq1 || !fastq2 || n_bytes1 < 0 || n_bytes2 < 0) {
        set_last_error("bwams_process_chunk2: batch, options and the two texts are required");
        return BWAMS_ERR_ARG;
    }
    bwams_fastq_t *f1 = nullptr, *f2 = nullptr, *fq = nullptr;
    int64_t n1 = 0, n2 = 0, nb = 0;
    int rc = bwams_fastq_decode(b->idx->device, fastq1, n_bytes1, &f1, &n1, &nb);
    if (rc) return rc;
    if ((rc = bwams_fastq_decode(b->idx->device, fastq2, n_bytes2, &f2, &n2, &nb))) { bwams_fastq_close(f1); return rc; }
    if (n1 != n2 || bwams_fastq_has_qual(f1) != bwams_fastq_has_qual(f2)) {
        bwams_fastq_close(f1); bwams_fastq_close(f2);
        set_last_error("bwams_process_chunk2: the two texts must hold the same number of records of one kind (" + std::to_string(n1) + " and " +
                       std::to_string(n2) + "): cut both files at the same record");
        return BWAMS_ERR_ARG;
    }
    rc = fastq_interleave(f1, f2, &fq);
    bwams_fastq_close(f1); bwams_fastq_close(f2);
    if (rc) return rc;
    rc = process_decoded(b, fq, 2 * n1, emf, ert, so, mo, sam_opt, 1, pes0, n_processed, flags, sam_bytes);
    if (!rc && n_reads) *n_reads = 2 * n1;
    return rc;
}

// process()'s MEM_F_SMARTPE branch (src/fastmap.cpp:378-414): bseq_classify splits the chunk into the reads that stand alone and the
// interleaved pairs; mem_process_seqs runs on the first set as single-end (ids from n_processed) and on the second as paired-end (ids
// from n_processed + the number of single reads, pes0); every read's text returns to its place in the chunk.
int bwams_process_chunk_smart(bwams_batch_t *b, bwams_emf_t *emf, bwams_ert_t *ert, const bwams_seed_opt_t *so, const bwams_mem_opt_t *mo,
                              const bwams_sam_opt_t *sam_opt, const char *fastq, int64_t n_bytes, const bwams_pestat_t *pes0,
                              int64_t n_processed, int32_t flags, int64_t *n_reads, int64_t *n_single, int64_t *sam_bytes) {
    if (!b || !so || !mo || !sam_opt || !fastq || n_bytes < 0) {
        set_last_error("bwams_process_chunk_smart: batch, options and text are required");
        return BWAMS_ERR_ARG;
    }
    bwams_fastq_t *fq = nullptr;
    int64_t n = 0, nb = 0;
    int rc = bwams_fastq_decode(b->idx->device, fastq, n_bytes, &fq, &n, &nb);
    if (rc) return rc;
    std::vector<uint8_t> which;
    if ((rc = fastq_classify(fq, &which))) { bwams_fastq_close(fq); return rc; }
    std::vector<int64_t> ids[2];
    for (int64_t i = 0; i < n; ++i) ids[which[(size_t)i]].push_back(i);
    struct Held {                                        // the text of one run, kept while the batch does the other
        char *text = nullptr;
        std::vector<int64_t> off;
        ~Held() { if (text) (void)hipFree(text); }
    } held[2];
    hipStream_t st = b->stream;
    for (int k = 0; k < 2; ++k) {
        held[k].off.assign(ids[k].size() + 1, 0);
        if (ids[k].empty()) continue;
        bwams_fastq_t *sub = nullptr;
        if ((rc = fastq_subset(fq, ids[k], &sub))) { bwams_fastq_close(fq); return rc; }
        int64_t bytes = 0;
        rc = process_decoded(b, sub, (int64_t)ids[k].size(), emf, ert, so, mo, sam_opt, k, k ? pes0 : nullptr,
                             n_processed + (k ? (int64_t)ids[0].size() : 0), flags, &bytes);
        if (rc) { bwams_fastq_close(fq); return rc; }
        ChainState *s = b->chain;
        hipError_t e = dev_malloc(reinterpret_cast<void **>(&held[k].text), (size_t)bytes + 16);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(held[k].text, s->sm_out.p, (size_t)bytes, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(held[k].off.data(), s->sm_off.p, held[k].off.size() * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { bwams_fastq_close(fq); BWAMS_HIP(e); }
    }
    bwams_fastq_close(fq);
    if (n_reads) *n_reads = n;
    if (n_single) *n_single = (int64_t)ids[0].size();
    ChainState *s;
    if ((rc = get_state(b, &s))) return rc;
    // every read's block back to its place in the chunk (ret->seqs[sep[k][i].id].sam = sep[k][i].sam)
    std::vector<int64_t> off((size_t)n + 1, 0);
    std::vector<int64_t> rank((size_t)n, 0);
    for (int k = 0; k < 2; ++k)
        for (size_t j = 0; j < ids[k].size(); ++j) rank[(size_t)ids[k][j]] = (int64_t)j;
    for (int64_t i = 0; i < n; ++i) {
        const Held &h = held[which[(size_t)i]];
        const size_t j = (size_t)rank[(size_t)i];
        off[(size_t)i + 1] = off[(size_t)i] + (h.off[j + 1] - h.off[j]);
    }
    const int64_t total = off[(size_t)n];
    BWAMS_HIP(s->sm_out.ensure((size_t)total + 16));
    BWAMS_HIP(s->sm_off.ensure((size_t)(n + 1) * 8));
    std::vector<SegMove> mv;
    mv.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const Held &h = held[which[(size_t)i]];
        const size_t j = (size_t)rank[(size_t)i];
        mv.push_back({h.text + h.off[j], s->sm_out.as<char>() + off[(size_t)i], h.off[j + 1] - h.off[j]});
    }
    if ((rc = segment_copy(mv, st))) return rc;
    BWAMS_HIP(hipMemcpyAsync(s->sm_off.p, off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    s->sm_bytes = total; s->sm_nregs = 0; s->sm_merged_n = n; s->sm_done = true;
    if (sam_bytes) *sam_bytes = total;
    return BWAMS_OK;
}

/* ------------------------------------------------------------ mem_perfect2reg ---- */

int bwams_emf_regs_run(bwams_batch_t *b, bwams_emf_t *e, const bwams_mem_opt_t *opt, int64_t *n_regs) {
    if (!b || !e || !b->d_emf_out || !b->d_emf_code) {
        set_last_error("bwams_emf_regs_run: run bwams_emf_run first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_emf_regs_run");
    if (rc) return rc;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    ChainState *s;
    if ((rc = get_state(b, &s))) return rc;
    s->er_done = false;
    hipStream_t st = b->stream;
    const int64_t nseq = b->nseq, n1 = nseq + 1;
    BWAMS_HIP(s->er_wide.ensure((size_t)n1 * 8)); BWAMS_HIP(s->er_off.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->er_ooff.ensure((size_t)n1 * 8));
    BWAMS_HIP(s->er_n.ensure((size_t)n1 * 4)); BWAMS_HIP(s->er_rev.ensure((size_t)n1));
    EmfRegArgs A;
    A.t = e->t; A.perfect = b->d_emf_out; A.code = b->d_emf_code; A.enc = b->d_enc; A.cum = b->d_cum; A.nseq = nseq;
    if ((rc = dev_bns(b->idx, &A.bns))) return rc;
    A.opt = *opt;
    A.scratch = nullptr;
    launch_emfregs_count(A, s->er_wide.as<int64_t>(), st);
    if ((rc = scan_rows(b, s->er_wide.as<int64_t>(), s->er_off.as<int64_t>(), 1, n1))) return rc;
    int64_t n_scr = 0, total = 0;
    BWAMS_HIP(hipMemcpyAsync(&n_scr, s->er_off.as<int64_t>() + nseq, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(s->er_scr.ensure(emfregs_scratch_bytes(n_scr)));
    A.scratch = s->er_scr.p;
    launch_emfregs_fill(A, s->er_off.as<int64_t>(), s->er_n.as<int32_t>(), s->er_rev.as<uint8_t>(), s->er_wide.as<int64_t>(), st);
    if ((rc = scan_rows(b, s->er_wide.as<int64_t>(), s->er_ooff.as<int64_t>(), 1, n1))) return rc;
    BWAMS_HIP(hipMemcpyAsync(&total, s->er_ooff.as<int64_t>() + nseq, 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(s->er_out.ensure((size_t)(total + 1) * sizeof(bwams_alnreg_t)));
    launch_emfregs_emit(A, s->er_off.as<int64_t>(), s->er_n.as<int32_t>(), s->er_ooff.as<int64_t>(), s->er_out.as<bwams_alnreg_t>(), st);
    BWAMS_HIP(hipGetLastError());
    s->er_total = total; s->er_nseq = nseq;
    s->er_done = true;
    if (n_regs) *n_regs = total;
    return BWAMS_OK;
}

int bwams_emf_regs_merge(bwams_batch_t *b, int64_t *n_regs) {
    if (!b || !b->chain || !b->chain->er_done || !b->chain->dedup_done || b->chain->er_nseq != b->chain->nseq) {
        set_last_error("bwams_emf_regs_merge: run bwams_emf_regs_run and bwams_dedup_run of this chunk first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    const int64_t nseq = s->nseq, n1 = nseq + 1;
    int rc;
    BWAMS_HIP(s->mg_wide.ensure((size_t)n1 * 8)); BWAMS_HIP(s->mg_off.ensure((size_t)n1 * 8));
    launch_emfregs_merge_count(s->dd_off.as<int64_t>(), s->er_ooff.as<int64_t>(), nseq, s->mg_wide.as<int64_t>(), st);
    if ((rc = scan_rows(b, s->mg_wide.as<int64_t>(), s->mg_off.as<int64_t>(), 1, n1))) return rc;
    const int64_t total = s->n_final + s->er_total;
    BWAMS_HIP(s->mg_out.ensure((size_t)(total + 1) * sizeof(bwams_alnreg_t)));
    launch_emfregs_merge(s->dd_out.as<bwams_alnreg_t>(), s->dd_off.as<int64_t>(), s->er_out.as<bwams_alnreg_t>(), s->er_ooff.as<int64_t>(), nseq,
                         s->mg_off.as<int64_t>(), s->mg_out.as<bwams_alnreg_t>(), st);
    BWAMS_HIP(hipStreamSynchronize(st));
    BWAMS_HIP(hipGetLastError());
    std::swap(s->dd_out, s->mg_out);
    std::swap(s->dd_off, s->mg_off);
    s->n_final = total;
    s->er_done = false;                                   // merged: a second call would add them again
    s->pair_done = s->al_done = s->sm_done = false;
    if (n_regs) *n_regs = total;
    return BWAMS_OK;
}

int bwams_emf_regs_fetch(bwams_batch_t *b, bwams_alnreg_t *regs, int64_t reg_cap, int64_t *reg_off, uint8_t *first_is_rev) {
    if (!b || !b->chain || !b->chain->er_done) {
        set_last_error("bwams_emf_regs_fetch: run bwams_emf_regs_run first");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    if (s->er_total > reg_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (s->er_total) BWAMS_HIP(hipMemcpyAsync(regs, s->er_out.p, (size_t)s->er_total * sizeof(bwams_alnreg_t), hipMemcpyDeviceToHost, st));
    if (reg_off) BWAMS_HIP(hipMemcpyAsync(reg_off, s->er_ooff.p, (size_t)(s->er_nseq + 1) * 8, hipMemcpyDeviceToHost, st));
    if (first_is_rev && s->er_nseq) BWAMS_HIP(hipMemcpyAsync(first_is_rev, s->er_rev.p, (size_t)s->er_nseq, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

/* ------------------------------------------------------------------ mem_pestat ---- */

// the insert-size keys of the qualifying pairs, sorted: orientation << 60 | insert size
static int pestat_keys(bwams_batch *b, ChainState *s, const bwams_mem_opt_t *opt, std::vector<unsigned long long> *keys) {
    hipStream_t st = b->stream;
    const int64_t n_pairs = s->nseq >> 1;
    const int64_t l_pac = (b->idx->fmi.ref_seq_len - 1) / 2;
    keys->assign((size_t)(n_pairs > 0 ? n_pairs : 0), ~0ull);
    if (n_pairs <= 0) return BWAMS_OK;
    BWAMS_HIP(s->pe_keys.ensure((size_t)n_pairs * 8));
    BWAMS_HIP(s->pe_keys2.ensure((size_t)n_pairs * 8));
    launch_pestat(s->dd_out.as<bwams_alnreg_t>(), s->dd_off.as<int64_t>(), n_pairs, l_pac, *opt,
                  s->pe_keys.as<unsigned long long>(), st);
    size_t tb = 0;
    BWAMS_HIP(rocprim::radix_sort_keys(nullptr, tb, s->pe_keys.as<unsigned long long>(), s->pe_keys2.as<unsigned long long>(),
                                       (size_t)n_pairs, 0, 64, st));
    if (tb > b->tmp_bytes) {
        BWAMS_HIP(hipStreamSynchronize(st));
        if (b->d_tmp) (void)hipFree(b->d_tmp);
        b->d_tmp = nullptr;
        BWAMS_HIP(dev_malloc(&b->d_tmp, tb));
        b->tmp_bytes = tb;
    }
    tb = b->tmp_bytes;
    BWAMS_HIP(rocprim::radix_sort_keys(b->d_tmp, tb, s->pe_keys.as<unsigned long long>(), s->pe_keys2.as<unsigned long long>(),
                                       (size_t)n_pairs, 0, 64, st));
    BWAMS_HIP(hipMemcpyAsync(keys->data(), s->pe_keys2.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    size_t k = keys->size();
    while (k > 0 && (*keys)[k - 1] == ~0ull) --k;           // pairs that do not qualify sort last
    keys->resize(k);
    return BWAMS_OK;
}

// the reference's arithmetic over each orientation's sorted insert sizes (bwamem_pair.cpp:111-155), as written
static void pestat_from_sorted(const unsigned long long *keys, size_t n, bwams_pestat_t pes[4]) {
    memset(pes, 0, 4 * sizeof(bwams_pestat_t));
    size_t beg[5] = {0, 0, 0, 0, 0};
    {
        size_t k = 0;
        for (int d = 0; d < 4; ++d) {
            beg[d] = k;
            while (k < n && (int)(keys[k] >> 60) == d) ++k;
        }
        beg[4] = k;
    }
    const unsigned long long mask = (1ull << 60) - 1ull;
    int max = 0;
    for (int d = 0; d < 4; ++d) {
        bwams_pestat_t *r = &pes[d];
        const unsigned long long *q = keys + beg[d];
        const size_t qn = beg[d + 1] - beg[d];
        max = max > (int)qn ? max : (int)qn;
        if (qn < 10) { r->failed = 1; continue; }
        const int p25 = (int)(q[(int)(.25 * qn + .499)] & mask);
        const int p75 = (int)(q[(int)(.75 * qn + .499)] & mask);
        r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
        if (r->low < 1) r->low = 1;
        r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
        int x = 0;
        size_t k;
        for (k = 0, r->avg = 0; k < qn; ++k) {
            const uint64_t v = q[k] & mask;
            if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->avg += v, ++x;
        }
        r->avg /= x;
        for (k = 0, r->std = 0; k < qn; ++k) {
            const uint64_t v = q[k] & mask;
            if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->std += (v - r->avg) * (v - r->avg);
        }
        r->std = sqrt(r->std / x);
        r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);
        r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
        if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);
        if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
        if (r->low < 1) r->low = 1;
    }
    for (int d = 0; d < 4; ++d)
        if (pes[d].failed == 0 && (double)(beg[d + 1] - beg[d]) < max * 0.05) pes[d].failed = 1;
}

int bwams_pestat(bwams_batch_t *b, const bwams_mem_opt_t *opt, bwams_pestat_t pes[4]) {
    if (!b || !b->chain || !b->chain->dedup_done || !pes) {
        set_last_error("bwams_pestat: run bwams_dedup_run first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_pestat");
    if (rc) return rc;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    std::vector<unsigned long long> keys;
    if ((rc = pestat_keys(b, b->chain, opt, &keys))) return rc;
    pestat_from_sorted(keys.data(), keys.size(), pes);
    return BWAMS_OK;
}

int bwams_pestat_keys(bwams_batch_t *b, const bwams_mem_opt_t *opt, uint64_t *keys_out, int64_t cap, int64_t *n_keys) {
    if (!b || !b->chain || !b->chain->dedup_done || !n_keys) {
        set_last_error("bwams_pestat_keys: run bwams_dedup_run first");
        return BWAMS_ERR_ARG;
    }
    int rc = check_opt(opt, "bwams_pestat_keys");
    if (rc) return rc;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    std::vector<unsigned long long> keys;
    if ((rc = pestat_keys(b, b->chain, opt, &keys))) return rc;
    *n_keys = (int64_t)keys.size();
    if ((int64_t)keys.size() > cap) return BWAMS_ERR_CAPACITY;
    if (!keys.empty()) memcpy(keys_out, keys.data(), keys.size() * 8);
    return BWAMS_OK;
}

int bwams_pestat_from_keys(const uint64_t *keys_in, int64_t n, bwams_pestat_t pes[4]) {
    if (n < 0 || (n && !keys_in) || !pes) return BWAMS_ERR_ARG;
    std::vector<unsigned long long> keys(keys_in, keys_in + n);
    std::sort(keys.begin(), keys.end());
    pestat_from_sorted(keys.data(), keys.size(), pes);
    return BWAMS_OK;
}

int bwams_extend_tasks_fetch(bwams_batch_t *b, int32_t side, bwams_seqpair_t *pairs, int64_t pair_cap, uint8_t *ref,
                             int64_t ref_cap, uint8_t *qer, int64_t qer_cap, int64_t *n_pairs, int64_t *ref_bytes,
                             int64_t *qer_bytes) {
    if (!b || !b->chain || !(b->chain->built || b->chain->ext_done) || (side != 0 && side != 1)) {
        set_last_error("bwams_extend_tasks_fetch: no task lists on the device");
        return BWAMS_ERR_ARG;
    }
    if (b->chain->tasks_inplace) {
        set_last_error("bwams_extend_tasks_fetch: bwams_extend_run extends in place and builds no flat task buffers; bwams_extend_build does");
        return BWAMS_ERR_ARG;
    }
    ChainState *s = b->chain;
    const int64_t n = side ? s->n_right : s->n_left, rb = side ? s->rref_b : s->lref_b, qb = side ? s->rqer_b : s->lqer_b;
    if (n_pairs) *n_pairs = n;
    if (ref_bytes) *ref_bytes = rb;
    if (qer_bytes) *qer_bytes = qb;
    if (n > pair_cap || rb > ref_cap || qb > qer_cap) return BWAMS_ERR_CAPACITY;
    BWAMS_HIP(hipSetDevice(b->idx->device));
    hipStream_t st = b->stream;
    if (n) BWAMS_HIP(hipMemcpyAsync(pairs, side ? s->rpairs.p : s->lpairs.p, (size_t)n * sizeof(bwams_seqpair_t), hipMemcpyDeviceToHost, st));
    if (rb) BWAMS_HIP(hipMemcpyAsync(ref, side ? s->rref.p : s->lref.p, (size_t)rb, hipMemcpyDeviceToHost, st));
    if (qb) BWAMS_HIP(hipMemcpyAsync(qer, side ? s->rqer.p : s->lqer.p, (size_t)qb, hipMemcpyDeviceToHost, st));
    BWAMS_HIP(hipStreamSynchronize(st));
    return BWAMS_OK;
}

/* Test hook: the region sorts of the de-duplication's wave tier on caller-given keys (order_out[i] = index of the i-th
 * record after the sort).  which: 0 = mem_ars2 (key k), 1 = mem_ars (s descending, k, q).  mode: 0 = as the kernels run
 * it, 1 = the operation-exact wave-parallel introsort even without ties, 2 = the sequential introsort on lane 0. */
int bwams_debug_sort(bwams_index_t *ix, const int64_t *k, const int32_t *s, const int32_t *q, int32_t n, int32_t which,
                     int32_t mode, int32_t *order_out) {
    if (!ix || n < 0 || (n && (!k || !s || !q || !order_out))) return BWAMS_ERR_ARG;
    BWAMS_HIP(hipSetDevice(ix->device));
    if (launch_sort_test(k, s, q, n, which, mode, order_out)) {
        set_last_error("bwams_debug_sort: n must be at most 1024");
        return BWAMS_ERR_ARG;
    }
    return BWAMS_OK;
}

}  // extern "C"

namespace bwams {
// timing and counts of the chain / extension stages for bwams_batch_stats (api.hip)
void chain_state_stats(const ChainState *s, bwams_stats_t *out) {
    if (!s) return;
    out->n_chains = s->n_chains; out->n_chain_seeds = s->n_seeds; out->n_chain_redo = s->n_chain_redo;
    out->n_left = s->n_left; out->n_right = s->n_right;
    out->n_retry_left = s->n_retry_left; out->n_retry_right = s->n_retry_right;
    auto el = [&](int a, int b, float *dst) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s->ev[a], s->ev[b]) == hipSuccess) *dst = ms;
    };
    el(0, 1, &out->ms_chain);
    el(2, 3, &out->ms_ext_plan);
    if (s->ext_done) {
        el(4, 5, &out->ms_ext_left);
        el(6, 7, &out->ms_ext_right);
        el(8, 9, &out->ms_ext_purge);
        el(10, 11, &out->ms_ext_total);
        out->n_ext_rounds = s->n_rounds;
    }
    if (s->dedup_done) { el(12, 13, &out->ms_dedup); out->n_final_regs = s->n_final; }
    if (s->pair_done) {
        el(14, 15, &out->ms_pair);
        out->n_pair_tasks = s->pr_tasks; out->n_pair_redone = s->pr_redone; out->n_pair_regs = s->pr_total;
    }
    (void)hipGetLastError();
}
}  // namespace bwams
