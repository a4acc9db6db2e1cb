// mem_process_seqs_hip.cpp — mem_process_seqs() (/root/reference/src/bwamem.cpp:1850-1903) with its body on the GPU(s): the
// reference's records in (bseq1_t), the reference's per-work-item SAM strings out, everything in between calls into the C-ABI of
// libbwams.so.  Host C++ only: no HIP header is needed here.
//
// The worker stands where worker_t stands.  It owns `depth` SLOTS; a slot is one chunk in flight: a bwams_multi over the worker's
// devices (one batch per device, the chunk cut on read / pair boundaries: host/chunk_multi.cpp) and the page-locked staging of what
// crosses PCIe.  The reference's kt_pipeline has three steps per chunk — read, mem_process_seqs, write — and overlaps them over
// consecutive chunks with its `-i` threads (src/fastmap.cpp:307-468; a step holds one chunk at a time: the ordering lock :475-491).
// The same three steps here:
//     mem_process_seqs_stage()    (optional; end of step 0, the reader's thread)  records -> page-locked arrays -> the slot's batches
//     mem_process_seqs()          (step 1)  the kernels; stages first if the chunk was not staged; collects at once unless deferred
//     mem_process_seqs_collect()  (optional; start of step 2, the writer's thread)  SAM text down, one string per 512-read work item
// so that chunk i + 1 goes up and chunk i - 1 comes down while chunk i computes.  A caller that only knows mem_process_seqs() gets the
// three in one synchronous call (depth 1).
#include "bwamem_hip.h"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

enum SlotState { SLOT_FREE = 0, SLOT_BUSY, SLOT_STAGED, SLOT_COMPUTED };

struct Slot {
    std::vector<bwams_batch_t *> batch;             // one per device
    bwams_multi_t *multi = nullptr;
    // page-locked staging (bwams_host_alloc): what goes up and what comes down every chunk
    uint8_t *enc = nullptr;
    char *qual = nullptr, *names = nullptr, *comments = nullptr, *sam = nullptr;
    int64_t *cum = nullptr, *name_off = nullptr, *comment_off = nullptr, *sam_off = nullptr;
    int64_t names_cap = 0, comments_cap = 0, sam_cap = 0;
    bwams_perfect_t *perfect = nullptr;
    uint8_t *code = nullptr;
    // the chunk it holds
    SlotState state = SLOT_FREE;
    const bseq1_t *owner = nullptr;
    int n = 0;
    bool refuse = false;
    int64_t bytes = 0;
};

}  // namespace

struct bwams_worker {
    int n_dev = 0, depth = 0;
    std::vector<bwams_index_t *> idx;
    std::vector<bwams_emf_t *> emf;
    std::vector<bwams_ert_t *> ert;
    int64_t max_reads = 0, max_bases = 0;
    char rg_id[256] = {0};
    bwams_host_path_t host_path = nullptr;
    void *host_user = nullptr;
    bool deferred_collect = false;
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    std::string err;
    // a chunk's text on its way into the work-item strings (the deferred collect): spare page-locked buffers that change places with the
    // slot's, so that the slot is free for the next chunk while the strings are cut (60 ms per 400 MB: first-touch faults of fresh blocks)
    std::mutex spare_mu;
    char *spare_sam = nullptr;
    int64_t spare_cap = 0;
    int64_t *spare_off = nullptr;
};

static void die(const char *what, int rc, const char *msg = nullptr) {
    // the reference's convention: a line on stderr, exit(EXIT_FAILURE) (e.g. src/bwamem.cpp:3638-3645)
    fprintf(stderr, "[bwams] %s: %s: %s\n", what, bwams_strerror(rc), msg && *msg ? msg : bwams_last_error());
    exit(EXIT_FAILURE);
}

template <class T>
static int pinned(T *&p, size_t n) {
    void *q = nullptr;
    const int rc = bwams_host_alloc(n * sizeof(T), &q);
    p = rc ? nullptr : static_cast<T *>(q);
    return rc;
}

template <class T>
static int grow(T *&p, int64_t &cap, int64_t need) {
    if (need <= cap) return BWAMS_OK;
    if (p) bwams_host_free(p);
    p = nullptr;
    cap = need + need / 4 + 4096;
    const int rc = pinned(p, (size_t)cap);
    if (rc) cap = 0;
    return rc;
}

void bwams_worker_destroy(bwams_worker *w) {
    if (!w) return;
    for (Slot &s : w->slots) {
        if (s.multi) bwams_multi_destroy(s.multi);
        for (bwams_batch_t *b : s.batch) if (b) bwams_batch_destroy(b);
        void *ps[] = {s.enc, s.qual, s.names, s.comments, s.sam, s.cum, s.name_off, s.comment_off, s.sam_off, s.perfect, s.code};
        for (void *p : ps) if (p) bwams_host_free(p);
    }
    if (w->spare_sam) bwams_host_free(w->spare_sam);
    if (w->spare_off) bwams_host_free(w->spare_off);
    delete w;
}

int bwams_worker_create_multi(bwams_index_t *const *idx, bwams_emf_t *const *emf, bwams_ert_t *const *ert, int n_dev, int depth,
                              int64_t max_reads, int64_t max_bases, const char *rg_id, bwams_worker **out) {
    if (!idx || !out || n_dev < 1 || depth < 1 || depth > 8 || max_reads <= 0 || max_bases <= 0) return BWAMS_ERR_ARG;
    for (int d = 0; d < n_dev; ++d) if (!idx[d]) return BWAMS_ERR_ARG;
    bwams_worker *w = new bwams_worker();
    w->n_dev = n_dev; w->depth = depth;
    w->max_reads = max_reads; w->max_bases = max_bases;
    if (rg_id) strncpy(w->rg_id, rg_id, sizeof w->rg_id - 1);
    for (int d = 0; d < n_dev; ++d) {
        w->idx.push_back(idx[d]);
        w->emf.push_back(emf ? emf[d] : nullptr);
        w->ert.push_back(ert ? ert[d] : nullptr);
    }
    w->slots.resize((size_t)depth);
    // a device's share of a chunk (shards differ by at most one pair) and of its bases (a share of the reads can hold more than its share
    // of the bases when read lengths differ: twice the even share, never more than the whole)
    const int64_t sh_reads = (max_reads + n_dev - 1) / n_dev + 2;
    const int64_t sh_bases = n_dev == 1 ? max_bases : std::min<int64_t>(max_bases, 2 * ((max_bases + n_dev - 1) / n_dev) + 1024);
    int rc = BWAMS_OK;
    for (Slot &s : w->slots) {
        s.batch.assign((size_t)n_dev, nullptr);
        for (int d = 0; d < n_dev && !rc; ++d) rc = bwams_batch_create(idx[d], sh_reads, sh_bases, 0, 0, &s.batch[(size_t)d]);
        if (!rc) rc = bwams_multi_create(s.batch.data(), emf ? w->emf.data() : nullptr, ert ? w->ert.data() : nullptr, n_dev, &s.multi);
        if (!rc) rc = pinned(s.enc, (size_t)max_bases);
        if (!rc) rc = pinned(s.qual, (size_t)max_bases);
        if (!rc) rc = pinned(s.cum, (size_t)max_reads + 1);
        if (!rc) rc = pinned(s.name_off, (size_t)max_reads + 1);
        if (!rc) rc = pinned(s.comment_off, (size_t)max_reads + 1);
        if (!rc) rc = pinned(s.sam_off, (size_t)max_reads + 1);
        if (!rc) rc = pinned(s.perfect, (size_t)max_reads);
        if (!rc) rc = pinned(s.code, (size_t)max_reads);
        if (rc) break;
    }
    if (!rc && depth > 1) rc = pinned(w->spare_off, (size_t)max_reads + 1);
    if (rc) { bwams_worker_destroy(w); return rc; }          // (the message of the failing call stays in bwams_last_error)
    *out = w;
    return BWAMS_OK;
}

int bwams_worker_create(bwams_index_t *idx, bwams_emf_t *emf, bwams_ert_t *ert, int64_t max_reads, int64_t max_bases, const char *rg_id,
                        bwams_worker **out) {
    if (!idx) return BWAMS_ERR_ARG;
    return bwams_worker_create_multi(&idx, emf ? &emf : nullptr, ert ? &ert : nullptr, 1, 1, max_reads, max_bases, rg_id, out);
}

void bwams_worker_set_host_path(bwams_worker *w, bwams_host_path_t f, void *user) { w->host_path = f; w->host_user = user; }
void bwams_worker_set_deferred_collect(bwams_worker *w, int on) { w->deferred_collect = on != 0; }
const char *bwams_worker_error(const bwams_worker *w) { return w ? w->err.c_str() : ""; }

// mem_opt_t -> the three option records of the C-ABI.  Every field the device path reads is a field of mem_opt_t
// (src/bwamem.h:89-124); n_threads and chunk_size are the driver's.
void bwams_map_options(const mem_opt_t *o, const char *rg_id, bwams_seed_opt_t *so, bwams_mem_opt_t *mo, bwams_sam_opt_t *sa) {
    memset(so, 0, sizeof *so); memset(mo, 0, sizeof *mo); memset(sa, 0, sizeof *sa);
    so->min_seed_len = o->min_seed_len; so->split_factor = o->split_factor; so->split_width = o->split_width;
    so->max_mem_intv = (int32_t)o->max_mem_intv; so->max_occ = o->max_occ;
    mo->a = o->a; mo->b = o->b; mo->o_del = o->o_del; mo->e_del = o->e_del; mo->o_ins = o->o_ins; mo->e_ins = o->e_ins;
    mo->pen_clip5 = o->pen_clip5; mo->pen_clip3 = o->pen_clip3; mo->w = o->w; mo->zdrop = o->zdrop;
    mo->min_seed_len = o->min_seed_len; mo->min_chain_weight = o->min_chain_weight; mo->max_chain_extend = o->max_chain_extend;
    mo->max_occ = o->max_occ; mo->max_chain_gap = o->max_chain_gap; mo->mask_level = o->mask_level; mo->drop_ratio = o->drop_ratio;
    memcpy(mo->mat, o->mat, 25);
    mo->extend_all = 0;
    mo->mask_level_redun = o->mask_level_redun; mo->max_ins = o->max_ins; mo->pen_unpaired = o->pen_unpaired;
    mo->max_matesw = o->max_matesw; mo->mapq_coef_len = (int32_t)o->mapQ_coef_len;
    sa->T = o->T;
    sa->flag = o->flag & (MEM_F_NOPAIRING | MEM_F_ALL | MEM_F_NO_MULTI | MEM_F_NO_RESCUE | MEM_F_REF_HDR | MEM_F_SOFTCLIP | MEM_F_PRIMARY5 |
                          MEM_F_KEEP_SUPP_MAPQ);
    sa->XA_drop_ratio = o->XA_drop_ratio; sa->max_XA_hits = o->max_XA_hits; sa->max_XA_hits_alt = o->max_XA_hits_alt;
    if (rg_id) strncpy(sa->rg_id, rg_id, sizeof sa->rg_id - 1);
}

// seq[i] < 4 ? seq[i] : nst_nt4_table[seq[i]] (src/bwamem.cpp:1232; the table: src/bntseq.cpp:64-81 — A C G T in either case -> 0..3,
// '-' -> 5, everything else -> 4) as one 256-entry table
struct Nt4Table {
    uint8_t t[256];
    Nt4Table() {
        for (int c = 0; c < 256; ++c) t[c] = 4;
        t[0] = 0; t[1] = 1; t[2] = 2; t[3] = 3;
        t[(unsigned char)'A'] = t[(unsigned char)'a'] = 0; t[(unsigned char)'C'] = t[(unsigned char)'c'] = 1;
        t[(unsigned char)'G'] = t[(unsigned char)'g'] = 2; t[(unsigned char)'T'] = t[(unsigned char)'t'] = 3;
        t[(unsigned char)'-'] = 5;
    }
};
static const Nt4Table kNt4;

// The host side of a chunk is byte shuffling over ~0.5 GB (records -> flat arrays, text -> work-item strings): a handful of threads,
// as the reference gives its own per-read host work to its `-t` threads.  f(first, last) over [0, n) in contiguous parts.
template <class F>
static void parallel_parts(int n, int max_threads, F f) {
    int nt = std::min(max_threads, std::max(1, n / 4096));
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + nt - 1) / nt;
    int started = 0;
    try {
        for (int t = 1; t < nt; ++t) { th.emplace_back(f, std::min(n, t * per), std::min(n, (t + 1) * per)); started = t; }
    } catch (...) {                                     // no thread to be had: the caller's thread does the rest
        for (auto &x : th) x.join();
        f(std::min(n, (started + 1) * per), n);
        f(0, std::min(n, per));
        return;
    }
    f(0, std::min(n, per));
    for (auto &x : th) x.join();
}
static int host_threads() {
    static const int n = [] { const char *e = getenv("BWAMS_HOST_THREADS"); int v = e ? atoi(e) : 6; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    return n;
}
static bool verbose() { static const bool v = getenv("BWAMS_VERBOSE") != nullptr; return v; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static Slot *take_slot(bwams_worker &w, SlotState want, const bseq1_t *owner, bool wait) {
    std::unique_lock<std::mutex> g(w.mu);
    for (;;) {
        for (Slot &s : w.slots)
            if (s.state == want && (want == SLOT_FREE || s.owner == owner)) { s.state = SLOT_BUSY; return &s; }
        if (!wait) return nullptr;
        w.cv.wait(g);
    }
}
static void put_slot(bwams_worker &w, Slot *s, SlotState st) {
    {
        std::lock_guard<std::mutex> g(w.mu);
        s->state = st;
        if (st == SLOT_FREE) s->owner = nullptr;
    }
    w.cv.notify_all();
}

// the records as flat page-locked arrays — seq becomes base codes in place (mem_kernel1_core, src/bwamem.cpp:1226-1237) — and up
static int stage_into(bwams_worker &w, Slot &s, const mem_opt_t *opt, int n, bseq1_t *seqs) {
    if ((int64_t)n > w.max_reads) { w.err = "chunk of " + std::to_string(n) + " reads, the worker was sized for " + std::to_string(w.max_reads); return BWAMS_ERR_CAPACITY; }
    const double t0 = now_ms();
    const int paired = (opt->flag & MEM_F_PE) ? 1 : 0;
    const int nt = host_threads();
    // pass 1: sizes -> offsets
    bool any_qual = false, all_qual = true, any_comment = false;
    {
        std::mutex mu;
        parallel_parts(n, nt, [&](int a, int b) {
            bool aq = false, lq = true, ac = false;
            for (int i = a; i < b; ++i) {
                s.cum[i] = seqs[i].l_seq;
                s.name_off[i] = (int64_t)strlen(seqs[i].name);
                s.comment_off[i] = seqs[i].comment ? (int64_t)strlen(seqs[i].comment) : 0;
                ac |= seqs[i].comment != nullptr;
                if (seqs[i].qual) aq = true; else lq = false;
            }
            std::lock_guard<std::mutex> g(mu);
            any_qual |= aq; all_qual &= lq; any_comment |= ac;
        });
    }
    int64_t nb = 0, nn = 0, nc = 0;
    for (int i = 0; i < n; ++i) {                       // lengths -> exclusive offsets
        const int64_t l = s.cum[i], a = s.name_off[i], c = s.comment_off[i];
        s.cum[i] = nb; s.name_off[i] = nn; s.comment_off[i] = nc;
        nb += l; nn += a; nc += c;
    }
    s.cum[n] = nb; s.name_off[n] = nn; s.comment_off[n] = nc;
    if (nb > w.max_bases) { w.err = "chunk of " + std::to_string(nb) + " bases, the worker was sized for " + std::to_string(w.max_bases); return BWAMS_ERR_CAPACITY; }
    int rc = grow(s.names, s.names_cap, nn + 1);
    if (!rc && any_comment) rc = grow(s.comments, s.comments_cap, nc + 1);
    if (rc) return rc;
    // pass 2: the bytes
    bool refuse = any_qual && !all_qual;                // a chunk mixing records with and without qualities: host path
    {
        std::mutex mu;
        const bool qual_up = all_qual && any_qual;
        parallel_parts(n, nt, [&](int a, int b) {
            bool bad = false;
            for (int i = a; i < b; ++i) {
                const int l = seqs[i].l_seq;
                unsigned char *q = reinterpret_cast<unsigned char *>(seqs[i].seq);
                uint8_t *e = s.enc + s.cum[i];
                uint8_t mx = 0;
                for (int j = 0; j < l; ++j) {
                    const uint8_t c = kNt4.t[q[j]];
                    q[j] = c;
                    e[j] = c;
                    mx |= c;                            // 5 ('-') has bits a code 0..4 pair cannot make: 1 | 4, yes — checked exactly below
                }
                if (mx >= 5) for (int j = 0; j < l; ++j) bad |= e[j] > 4;      // '-': the device path has no code for it
                if (qual_up) memcpy(s.qual + s.cum[i], seqs[i].qual, (size_t)l);
                memcpy(s.names + s.name_off[i], seqs[i].name, (size_t)(s.name_off[i + 1] - s.name_off[i]));
                if (seqs[i].comment) memcpy(s.comments + s.comment_off[i], seqs[i].comment, (size_t)(s.comment_off[i + 1] - s.comment_off[i]));
                seqs[i].sam = nullptr;
            }
            if (bad) { std::lock_guard<std::mutex> g(mu); refuse = true; }
        });
    }
    s.owner = seqs; s.n = n; s.refuse = refuse; s.bytes = 0;
    const double t1 = now_ms();
    if (refuse) return BWAMS_OK;                        // decided in mem_process_seqs (host path, or the run ends)
    rc = bwams_multi_upload(s.multi, s.enc, s.cum, n, s.names, s.name_off, (any_qual && all_qual) ? s.qual : nullptr,
                            any_comment ? s.comments : nullptr, any_comment ? s.comment_off : nullptr, paired);
    if (rc) w.err = bwams_multi_error(s.multi);
    if (verbose()) fprintf(stderr, "[bwams_worker] stage: %d reads, records -> page-locked arrays %.1f ms (%d threads), upload %.1f ms\n", n, t1 - t0, nt, now_ms() - t1);
    return rc;
}

// the text back, one string per 512-read work item as worker_sam leaves it (src/bwamem.cpp:1722, :1823); find_perfect_match_entry's
// record of every read (src/perfect_map.cpp:638-659)
static int fetch_text(bwams_worker &w, Slot &s) {
    int rc = grow(s.sam, s.sam_cap, s.bytes + 1);
    if (rc) return rc;
    if ((rc = bwams_multi_fetch(s.multi, s.sam, s.sam_cap, s.sam_off))) w.err = bwams_multi_error(s.multi);
    return rc;
}
static int cut_strings(bwams_worker &w, const char *sam, const int64_t *sam_off, int n, bseq1_t *seqs) {
    const int n_items = (n + BATCH_SIZE - 1) / BATCH_SIZE;
    std::mutex mu;
    bool oom = false;
    parallel_parts(n_items, host_threads(), [&](int a, int b) {
        for (int k = a; k < b; ++k) {
            const int i = k * BATCH_SIZE, e = i + BATCH_SIZE < n ? i + BATCH_SIZE : n;
            const int64_t len = sam_off[e] - sam_off[i];
            char *t = static_cast<char *>(malloc((size_t)len + 1));
            if (!t) { std::lock_guard<std::mutex> g(mu); oom = true; return; }
            memcpy(t, sam + sam_off[i], (size_t)len);
            t[len] = 0;
            seqs[i].sam = t;
        }
    });
    if (oom) {
        for (int i = 0; i < n; i += BATCH_SIZE) { free(seqs[i].sam); seqs[i].sam = nullptr; }
        w.err = "out of memory";
        return BWAMS_ERR_NOMEM;
    }
    return BWAMS_OK;
}
static int collect_from(bwams_worker &w, Slot &s, int n, bseq1_t *seqs) {
    const double t0 = now_ms();
    int rc = fetch_text(w, s);
    if (rc) return rc;
    const double t1 = now_ms();
    rc = cut_strings(w, s.sam, s.sam_off, n, seqs);
    if (verbose()) fprintf(stderr, "[bwams_worker] collect: %lld bytes down %.1f ms, %d work-item strings %.1f ms\n", (long long)s.bytes, t1 - t0, (n + BATCH_SIZE - 1) / BATCH_SIZE, now_ms() - t1);
    return rc;
}

int mem_process_seqs_stage(mem_opt_t *opt, int n, bseq1_t *seqs, bwams_worker &w) {
    if (n <= 0) return BWAMS_OK;
    Slot *s = take_slot(w, SLOT_FREE, nullptr, true);
    const int rc = stage_into(w, *s, opt, n, seqs);
    put_slot(w, s, rc ? SLOT_FREE : SLOT_STAGED);
    return rc;
}

static int emf_records(bwams_worker &w, Slot &s, int n, int paired, bseq1_t *seqs) {
    if (!w.emf[0]) return BWAMS_OK;
    std::vector<int64_t> bounds((size_t)w.n_dev + 1);
    int rc = bwams_shard_bounds(n, w.n_dev, paired, bounds.data());
    for (int d = 0; d < w.n_dev && !rc; ++d)
        if (bounds[(size_t)d + 1] > bounds[(size_t)d]) rc = bwams_emf_fetch(s.batch[(size_t)d], s.perfect + bounds[(size_t)d], s.code + bounds[(size_t)d]);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        seqs[i].perfect.exist = 0;
        if (s.code[i] == 3 || s.code[i] == 4) { seqs[i].perfect.flags = s.perfect[i].flags; seqs[i].perfect.location = s.perfect[i].location; }
    }
    return BWAMS_OK;
}

int mem_process_seqs_collect(mem_opt_t *opt, int n, bseq1_t *seqs, bwams_worker &w) {
    if (n <= 0) return BWAMS_OK;
    Slot *s = take_slot(w, SLOT_COMPUTED, seqs, false);
    if (!s) return BWAMS_OK;                            // collected inside mem_process_seqs (not deferred), or the host path ran the chunk
    if (!w.spare_off) {                                 // depth 1: nothing to overlap with
        int rc = collect_from(w, *s, n, seqs);
        if (!rc) rc = emf_records(w, *s, n, (opt->flag & MEM_F_PE) ? 1 : 0, seqs);
        put_slot(w, s, SLOT_FREE);
        return rc;
    }
    // the text comes down into the slot's buffers, which then change places with the spare ones: the slot — its devices' buffers, its
    // staging arrays — is free for the next chunk while this one's strings are cut
    std::lock_guard<std::mutex> sp(w.spare_mu);
    const double t0 = now_ms();
    int rc = fetch_text(w, *s);
    if (!rc) rc = emf_records(w, *s, n, (opt->flag & MEM_F_PE) ? 1 : 0, seqs);
    const long long bytes = (long long)s->bytes;
    if (!rc) { std::swap(s->sam, w.spare_sam); std::swap(s->sam_cap, w.spare_cap); std::swap(s->sam_off, w.spare_off); }
    put_slot(w, s, SLOT_FREE);
    const double t1 = now_ms();
    if (!rc) rc = cut_strings(w, w.spare_sam, w.spare_off, n, seqs);
    if (verbose()) fprintf(stderr, "[bwams_worker] collect: %lld bytes down %.1f ms (slot released), %d work-item strings %.1f ms\n", bytes, t1 - t0, (n + BATCH_SIZE - 1) / BATCH_SIZE, now_ms() - t1);
    return rc;
}

// step 1 without the reference's way of ending the run: 0, or a BWAMS_ERR_* with bwams_worker_error()
int bwams_worker_process(mem_opt_t *opt, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0, bwams_worker &w) {
    if (n <= 0) return BWAMS_OK;
    Slot *s = take_slot(w, SLOT_STAGED, seqs, false);
    int rc = BWAMS_OK;
    if (!s) {
        s = take_slot(w, SLOT_FREE, nullptr, true);
        if ((rc = stage_into(w, *s, opt, n, seqs))) { put_slot(w, s, SLOT_FREE); return rc; }
    }
    if (s->refuse) {                                    // an input the device path refuses: the reference's own code runs the chunk
        put_slot(w, s, SLOT_FREE);
        if (w.host_path) { w.host_path(opt, n_processed, n, seqs, pes0, w.host_user); return BWAMS_OK; }
        w.err = "a chunk the device path refuses (records with and without qualities, or a '-' base) and no host path is set";
        return BWAMS_ERR_UNSUPPORTED;
    }
    bwams_seed_opt_t so; bwams_mem_opt_t mo; bwams_sam_opt_t sa;
    bwams_map_options(opt, w.rg_id, &so, &mo, &sa);
    rc = bwams_multi_compute(s->multi, &so, &mo, &sa, reinterpret_cast<const bwams_pestat_t *>(pes0), n_processed,
                             (opt->flag & MEM_F_NO_RESCUE) ? BWAMS_PAIR_NO_RESCUE : 0, &s->bytes);
    if (rc == BWAMS_ERR_UNSUPPORTED && w.host_path) {
        put_slot(w, s, SLOT_FREE);
        w.host_path(opt, n_processed, n, seqs, pes0, w.host_user);
        return BWAMS_OK;
    }
    if (rc) { w.err = bwams_multi_error(s->multi); put_slot(w, s, SLOT_FREE); return rc; }
    if (w.deferred_collect) { put_slot(w, s, SLOT_COMPUTED); return BWAMS_OK; }
    rc = collect_from(w, *s, n, seqs);
    if (!rc) rc = emf_records(w, *s, n, (opt->flag & MEM_F_PE) ? 1 : 0, seqs);
    put_slot(w, s, SLOT_FREE);
    return rc;
}

void mem_process_seqs(mem_opt_t *opt, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0, bwams_worker &w) {
    const int rc = bwams_worker_process(opt, n_processed, n, seqs, pes0, w);
    if (rc) die("mem_process_seqs", rc, w.err.c_str());
}
