// mem_process_seqs_hip.cpp — mem_process_seqs() (/root/reference/src/bwamem.cpp:1850-1903) with its body on the GPU: the
// reference's records in (bseq1_t), the reference's per-work-item SAM strings out, everything in between one call into the
// C-ABI of libbwams.so (bwams_process_reads).  Host C++ only: no HIP header is needed here.
#include "bwamem_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

struct bwams_worker {
    bwams_index_t *idx = nullptr;
    bwams_emf_t *emf = nullptr;
    bwams_ert_t *ert = nullptr;
    bwams_batch_t *batch = nullptr;
    int64_t max_reads = 0, max_bases = 0;
    char rg_id[256] = {0};
    bwams_host_path_t host_path = nullptr;
    void *host_user = nullptr;
    // page-locked staging (bwams_host_alloc): what goes up and what comes down every chunk
    uint8_t *enc = nullptr;
    char *qual = nullptr, *names = nullptr, *comments = nullptr, *sam = nullptr;
    int64_t *cum = nullptr, *name_off = nullptr, *comment_off = nullptr, *sam_off = nullptr;
    int64_t names_cap = 0, comments_cap = 0, sam_cap = 0;
    bwams_perfect_t *perfect = nullptr;
    uint8_t *code = nullptr;
};

static void die(const char *what, int rc) {
    // the reference's convention: a line on stderr, exit(EXIT_FAILURE) (e.g. src/bwamem.cpp:3638-3645)
    fprintf(stderr, "[bwams] %s: %s: %s\n", what, bwams_strerror(rc), bwams_last_error());
    exit(EXIT_FAILURE);
}

template <class T>
static void pinned(T *&p, size_t n) {
    void *q = nullptr;
    const int rc = bwams_host_alloc(n * sizeof(T), &q);
    if (rc) die("bwams_host_alloc", rc);
    p = static_cast<T *>(q);
}

template <class T>
static void grow(T *&p, int64_t &cap, int64_t need) {
    if (need <= cap) return;
    if (p) bwams_host_free(p);
    p = nullptr;
    cap = need + need / 4 + 4096;
    pinned(p, (size_t)cap);
}

int bwams_worker_create(bwams_index_t *idx, bwams_emf_t *emf, bwams_ert_t *ert, int64_t max_reads, int64_t max_bases, const char *rg_id,
                        bwams_worker **out) {
    if (!idx || !out || max_reads <= 0 || max_bases <= 0) return BWAMS_ERR_ARG;
    bwams_worker *w = new bwams_worker();
    w->idx = idx; w->emf = emf; w->ert = ert;
    w->max_reads = max_reads; w->max_bases = max_bases;
    if (rg_id) { strncpy(w->rg_id, rg_id, sizeof w->rg_id - 1); }
    int rc = bwams_batch_create(idx, max_reads, max_bases, 0, 0, &w->batch);
    if (rc) { delete w; return rc; }
    pinned(w->enc, (size_t)max_bases);
    pinned(w->qual, (size_t)max_bases);
    pinned(w->cum, (size_t)max_reads + 1);
    pinned(w->name_off, (size_t)max_reads + 1);
    pinned(w->comment_off, (size_t)max_reads + 1);
    pinned(w->sam_off, (size_t)max_reads + 1);
    pinned(w->perfect, (size_t)max_reads);
    pinned(w->code, (size_t)max_reads);
    *out = w;
    return BWAMS_OK;
}

void bwams_worker_destroy(bwams_worker *w) {
    if (!w) return;
    if (w->batch) bwams_batch_destroy(w->batch);
    void *ps[] = {w->enc, w->qual, w->names, w->comments, w->sam, w->cum, w->name_off, w->comment_off, w->sam_off, w->perfect, w->code};
    for (void *p : ps) if (p) bwams_host_free(p);
    delete w;
}

void bwams_worker_set_host_path(bwams_worker *w, bwams_host_path_t f, void *user) { w->host_path = f; w->host_user = user; }

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

// nst_nt4_table (src/bntseq.cpp:64-81): A C G T in either case -> 0..3, '-' -> 5, everything else -> 4
static inline uint8_t nt4(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        case '-': return 5;
        default: return 4;
    }
}

void mem_process_seqs(mem_opt_t *opt, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0, bwams_worker &w) {
    if (n <= 0) return;
    const int paired = (opt->flag & MEM_F_PE) ? 1 : 0;
    if ((int64_t)n > w.max_reads) { fprintf(stderr, "[bwams] mem_process_seqs: %d reads, the worker was sized for %lld\n", n, (long long)w.max_reads); exit(EXIT_FAILURE); }
    // ---- the records as flat arrays; seq becomes base codes in place (mem_kernel1_core, src/bwamem.cpp:1226-1237)
    int64_t nb = 0, nn = 0, nc = 0;
    bool any_qual = false, all_qual = true, any_comment = false;
    for (int i = 0; i < n; ++i) {
        nb += seqs[i].l_seq;
        nn += (int64_t)strlen(seqs[i].name);
        if (seqs[i].comment) { nc += (int64_t)strlen(seqs[i].comment); any_comment = true; }
        if (seqs[i].qual) any_qual = true; else all_qual = false;
    }
    if (nb > w.max_bases) { fprintf(stderr, "[bwams] mem_process_seqs: %lld bases, the worker was sized for %lld\n", (long long)nb, (long long)w.max_bases); exit(EXIT_FAILURE); }
    bool refuse = any_qual && !all_qual;                // a chunk mixing records with and without qualities: host path
    grow(w.names, w.names_cap, nn + 1);
    if (any_comment) grow(w.comments, w.comments_cap, nc + 1);
    int64_t ob = 0, on = 0, oc = 0;
    for (int i = 0; i < n; ++i) {
        const int l = seqs[i].l_seq;
        w.cum[i] = ob; w.name_off[i] = on; w.comment_off[i] = oc;
        unsigned char *s = reinterpret_cast<unsigned char *>(seqs[i].seq);
        for (int j = 0; j < l; ++j) {
            const uint8_t c = s[j] < 4 ? s[j] : nt4(s[j]);
            s[j] = c;
            w.enc[ob + j] = c;
            refuse |= c > 4;                            // '-': the device path has no code for it
        }
        if (all_qual && any_qual) memcpy(w.qual + ob, seqs[i].qual, (size_t)l);
        const size_t ln = strlen(seqs[i].name);
        memcpy(w.names + on, seqs[i].name, ln);
        on += (int64_t)ln;
        if (seqs[i].comment) { const size_t lc = strlen(seqs[i].comment); memcpy(w.comments + oc, seqs[i].comment, lc); oc += (int64_t)lc; }
        ob += l;
        seqs[i].sam = nullptr;
    }
    w.cum[n] = ob; w.name_off[n] = on; w.comment_off[n] = oc;

    bwams_seed_opt_t so; bwams_mem_opt_t mo; bwams_sam_opt_t sa;
    bwams_map_options(opt, w.rg_id, &so, &mo, &sa);
    int64_t bytes = 0;
    int rc = refuse ? BWAMS_ERR_UNSUPPORTED
                    : bwams_process_reads(w.batch, w.emf, w.ert, &so, &mo, &sa, w.enc, w.cum, n, w.names, w.name_off,
                                          (any_qual && all_qual) ? w.qual : nullptr, any_comment ? w.comments : nullptr,
                                          any_comment ? w.comment_off : nullptr, paired, reinterpret_cast<const bwams_pestat_t *>(pes0),
                                          n_processed, (opt->flag & MEM_F_NO_RESCUE) ? BWAMS_PAIR_NO_RESCUE : 0, &bytes);
    if (rc == BWAMS_ERR_UNSUPPORTED && w.host_path) {   // an input or option the device path refuses: the reference's own code runs the chunk
        w.host_path(opt, n_processed, n, seqs, pes0, w.host_user);
        return;
    }
    if (rc) die("mem_process_seqs", rc);
    // ---- the text back, one string per 512-read work item as worker_sam leaves it (src/bwamem.cpp:1722, :1823)
    grow(w.sam, w.sam_cap, bytes + 1);
    if ((rc = bwams_sam_fetch(w.batch, w.sam, w.sam_cap, w.sam_off, nullptr, 0))) die("bwams_sam_fetch", rc);
    for (int i = 0; i < n; i += BATCH_SIZE) {
        const int e = i + BATCH_SIZE < n ? i + BATCH_SIZE : n;
        const int64_t len = w.sam_off[e] - w.sam_off[i];
        char *s = static_cast<char *>(malloc((size_t)len + 1));
        if (!s) { fprintf(stderr, "[bwams] mem_process_seqs: out of memory\n"); exit(EXIT_FAILURE); }
        memcpy(s, w.sam + w.sam_off[i], (size_t)len);
        s[len] = 0;
        seqs[i].sam = s;
    }
    if (w.emf) {                                        // find_perfect_match_entry's record of every read (src/perfect_map.cpp:638-659)
        if ((rc = bwams_emf_fetch(w.batch, w.perfect, w.code))) die("bwams_emf_fetch", rc);
        for (int i = 0; i < n; ++i) {
            seqs[i].perfect.exist = 0;
            if (w.code[i] == 3 || w.code[i] == 4) { seqs[i].perfect.flags = w.perfect[i].flags; seqs[i].perfect.location = w.perfect[i].location; }
        }
    }
}
