/*
 * ert_oracle.c — CPU restatement of the ERT (enumerated radix tree) index: its writer, its
 * decoding, and seeding over it (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * PARITY UNPINNED: src/ertindex.cpp and src/ertseeding.cpp include the un-vendored safestringlib and
 * cannot be compiled here; the reference ships no ERT fixtures.  The layout below is restated from
 * the writer (the only definition of the format); the decoder is checked from first principles
 * (tests/test_oracle_ert.py: longest matches and hit lists by brute force over the text) and the
 * seeding result against the FM-index restatement (orc_collect_smem + orc_sa_lookup), which the
 * reference's ERT mode is written to reproduce ("for equivalency with BWA-MEM",
 * src/ertseeding.cpp:2891, :3444-3461; SURVEY.md §8c addendum: 0 differing SAM lines).
 *
 * Writer, function by function:
 *   buildIndex                 /root/reference/src/ertindex.cpp:490-771   (k-mer entry, LEP bits, 2/3/4-byte pointers)
 *   ert_build_kmertree         /root/reference/src/ertindex.cpp:148-211
 *   handleDivergence/Leaf      /root/reference/src/ertindex.cpp:88-146
 *   ert_build_table            /root/reference/src/ertindex.cpp:213-311   (x-mer table of FREQUENT k-mers)
 *   ert_traverse_kmertree      /root/reference/src/ertindex.cpp:380-476   (+ add* :313-378)
 *   buildKmerTrees             /root/reference/src/ertindex.cpp:773-943   (offsets; the per-thread split does not
 *                                                                          change the bytes)
 * Decoder (entry / node fields only; the reference's LEP-driven search order is not restated):
 *   rightExtend                /root/reference/src/ertseeding.cpp:2142-2305  (k-mer / x-mer entry fields)
 *   getNextByteIdx             /root/reference/src/ertseeding.cpp:836-975    (LEAF / UNIFORM / DIVERGE steps)
 *   getOffsetToChildNode       /root/reference/src/ertseeding.cpp:485-497    (pointer = offset << 6 | hits)
 *   getNextByteIdx_dfs         /root/reference/src/ertseeding.cpp:521-587    (leaves in A, C, G, T order)
 *   get_seeds_prefix           /root/reference/src/ertseeding.cpp:2940-2965  (leaf expansion against .0123)
 * Seeding rounds (what the walk has to deliver):
 *   mem_kernel1_core_ert       /root/reference/src/bwamem.cpp:1122-1204     (seeds, reseed, last, sort)
 *   mem_chain_new              /root/reference/src/bwamem.cpp:961-1050      (hit sampling by max_occ)
 *
 * The k-mer size, x-mer size, read length and FREQUENT threshold are macros in the reference
 * (src/macro.h:204-220: 15, 4, READ_LEN, 256); they are arguments here so that tests can build an index
 * of a toy genome in milliseconds.
 */
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

enum { N_EMPTY = 0, N_LEAF = 1, N_UNIFORM = 2, N_DIVERGE = 3 };       /* node_type_t, ertindex.h:12 */
enum { E_INVALID = 0, E_SINGLE = 1, E_INFREQUENT = 2, E_FREQUENT = 3 }; /* macro.h:216-219 */

typedef struct { int64_t k, l, s; } iv_t;

/* bwt_extend(bwt, ik, ok, 0): ok[i] is the interval of the pattern followed by base 3 - i */
static void ext4(const orc_fmi_t *f, iv_t ik, iv_t ok[4])
{
    for (int b = 0; b < 4; ++b) {
        bwams_smem_t in, out;
        memset(&in, 0, sizeof in);
        in.k = ik.k; in.l = ik.l; in.s = ik.s;
        orc_forward_ext(f, &in, b, &out);
        ok[3 - b].k = out.k; ok[3 - b].l = out.l; ok[3 - b].s = out.s;
    }
}

/* bwt_sa of the classic index: the text position of BWT row pos (no sentinel quirk) */
static int64_t sa_true(const orc_fmi_t *f, int64_t pos)
{
    int64_t sp = pos, offset = 0;
    for (;;) {
        if ((sp & 7) == 0) {
            int64_t v = (int64_t)f->sa_ms_byte[sp >> 3];
            v = (v << 32) + (int64_t)f->sa_ls_word[sp >> 3];
            return v + offset;
        }
        const bwams_cp_occ_t *b = &f->cp_occ[sp >> 6];
        const int sh = 63 - (int)(sp & 63);
        int c = 4;
        for (int a = 0; a < 4; ++a)
            if ((b->one_hot_bwt_str[a] >> sh) & 1) c = a;
        if (c == 4) return offset;          /* the row of the whole text: position 0 */
        sp = f->count[c] + orc_fmi_occ(f, sp, c);
        offset++;
    }
}

/* ------------------------------------------------------------------ writer -- */

typedef struct enode {
    uint8_t type, c;            /* c = seq[pos]: first edge symbol, in the 3 - base convention */
    int nchild, num_bp, cap_bp;
    uint8_t *bases;             /* UNIFORM run, 3 - base each */
    int64_t n_hits;
    int64_t *hits;
    struct enode *child[4];
} enode;

typedef struct { uint8_t *p; int64_t n, cap; } bytes_t;

static void b_reserve(bytes_t *b, int64_t n)
{
    if (n <= b->cap) return;
    int64_t c = b->cap ? b->cap : 256;
    while (c < n) c *= 2;
    b->p = (uint8_t *)realloc(b->p, (size_t)c);
    memset(b->p + b->cap, 0, (size_t)(c - b->cap));
    b->cap = c;
}
static void b_put_at(bytes_t *b, int64_t off, uint64_t v, int nbytes)
{
    b_reserve(b, off + nbytes);
    for (int i = 0; i < nbytes; ++i) b->p[off + i] = (uint8_t)(v >> (8 * i));
    if (off + nbytes > b->n) b->n = off + nbytes;
}
static void b_put(bytes_t *b, uint64_t v, int nbytes) { b_put_at(b, b->n, v, nbytes); }

static enode *node_new(void) { return (enode *)calloc(1, sizeof(enode)); }
static void node_free(enode *n)
{
    if (!n) return;
    for (int j = 0; j < n->nchild; ++j) node_free(n->child[j]);
    free(n->bases); free(n->hits); free(n);
}
static void node_push_bp(enode *n, uint8_t c)
{
    if (n->num_bp == n->cap_bp) {
        n->cap_bp = n->cap_bp ? 2 * n->cap_bp : 8;
        n->bases = (uint8_t *)realloc(n->bases, (size_t)n->cap_bp);
    }
    n->bases[n->num_bp++] = c;
}

static void handle_leaf(const orc_fmi_t *f, iv_t ik, enode *n)
{
    n->type = N_LEAF;
    n->n_hits = ik.s;
    n->hits = (int64_t *)calloc((size_t)(ik.s > 0 ? ik.s : 1), sizeof(int64_t));
    for (int64_t j = 0; j < ik.s; ++j) n->hits[j] = sa_true(f, ik.k + j);
}

static void build_tree(const orc_fmi_t *f, iv_t ik, int depth, enode *parent, int max_depth);

static void handle_divergence(const orc_fmi_t *f, const iv_t ok_in[4], int depth, enode *parent, int max_depth)
{
    iv_t ok[4];
    memcpy(ok, ok_in, sizeof ok);
    for (int i = 3; i >= 0; --i) {
        enode *n = node_new();
        parent->child[parent->nchild++] = n;
        if (ok[i].s == 0) {
            n->type = N_EMPTY;
        } else if (ok[i].s > 1 && depth != max_depth) {
            n->type = N_DIVERGE;
            n->c = (uint8_t)i;
            n->num_bp = 0;
            n->n_hits = ok[i].s;
            build_tree(f, ok[i], depth + 1, n, max_depth);
        } else {
            n->c = (uint8_t)i;
            handle_leaf(f, ok[i], n);
        }
    }
}

static void build_tree(const orc_fmi_t *f, iv_t ik, int depth, enode *parent, int max_depth)
{
    iv_t ok[4];
    ext4(f, ik, ok);
    int nb = 0, ub = 0;
    for (int i = 0; i < 4; ++i)
        if (ok[i].s > 0) { nb++; ub = i; }
    if (nb == 1) {
        enode *n = node_new();
        parent->child[parent->nchild++] = n;
        n->c = (uint8_t)ub;
        node_push_bp(n, (uint8_t)ub);
        n->n_hits = ok[ub].s;
        if (depth < max_depth) {
            const iv_t ok_init = ok[ub];
            iv_t ik_new = ok[ub];
            for (;;) {
                depth += 1;
                ext4(f, ik_new, ok);
                nb = 0; ub = 0;
                for (int i = 0; i < 4; ++i)
                    if (ok[i].s > 0) { nb++; ub = i; }
                if (nb == 1) {
                    ik_new = ok[ub];
                    node_push_bp(n, (uint8_t)ub);
                    if (depth == max_depth) { handle_leaf(f, ok_init, n); break; }   /* multi-hit leaf */
                } else {
                    n->type = N_UNIFORM;
                    handle_divergence(f, ok, depth, n, max_depth);
                    break;
                }
            }
        } else {
            handle_leaf(f, ok[ub], n);
        }
    } else {
        handle_divergence(f, ok, depth, parent, max_depth);
    }
}

static void put_mh(bytes_t *mlt, bytes_t *mh, const enode *ch)
{
    b_put(mlt, ((uint64_t)mh->n << 1) | 1ULL, 5);
    b_put(mh, (uint64_t)ch->n_hits, 2);
    for (int64_t k = 0; k < (int64_t)(uint16_t)ch->n_hits; ++k) b_put(mh, ((uint64_t)ch->hits[k] << 1) | 1ULL, 5);
}

static void emit_tree(const enode *n, bytes_t *mlt, bytes_t *mh, int w, uint64_t *max_ptr)
{
    if (n->nchild == 1) {
        const enode *ch = n->child[0];
        if (ch->type == N_LEAF) {
            b_put(mlt, (uint64_t)N_LEAF << (ch->c << 1), 1);
            put_mh(mlt, mh, ch);
        } else {                                   /* UNIFORM */
            b_put(mlt, (uint64_t)N_UNIFORM << (ch->c << 1), 1);
            b_put(mlt, (uint64_t)(uint8_t)ch->num_bp, 1);
            const int nby = (ch->num_bp + 3) >> 2;
            const int64_t at = mlt->n;
            b_reserve(mlt, at + nby);
            memset(mlt->p + at, 0, (size_t)nby);
            for (int j = 0; j < ch->num_bp; ++j) mlt->p[at + (j >> 2)] |= (uint8_t)(ch->bases[j] << ((~j & 3) << 1));
            mlt->n = at + nby;
            emit_tree(ch, mlt, mh, w, max_ptr);
        }
        return;
    }
    int n_empty = 0, n_leaf = 0;
    uint8_t code = 0;
    for (int j = 0; j < n->nchild; ++j) {
        const enode *ch = n->child[j];
        if (ch->type == N_EMPTY) n_empty++;
        else if (ch->type == N_LEAF) { n_leaf++; code |= (uint8_t)(N_LEAF << (ch->c << 1)); }
        else code |= (uint8_t)(N_DIVERGE << (ch->c << 1));
    }
    const int n_ptr = 4 - n_empty - n_leaf > 0 ? 4 - n_empty - n_leaf : 0;
    const int64_t start = mlt->n;
    b_put(mlt, code, 1);
    const int64_t ptr_at = mlt->n;
    if (n_ptr > 0) { b_reserve(mlt, ptr_at + n_ptr * w); mlt->n = ptr_at + n_ptr * w; }
    for (int j = 0; j < n->nchild; ++j) {
        const enode *ch = n->child[j];
        if (ch->type != N_LEAF) continue;
        if (ch->n_hits == 1) b_put(mlt, (uint64_t)ch->hits[0] << 1, 5);
        else put_mh(mlt, mh, ch);
    }
    int64_t to[5] = {0, 0, 0, 0, 0}, cnt[5] = {0, 0, 0, 0, 0};
    int oi = 0;
    if (n_ptr > 0) to[0] = mlt->n;
    for (int j = 0; j < n->nchild; ++j) {
        const enode *ch = n->child[j];
        if (ch->type != N_DIVERGE) continue;
        emit_tree(ch, mlt, mh, w, max_ptr);
        cnt[oi] = ch->n_hits;
        oi++;
        to[oi] = mlt->n;
    }
    for (int j = 0; j < n_ptr; ++j) {
        const uint64_t p = (uint64_t)(to[j] - start);
        if (p > *max_ptr) *max_ptr = p;
        const uint64_t v = cnt[j] < 20 ? (p << 6) | (uint64_t)cnt[j] : (p << 6);
        b_put_at(mlt, ptr_at + (int64_t)j * w, v, w);
    }
}

/* radix tree of one k-mer (INFREQUENT, ertindex.cpp:598-655): 4-byte offset of the multi-hit area, the tree, the area */
static int blob_tree(const orc_fmi_t *f, iv_t ik, int kmer, int max_depth, bytes_t *out)
{
    enode *root = node_new();
    root->type = N_DIVERGE;
    root->n_hits = ik.s;
    build_tree(f, ik, kmer, root, max_depth);
    bytes_t mlt = {0, 0, 0}, mh = {0, 0, 0};
    uint64_t max_ptr = 0;
    int w = 2;
    for (;;) {
        mlt.n = 0; mh.n = 0; max_ptr = 0;
        b_put(&mlt, 0, 4);
        emit_tree(root, &mlt, &mh, w, &max_ptr);
        if (w == 2 && max_ptr >= 1024 && max_ptr < 262144) { w = 3; continue; }
        if (w < 4 && max_ptr >= 262144) { w = 4; continue; }
        break;
    }
    b_put_at(&mlt, 0, (uint64_t)mlt.n, 4);
    const int64_t at = out->n;
    b_reserve(out, at + mlt.n + mh.n);
    memcpy(out->p + at, mlt.p, (size_t)mlt.n);
    if (mh.n) memcpy(out->p + at + mlt.n, mh.p, (size_t)mh.n);
    out->n = at + mlt.n + mh.n;
    free(mlt.p); free(mh.p);
    node_free(root);
    return w;
}

/* x-mer table of one FREQUENT k-mer (ert_build_table) */
static int blob_table(const orc_fmi_t *f, iv_t ik0, int kmer, int xmer, int max_depth, bytes_t *out)
{
    const int n_x = 1 << (2 * xmer);
    bytes_t mlt = {0, 0, 0}, mh = {0, 0, 0};
    uint64_t max_ptr = 0;
    int w = 2;
    for (;;) {
        mlt.n = 0; mh.n = 0; max_ptr = 0;
        b_put(&mlt, 0, 4);
        b_reserve(&mlt, 4 + 8 * (int64_t)n_x);
        memset(mlt.p + 4, 0, (size_t)(8 * n_x));
        mlt.n = 4 + 8 * (int64_t)n_x;
        uint64_t lep1 = 0;                       /* not reset between x-mers in the reference either (:227) */
        for (int x = 0; x < n_x; ++x) {
            iv_t ik = ik0, ok[4];
            int64_t prev = ik0.s;
            int j, c = 0;
            for (j = 0; j < xmer; ++j) {
                c = 3 - ((x >> (2 * j)) & 3);
                ext4(f, ik, ok);
                if (ok[c].s != prev) lep1 |= 1ULL << j;
                if (ok[c].s >= 1) { prev = ok[c].s; ik = ok[c]; } else break;
            }
            const int64_t num_hits = ok[c].s;
            const int64_t mlt_offset = mlt.n;
            uint16_t xdata;
            if (num_hits == 0) {
                xdata = (uint16_t)(((lep1 & 0x3FFF) << 2) | E_INVALID);
            } else if (num_hits == 1) {
                xdata = (uint16_t)(((lep1 & 0x3FFF) << 2) | E_SINGLE);
                b_put(&mlt, 0, 1);
                b_put(&mlt, (uint64_t)sa_true(f, ok[c].k) << 1, 5);
            } else {
                xdata = (uint16_t)(((lep1 & 0x3FFF) << 2) | E_INFREQUENT);
                enode *root = node_new();
                root->type = N_DIVERGE;
                build_tree(f, ik, kmer + j, root, max_depth);
                emit_tree(root, &mlt, &mh, w, &max_ptr);
                node_free(root);
            }
            uint64_t e = ((uint64_t)mlt_offset << 24) | xdata;
            if (num_hits < 20) e |= (uint64_t)num_hits << 17;
            e |= (uint64_t)(w < 4 ? w : 0) << 22;
            b_put_at(&mlt, 4 + 8 * (int64_t)x, e, 8);
        }
        if (w == 2 && max_ptr >= 1024 && max_ptr < 262144) { w = 3; continue; }
        if (w < 4 && max_ptr >= 262144) { w = 4; continue; }
        break;
    }
    b_put_at(&mlt, 0, (uint64_t)mlt.n, 4);
    const int64_t at = out->n;
    b_reserve(out, at + mlt.n + mh.n);
    memcpy(out->p + at, mlt.p, (size_t)mlt.n);
    if (mh.n) memcpy(out->p + at + mlt.n, mh.p, (size_t)mh.n);
    out->n = at + mlt.n + mh.n;
    free(mlt.p); free(mh.p);
    return w;
}

typedef struct { uint64_t idx; iv_t ik; uint64_t lep; } alive_t;
typedef struct {
    const orc_fmi_t *f;
    int kmer;
    uint64_t lep_mask;
    uint64_t *table;
    alive_t *alive;
    int64_t n_alive, cap_alive;
} kdfs_t;

/* the k-mer loop of buildIndex (:528-552) for every k-mer that shares the prefix aq[0..i): k-mers that die at
 * step i share their LEP bits, so they are filled in one sweep instead of being searched one by one */
static void kmer_dfs(kdfs_t *d, int i, uint64_t idx, iv_t ik, uint64_t lep, int64_t prev)
{
    if (i == d->kmer) {
        if (d->n_alive == d->cap_alive) {
            d->cap_alive = d->cap_alive ? 2 * d->cap_alive : 1024;
            d->alive = (alive_t *)realloc(d->alive, (size_t)d->cap_alive * sizeof(alive_t));
        }
        alive_t a; a.idx = idx; a.ik = ik; a.lep = lep;
        d->alive[d->n_alive++] = a;
        return;
    }
    iv_t ok[4];
    ext4(d->f, ik, ok);
    for (int b = 0; b < 4; ++b) {
        const int c = 3 - b;
        uint64_t lp = lep;
        if (ok[c].s != prev) lp |= 1ULL << (i - 1);
        const uint64_t id = idx | ((uint64_t)b << (2 * i));
        if (ok[c].s >= 1) {
            kmer_dfs(d, i + 1, id, ok[c], lp, ok[c].s);
        } else {
            const uint64_t v = ((lp & d->lep_mask) << 2) | E_INVALID;
            const uint64_t stride = 1ULL << (2 * (i + 1)), total = 1ULL << (2 * d->kmer);
            for (uint64_t t = id; t < total; t += stride) d->table[t] = v;
        }
    }
}

static int cmp_alive(const void *a, const void *b)
{
    const uint64_t x = ((const alive_t *)a)->idx, y = ((const alive_t *)b)->idx;
    return x < y ? -1 : x > y;
}

uint8_t *orc_ert_build(const orc_fmi_t *f, int kmer, int xmer, int read_len, int hit_threshold,
                       uint64_t *kmer_table, int64_t *mlt_bytes)
{
    kdfs_t d;
    memset(&d, 0, sizeof d);
    d.f = f; d.kmer = kmer; d.table = kmer_table;
    d.lep_mask = (1ULL << (kmer - 1)) - 1;
    for (int a = 0; a < 4; ++a) {
        iv_t ik;
        ik.k = f->count[a]; ik.l = f->count[3 - a]; ik.s = f->count[a + 1] - f->count[a];
        kmer_dfs(&d, 1, (uint64_t)a, ik, 0, ik.s);
    }
    qsort(d.alive, (size_t)d.n_alive, sizeof(alive_t), cmp_alive);
    bytes_t out = {0, 0, 0};
    const uint64_t total = 1ULL << (2 * kmer);
    int64_t ai = 0;
    const int max_depth = read_len - 1;
    for (uint64_t idx = 0; idx < total; ++idx) {
        const uint64_t ptr = (uint64_t)out.n;
        if (ai < d.n_alive && d.alive[ai].idx == idx) {
            const alive_t *a = &d.alive[ai++];
            const int64_t num_hits = a->ik.s;
            uint64_t data, w = 0;
            if (num_hits == 1) {
                data = ((a->lep & d.lep_mask) << 2) | E_SINGLE;
                b_put(&out, 0, 1);
                b_put(&out, (uint64_t)sa_true(f, a->ik.k) << 1, 5);
            } else if (num_hits <= hit_threshold) {
                data = ((a->lep & d.lep_mask) << 2) | E_INFREQUENT;
                w = (uint64_t)blob_tree(f, a->ik, kmer, max_depth, &out);
            } else {
                data = ((a->lep & d.lep_mask) << 2) | E_FREQUENT;
                w = (uint64_t)blob_table(f, a->ik, kmer, xmer, max_depth, &out);
            }
            uint64_t e = (ptr << 24) | data;
            if (num_hits < 20) e |= (uint64_t)num_hits << 17;
            e |= (w < 4 ? w : 0) << 22;
            kmer_table[idx] = e;
        } else {
            kmer_table[idx] = (ptr << 24) | kmer_table[idx];
        }
    }
    free(d.alive);
    *mlt_bytes = out.n;
    if (!out.p) out.p = (uint8_t *)calloc(1, 8);
    return out.p;
}

void orc_ert_free(uint8_t *p) { free(p); }

/* ----------------------------------------------------------------- decoder -- */

static inline uint64_t rd(const uint8_t *p, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

typedef struct {
    int kind;            /* 0 nothing, 1 one position, 2 multi-hit list at `at`, 3 subtree at `at` */
    int64_t at, pos;
    int64_t root;        /* the k-mer's blob (multi-hit offsets are relative to root + u32 at root) */
    int w;
} ert_where_t;

#define MANY 255

/* longest-match profile of read[i..): L[m-1] = the longest prefix with at least m occurrences, m = 1..M
 * (0 when shorter than the table lookups resolve: < kmer, or < kmer + xmer under a FREQUENT k-mer).
 * When stop_len > 0 the walk ends at that depth and `wh` describes where the hits of read[i, i+stop_len) are. */
static void ert_walk(const orc_ert_t *e, const uint8_t *q, int len, int i, int M, uint8_t *L, int stop_len, ert_where_t *wh)
{
    for (int m = 0; m < M; ++m) L[m] = 0;
    if (wh) memset(wh, 0, sizeof *wh);
    const int K = e->kmer, X = e->xmer;
    if (i + K > len) return;
    uint64_t key = 0;
    for (int j = 0; j < K; ++j) {
        if (q[i + j] > 3) return;
        key |= (uint64_t)q[i + j] << (2 * j);
    }
    const uint64_t ent = e->kmer_table[key];
    int code = (int)(ent & 3);
    const int64_t root = (int64_t)(ent >> 24);
    const int w = ((ent >> 22) & 3) == 0 ? 4 : (int)((ent >> 22) & 3);
    int cur = (int)((ent >> 17) & 31);      /* occurrences of the pattern matched so far; MANY = 20 or more */
    int d = K;                              /* bases matched */
    int64_t node = -1, leaf_pos = -1, mh_at = -1;
    const uint8_t *mlt = e->mlt;
    if (code == E_INVALID) return;
    if (cur == 0) cur = MANY;
    if (code == E_SINGLE) {
        leaf_pos = (int64_t)(rd(mlt + root + 1, 5) >> 1);
        cur = 1;
    } else if (code == E_INFREQUENT) {
        node = root + 4;
    } else {
        if (i + K + X > len) return;
        uint64_t xk = 0;
        for (int j = 0; j < X; ++j) {
            if (q[i + K + j] > 3) return;
            xk |= (uint64_t)q[i + K + j] << (2 * j);
        }
        const uint64_t xe = rd(mlt + root + 4 + 8 * (int64_t)xk, 8);
        code = (int)(xe & 3);
        if (code == E_INVALID) return;
        d = K + X;
        cur = (int)((xe >> 17) & 31);
        if (cur == 0) cur = MANY;
        if (code == E_SINGLE) {
            leaf_pos = (int64_t)(rd(mlt + root + (int64_t)(xe >> 24) + 1, 5) >> 1);
            cur = 1;
        } else {
            node = root + (int64_t)(xe >> 24);
        }
    }
    const int64_t mh_base = root + (int64_t)rd(mlt + root, 4);
#define DROP_TO(newc)                                                              \
    do {                                                                           \
        for (int m_ = (newc) + 1; m_ <= (cur < M ? cur : M); ++m_) L[m_ - 1] = (uint8_t)d; \
        cur = (newc);                                                              \
    } while (0)
    while (leaf_pos < 0) {
        if (stop_len > 0 && d >= stop_len) break;
        if (i + d >= len || q[i + d] > 3) break;
        const int c = 3 - q[i + d];
        const uint8_t cd = mlt[node];
        const int t = (cd >> (c << 1)) & 3;
        if (t == N_EMPTY) break;
        if (t == N_UNIFORM) {
            const int nbp = mlt[node + 1];
            int j = 0;
            for (; j < nbp; ++j) {
                if (stop_len > 0 && d + j >= stop_len) break;
                if (i + d + j >= len) break;
                const int bp = (mlt[node + 2 + (j >> 2)] >> ((~j & 3) << 1)) & 3;
                if (q[i + d + j] > 3 || 3 - q[i + d + j] != bp) break;
            }
            d += j;
            node = node + 2 + ((nbp + 3) >> 2);
            if (j < nbp) break;                 /* the subtree below the run holds the hits */
            continue;
        }
        int n_ptr = 0, before_leaf = 0, before_ptr = 0;
        for (int cc = 0; cc < 4; ++cc) {
            const int tt = (cd >> (cc << 1)) & 3;
            if (tt == N_DIVERGE) { n_ptr++; if (cc > c) before_ptr++; }
            if (tt == N_LEAF && cc > c) before_leaf++;
        }
        if (t == N_LEAF) {
            const uint64_t rec = rd(mlt + node + 1 + (int64_t)n_ptr * w + 5 * (int64_t)before_leaf, 5);
            int nc = 1;
            if (rec & 1) {
                mh_at = mh_base + (int64_t)(rec >> 1);
                nc = (int)rd(mlt + mh_at, 2);
                leaf_pos = (int64_t)(rd(mlt + mh_at + 2, 5) >> 1);
                if (nc >= 20) nc = MANY;
            } else {
                leaf_pos = (int64_t)(rec >> 1);
            }
            DROP_TO(nc);
            d += 1;
        } else {
            const uint64_t v = rd(mlt + node + 1 + (int64_t)before_ptr * w, w);
            int nc = (int)(v & 63);
            if (nc == 0) nc = MANY;
            DROP_TO(nc);
            d += 1;
            node = node + (int64_t)(v >> 6);
        }
    }
    if (leaf_pos >= 0) {
        /* lazy expansion of the leaf against the text (get_seeds_prefix :2940-2965) */
        while ((stop_len <= 0 || d < stop_len) && i + d < len && leaf_pos + d < e->ref_len && q[i + d] < 4 &&
               e->ref[leaf_pos + d] == q[i + d])
            d++;
    }
    for (int m = 1; m <= (cur < M ? cur : M); ++m) L[m - 1] = (uint8_t)d;
#undef DROP_TO
    if (wh) {
        wh->root = root; wh->w = w;
        if (leaf_pos >= 0 && mh_at >= 0) { wh->kind = 2; wh->at = mh_at; }
        else if (leaf_pos >= 0) { wh->kind = 1; wh->pos = leaf_pos; }
        else { wh->kind = 3; wh->at = node; }
    }
}

void orc_ert_profile(const orc_ert_t *e, const uint8_t *q, int len, int i, int M, uint8_t *L)
{
    ert_walk(e, q, len, i, M, L, 0, NULL);
}

/* leaves below a node in A, C, G, T order (getNextByteIdx_dfs); hits may be NULL to count only */
static int64_t dfs(const orc_ert_t *e, int64_t node, int64_t mh_base, int w, int64_t *hits, int64_t n, int64_t cap)
{
    const uint8_t *mlt = e->mlt;
    const uint8_t cd = mlt[node];
    int n_ptr = 0;
    for (int cc = 0; cc < 4; ++cc)
        if (((cd >> (cc << 1)) & 3) == N_DIVERGE) n_ptr++;
    int li = 0, pi = 0;
    for (int c = 3; c >= 0; --c) {
        const int t = (cd >> (c << 1)) & 3;
        if (t == N_UNIFORM) {
            const int nbp = mlt[node + 1];
            n = dfs(e, node + 2 + ((nbp + 3) >> 2), mh_base, w, hits, n, cap);
        } else if (t == N_LEAF) {
            const uint64_t rec = rd(mlt + node + 1 + (int64_t)n_ptr * w + 5 * (int64_t)li, 5);
            li++;
            if (rec & 1) {
                const int64_t at = mh_base + (int64_t)(rec >> 1);
                const int nc = (int)rd(mlt + at, 2);
                for (int k = 0; k < nc; ++k) {
                    if (hits && n < cap) hits[n] = (int64_t)(rd(mlt + at + 2 + 5 * (int64_t)k, 5) >> 1);
                    n++;
                }
            } else {
                if (hits && n < cap) hits[n] = (int64_t)(rec >> 1);
                n++;
            }
        } else if (t == N_DIVERGE) {
            const uint64_t v = rd(mlt + node + 1 + (int64_t)pi * w, w);
            pi++;
            n = dfs(e, node + (int64_t)(v >> 6), mh_base, w, hits, n, cap);
        }
    }
    return n;
}

/* every occurrence of read[i, i+mlen) in right-context order; returns the count (hits beyond cap are counted only) */
int64_t orc_ert_hits(const orc_ert_t *e, const uint8_t *q, int len, int i, int mlen, int64_t *hits, int64_t cap)
{
    uint8_t L[1];
    ert_where_t wh;
    ert_walk(e, q, len, i, 1, L, mlen, &wh);
    if (L[0] < mlen) return 0;
    if (wh.kind == 1) { if (hits && cap > 0) hits[0] = wh.pos; return 1; }
    if (wh.kind == 2) {
        const int nc = (int)rd(e->mlt + wh.at, 2);
        for (int k = 0; k < nc; ++k)
            if (hits && k < cap) hits[k] = (int64_t)(rd(e->mlt + wh.at + 2 + 5 * (int64_t)k, 5) >> 1);
        return nc;
    }
    if (wh.kind == 3) return dfs(e, wh.at, wh.root + (int64_t)rd(e->mlt + wh.root, 4), wh.w, hits, 0, cap);
    return 0;
}

/* ----------------------------------------------------------------- seeding -- */

static int cmp_smem3(const void *pa, const void *pb)
{
    const bwams_smem_t *a = (const bwams_smem_t *)pa, *b = (const bwams_smem_t *)pb;
    if (a->rid != b->rid) return a->rid < b->rid ? -1 : 1;
    if (a->m != b->m) return a->m < b->m ? -1 : 1;
    if (a->n != b->n) return a->n < b->n ? -1 : 1;
    return 0;
}

/* The three seeding rounds of mem_kernel1_core_ert (bwamem.cpp:1153-1193) expressed through the per-position
 * match profile: a match of read[i..) that can be lengthened neither way is an SMEM; reseeding keeps the matches
 * with at least `hitcount + 1` occurrences that cover the middle of a long SMEM; `last` walks forward until fewer
 * than max_mem_intv occurrences remain.  Output as orc_collect_smem + orc_sa_lookup leave it: SMEMs in
 * (rid, m, n) order with k = l = 0, s = occurrences; sa_coord = the occurrences mem_chain_new samples.
 * Returns the SMEM count, -1 when a buffer is too small, -2 for option values the index cannot answer. */
int64_t orc_ert_collect(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *enc, const int64_t *cum,
                        const uint8_t *skip, int32_t nseq, bwams_smem_t *out, int64_t cap, int64_t *sa_coord,
                        int64_t sa_cap, int64_t *sa_off)
{
    const int msl = opt->min_seed_len;
    const int M = opt->split_width + 1 > opt->max_mem_intv ? opt->split_width + 1 : opt->max_mem_intv;
    if (msl < e->kmer + e->xmer || M > 20 || M < 1) return -2;
    const int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
    int64_t n_out = 0;
    uint8_t *P = NULL;
    int cap_len = 0;
    for (int32_t r = 0; r < nseq; ++r) {
        if (skip && skip[r]) continue;
        const uint8_t *q = enc + cum[r];
        const int len = (int)(cum[r + 1] - cum[r]);
        if (len > e->read_len || len > 255) { free(P); return -2; }
        if (len > cap_len) { cap_len = len; P = (uint8_t *)realloc(P, (size_t)cap_len * (size_t)M); }
#define LM(m, i) P[(size_t)(i) * (size_t)M + (size_t)((m) - 1)]
        for (int i = 0; i < len; ++i) ert_walk(e, q, len, i, M, &LM(1, i), 0, NULL);
#define EMIT(st, ln)                                                                     \
    do {                                                                                 \
        if (n_out >= cap) { free(P); return -1; }                                        \
        bwams_smem_t *o_ = &out[n_out++];                                                \
        memset(o_, 0, sizeof *o_);                                                       \
        o_->rid = (uint32_t)r; o_->m = (uint32_t)(st); o_->n = (uint32_t)((st) + (ln) - 1); \
        int c_ = 0;                                                                      \
        for (int m_ = 1; m_ <= M; ++m_) if (LM(m_, st) >= (ln)) c_ = m_;                 \
        o_->s = c_ >= M ? -1 : c_;                                                       \
    } while (0)
        const int64_t first = n_out;
        /* round 1 */
        for (int i = 0; i < len; ++i) {
            const int l1 = LM(1, i);
            if (l1 < msl) continue;
            if (i > 0 && i + l1 <= i - 1 + LM(1, i - 1)) continue;
            EMIT(i, l1);
        }
        /* occurrences of the long round-1 SMEMs are needed now when they saturate the stored counters */
        const int64_t n1 = n_out;
        for (int64_t t = first; t < n1; ++t) {
            if (out[t].s < 0) out[t].s = orc_ert_hits(e, q, len, (int)out[t].m, (int)(out[t].n - out[t].m + 1), NULL, 0);
        }
        /* round 2 */
        for (int64_t t = first; t < n1; ++t) {
            const int st = (int)out[t].m, en = (int)out[t].n + 1;
            if (en - st < split_len || out[t].s > opt->split_width) continue;
            const int x = (st + en) >> 1, m = (int)out[t].s + 1;
            for (int i = x; i >= 0; --i) {
                const int l = LM(m, i);
                if (l > 0 && i + l <= x) break;
                if (l < msl || i + l <= x) continue;
                if (i > 0 && i + l <= i - 1 + LM(m, i - 1)) continue;
                EMIT(i, l);
            }
        }
        /* round 3 */
        if (opt->max_mem_intv > 0) {
            int x = 0;
            while (x < len) {
                if (q[x] > 3) { x++; continue; }
                /* forward loop of bwtSeedStrategyAllPosOneThread: ends at the first length >= min_seed_len + 1 whose
                 * interval is below max_mem_intv */
                int want = LM(opt->max_mem_intv, x) + 1;
                if (want < msl + 1) want = msl + 1;
                int nn = -1;
                for (int j = x + 1; j < x + want && j < len; ++j)
                    if (q[j] > 3) { nn = j; break; }
                if (nn >= 0) { x = nn + 1; continue; }
                if (x + want > len) { x = len; continue; }
                if (LM(1, x) >= want) EMIT(x, want);
                x = x + want;
            }
        }
        for (int64_t t = n1; t < n_out; ++t)
            if (out[t].s < 0) out[t].s = orc_ert_hits(e, q, len, (int)out[t].m, (int)(out[t].n - out[t].m + 1), NULL, 0);
#undef EMIT
#undef LM
    }
    free(P);
    qsort(out, (size_t)n_out, sizeof(bwams_smem_t), cmp_smem3);
    if (sa_coord && sa_off) {
        int64_t tot = 0;
        int64_t *tmp = NULL;
        int64_t tmp_cap = 0;
        for (int64_t t = 0; t < n_out; ++t) {
            const bwams_smem_t *p = &out[t];
            const uint8_t *q = enc + cum[p->rid];
            const int len = (int)(cum[p->rid + 1] - cum[p->rid]);
            sa_off[t] = tot;
            if (p->s > tmp_cap) { tmp_cap = p->s; tmp = (int64_t *)realloc(tmp, (size_t)tmp_cap * sizeof(int64_t)); }
            const int64_t got = orc_ert_hits(e, q, len, (int)p->m, (int)(p->n - p->m + 1), tmp, tmp_cap);
            if (got != p->s) { free(tmp); return -3; }
            const int64_t step = p->s > opt->max_occ ? p->s / opt->max_occ : 1;
            int64_t c = 0;
            for (int64_t k = 0; k < p->s && c < opt->max_occ; k += step, ++c) {
                if (tot + c >= sa_cap) { free(tmp); return -1; }
                sa_coord[tot + c] = tmp[k];
            }
            tot += c;
        }
        sa_off[n_out] = tot;
        free(tmp);
    }
    return n_out;
}
