/*
 * emf_oracle.c — CPU restatement of the exact-match filter (EMF) probe
 * (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).
 *
 * Follows, /root/reference/src:
 *   __get_hash_idx_fw / _rc, __fmix64        perfect.h:494-707
 *   __compare_fw_rc                          perfect.h:362-368
 *   ____seedcmp / __seedcmp                  perfect.h:273-360
 *   __seedmatch_further(_fw/_rc)             perfect.h:415-491
 *   GET_MULTI_FW_AND_RC                      perfect.h:170-186
 *   seedmatch_further                        perfect_map.cpp:528-581
 *   __find_perfect_match_entry               perfect_map.cpp:583-629
 *   seed_with_N, find_perfect_match_entry    perfect_map.cpp:631-659
 * Pinning: the inline primitives (hash, canonical-strand test, ordered compare, tail match)
 * are PINNED against perfect.h compiled where it lies (oracle/_ref/libref_emf.so,
 * tests/test_oracle_emf.py).  The probe loop lives in perfect_map.cpp, which is not buildable
 * here: PARITY UNPINNED for the loop itself; it is checked against brute force (does the read
 * occur exactly in the reference, and where).
 */
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

#define FLAG_FW_LESS 0x1u
#define FLAG_COLLISION 0x2u
#define NO_ENTRY 0xffffffffu
#define FLAG_VALID 0x1u
#define FLAG_RC 0x2u

static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

/* base i of the canonical string of s[0..len): s itself when fw, else its reverse complement */
static inline int canon_at(const uint8_t *s, int len, int fw, int i)
{
    return fw ? (s[i] & 3) : 3 - (s[len - 1 - i] & 3);
}

/* hash of the string read in the given orientation: XOR of 32-base words (first base most
 * significant), the last partial word right-aligned; then fmix64 mod table size */
int64_t orc_emf_hash(uint32_t num_seed_entry, const uint8_t *s, int len, int fw)
{
    uint64_t h = 0, w = 0;
    int i, full = len - len % 32;
    for (i = 0; i < full; ++i) {
        w = (w << 2) | (uint64_t)canon_at(s, len, fw, i);
        if ((i & 31) == 31) { h ^= w; w = 0; }
    }
    if (len % 32) {
        w = 0;
        for (; i < len; ++i) w = (w << 2) | (uint64_t)canon_at(s, len, fw, i);
        h ^= w;
    }
    return (int64_t)(fmix64(h) % num_seed_entry);
}

/* lexicographic comparison of two strings, each read forward (fl = 1) or as reverse complement */
int orc_emf_seedcmp(const uint8_t *a, int afl, const uint8_t *b, int bfl, int len)
{
    for (int i = 0; i < len; ++i) {
        const int x = afl ? a[i] : 3 - a[len - 1 - i];
        const int y = bfl ? b[i] : 3 - b[len - 1 - i];
        if (x != y) return x > y ? 1 : -1;
    }
    return 0;
}

/* 1 if the forward string is <= its reverse complement, judged on the first half only */
int orc_emf_compare_fw_rc(const uint8_t *seed, int len)
{
    const int half = (len + 1) / 2;
    return orc_emf_seedcmp(seed, 1, seed + (len - half), 0, half) <= 0 ? 1 : 0;
}

/* does the read continue to match beyond the seed_len bases at loc? (reads longer than the table's L) */
int orc_emf_match_further(const orc_emf_t *t, uint32_t loc, const uint8_t *seed, int is_rev, int len)
{
    const int L = t->seed_len;
    len -= L;
    if (!is_rev) {
        if (loc + (uint32_t)len >= t->seq_len) return 0;
        return memcmp(t->ref + loc + L, seed + L, (size_t)len) == 0;
    }
    if (loc < (uint32_t)len) return 0;
    for (int i = 0; i < len; ++i)
        if (t->ref[loc - len + i] != 3 - seed[L + len - 1 - i]) return 0;
    return 1;
}

static int match_further_all(const orc_emf_t *t, const orc_seed_entry_t *ent, const uint8_t *seed, int fw_less,
                             int len, uint32_t *flags, uint32_t *location)
{
    int is_rev = ((ent->flags & FLAG_FW_LESS) != 0) == (fw_less != 0) ? 0 : 1;
    uint32_t loc = NO_ENTRY;
    if (orc_emf_match_further(t, ent->location, seed, is_rev, len)) {
        loc = ent->location;
    } else {
        const uint32_t multi = ent->flags >> 2;
        if (multi) {
            const uint32_t *lt = t->loc_table;
            const int many = (lt[multi] & 0x80000000u) != 0;
            const uint32_t st = many ? (lt[multi] & 0x7fffffffu) : multi;
            uint32_t nfw, nrc;
            const uint32_t *lfw, *lrc;
            if (!many) { nfw = (lt[st] >> 16) & 0xffff; nrc = lt[st] & 0xffff; lfw = &lt[st + 1]; lrc = &lt[st + 1 + nfw]; }
            else { nfw = lt[st]; nrc = lt[st + 1]; lfw = &lt[st + 2]; lrc = &lt[st + 2 + nfw]; }
            uint32_t i;
            for (i = 0; i < nfw && loc == NO_ENTRY; ++i)
                if (orc_emf_match_further(t, lfw[i], seed, is_rev, len)) loc = lfw[i];
            if (loc == NO_ENTRY) {
                is_rev = !is_rev;
                for (i = 0; i < nrc && loc == NO_ENTRY; ++i)
                    if (orc_emf_match_further(t, lrc[i], seed, is_rev, len)) loc = lrc[i];
            }
        }
    }
    if (loc == NO_ENTRY) return 5;                      /* FIND_PERFECT_SEED_ONLY_MATCHED */
    *location = loc;
    if (!is_rev) { *flags = (ent->flags & ~FLAG_RC) | FLAG_VALID; return 3; }
    *flags = ent->flags | FLAG_RC | FLAG_VALID;
    return 4;
}

/* find_perfect_match_entry: code 0..5 (perfect.h:902-907); flags/location set on 3 and 4 */
int orc_emf_probe(const orc_emf_t *t, const uint8_t *seed, int len, uint32_t *flags, uint32_t *location)
{
    *flags = 0; *location = 0;
    if (!t || len < t->seed_len) return 0;              /* FIND_PERFECT_NO_TABLE */
    int n = 0;
    for (int i = 0; i < len; ++i) n |= seed[i] & 0xC;
    if (n) return 1;                                    /* FIND_PERFECT_WITH_N */
    const int L = t->seed_len;
    const int fw_less = orc_emf_compare_fw_rc(seed, L);
    uint32_t idx = (uint32_t)orc_emf_hash(t->num_seed_entry, seed, L, fw_less);
    const orc_seed_entry_t *ent = &t->seed_table[idx];
    if (ent->location == NO_ENTRY || (ent->flags & FLAG_COLLISION)) return 2;
    while (ent) {
        const int efl = (ent->flags & FLAG_FW_LESS) != 0;
        const int cmp = orc_emf_seedcmp(t->ref + ent->location, efl, seed, fw_less, L);
        if (cmp == 0) {
            if (len == L) {
                *location = ent->location;
                if (efl == fw_less) { *flags = (ent->flags & ~FLAG_RC) | FLAG_VALID; return 3; }
                *flags = ent->flags | FLAG_RC | FLAG_VALID;
                return 4;
            }
            return match_further_all(t, ent, seed, fw_less, len, flags, location);
        }
        idx = cmp > 0 ? ent->left : ent->right;
        ent = idx == NO_ENTRY ? NULL : &t->seed_table[idx];
    }
    return 2;                                           /* FIND_PERFECT_NOT_MATCHED */
}

/* ---- EMF result consumers: get_perfect_locations, perfect_dedup_patch, mem_perfect2reg
 *      (perfect_map.cpp:659-869): every location of an exactly matching read as a mem_alnreg_t.
 *      PARITY UNPINNED (perfect_map.cpp is not buildable here); the primitives it calls are pinned. ---- */
typedef struct { int64_t loc, pos; int rid, is_rev, is_alt; } aln_perfect_t;

static int emf_pos2rid(const orc_bns_t *b, int64_t pos_f)
{
    int left = 0, mid = 0, right = b->n_seqs;
    if (pos_f >= b->l_pac) return -1;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= b->contigs[mid].offset) {
            if (mid == b->n_seqs - 1) break;
            if (pos_f < b->contigs[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
static void init_aln(aln_perfect_t *a, int64_t pos, int len, int is_rev, const orc_bns_t *bns, int seed_len)
{
    a->loc = pos;
    a->rid = emf_pos2rid(bns, pos);
    if (len != seed_len && is_rev) pos = pos - (len - seed_len);
    a->pos = pos - bns->contigs[a->rid].offset;
    a->is_rev = is_rev;
    a->is_alt = bns->contigs[a->rid].is_alt != 0;
}
static int init_multi(aln_perfect_t *av, int n, uint32_t num_loc, const uint32_t *locs, const uint8_t *seq, int l_seq, int is_rev,
                      uint32_t matched_loc, const orc_bns_t *bns, const orc_emf_t *t)
{
    for (uint32_t i = 0; i < num_loc; ++i) {
        const uint32_t loc = locs[is_rev ? num_loc - 1 - i : i];        /* keeps the vector sorted by rb */
        if (loc == matched_loc) continue;
        if (t->seed_len == l_seq || orc_emf_match_further(t, loc, seq, is_rev, l_seq))
            init_aln(&av[n++], (int64_t)loc, l_seq, is_rev, bns, t->seed_len);
    }
    return n;
}
int orc_perfect2reg(const bwams_mem_opt_t *opt, const orc_emf_t *t, const orc_bns_t *bns, const uint8_t *seq, int l_seq,
                    uint32_t flags, uint32_t location, bwams_alnreg_t *out, int cap, int *first_is_rev)
{
    const int rc_matched = (flags & FLAG_RC) != 0;
    const uint32_t multi = flags >> 2;
    const uint32_t *lt = t->loc_table;
    int m = 1;
    if (multi) {
        const uint32_t first = lt[multi];
        m = (first & 0x80000000u) ? 1 + (int)lt[first & 0x7fffffffu] + (int)lt[(first & 0x7fffffffu) + 1]
                                  : 1 + (int)((first >> 16) & 0xffff) + (int)(first & 0xffff);
    }
    aln_perfect_t *av = (aln_perfect_t *)calloc((size_t)m, sizeof *av);
    int n = 0;
    if (!multi) init_aln(&av[n++], (int64_t)location, l_seq, rc_matched ? 1 : 0, bns, t->seed_len);
    else {
        const int many = (lt[multi] & 0x80000000u) != 0;
        const uint32_t st = many ? (lt[multi] & 0x7fffffffu) : multi;
        uint32_t nfw, nrc;
        const uint32_t *lfw, *lrc;
        if (!many) { nfw = (lt[st] >> 16) & 0xffff; nrc = lt[st] & 0xffff; lfw = &lt[st + 1]; lrc = &lt[st + 1 + nfw]; }
        else { nfw = lt[st]; nrc = lt[st + 1]; lfw = &lt[st + 2]; lrc = &lt[st + 2 + nfw]; }
        if (!rc_matched) {
            init_aln(&av[n++], (int64_t)location, l_seq, 0, bns, t->seed_len);
            n = init_multi(av, n, nfw, lfw, seq, l_seq, 0, location, bns, t);
            n = init_multi(av, n, nrc, lrc, seq, l_seq, 1, location, bns, t);
        } else {
            n = init_multi(av, n, nrc, lrc, seq, l_seq, 0, location, bns, t);
            init_aln(&av[n++], (int64_t)location, l_seq, 1, bns, t->seed_len);
            n = init_multi(av, n, nfw, lfw, seq, l_seq, 1, location, bns, t);
        }
    }
    /* perfect_dedup_patch (perfect_map.cpp:781-815) */
    if (n > 1) {
        int i, j, k;
        for (i = 1; i < n; ++i) {
            aln_perfect_t *p = &av[i];
            if (p->rid != av[i - 1].rid || p->is_rev != av[i - 1].is_rev || p->pos >= av[i - 1].pos + l_seq + opt->max_chain_gap) continue;
            for (j = i - 1; j >= 0 && p->rid == av[j].rid && p->is_rev == av[j].is_rev && p->pos < av[j].pos + l_seq + opt->max_chain_gap; --j) {
                aln_perfect_t *q = &av[j];
                if (q->rid < 0) continue;
                if (q->pos + l_seq - p->pos > opt->mask_level_redun * l_seq) q->rid = -1;
            }
        }
        for (i = 0, k = 0; i < n; ++i)
            if (av[i].rid >= 0) { if (k != i) av[k++] = av[i]; else ++k; }
        n = k;
    }
    if (first_is_rev) *first_is_rev = av[0].is_rev;
    if (n > cap) { free(av); return -1; }
    for (int i = 0; i < n; ++i) {                         /* mem_perfect2reg (perfect_map.cpp:817-867) */
        const aln_perfect_t *p = &av[i];
        bwams_alnreg_t *r = &out[i];
        memset(r, 0, sizeof *r);
        if (!p->is_rev) { r->rb = p->loc; r->re = p->loc + l_seq; }
        else { r->rb = (bns->l_pac << 1) - (p->loc + l_seq); r->re = (bns->l_pac << 1) - p->loc; }
        r->qb = 0; r->qe = l_seq;
        r->rid = p->rid;
        r->score = r->truesc = l_seq * opt->a;
        r->w = opt->w;
        r->seedlen0 = l_seq;
        r->n_comp_is_alt = 1 | (p->is_alt << 30);
    }
    free(av);
    return n;
}
