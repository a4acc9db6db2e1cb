/*
 * fmi_oracle.c — CPU restatement of FM-index seeding (TEST INFRASTRUCTURE ONLY,
 * see bwams_oracle.h for the rules and the pinning status: PARITY UNPINNED for
 * this file; validated from first principles in tests/test_oracle_fmi.py).
 *
 * Follows, function by function:
 *   GET_OCC                          /root/reference/src/FMI_search.h:76-83
 *   one_hot_mask_array               /root/reference/src/FMI_search.cpp:1253-1261
 *   FMI_search::backwardExt          /root/reference/src/FMI_search.cpp:2029-2056
 *   getSMEMsOnePosOneThread          /root/reference/src/FMI_search.cpp:1372-1606
 *   getSMEMsAllPosOneThread          /root/reference/src/FMI_search.cpp:1608-1660
 *   bwtSeedStrategyAllPosOneThread   /root/reference/src/FMI_search.cpp:1662-1816
 *   mem_collect_smem                 /root/reference/src/bwamem.cpp:648-786
 *   __build_all_smem_table           /root/reference/src/FMI_search.cpp:78-120
 *   __build_last_smem_table          /root/reference/src/FMI_search.cpp:155-194
 *   all_smem use in round 1/2        /root/reference/src/FMI_search.cpp:1414-1463
 *   last_smem use in round 3         /root/reference/src/FMI_search.cpp:1705-1750
 *   call_one_step                    /root/reference/src/FMI_search.cpp:2206-2259
 *   get_sa_entries_prefetch          /root/reference/src/FMI_search.cpp:2261-2379
 */
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

/* Mask with the top y bits set: the rows of a block that precede offset y. */
static inline uint64_t top_bits(int64_t y)
{
    return y == 0 ? 0ULL : (~0ULL) << (64 - y);
}

int64_t orc_fmi_occ(const orc_fmi_t *f, int64_t pos, int c)
{
    const bwams_cp_occ_t *b = &f->cp_occ[pos >> 6];
    uint64_t hit = b->one_hot_bwt_str[c] & top_bits(pos & 63);
    return b->cp_count[c] + __builtin_popcountll(hit);
}

void orc_backward_ext(const orc_fmi_t *f, const bwams_smem_t *in, int a,
                      bwams_smem_t *out, orc_counters_t *ctr)
{
    int64_t sp = in->k, ep = in->k + in->s;
    int64_t k[4], l[4], s[4];
    for (int b = 0; b < 4; ++b) {
        int64_t occ_sp = orc_fmi_occ(f, sp, b);
        int64_t occ_ep = orc_fmi_occ(f, ep, b);
        k[b] = f->count[b] + occ_sp;
        s[b] = occ_ep - occ_sp;
    }
    int64_t sentinel_offset =
        (in->k <= f->sentinel_index && in->k + in->s > f->sentinel_index) ? 1 : 0;
    l[3] = in->l + sentinel_offset;
    l[2] = l[3] + s[3];
    l[1] = l[2] + s[2];
    l[0] = l[1] + s[1];
    *out = *in;
    out->k = k[a];
    out->l = l[a];
    out->s = s[a];
    if (ctr) {
        ctr->n_ext += 1;
        ctr->n_ext_blocks += ((sp >> 6) == (ep >> 6)) ? 1 : 2;
    }
}

/* Forward extension by base a: swap strands, extend backward by the
 * complement, swap back (src/FMI_search.cpp:1475-1485). */
static void forward_ext(const orc_fmi_t *f, const bwams_smem_t *in, int a,
                        bwams_smem_t *out, orc_counters_t *ctr)
{
    bwams_smem_t t = *in, r;
    t.k = in->l;
    t.l = in->k;
    orc_backward_ext(f, &t, 3 - a, &r, ctr);
    *out = r;
    out->k = r.l;
    out->l = r.k;
}

void orc_forward_ext(const orc_fmi_t *f, const bwams_smem_t *in, int a, bwams_smem_t *out)
{
    forward_ext(f, in, a, out, NULL);
}

/* all_smem_t / last_smem_t exactly as the reference packs them (src/FMI_search.h:108-133) */
typedef struct __attribute__((packed)) {
    uint32_t last_avail;
    struct { uint32_t k32, l32, s32; } list[10];
    uint8_t pad_[4];
} all_smem_rec;
typedef struct __attribute__((packed)) {
    uint8_t bp;
    int8_t kms, lms, sms;
    uint32_t kls, lls, sls;
} last_smem_rec;
_Static_assert(sizeof(all_smem_rec) == 128 && sizeof(last_smem_rec) == 16, "FMA record sizes");

void orc_build_all_smem(const orc_fmi_t *f, int bp, void *table)
{
    all_smem_rec *t = (all_smem_rec *)table;
    const int64_t n = (int64_t)1 << (2 * bp);
    for (int64_t idx = 0; idx < n; ++idx) {
        all_smem_rec *ent = &t[idx];
        memset(ent, 0, sizeof *ent);
        int a = (int)((idx >> (2 * (bp - 1))) & 3);
        bwams_smem_t sm;
        memset(&sm, 0, sizeof sm);
        sm.k = f->count[a]; sm.l = f->count[3 - a]; sm.s = f->count[a + 1] - f->count[a];
        for (int i = 1; i < bp; ++i) {
            a = (int)((idx >> (2 * (bp - 1 - i))) & 3);
            bwams_smem_t nw;
            forward_ext(f, &sm, a, &nw, NULL);
            ent->list[i - 1].l32 = (uint32_t)(nw.l - f->count[3 - a]);
            ent->list[i - 1].k32 = (uint32_t)(nw.k - sm.k);
            ent->list[i - 1].s32 = (uint32_t)nw.s;
            if (nw.s > 0) ent->last_avail = (uint32_t)i; else break;
            sm = nw;
        }
    }
}

void orc_build_last_smem(const orc_fmi_t *f, int bp, void *table)
{
    last_smem_rec *t = (last_smem_rec *)table;
    const int64_t n = (int64_t)1 << (2 * bp);
    for (int64_t idx = 0; idx < n; ++idx) {
        int a = (int)((idx >> (2 * (bp - 1))) & 3);
        bwams_smem_t sm;
        memset(&sm, 0, sizeof sm);
        sm.k = f->count[a]; sm.l = f->count[3 - a]; sm.s = f->count[a + 1] - f->count[a];
        int i;
        for (i = 1; i < bp; ++i) {
            a = (int)((idx >> (2 * (bp - 1 - i))) & 3);
            bwams_smem_t nw;
            forward_ext(f, &sm, a, &nw, NULL);
            if (nw.s == 0) break;
            sm = nw;
        }
        last_smem_rec *ent = &t[idx];
        ent->bp = (uint8_t)i;
        ent->kms = (int8_t)(sm.k >> 32); ent->kls = (uint32_t)(sm.k & 0xffffffff);
        ent->lms = (int8_t)(sm.l >> 32); ent->lls = (uint32_t)(sm.l & 0xffffffff);
        ent->sms = (int8_t)(sm.s >> 32); ent->sls = (uint32_t)(sm.s & 0xffffffff);
    }
}

int64_t orc_smem_one_pos(const orc_fmi_t *f, const uint8_t *enc_qdb,
                         int16_t *query_pos, const int32_t *min_intv,
                         const int32_t *rid, int32_t num_reads,
                         const int64_t *cum_len, int32_t min_seed_len,
                         bwams_smem_t *out, orc_counters_t *ctr)
{
    int64_t n_out = 0;
    int max_len = 0;
    for (int32_t i = 0; i < num_reads; ++i) {
        int len = (int)(cum_len[rid[i] + 1] - cum_len[rid[i]]);
        if (len > max_len) max_len = len;
    }
    bwams_smem_t *prev = (bwams_smem_t *)malloc(sizeof(bwams_smem_t) * (size_t)(max_len + 1));

    for (int32_t i = 0; i < num_reads; ++i) {
        int x = query_pos[i];
        int32_t r = rid[i];
        int next_x = x + 1;
        int readlength = (int)(cum_len[r + 1] - cum_len[r]);
        const uint8_t *q = enc_qdb + cum_len[r];
        uint8_t a = q[x];

        if (a < 4) {
            bwams_smem_t cur;
            memset(&cur, 0, sizeof cur);
            cur.rid = (uint32_t)r;
            cur.m = (uint32_t)x;
            cur.n = (uint32_t)x;
            cur.k = f->count[a];
            cur.l = f->count[3 - a];
            cur.s = f->count[a + 1] - f->count[a];
            int num_prev = 0;
            int j;

            /* FMA: the first forward steps come from the all_smem table (FMI_search.cpp:1414-1463) */
            j = x + 1;
            if (f->all_smem && readlength - x >= f->all_bp) {
                const int bp = f->all_bp;
                uint64_t idx = 0;
                int kk;
                for (kk = 0; kk < bp; ++kk) {
                    if (q[x + kk] >= 4) break;
                    idx |= (uint64_t)q[x + kk] << ((bp - 1 - kk) * 2);
                }
                const all_smem_rec *ent = &((const all_smem_rec *)f->all_smem)[idx];
                const int with_n = kk < bp;
                const int last_idx = (kk > (int)ent->last_avail ? (int)ent->last_avail : kk) - 1;
                int t;
                for (t = 0; t < last_idx; ++j, ++t) {
                    a = q[j];
                    next_x = j + 1;
                    bwams_smem_t nw = cur;
                    nw.k = cur.k + ent->list[t].k32;
                    nw.l = f->count[3 - a] + ent->list[t].l32;
                    nw.s = ent->list[t].s32;
                    nw.n = (uint32_t)j;
                    prev[num_prev] = cur;
                    if (nw.s != cur.s) num_prev++;
                    if (nw.s < min_intv[i]) {
                        next_x = j;
                        j = readlength;       /* skips the loop below */
                        break;
                    }
                    cur = nw;
                }
                if (with_n) {
                    next_x = j + 1;           /* reference quirk: also after the early exit above */
                    j = readlength;
                }
            }
            /* forward phase: collect the intervals whose size changes */
            for (; j < readlength; ++j) {
                a = q[j];
                next_x = j + 1;
                if (a >= 4) break;
                bwams_smem_t nw;
                forward_ext(f, &cur, a, &nw, ctr);
                nw.n = (uint32_t)j;
                prev[num_prev] = cur;
                if (nw.s != cur.s) num_prev++;
                if (nw.s < min_intv[i]) {
                    next_x = j;
                    break;
                }
                cur = nw;
            }
            if (cur.s >= min_intv[i]) prev[num_prev++] = cur;

            /* longest match first */
            for (int p = 0; p < num_prev / 2; ++p) {
                bwams_smem_t t = prev[p];
                prev[p] = prev[num_prev - p - 1];
                prev[num_prev - p - 1] = t;
            }

            /* backward phase */
            for (j = x - 1; j >= 0; --j) {
                int num_curr = 0;
                int32_t curr_s = -1;      /* 32-bit in the reference */
                a = q[j];
                if (a > 3) break;
                int p;
                for (p = 0; p < num_prev; ++p) {
                    bwams_smem_t sm = prev[p], nw;
                    orc_backward_ext(f, &sm, a, &nw, ctr);
                    nw.m = (uint32_t)j;
                    if (nw.s < min_intv[i] && (sm.n - sm.m + 1) >= (uint32_t)min_seed_len) {
                        out[n_out++] = sm;
                        break;
                    }
                    if (nw.s >= min_intv[i] && nw.s != curr_s) {
                        curr_s = (int32_t)nw.s;
                        prev[num_curr++] = nw;
                        break;
                    }
                }
                p++;
                for (; p < num_prev; ++p) {
                    bwams_smem_t sm = prev[p], nw;
                    orc_backward_ext(f, &sm, a, &nw, ctr);
                    nw.m = (uint32_t)j;
                    if (nw.s >= min_intv[i] && nw.s != curr_s) {
                        curr_s = (int32_t)nw.s;
                        prev[num_curr++] = nw;
                    }
                }
                num_prev = num_curr;
                if (num_curr == 0) break;
            }
            if (num_prev != 0) {
                bwams_smem_t sm = prev[0];
                if ((sm.n - sm.m + 1) >= (uint32_t)min_seed_len) out[n_out++] = sm;
            }
        }
        query_pos[i] = (int16_t)next_x;
    }
    free(prev);
    return n_out;
}

int64_t orc_smem_all_pos(const orc_fmi_t *f, const uint8_t *enc_qdb,
                         int32_t *min_intv, int32_t *rid, int32_t num_reads,
                         const int64_t *cum_len, int32_t min_seed_len,
                         bwams_smem_t *out, orc_counters_t *ctr)
{
    int16_t *query_pos = (int16_t *)calloc((size_t)(num_reads > 0 ? num_reads : 1), sizeof(int16_t));
    int32_t num_active = num_reads;
    int64_t total = 0;
    do {
        int32_t tail = 0;
        for (int32_t head = 0; head < num_active; ++head) {
            int readlength = (int)(cum_len[rid[head] + 1] - cum_len[rid[head]]);
            if (query_pos[head] < readlength) {
                rid[tail] = rid[head];
                query_pos[tail] = query_pos[head];
                min_intv[tail] = min_intv[head];
                tail++;
            }
        }
        total += orc_smem_one_pos(f, enc_qdb, query_pos, min_intv, rid, tail,
                                  cum_len, min_seed_len, out + total, ctr);
        num_active = tail;
    } while (num_active > 0);
    free(query_pos);
    return total;
}

int64_t orc_seed_strategy(const orc_fmi_t *f, const uint8_t *enc_qdb,
                          const uint8_t *skip, int32_t max_intv,
                          int32_t num_reads, const int64_t *cum_len,
                          int32_t min_seed_len, bwams_smem_t *out,
                          orc_counters_t *ctr)
{
    int64_t n_out = 0;
    for (int32_t i = 0; i < num_reads; ++i) {
        if (skip && skip[i]) continue;
        int readlength = (int)(cum_len[i + 1] - cum_len[i]);
        const uint8_t *q = enc_qdb + cum_len[i];
        int16_t x = 0;
        while (x < readlength) {
            int next_x = x + 1;
            uint8_t a = q[x];
            if (a < 4) {
                bwams_smem_t cur;
                memset(&cur, 0, sizeof cur);
                cur.rid = (uint32_t)i;
                cur.m = (uint32_t)x;
                cur.n = (uint32_t)x;
                cur.k = f->count[a];
                cur.l = f->count[3 - a];
                cur.s = f->count[a + 1] - f->count[a];
                int j = x + 1;
                /* FMA: jump over the longest non-empty prefix of the next last_bp bases
                 * (FMI_search.cpp:1705-1750); an emitted seed does NOT end the pivot here */
                if (f->last_smem && readlength - x >= f->last_bp) {
                    const int bp = f->last_bp;
                    uint64_t idx = 0;
                    int with_n = 0;
                    for (int kk = 0; kk < bp; ++kk) {
                        idx |= (uint64_t)q[x + kk] << ((bp - 1 - kk) * 2);
                        with_n += q[x + kk] >> 2;
                    }
                    if (with_n == 0) {
                        const last_smem_rec *ent = &((const last_smem_rec *)f->last_smem)[idx];
                        j = x + ent->bp;
                        next_x = j;
                        cur.k = ((int64_t)ent->kms << 32) | (int64_t)ent->kls;
                        cur.l = ((int64_t)ent->lms << 32) | (int64_t)ent->lls;
                        cur.s = ((int64_t)ent->sms << 32) | (int64_t)ent->sls;
                        cur.n = (uint32_t)(j - 1);
                        if (cur.s < max_intv && (cur.n - cur.m + 1) >= (uint32_t)min_seed_len) {
                            if (cur.s > 0) out[n_out++] = cur;
                        }
                    }
                }
                for (; j < readlength; ++j) {
                    next_x = j + 1;
                    a = q[j];
                    if (a >= 4) break;
                    bwams_smem_t nw;
                    forward_ext(f, &cur, a, &nw, ctr);
                    nw.n = (uint32_t)j;
                    cur = nw;
                    if (cur.s < max_intv && (cur.n - cur.m + 1) >= (uint32_t)min_seed_len) {
                        if (cur.s > 0) out[n_out++] = cur;
                        break;
                    }
                }
            }
            x = (int16_t)next_x;
        }
    }
    return n_out;
}

static int cmp_smem(const void *pa, const void *pb)
{
    const bwams_smem_t *a = (const bwams_smem_t *)pa, *b = (const bwams_smem_t *)pb;
    if (a->rid != b->rid) return a->rid < b->rid ? -1 : 1;
    if (a->m != b->m) return a->m < b->m ? -1 : 1;
    if (a->n != b->n) return a->n < b->n ? -1 : 1;
    return 0;
}

int64_t orc_collect_smem(const orc_fmi_t *f, const bwams_seed_opt_t *opt,
                         const uint8_t *enc_qdb, const int64_t *cum_len,
                         const uint8_t *skip, int32_t nseq,
                         bwams_smem_t *out, int64_t cap, orc_counters_t *ctr)
{
    /* Work in a private, generously sized array (the reference pre-allocates
     * BATCH_MUL * readLen slots per read, fastmap.cpp:288-289), then copy. */
    int64_t total_bases = cum_len[nseq] - cum_len[0];
    int64_t wcap = 3 * total_bases + 64;
    bwams_smem_t *w = (bwams_smem_t *)malloc(sizeof(bwams_smem_t) * (size_t)wcap);
    int32_t *min_intv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(wcap));
    int32_t *rid = (int32_t *)malloc(sizeof(int32_t) * (size_t)(wcap));
    int16_t *qpos = (int16_t *)malloc(sizeof(int16_t) * (size_t)(wcap));
    int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);

    /* round 1: every pivot of every non-filtered read, min_intv = 1 */
    int32_t n_active = 0;
    for (int32_t l = 0; l < nseq; ++l) {
        if (skip && skip[l]) continue;
        min_intv[n_active] = 1;
        rid[n_active] = l;
        n_active++;
    }
    int64_t n1 = orc_smem_all_pos(f, enc_qdb, min_intv, rid, n_active, cum_len,
                                  opt->min_seed_len, w, ctr);

    /* round 2: re-seed long, low-occurrence SMEMs at their midpoint */
    int64_t pos = 0;
    for (int64_t i = 0; i < n1; ++i) {
        const bwams_smem_t *p = &w[i];
        int start = (int)p->m, end = (int)p->n + 1;
        if (end - start < split_len || p->s > opt->split_width) continue;
        rid[pos] = (int32_t)p->rid;
        qpos[pos] = (int16_t)((end + start) >> 1);
        min_intv[pos] = (int32_t)(p->s + 1);
        pos++;
    }
    int64_t n2 = orc_smem_one_pos(f, enc_qdb, qpos, min_intv, rid, (int32_t)pos,
                                  cum_len, opt->min_seed_len, w + n1, ctr);

    /* round 3: forward-only seeds bounded by max_mem_intv */
    int64_t n3 = 0;
    if (opt->max_mem_intv > 0)
        n3 = orc_seed_strategy(f, enc_qdb, skip, opt->max_mem_intv, nseq, cum_len,
                               opt->min_seed_len + 1, w + n1 + n2, ctr);
    int64_t tot = n1 + n2 + n3;
    if (ctr) {
        ctr->n_smem[0] += n1;
        ctr->n_smem[1] += n2;
        ctr->n_smem[2] += n3;
    }

    /* sortSMEMs (rid, m, n desc) followed by the per-read introsort on
     * (m << 32 | n): final order is (rid, m, n) ascending.  An SMEM's (k,l,s)
     * is a function of (rid, m, n), so ties are identical records. */
    qsort(w, (size_t)tot, sizeof(bwams_smem_t), cmp_smem);

    int64_t ret = tot;
    if (tot > cap) ret = -1;
    else memcpy(out, w, sizeof(bwams_smem_t) * (size_t)tot);
    free(w); free(min_intv); free(rid); free(qpos);
    return ret;
}

static inline int bwt_char(const orc_fmi_t *f, int64_t pos)
{
    const bwams_cp_occ_t *b = &f->cp_occ[pos >> 6];
    int sh = 63 - (int)(pos & 63);
    for (int c = 0; c < 4; ++c)
        if ((b->one_hot_bwt_str[c] >> sh) & 1) return c;
    return 4;
}

int64_t orc_sa_entry(const orc_fmi_t *f, int64_t pos, orc_counters_t *ctr)
{
    int64_t sp = pos, offset = 0;
    for (;;) {
        if ((sp & 7) == 0) {
            int64_t v = (int64_t)f->sa_ms_byte[sp >> 3];
            v = (v << 32) + (int64_t)f->sa_ls_word[sp >> 3];
            return v + offset;
        }
        int b = bwt_char(f, sp);
        if (b == 4) return 0;      /* sentinel row: the reference drops the offset */
        sp = f->count[b] + orc_fmi_occ(f, sp, b);
        offset++;
        if (ctr) ctr->n_lf_steps++;
    }
}

int64_t orc_sa_lookup(const orc_fmi_t *f, const bwams_smem_t *smem, int64_t n,
                      int32_t max_occ, int64_t *coord, int64_t cap,
                      int64_t *sa_off, orc_counters_t *ctr)
{
    int64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) {
        const bwams_smem_t *p = &smem[i];
        int64_t hi = p->k + p->s;
        int64_t step = p->s > max_occ ? p->s / max_occ : 1;
        int32_t c = 0;
        sa_off[i] = tot;
        for (int64_t j = p->k; j < hi && c < max_occ; j += step, ++c) {
            if (tot + c >= cap) return -1;
            coord[tot + c] = orc_sa_entry(f, j, ctr);
        }
        tot += c;
    }
    sa_off[n] = tot;
    if (ctr) ctr->n_sa_lookups += tot;
    return tot;
}
