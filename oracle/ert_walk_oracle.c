/*
 * ert_walk_oracle.c — CPU restatement of the reference's OWN ERT seeding walk, function by function
 * (TEST INFRASTRUCTURE ONLY, see bwams_oracle.h).  Where ert_oracle.c states WHAT ERT-mode seeding has to
 * deliver (the match-profile formulation, pinned to FM-index seeding), this file restates HOW
 * src/ertseeding.cpp gets there: the LEP-driven forward / backward search order, the five tree-walk variants,
 * lazy leaf expansion, the re-gathering of a backward-found MEM's hits by a forward traversal
 * ("to report hits in the same order as BWA-MEM", ertseeding.cpp:644-648, :2058-2062), `forward` /
 * `fetch_leaves` / `end_correction` per MEM, and the hit array in the order the walk pushes it — i.e. exactly
 * what mem_kernel1_core_ert hands to mem_chain_new (bwamem.cpp:1122-1193, consumer :993-1006).
 *
 * PARITY UNPINNED in the contract's sense: src/ertseeding.cpp includes the un-vendored safestringlib and cannot
 * be compiled here, and the reference ships no ERT fixtures.  What this file adds is a second, independent
 * statement of the cited functions' observable output; tests/test_oracle_ert_walk.py compares it with the
 * profile formulation (orc_ert_collect) and with FM-index seeding on seeds with 1 < hits <= max_occ and
 * hits > max_occ, and tests/test_gpu_ert.py compares the HIP path with it directly.
 *
 * Restated, in file order (all line numbers: /root/reference/src/ertseeding.cpp):
 *   getHashKey :435   get_seq :455   getOffsetToChildNode :485   getOffsetToLeafData :507 (code_table / leaf_table
 *   :24-421 are the number of bytes in front of child c's pointer / leaf record; computed here from the code byte)
 *   getNextByteIdx_dfs :521   leaf_gather :589   getNextByteIdx_backward :609   getNextByteIdx_backward_wlimit :718
 *   getNextByteIdx :836   getNextByteIdx_wlimit :989   getNextByteIdx_last :1168   leftExtend :1288
 *   leftExtend_wlimit :1392   getNextByteIdx_fetch_leaves_prefix_reseed :1505   getNextByteIdx_fetch_leaves_prefix :1644
 *   getNextByteIdx_fetch_leaves :1759   rightExtend_fetch_leaves_prefix_reseed :1854   rightExtend_fetch_leaves_prefix :1967
 *   rightExtend_fetch_leaves :2071   rightExtend :2142   rightExtend_wlimit :2317   rightExtend_last :2500
 *   init_mem :2609   check_and_add_smem_prefix_reseed :2639   check_and_add_smem_prefix :2756   check_and_add_smem :2867
 *   get_seeds_prefix :2925   get_seeds :3062   reseed_prefix :3200   reseed :3315   last :3425
 * and the per-read driver mem_kernel1_core_ert (/root/reference/src/bwamem.cpp:1122-1191).
 *
 * kmerSize / xmerSize are macros in the reference (macro.h:204-206); they are fields of orc_ert_t here, as in
 * ert_oracle.c.  Places where the reference would touch memory it does not own or trip one of its asserts set
 * a flag bit (ORC_ERTW_*) instead; the tests require the flags to stay clear on genome-like data.
 */
#include <stdlib.h>
#include <string.h>
#include "bwams_oracle.h"

enum { N_EMPTY = 0, N_LEAF = 1, N_UNIFORM = 2, N_DIVERGE = 3 };
enum { E_INVALID = 0, E_SINGLE = 1, E_INFREQUENT = 2, E_FREQUENT = 3 };

typedef bwams_ert_mem_t mem_t;
typedef struct { uint64_t *a; int64_t n, m; } hv_t;
typedef struct { mem_t *a; int64_t n, m; } mv_t;
typedef struct { uint64_t byte_idx; int num_hits; } ninfo_t;
typedef struct { ninfo_t a[1024]; int n; } path_t;

typedef struct {
    const orc_ert_t *e;
    int K, X;
    int64_t l_pac;
    int min_seed_len, l_seq, ptr_width, num_hits, limit;
    uint64_t lep[5];
    uint64_t nextLEPBit;
    uint64_t mlt_start_addr, mh_start_addr;
    const uint8_t *fw, *rc, *read_buf;
    int flags;
} raux_t;

typedef struct {
    int prevMemStart, prevMemEnd, curr_pivot, prev_pivot, prev_prev_pivot, stop_be, mem_end_limit;
} sh_t;

static void hv_push(hv_t *v, uint64_t x)
{
    if (v->n == v->m) {
        v->m = v->m ? v->m << 1 : 256;
        v->a = (uint64_t *)realloc(v->a, (size_t)v->m * sizeof(uint64_t));
    }
    v->a[v->n++] = x;
}

static void mv_push(mv_t *v, const mem_t *x)
{
    if (v->n == v->m) {
        v->m = v->m ? v->m << 1 : 32;
        v->a = (mem_t *)realloc(v->a, (size_t)v->m * sizeof(mem_t));
    }
    v->a[v->n++] = *x;
}

static inline uint64_t rd(const uint8_t *p, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

static inline void lep_set(raux_t *r, uint64_t bit)
{
    if (bit >= 320) { r->flags |= ORC_ERTW_LEP_RANGE; return; }
    r->lep[bit >> 6] |= 1ULL << (bit & 63);
}

static void path_push(raux_t *r, path_t *p, uint64_t byte_idx, int num_hits)
{
    if (p->n >= 1024) { r->flags |= ORC_ERTW_STACK; return; }
    p->a[p->n].byte_idx = byte_idx;
    p->a[p->n].num_hits = num_hits;
    p->n++;
}

static uint64_t path_pop(raux_t *r, path_t *p)
{
    if (p->n <= 0) { r->flags |= ORC_ERTW_STACK; return 0; }
    return p->a[--p->n].byte_idx;
}

/* :435 — key of the K-mer at str, LSB first; stops at an ambiguous base */
static uint32_t hash_key(const uint8_t *str, int keysize, int index, int seq_len, int *end_flag, int *idx_first_N)
{
    uint32_t key = 0;
    int len = keysize;
    if (index + keysize > seq_len) {
        if (end_flag) *end_flag = 1;        /* the reference dereferences a null end_flag here when one is passed as 0 */
        len = seq_len - index;
    }
    for (int i = 0; i < len; ++i) {
        if (str[i] != 4) key |= (uint32_t)str[i] << (i << 1);
        else { *idx_first_N = i; break; }
    }
    return key;
}

/* :455 — window [beg, end) of the fw‖rc text; nothing when it bridges the strands */
static const uint8_t *get_seq(const raux_t *r, int64_t beg, int64_t end, int64_t *len)
{
    const int64_t l_pac = r->l_pac;
    if (end < beg) { int64_t t = beg; beg = end; end = t; }
    if (end > (l_pac << 1)) end = l_pac << 1;
    if (beg < 0) beg = 0;
    if (beg >= l_pac || end <= l_pac) {
        *len = end - beg;
        return r->e->ref + beg;
    }
    *len = 0;
    return NULL;
}

static inline int cnt_above(uint8_t code, int c, int type)
{
    int n = 0;
    for (int cc = c + 1; cc < 4; ++cc) n += ((code >> (cc << 1)) & 3) == type;
    return n;
}

/* :485 — byteIdx points just behind the code byte; jumps to child c's node */
static void to_child(raux_t *r, const uint8_t *mlt_data, uint8_t code, int c, uint64_t *byteIdx)
{
    uint64_t next = *byteIdx;
    const uint64_t start = next - 1;
    next += (uint64_t)(cnt_above(code, c, N_DIVERGE) * r->ptr_width);
    const uint32_t v = (uint32_t)rd(mlt_data + next, r->ptr_width);
    r->num_hits = (int)(v & 0x3F);
    *byteIdx = start + (v >> 6);
}

/* :507 */
static inline int to_leaf(const raux_t *r, uint8_t code, int c)
{
    return (cnt_above(code, -1, N_DIVERGE)) * r->ptr_width + 5 * cnt_above(code, c, N_LEAF);
}

/* the leaf record of child c: pushes its hit(s); returns 1 for a multi-hit record */
static int leaf_hits(raux_t *r, const uint8_t *mlt_data, uint64_t at, mem_t *mem, hv_t *hits)
{
    const uint64_t leaf_data = rd(mlt_data + at, 5);
    if (leaf_data & 1) {
        uint64_t next = r->mh_start_addr + (leaf_data >> 1);
        r->num_hits = (int)rd(mlt_data + next, 2);
        next += 2;
        mem->hitcount += r->num_hits;
        for (int k = 0; k < r->num_hits; ++k) {
            hv_push(hits, rd(mlt_data + next, 5) >> 1);
            next += 5;
        }
        return 1;
    }
    r->num_hits = 1;
    mem->hitcount += 1;
    hv_push(hits, leaf_data >> 1);
    return 0;
}

/* the run of a UNIFORM node at *next (behind the code byte): returns the number of run bases the read matches from
 * position i and leaves *next at the child node; *count = the run's length */
static int uniform_match(const raux_t *r, const uint8_t *mlt_data, uint64_t *next, int i, int *count)
{
    const int countBP = mlt_data[(*next)++];
    const int numBytes = (countBP * 2 + 7) / 8;
    const uint8_t *packed = mlt_data + *next;
    *next += (uint64_t)numBytes;
    int j;
    for (j = 0; j < countBP; ++j) {
        if (i + j >= r->l_seq) break;
        if (r->read_buf[i + j] == 4) break;
        const int bp = (packed[j >> 2] >> ((~j & 3) << 1)) & 3;
        if (3 - r->read_buf[i + j] != bp) break;
    }
    *count = countBP;
    return j;
}

/* :521 */
static void gnb_dfs(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, mem_t *mem, int bc, hv_t *hits)
{
    uint64_t next = *byte_idx;
    const int c = 3 - bc;
    mem->skip_ref_fetch = 1;
    const uint8_t code = mlt_data[next++];
    const int code_c = (code >> (c << 1)) & 3;
    if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        const uint64_t leaf_data = rd(mlt_data + next, 5);
        if (leaf_data & 1) {
            next = r->mh_start_addr + (leaf_data >> 1);
            r->num_hits = (int)rd(mlt_data + next, 2);
            next += 2;
            mem->hitcount += r->num_hits;
            for (int k = 0; k < r->num_hits; ++k) {
                hv_push(hits, rd(mlt_data + next, 5) >> 1);
                next += 5;
            }
        } else {
            r->num_hits = 1;
            mem->hitcount += 1;
            hv_push(hits, leaf_data >> 1);
        }
    } else if (code_c == N_UNIFORM) {
        const int countBP = mlt_data[next++];
        next += (uint64_t)((countBP * 2 + 7) / 8);
        uint64_t start = next;
        for (int k = 0; k < 4; ++k) {
            gnb_dfs(r, mlt_data, &start, mem, k, hits);
            start = next;
        }
    } else if (code_c == N_DIVERGE) {
        to_child(r, mlt_data, code, c, &next);
        uint64_t start = next;
        for (int k = 0; k < 4; ++k) {
            gnb_dfs(r, mlt_data, &start, mem, k, hits);
            start = next;
        }
    }
    *byte_idx = next;
}

/* :589 */
static void leaf_gather(raux_t *r, const uint8_t *mlt_data, const uint64_t *byte_idx, mem_t *mem, hv_t *hits)
{
    const uint64_t start = *byte_idx;
    uint64_t tmp = start;
    for (int k = 0; k < 4; ++k) {
        gnb_dfs(r, mlt_data, &tmp, mem, k, hits);
        tmp = start;
    }
}

/* :609 */
static void gnb_backward(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int *i, mem_t *mem, hv_t *hits)
{
    uint64_t next = *byte_idx;
    int c = 0, code_c;
    uint8_t code = 0;
    if (r->read_buf[*i] != 4) {
        c = 3 - r->read_buf[*i];
        code = mlt_data[next++];
        code_c = (code >> (c << 1)) & 3;
        if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    } else code_c = N_EMPTY;
    if (code_c == N_EMPTY) {
        mem->rc_end = *i;
        mem->fetch_leaves = 1;
    } else if (code_c == N_LEAF) {
        *i += 1;
        mem->rc_end = *i;
        next += (uint64_t)to_leaf(r, code, c);
        if (leaf_hits(r, mlt_data, next, mem, hits)) mem->fetch_leaves = 1;   /* re-fetched later in forward order */
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, *i, &countBP);
        *i += j;
        if (j == countBP) {
            if (*i < r->l_seq) gnb_backward(r, mlt_data, &next, i, mem, hits);
            else mem->rc_end = *i;
        } else {
            mem->rc_end = *i;
            mem->fetch_leaves = 1;
        }
    } else {
        to_child(r, mlt_data, code, c, &next);
        *i += 1;
        if (*i < r->l_seq) gnb_backward(r, mlt_data, &next, i, mem, hits);
        else mem->rc_end = *i;
    }
    *byte_idx = next;
}

/* :718 */
static void gnb_backward_wlimit(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int *i, mem_t *mem, hv_t *hits)
{
    uint64_t next = *byte_idx;
    int c = 0, code_c;
    uint8_t code = 0;
    if (r->read_buf[*i] != 4) {
        c = 3 - r->read_buf[*i];
        code = mlt_data[next++];
        code_c = (code >> (c << 1)) & 3;
        if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    } else code_c = N_EMPTY;
    if (code_c == N_EMPTY) {
        mem->rc_end = *i;
        mem->fetch_leaves = 1;
    } else if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        const uint64_t leaf_data = rd(mlt_data + next, 5);
        if (leaf_data & 1) {
            next = r->mh_start_addr + (leaf_data >> 1);
            r->num_hits = (int)rd(mlt_data + next, 2);
            next += 2;
            if (r->num_hits >= r->limit) {
                mem->hitcount += r->num_hits;
                for (int k = 0; k < r->num_hits; ++k) {
                    hv_push(hits, rd(mlt_data + next, 5) >> 1);
                    next += 5;
                }
                *i += 1;
            }
        }
        mem->fetch_leaves = 1;
        mem->rc_end = *i;
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, *i, &countBP);
        *i += j;
        if (j == countBP) {
            if (*i < r->l_seq) gnb_backward_wlimit(r, mlt_data, &next, i, mem, hits);
            else { mem->rc_end = *i; mem->fetch_leaves = 1; }
        } else {
            mem->rc_end = *i;
            mem->fetch_leaves = 1;
        }
    } else {
        r->num_hits = 0;
        to_child(r, mlt_data, code, c, &next);
        if (r->num_hits == 0 || r->num_hits >= r->limit) {
            *i += 1;
            if (*i < r->l_seq) gnb_backward_wlimit(r, mlt_data, &next, i, mem, hits);
            else { mem->rc_end = *i; mem->fetch_leaves = 1; }
        } else {
            mem->rc_end = *i;
            mem->fetch_leaves = 1;
        }
    }
    *byte_idx = next;
}

/* :836 */
static void gnb(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int *i, mem_t *mem, hv_t *hits)
{
    uint64_t next = *byte_idx;
    const uint64_t parent = next;
    int c = 0, code_c;
    uint8_t code = 0;
    if (r->read_buf[*i] != 4) {
        c = 3 - r->read_buf[*i];
        code = mlt_data[next++];
        code_c = (code >> (c << 1)) & 3;
        if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    } else code_c = N_EMPTY;
    if (code_c == N_EMPTY) {
        if (mem->start == 0) {
            if (*i >= r->min_seed_len) leaf_gather(r, mlt_data, &parent, mem, hits);
        }
        lep_set(r, r->nextLEPBit);
        r->nextLEPBit += 1;
    } else if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        leaf_hits(r, mlt_data, next, mem, hits);
        lep_set(r, r->nextLEPBit);
        r->nextLEPBit += 1;
        *i += 1;
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, *i, &countBP);
        r->nextLEPBit += (uint64_t)j;
        *i += j;
        if (j == countBP) {
            if (*i == r->l_seq) {
                if (mem->start == 0) leaf_gather(r, mlt_data, &next, mem, hits);
                lep_set(r, r->nextLEPBit);
            } else if (*i < r->l_seq) {
                gnb(r, mlt_data, &next, i, mem, hits);
            }
        } else {
            if (mem->start == 0) {
                if (*i >= r->min_seed_len) leaf_gather(r, mlt_data, &next, mem, hits);
            }
            lep_set(r, r->nextLEPBit);
        }
    } else {
        to_child(r, mlt_data, code, c, &next);
        lep_set(r, r->nextLEPBit);
        r->nextLEPBit += 1;
        *i += 1;
        if (*i < r->l_seq) {
            gnb(r, mlt_data, &next, i, mem, hits);
        } else {
            if (mem->start == 0) leaf_gather(r, mlt_data, &next, mem, hits);
            lep_set(r, r->nextLEPBit);
            r->nextLEPBit += 1;
        }
    }
    *byte_idx = next;
}

/* :989 */
static void gnb_wlimit(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int *i, mem_t *mem, path_t *visited, hv_t *hits)
{
    uint64_t next = *byte_idx;
    const uint64_t parent = next;
    int c = 0, code_c;
    uint8_t code = 0;
    if (r->read_buf[*i] != 4) {
        c = 3 - r->read_buf[*i];
        code = mlt_data[next++];
        code_c = (code >> (c << 1)) & 3;
        if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    } else code_c = N_EMPTY;
    if (code_c == N_EMPTY) {
        if (mem->start == 0) {
            if (*i >= r->min_seed_len) leaf_gather(r, mlt_data, &parent, mem, hits);
        }
        lep_set(r, r->nextLEPBit);
        r->nextLEPBit += 1;
    } else if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        const uint64_t leaf_data = rd(mlt_data + next, 5);
        if (leaf_data & 1) {
            next = r->mh_start_addr + (leaf_data >> 1);
            r->num_hits = (int)rd(mlt_data + next, 2);
            next += 2;
        } else {
            r->num_hits = 1;
        }
        if (r->num_hits >= r->limit) {
            mem->hitcount += r->num_hits;
            for (int k = 0; k < r->num_hits; ++k) {
                /* a single-hit record is read again as (ref_pos >> 1) from the record itself, :1028-1033 */
                hv_push(hits, rd(mlt_data + next, 5) >> 1);
                next += 5;
            }
            *i += 1;
        } else {
            if (mem->start == 0) {
                if (*i >= r->min_seed_len) {
                    const uint64_t at = path_pop(r, visited);
                    leaf_gather(r, mlt_data, &at, mem, hits);
                }
            }
        }
        lep_set(r, r->nextLEPBit);
        r->nextLEPBit += 1;
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, *i, &countBP);
        r->nextLEPBit += (uint64_t)j;
        *i += j;
        if (j == countBP) {
            if (*i == r->l_seq) {
                if (mem->start == 0) leaf_gather(r, mlt_data, &next, mem, hits);
                lep_set(r, r->nextLEPBit);
            } else if (*i < r->l_seq) {
                gnb_wlimit(r, mlt_data, &next, i, mem, visited, hits);
            }
        } else {
            if (mem->start == 0) {
                if (*i >= r->min_seed_len) leaf_gather(r, mlt_data, &next, mem, hits);
            }
            lep_set(r, r->nextLEPBit);
        }
    } else {
        to_child(r, mlt_data, code, c, &next);
        lep_set(r, r->nextLEPBit);
        r->nextLEPBit += 1;
        if (r->num_hits == 0 || r->num_hits >= r->limit) {
            path_push(r, visited, next, r->num_hits);
            *i += 1;
            if (*i < r->l_seq) {
                gnb_wlimit(r, mlt_data, &next, i, mem, visited, hits);
            } else {
                if (mem->start == 0) leaf_gather(r, mlt_data, &next, mem, hits);
                lep_set(r, r->nextLEPBit);
                r->nextLEPBit += 1;
            }
        } else {
            if (mem->start == 0) {
                if (*i >= r->min_seed_len) {
                    const uint64_t at = path_pop(r, visited);
                    leaf_gather(r, mlt_data, &at, mem, hits);
                }
            }
        }
    }
    *byte_idx = next;
}

/* :1168 */
static void gnb_last(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int *i, mem_t *mem, hv_t *hits)
{
    uint64_t next = *byte_idx;
    int c = 0, code_c;
    uint8_t code = 0;
    if (r->read_buf[*i] != 4) {
        c = 3 - r->read_buf[*i];
        code = mlt_data[next++];
        code_c = (code >> (c << 1)) & 3;
        if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    } else code_c = N_EMPTY;
    if (code_c == N_EMPTY) {
        *i += 1;
    } else if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        leaf_hits(r, mlt_data, next, mem, hits);
        *i += 1;
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, *i, &countBP);
        *i += j;
        const int len = *i - mem->start;
        const int stop = (r->num_hits > 0 && r->num_hits < r->limit && len >= r->min_seed_len + 1) ? 1 : 0;
        if (stop) {
            leaf_gather(r, mlt_data, &next, mem, hits);
            *i = mem->start + (r->min_seed_len + 1);
        } else if (j == countBP) {
            if (*i < r->l_seq) gnb_last(r, mlt_data, &next, i, mem, hits);
        } else {
            *i += 1;
        }
    } else {
        to_child(r, mlt_data, code, c, &next);
        *i += 1;
        const int len = *i - mem->start;
        const int stop = (r->num_hits > 0 && r->num_hits < r->limit && len >= r->min_seed_len + 1) ? 1 : 0;
        if (stop) leaf_gather(r, mlt_data, &next, mem, hits);
        else if (*i < r->l_seq) gnb_last(r, mlt_data, &next, i, mem, hits);
    }
    *byte_idx = next;
}

/* the k-mer entry at read_buf[i..): fields shared by every *Extend* routine */
typedef struct { uint64_t entry; int code; uint64_t start_addr; } kent_t;

static kent_t kmer_entry(raux_t *r, uint32_t hashval)
{
    kent_t k;
    k.entry = r->e->kmer_table[hashval];
    k.code = (int)(k.entry & 3);
    k.start_addr = k.entry >> 24;
    r->ptr_width = ((k.entry >> 22) & 3) == 0 ? 4 : (int)((k.entry >> 22) & 3);
    return k;
}

/* :1288 / :1392 — wlimit = 0: leftExtend, 1: leftExtend_wlimit */
static void left_extend(raux_t *r, int *i, mem_t *mem, hv_t *hits, int wlimit)
{
    const int K = r->K, X = r->X;
    int idx_first_N = -1;
    uint32_t hashval = hash_key(r->read_buf + *i, K, *i, r->l_seq, NULL, &idx_first_N);
    if (idx_first_N != -1) {
        *i += K + X;
        mem->rc_end = *i;
        return;
    }
    const kent_t ke = kmer_entry(r, hashval);
    if (wlimit) r->num_hits = (int)((ke.entry >> 17) & 0x1F);
    uint64_t byte_idx = 0;
    const uint8_t *mlt_data = r->e->mlt + ke.start_addr;
    if (ke.code == E_INVALID) {
        *i += K + X;
        mem->rc_end = *i;
    } else if (ke.code == E_SINGLE) {
        if (wlimit) {
            *i += K + X;
            mem->rc_end = *i;
        } else {
            mem->hitcount += 1;
            hv_push(hits, rd(mlt_data + 1, 5) >> 1);
            *i += K;
            mem->rc_end = *i;
        }
    } else if (ke.code == E_INFREQUENT) {
        *i += K;
        if (!wlimit) {
            if (*i < r->l_seq) {
                r->mh_start_addr = rd(mlt_data + byte_idx, 4);
                byte_idx += 4;
                gnb_backward(r, mlt_data, &byte_idx, i, mem, hits);
            } else mem->rc_end = *i;
        } else if (r->num_hits == 0 || r->num_hits >= r->limit) {
            if (*i < r->l_seq) {
                r->mh_start_addr = rd(mlt_data + byte_idx, 4);
                byte_idx += 4;
                gnb_backward_wlimit(r, mlt_data, &byte_idx, i, mem, hits);
            } else { mem->rc_end = *i; mem->fetch_leaves = 1; }
        } else {
            mem->rc_end = *i;
        }
    } else {
        *i += K;
        /* the x-mer key is taken wherever *i stands, also past the end of the read (the reference reads its buffer there) */
        hashval = hash_key(r->read_buf + *i, X, *i, r->l_seq, NULL, &idx_first_N);
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        const uint64_t xe = rd(mlt_data + byte_idx + ((uint64_t)hashval << 3), 8);
        const int code = (int)(xe & 3);
        const uint64_t ptr = xe >> 24;
        if (wlimit) r->num_hits = (int)((xe >> 17) & 0x1F);
        if (idx_first_N != -1) {
            *i += X;
            mem->rc_end = *i;
            return;
        }
        if (code == E_INVALID) {
            *i += X;
            mem->rc_end = *i;
        } else if (code == E_SINGLE) {
            if (!wlimit) {
                mem->hitcount += 1;
                hv_push(hits, rd(mlt_data + ptr + 1, 5) >> 1);
            }
            *i += X;
            mem->rc_end = *i;
        } else {
            byte_idx = ptr;
            *i += X;
            if (!wlimit) {
                if (*i < r->l_seq) gnb_backward(r, mlt_data, &byte_idx, i, mem, hits);
                else mem->rc_end = *i;
            } else if (r->num_hits == 0 || r->num_hits >= r->limit) {
                if (*i < r->l_seq) gnb_backward_wlimit(r, mlt_data, &byte_idx, i, mem, hits);
                else { mem->rc_end = *i; mem->fetch_leaves = 1; }
            } else {
                mem->rc_end = *i;
            }
        }
    }
}

static void end_gather(raux_t *r, const uint8_t *mlt_data, const uint64_t *at, int i, mem_t *mem, hv_t *hits)
{
    mem->end = i;
    if (mem->end - mem->start >= r->min_seed_len) leaf_gather(r, mlt_data, at, mem, hits);
}

/* :1505 */
static void gnb_fl_prefix_reseed(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int idx, mem_t *mem, path_t *visited,
                                 hv_t *hits)
{
    uint64_t next = *byte_idx;
    const uint64_t parent = next;
    int i = idx;
    if (r->read_buf[i] == 4) { r->flags |= ORC_ERTW_ASSERT; return; }
    const int c = 3 - r->read_buf[i];
    const uint8_t code = mlt_data[next++];
    const int code_c = (code >> (c << 1)) & 3;
    if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    if (code_c == N_EMPTY) {
        end_gather(r, mlt_data, &parent, i, mem, hits);
    } else if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        const uint64_t leaf_data = rd(mlt_data + next, 5);
        if (leaf_data & 1) {
            next = r->mh_start_addr + (leaf_data >> 1);
            r->num_hits = (int)rd(mlt_data + next, 2);
            next += 2;
        } else {
            r->num_hits = 1;
        }
        if (r->num_hits >= r->limit) {
            mem->hitcount += r->num_hits;
            for (int k = 0; k < r->num_hits; ++k) {
                hv_push(hits, rd(mlt_data + next, 5) >> 1);
                next += 5;
            }
            i += 1;
            mem->end = i;
            mem->is_multi_hit = 1;
        } else {
            mem->end = i;
            if (mem->end - mem->start >= r->min_seed_len) {
                const uint64_t at = path_pop(r, visited);
                leaf_gather(r, mlt_data, &at, mem, hits);
            }
        }
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, i, &countBP);
        i += j;
        if (j == countBP && i < r->l_seq) gnb_fl_prefix_reseed(r, mlt_data, &next, i, mem, visited, hits);
        else end_gather(r, mlt_data, &next, i, mem, hits);
    } else {
        r->num_hits = 0;
        to_child(r, mlt_data, code, c, &next);
        if (r->num_hits == 0 || r->num_hits >= r->limit) {
            path_push(r, visited, next, r->num_hits);
            i += 1;
            if (i < r->l_seq) gnb_fl_prefix_reseed(r, mlt_data, &next, i, mem, visited, hits);
            else end_gather(r, mlt_data, &next, i, mem, hits);
        } else {
            mem->end = i;
            if (mem->end - mem->start >= r->min_seed_len) {
                const uint64_t at = path_pop(r, visited);
                leaf_gather(r, mlt_data, &at, mem, hits);
            }
        }
    }
    *byte_idx = next;
}

/* :1644 */
static void gnb_fl_prefix(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int idx, mem_t *mem, hv_t *hits)
{
    uint64_t next = *byte_idx;
    const uint64_t parent = next;
    int i = idx;
    if (r->read_buf[i] == 4) { r->flags |= ORC_ERTW_ASSERT; return; }
    const int c = 3 - r->read_buf[i];
    const uint8_t code = mlt_data[next++];
    const int code_c = (code >> (c << 1)) & 3;
    if (code == 0) r->flags |= ORC_ERTW_ASSERT;
    if (code_c == N_EMPTY) {
        end_gather(r, mlt_data, &parent, i, mem, hits);
    } else if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        leaf_hits(r, mlt_data, next, mem, hits);
        i += 1;
        mem->end = i;
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, i, &countBP);
        i += j;
        if (j == countBP && i < r->l_seq) gnb_fl_prefix(r, mlt_data, &next, i, mem, hits);
        else end_gather(r, mlt_data, &next, i, mem, hits);
    } else {
        r->num_hits = 0;
        to_child(r, mlt_data, code, c, &next);
        i += 1;
        if (i < r->l_seq) gnb_fl_prefix(r, mlt_data, &next, i, mem, hits);
        else end_gather(r, mlt_data, &next, i, mem, hits);
    }
}

/* :1759 */
static void gnb_fl(raux_t *r, const uint8_t *mlt_data, uint64_t *byte_idx, int idx, mem_t *mem, hv_t *hits)
{
    uint64_t next = *byte_idx;
    int i = idx;
    if (r->read_buf[i] == 4) { r->flags |= ORC_ERTW_ASSERT; return; }
    const int c = 3 - r->read_buf[i];
    const uint8_t code = mlt_data[next++];
    const int code_c = (code >> (c << 1)) & 3;
    if (code == 0 || code_c == N_EMPTY) { r->flags |= ORC_ERTW_ASSERT; return; }
    if (code_c == N_LEAF) {
        next += (uint64_t)to_leaf(r, code, c);
        leaf_hits(r, mlt_data, next, mem, hits);
    } else if (code_c == N_UNIFORM) {
        int countBP;
        const int j = uniform_match(r, mlt_data, &next, i, &countBP);
        i += j;
        if (j == countBP && i < mem->end) gnb_fl(r, mlt_data, &next, i, mem, hits);
        else leaf_gather(r, mlt_data, &next, mem, hits);
    } else {
        r->num_hits = 0;
        to_child(r, mlt_data, code, c, &next);
        i += 1;
        if (i < mem->end) gnb_fl(r, mlt_data, &next, i, mem, hits);
        else leaf_gather(r, mlt_data, &next, mem, hits);
    }
}

/* :1854 */
static void rx_fl_prefix_reseed(raux_t *r, mem_t *mem, hv_t *hits)
{
    const int K = r->K, X = r->X;
    int flag = 0, idx_first_N = -1;
    int i = mem->start;
    uint32_t hashval = hash_key(r->read_buf + i, K, i, r->l_seq, &flag, &idx_first_N);
    const kent_t ke = kmer_entry(r, hashval);
    r->mh_start_addr = 0;
    r->num_hits = (int)((ke.entry >> 17) & 0x1F);
    uint64_t byte_idx = 0;
    const uint8_t *mlt_data = r->e->mlt + ke.start_addr;
    if (ke.code == E_INVALID) { r->flags |= ORC_ERTW_ASSERT; return; }
    if (ke.code == E_SINGLE) {
        mem->end = i;
    } else if (ke.code == E_INFREQUENT) {
        if (r->num_hits == 0 || r->num_hits >= r->limit) {
            i += K;
            r->mh_start_addr = rd(mlt_data + byte_idx, 4);
            byte_idx += 4;
            if (i < r->l_seq) {
                path_t visited;
                visited.n = 0;
                path_push(r, &visited, byte_idx, r->num_hits);
                gnb_fl_prefix_reseed(r, mlt_data, &byte_idx, i, mem, &visited, hits);
            } else end_gather(r, mlt_data, &byte_idx, i, mem, hits);
        } else {
            mem->end = i;
        }
    } else {
        hashval = hash_key(r->read_buf + i + K, X, i + K, r->l_seq, NULL, &idx_first_N);
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        const uint64_t xe = rd(mlt_data + byte_idx + ((uint64_t)hashval << 3), 8);
        const int code = (int)(xe & 3);
        const uint64_t ptr = xe >> 24;
        r->num_hits = (int)((xe >> 17) & 0x1F);
        if (code == E_INVALID || code == E_SINGLE) {
            mem->end = i;
        } else if (r->num_hits == 0 || r->num_hits >= r->limit) {
            byte_idx = ptr;
            i += K + X;
            if (i < r->l_seq) {
                path_t visited;
                visited.n = 0;
                path_push(r, &visited, byte_idx, r->num_hits);
                gnb_fl_prefix_reseed(r, mlt_data, &byte_idx, i, mem, &visited, hits);
            } else end_gather(r, mlt_data, &byte_idx, i, mem, hits);
        } else {
            mem->end = i;
        }
    }
}

/* :1967 */
static void rx_fl_prefix(raux_t *r, mem_t *mem, hv_t *hits)
{
    const int K = r->K, X = r->X;
    int flag = 0, idx_first_N = -1;
    int i = mem->start;
    uint32_t hashval = hash_key(r->read_buf + i, K, i, r->l_seq, &flag, &idx_first_N);
    const kent_t ke = kmer_entry(r, hashval);
    r->mh_start_addr = 0;
    uint64_t byte_idx = 0;
    const uint8_t *mlt_data = r->e->mlt + ke.start_addr;
    if (ke.code == E_INVALID) { r->flags |= ORC_ERTW_ASSERT; return; }
    if (ke.code == E_SINGLE) {
        mem->hitcount += 1;
        hv_push(hits, rd(mlt_data + 1, 5) >> 1);
        i += K;
        mem->end = i;
    } else if (ke.code == E_INFREQUENT) {
        i += K;
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        if (i < r->l_seq) gnb_fl_prefix(r, mlt_data, &byte_idx, i, mem, hits);
        else end_gather(r, mlt_data, &byte_idx, i, mem, hits);
    } else {
        hashval = hash_key(r->read_buf + i + K, X, i + K, r->l_seq, NULL, &idx_first_N);
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        const uint64_t xe = rd(mlt_data + byte_idx + ((uint64_t)hashval << 3), 8);
        const int code = (int)(xe & 3);
        const uint64_t ptr = xe >> 24;
        if (code == E_INVALID) {
            mem->end = i;
        } else if (code == E_SINGLE) {
            mem->hitcount += 1;
            hv_push(hits, rd(mlt_data + ptr + 1, 5) >> 1);
            i += K + X;
            mem->end = i;
        } else {
            byte_idx = ptr;
            i += K + X;
            if (i < r->l_seq) gnb_fl_prefix(r, mlt_data, &byte_idx, i, mem, hits);
            else end_gather(r, mlt_data, &byte_idx, i, mem, hits);
        }
    }
}

/* :2071 */
static void rx_fl(raux_t *r, mem_t *mem, hv_t *hits)
{
    const int K = r->K, X = r->X;
    int flag = 0, idx_first_N = -1;
    int i = mem->start;
    const int end = mem->end;
    uint32_t hashval = hash_key(r->read_buf + i, K, i, r->l_seq, &flag, &idx_first_N);
    const kent_t ke = kmer_entry(r, hashval);
    r->mh_start_addr = 0;
    uint64_t byte_idx = 0;
    const uint8_t *mlt_data = r->e->mlt + ke.start_addr;
    if (ke.code == E_INVALID || ke.code == E_SINGLE) { r->flags |= ORC_ERTW_ASSERT; return; }
    if (ke.code == E_INFREQUENT) {
        i += K;
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        if (i < end) gnb_fl(r, mlt_data, &byte_idx, i, mem, hits);
        else leaf_gather(r, mlt_data, &byte_idx, mem, hits);
    } else {
        i += K;
        hashval = hash_key(r->read_buf + i, X, i, r->l_seq, NULL, &idx_first_N);
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        const uint64_t xe = rd(mlt_data + byte_idx + ((uint64_t)hashval << 3), 8);
        const int code = (int)(xe & 3);
        if (code == E_INVALID || code == E_SINGLE) { r->flags |= ORC_ERTW_ASSERT; return; }
        byte_idx = xe >> 24;
        i += X;
        if (i < end) gnb_fl(r, mlt_data, &byte_idx, i, mem, hits);
        else leaf_gather(r, mlt_data, &byte_idx, mem, hits);
    }
}

/* the k-mer's LEP bits ORed in at read position i (:2164-2193: the nine-way case split is this shift) */
static void lep_or_kmer(raux_t *r, int i, uint64_t lep_data)
{
    const int w = i >> 6, sh = i & 63;
    if (w > 4) { r->flags |= ORC_ERTW_LEP_RANGE; return; }
    r->lep[w] |= lep_data << sh;
    if (sh && w + 1 <= 4 && i < 256) r->lep[w + 1] |= lep_data >> (64 - sh);
}

/* :2142 / :2317 — wlimit = 0: rightExtend, 1: rightExtend_wlimit */
static void right_extend(raux_t *r, int *i, mem_t *mem, hv_t *hits, int wlimit)
{
    const int K = r->K, X = r->X;
    int flag = 0, idx_first_N = -1;
    uint32_t hashval = hash_key(r->read_buf + *i, K, *i, r->l_seq, &flag, &idx_first_N);
    const kent_t ke = kmer_entry(r, hashval);
    uint64_t lep_data = (ke.entry >> 2) & ((1ULL << (K - 1)) - 1);
    r->mlt_start_addr = ke.start_addr;
    r->mh_start_addr = 0;
    if (wlimit) r->num_hits = (int)((ke.entry >> 17) & 0x1F);
    lep_or_kmer(r, *i, lep_data);
    r->nextLEPBit = (uint64_t)(*i + K - 1);
    uint64_t byte_idx = 0;
    if (idx_first_N != -1) {
        if (*i != 0) {
            r->nextLEPBit = (uint64_t)(*i + idx_first_N - 1);
            lep_set(r, r->nextLEPBit);
        }
        *i += idx_first_N;
        return;
    }
    if (flag) {
        r->nextLEPBit = (uint64_t)(r->l_seq - 1);
        *i = r->l_seq;
        lep_set(r, r->nextLEPBit);
        return;
    }
    const uint8_t *mlt_data = r->e->mlt + ke.start_addr;
    if (ke.code == E_INVALID) {
        *i += K + X;
    } else if (ke.code == E_SINGLE) {
        if (wlimit) {
            *i += K + X;
        } else {
            mem->hitcount += 1;
            hv_push(hits, rd(mlt_data + 1, 5) >> 1);
            *i += K;
        }
    } else if (ke.code == E_INFREQUENT) {
        *i += K;
        if (!wlimit) {
            if (*i < r->l_seq) {
                r->mh_start_addr = rd(mlt_data + byte_idx, 4);
                byte_idx += 4;
                gnb(r, mlt_data, &byte_idx, i, mem, hits);
            } else {
                lep_set(r, r->nextLEPBit);
                r->nextLEPBit += 1;
            }
        } else if (r->num_hits == 0 || r->num_hits >= r->limit) {
            if (*i < r->l_seq) {
                path_t visited;
                visited.n = 0;
                path_push(r, &visited, byte_idx, r->num_hits);        /* byte_idx 0: before the 4-byte header, as written */
                r->mh_start_addr = rd(mlt_data + byte_idx, 4);
                byte_idx += 4;
                gnb_wlimit(r, mlt_data, &byte_idx, i, mem, &visited, hits);
            } else {
                lep_set(r, r->nextLEPBit);
                r->nextLEPBit += 1;
            }
        }
    } else {
        *i += K;
        flag = 0;
        hashval = hash_key(r->read_buf + *i, X, *i, r->l_seq, &flag, &idx_first_N);
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        const uint64_t xe = rd(mlt_data + byte_idx + ((uint64_t)hashval << 3), 8);
        const int code = (int)(xe & 3);
        lep_data = (xe >> 2) & ((1ULL << X) - 1);
        const uint64_t ptr = xe >> 24;
        if (wlimit) r->num_hits = (int)((xe >> 17) & 0x1F);
        const int xmerLen = r->l_seq - *i > X ? X : r->l_seq - *i;
        for (int k = 0; k < xmerLen; ++k) {
            if ((lep_data >> k) & 1) lep_set(r, r->nextLEPBit);
            r->nextLEPBit++;
        }
        if (idx_first_N != -1) {
            r->nextLEPBit = (uint64_t)(*i + idx_first_N - 1);
            lep_set(r, r->nextLEPBit);
            *i += idx_first_N;
            return;
        }
        if (flag) {
            r->nextLEPBit = (uint64_t)(r->l_seq - 1);
            *i = r->l_seq;
            lep_set(r, r->nextLEPBit);
            return;
        }
        if (code == E_INVALID) {
            *i += X;
        } else if (code == E_SINGLE) {
            if (!wlimit) {
                mem->hitcount += 1;
                hv_push(hits, rd(mlt_data + ptr + 1, 5) >> 1);
            }
            *i += X;
        } else {
            byte_idx = ptr;
            *i += X;
            if (!wlimit) {
                if (*i < r->l_seq) gnb(r, mlt_data, &byte_idx, i, mem, hits);
                else { lep_set(r, r->nextLEPBit); r->nextLEPBit += 1; }
            } else if (r->num_hits == 0 || r->num_hits >= r->limit) {
                if (*i < r->l_seq) {
                    path_t visited;
                    visited.n = 0;
                    path_push(r, &visited, byte_idx, r->num_hits);
                    gnb_wlimit(r, mlt_data, &byte_idx, i, mem, &visited, hits);
                } else { lep_set(r, r->nextLEPBit); r->nextLEPBit += 1; }
            }
        }
    }
}

/* :2500 */
static void right_extend_last(raux_t *r, int *i, mem_t *mem, hv_t *hits)
{
    const int K = r->K, X = r->X;
    int flag = 0, idx_first_N = -1;
    uint32_t hashval = hash_key(r->read_buf + *i, K, *i, r->l_seq, &flag, &idx_first_N);
    if (idx_first_N != -1) { *i += idx_first_N + 1; return; }
    if (flag) { *i = r->l_seq; return; }
    const kent_t ke = kmer_entry(r, hashval);
    r->mlt_start_addr = ke.start_addr;
    r->mh_start_addr = 0;
    r->num_hits = (int)((ke.entry >> 17) & 0x1F);
    uint64_t byte_idx = 0;
    const uint8_t *mlt_data = r->e->mlt + ke.start_addr;
    if (ke.code == E_INVALID) {
        *i += K;
    } else if (ke.code == E_SINGLE) {
        mem->hitcount += 1;
        hv_push(hits, rd(mlt_data + 1, 5) >> 1);
        *i += K;
    } else if (ke.code == E_INFREQUENT) {
        *i += K;
        if (*i < r->l_seq) {
            r->mh_start_addr = rd(mlt_data + byte_idx, 4);
            byte_idx += 4;
            gnb_last(r, mlt_data, &byte_idx, i, mem, hits);
        }
    } else {
        *i += K;
        flag = 0;
        hashval = hash_key(r->read_buf + *i, X, *i, r->l_seq, &flag, &idx_first_N);
        if (idx_first_N != -1) { *i += idx_first_N + 1; return; }
        if (flag) { *i = r->l_seq; return; }
        r->mh_start_addr = rd(mlt_data + byte_idx, 4);
        byte_idx += 4;
        const uint64_t xe = rd(mlt_data + byte_idx + ((uint64_t)hashval << 3), 8);
        const int code = (int)(xe & 3);
        const uint64_t ptr = xe >> 24;
        r->num_hits = (int)((xe >> 17) & 0x1F);
        if (code == E_INVALID) {
            *i += X;
        } else if (code == E_SINGLE) {
            mem->hitcount += 1;
            hv_push(hits, rd(mlt_data + ptr + 1, 5) >> 1);
            *i += X;
        } else {
            byte_idx = ptr;
            *i += X;
            if (r->num_hits == 0 || r->num_hits >= r->limit || (*i - mem->start) < (r->min_seed_len + 1)) {
                if (*i < r->l_seq) gnb_last(r, mlt_data, &byte_idx, i, mem, hits);
            } else {
                leaf_gather(r, mlt_data, &byte_idx, mem, hits);
            }
        }
    }
}

/* :2609 */
static int init_mem(const uint64_t *lep, mem_t *mem, int j, int seq_len, int min_seed_len)
{
    const int lep_bit_set = (int)((lep[j >> 6] >> (j & 63)) & 1);
    const int in_valid_range = j >= min_seed_len - 1;
    memset(mem, 0, sizeof *mem);            /* start and pt are left unset by the reference; zero here */
    mem->end = j + 1;
    mem->rc_start = seq_len - j - 1;
    mem->rc_end = mem->rc_start;
    return lep_bit_set && in_valid_range;
}

/* bases of fw[from..) that continue the match at text position pos (lazy leaf expansion to the right) */
static int expand_right(const raux_t *r, int64_t start_ref_pos, int64_t end_ref_pos, int from)
{
    int64_t len;
    const uint8_t *rseq = get_seq(r, start_ref_pos, end_ref_pos, &len);
    int n = 0;
    for (int64_t m = 0; m < len; ++m) {
        if (rseq[m] == r->fw[from + m]) n++;
        else break;
    }
    return n;
}

/* the first half of check_and_add_smem_prefix[_reseed] (:2761-2793, :2644-2673): leaf expansion of a backward-found
 * match on both of its sides, in the coordinates of the reverse-complemented read */
static void expand_lmem_prefix(raux_t *r, mem_t *mem, const hv_t *hits, int lmemLen)
{
    int64_t len;
    const uint64_t h = hits->a[mem->hitbeg];
    const uint8_t *rseq = get_seq(r, (int64_t)h - mem->rc_start, (int64_t)h, &len);
    int n = 0;
    for (int64_t m = 1; m <= len; ++m) {
        if (rseq[mem->rc_start - m] == r->read_buf[mem->rc_start - m]) n++;
        else break;
    }
    mem->end += n;
    mem->end_correction += n;
    const int64_t s = (int64_t)h + lmemLen;
    rseq = get_seq(r, s, s + mem->start, &len);
    n = 0;
    for (int64_t m = 0; m < len; ++m) {
        if (rseq[m] == r->read_buf[mem->rc_end + m]) n++;
        else break;
    }
    mem->start -= n;
}

/* :2639 */
static int caas_prefix_reseed(raux_t *r, mem_t *mem, sh_t *sh, mv_t *smems, hv_t *hits)
{
    mem->start = r->l_seq - mem->rc_end;
    int lmemLen = mem->end - mem->start, rmemLen = -1, next_be_point;
    if (mem->hitcount > 0 && !mem->skip_ref_fetch) expand_lmem_prefix(r, mem, hits, lmemLen);
    lmemLen = mem->end - mem->start;
    next_be_point = mem->end;
    if (mem->hitcount == 1) {
        if (lmemLen >= r->min_seed_len) mv_push(smems, mem);
        else next_be_point += r->min_seed_len - lmemLen;
    } else if (mem->fetch_leaves && mem->start <= r->l_seq - r->min_seed_len) {
        hits->n -= mem->hitcount;
        mem->hitbeg = (int32_t)hits->n;
        mem->hitcount = 0;
        r->read_buf = r->fw;
        rx_fl_prefix_reseed(r, mem, hits);
        r->read_buf = r->rc;
        rmemLen = mem->end - mem->start;
        next_be_point = mem->end;
        if (mem->hitcount > 0) {
            if (mem->is_multi_hit) {
                const int64_t h = (int64_t)hits->a[mem->hitbeg];
                mem->end += expand_right(r, h + rmemLen, h + r->l_seq - mem->start, mem->end);
                rmemLen = mem->end - mem->start;
                next_be_point = mem->end;
            }
            if (rmemLen >= r->min_seed_len && mem->end <= sh->mem_end_limit) mv_push(smems, mem);
            else next_be_point += r->min_seed_len - rmemLen;
        } else {
            if (rmemLen > r->min_seed_len) r->flags |= ORC_ERTW_ASSERT;
            next_be_point += r->min_seed_len - rmemLen;
        }
    } else {
        if (lmemLen <= r->min_seed_len) next_be_point += r->min_seed_len - lmemLen;
    }
    return next_be_point;
}

/* :2756 */
static int caas_prefix(raux_t *r, mem_t *mem, sh_t *sh, mv_t *smems, hv_t *hits)
{
    (void)sh;
    mem->start = r->l_seq - mem->rc_end;
    int lmemLen = mem->end - mem->start, rmemLen = -1, next_be_point;
    if (mem->hitcount > 0 && !mem->skip_ref_fetch) expand_lmem_prefix(r, mem, hits, lmemLen);
    lmemLen = mem->end - mem->start;
    next_be_point = mem->end;
    if (mem->hitcount == 1) {
        if (lmemLen >= r->min_seed_len) mv_push(smems, mem);
        else next_be_point += r->min_seed_len - lmemLen;
    } else if (mem->fetch_leaves && mem->start <= r->l_seq - r->min_seed_len) {
        hits->n -= mem->hitcount;
        mem->hitbeg = (int32_t)hits->n;
        mem->hitcount = 0;
        r->read_buf = r->fw;
        rx_fl_prefix(r, mem, hits);
        r->read_buf = r->rc;
        rmemLen = mem->end - mem->start;
        next_be_point = mem->end;
        if (mem->hitcount > 0) {
            const int64_t h = (int64_t)hits->a[mem->hitbeg];
            mem->end += expand_right(r, h + rmemLen, h + r->l_seq - mem->start, mem->end);
            rmemLen = mem->end - mem->start;
            next_be_point = mem->end;
            if (rmemLen >= r->min_seed_len) mv_push(smems, mem);
            else next_be_point += r->min_seed_len - rmemLen;
        } else {
            if (rmemLen > r->min_seed_len) r->flags |= ORC_ERTW_ASSERT;
            next_be_point += r->min_seed_len - rmemLen;
        }
    } else {
        if (lmemLen > r->min_seed_len) r->flags |= ORC_ERTW_ASSERT;
        next_be_point += r->min_seed_len - lmemLen;
    }
    return next_be_point;
}

/* :2867 */
static void caas(raux_t *r, mem_t *mem, sh_t *sh, mv_t *smems, hv_t *hits)
{
    mem->start = r->l_seq - mem->rc_end;
    int lmemLen = mem->end - mem->start;
    if (mem->hitcount > 0 && !mem->skip_ref_fetch) {
        int64_t len;
        const int64_t s = (int64_t)hits->a[mem->hitbeg] + lmemLen;
        const uint8_t *rseq = get_seq(r, s, s + mem->start, &len);
        int n = 0;
        for (int64_t m = 0; m < len; ++m) {
            if (rseq[m] == r->read_buf[mem->rc_end + m]) n++;
            else break;
        }
        mem->start -= n;
    }
    lmemLen = mem->end - mem->start;
    if (lmemLen >= r->min_seed_len) {
        if (mem->start < sh->prevMemStart || mem->end > sh->prevMemEnd) {
            if (mem->fetch_leaves) {
                hits->n -= mem->hitcount;
                mem->hitbeg = (int32_t)hits->n;
                mem->hitcount = 0;
                r->read_buf = r->fw;
                rx_fl(r, mem, hits);
                r->read_buf = r->rc;
            }
            if (mem->hitcount > 0) {
                mem->c_pivot = sh->curr_pivot;
                mem->p_pivot = sh->prev_pivot;
                mem->pp_pivot = sh->prev_prev_pivot;
                mv_push(smems, mem);
                if (mem->start <= sh->prev_pivot + 1) sh->stop_be = 1;
            }
            sh->prevMemStart = mem->start;
            sh->prevMemEnd = mem->end;
        }
    }
}

/* the forward match of a pivot with its lazy leaf expansion and LEP marks (:2942-2966 and its three twins) */
static void rmem_expand(raux_t *r, mem_t *rm, int *i, const hv_t *hits)
{
    if (rm->hitcount > 0 && !rm->skip_ref_fetch) {
        int64_t len;
        const int64_t h = (int64_t)hits->a[rm->hitbeg];
        const uint8_t *rseq = get_seq(r, h + *i - rm->start, h + r->l_seq - rm->start, &len);
        int64_t m;
        int n = 0;
        for (m = 0; m < len; ++m) {
            if (rseq[m] == r->fw[*i + m]) n++;
            else { lep_set(r, (uint64_t)(*i + m - 1)); break; }
        }
        if (m == len) lep_set(r, (uint64_t)(*i + m - 1));
        *i += n;
    }
}

/* the pivot advance shared by get_seeds and get_seeds_prefix (:3018-3033) */
static void skip_ambiguous(const raux_t *r, int *i, int rm_start)
{
    while (*i < r->l_seq) {
        if (r->fw[*i] == 4) ++*i;
        else break;
    }
    while (*i < r->l_seq && (*i - rm_start) < r->min_seed_len) {
        if (r->fw[*i] == 4) { ++*i; break; }
        ++*i;
    }
}

/* :2925 (prefix = 1) and :3062 (prefix = 0) */
static void get_seeds(raux_t *r, mv_t *smems, hv_t *hits, int prefix)
{
    sh_t sh;
    memset(&sh, 0, sizeof sh);
    sh.prevMemStart = r->l_seq;
    sh.prevMemEnd = 0;
    int i = 0, j = 0;
    sh.prev_pivot = -1;
    sh.prev_prev_pivot = -1;
    memset(r->lep, 0, sizeof r->lep);
    while (i < r->l_seq) {
        mem_t rm;
        memset(&rm, 0, sizeof rm);
        rm.start = i;
        rm.forward = 1;
        rm.hitbeg = (int32_t)hits->n;
        sh.curr_pivot = rm.start;
        r->read_buf = r->fw;
        right_extend(r, &i, &rm, hits, 0);
        rmem_expand(r, &rm, &i, hits);
        rm.end = i;
        const int rmemLen = rm.end - rm.start;
        if (rm.start == 0) {
            if (rmemLen >= r->min_seed_len) {
                if (rm.hitcount > 0) {
                    if (!prefix) { rm.c_pivot = sh.curr_pivot; rm.p_pivot = sh.prev_pivot; rm.pp_pivot = sh.prev_prev_pivot; }
                    mv_push(smems, &rm);
                }
            } else {
                hits->n -= rm.hitcount;
            }
            memset(r->lep, 0, sizeof r->lep);
        } else {
            hits->n -= rm.hitcount;
            const int seq_len = r->l_seq, msl = r->min_seed_len;
            sh.stop_be = 0;
            const int min_j = rm.start > msl ? rm.start - 1 : msl - 1;
            if (prefix) {
                const int max_j = rm.end - 1;
                j = min_j;
                sh.prev_pivot = rm.start;
                while (j <= max_j) {
                    mem_t m;
                    const int valid = init_mem(r->lep, &m, j, seq_len, msl);
                    m.hitbeg = (int32_t)hits->n;
                    int next_j = j + 1;
                    if (valid) {
                        const int be_point = j + 1;
                        if (be_point >= msl) {
                            int rc_i = seq_len - be_point;
                            r->read_buf = r->rc;
                            left_extend(r, &rc_i, &m, hits, 0);
                            next_j = caas_prefix(r, &m, &sh, smems, hits);
                        }
                    }
                    j = next_j;
                    if (m.end > i) i = m.end;
                }
            } else {
                j = rm.end - 1;
                while (j >= min_j) {
                    mem_t m;
                    const int valid = init_mem(r->lep, &m, j, seq_len, msl);
                    m.hitbeg = (int32_t)hits->n;
                    if (valid) {
                        const int be_point = j + 1;
                        if (be_point >= msl) {
                            int rc_i = seq_len - be_point;
                            r->read_buf = r->rc;
                            left_extend(r, &rc_i, &m, hits, 0);
                            caas(r, &m, &sh, smems, hits);
                            if (sh.stop_be) break;
                        }
                    }
                    j -= 1;
                }
            }
        }
        r->read_buf = r->fw;
        skip_ambiguous(r, &i, rm.start);
        sh.prev_prev_pivot = sh.prev_pivot;
        sh.prev_pivot = rm.start;
        memset(r->lep, 0, sizeof r->lep);
    }
}

/* :3200 (prefix = 1) and :3315 (prefix = 0) */
static void reseed(raux_t *r, mv_t *smems, int start, int limit, const mem_t *pt, hv_t *hits, int prefix)
{
    sh_t sh;
    memset(&sh, 0, sizeof sh);
    sh.prevMemStart = r->l_seq;
    sh.prevMemEnd = 0;
    int i = start, j = 0;
    memset(r->lep, 0, sizeof r->lep);
    mem_t rm;
    memset(&rm, 0, sizeof rm);
    rm.start = i;
    rm.forward = 1;
    rm.hitbeg = (int32_t)hits->n;
    sh.prev_pivot = rm.start >= pt->c_pivot ? pt->p_pivot : pt->pp_pivot;
    r->read_buf = r->fw;
    r->limit = limit;
    right_extend(r, &i, &rm, hits, 1);
    rmem_expand(r, &rm, &i, hits);
    rm.end = i;
    const int rmemLen = rm.end - rm.start;
    if (rm.start == 0) {
        if (rmemLen >= r->min_seed_len) {
            if (rm.hitcount > 0) mv_push(smems, &rm);
        } else {
            hits->n -= rm.hitcount;
        }
        memset(r->lep, 0, sizeof r->lep);
    } else {
        hits->n -= rm.hitcount;
        const int seq_len = r->l_seq, msl = r->min_seed_len;
        sh.stop_be = 0;
        const int min_j = rm.start > msl ? rm.start - 1 : msl - 1;
        if (prefix) {
            const int max_j = rm.end - 1;
            j = min_j;
            sh.prev_pivot = rm.start;
            sh.mem_end_limit = rm.end;
            while (j <= max_j) {
                mem_t m;
                const int valid = init_mem(r->lep, &m, j, seq_len, msl);
                m.hitbeg = (int32_t)hits->n;
                int next_j = j + 1;
                if (valid) {
                    const int be_point = j + 1;
                    if (be_point >= msl) {
                        int rc_i = seq_len - be_point;
                        r->read_buf = r->rc;
                        left_extend(r, &rc_i, &m, hits, 1);
                        next_j = caas_prefix_reseed(r, &m, &sh, smems, hits);
                    }
                }
                j = next_j;
            }
        } else {
            j = rm.end - 1;
            while (j >= min_j) {
                mem_t m;
                const int valid = init_mem(r->lep, &m, j, seq_len, msl);
                m.hitbeg = (int32_t)hits->n;
                if (valid) {
                    const int be_point = j + 1;
                    if (be_point >= msl) {
                        int rc_i = seq_len - be_point;
                        r->read_buf = r->rc;
                        left_extend(r, &rc_i, &m, hits, 1);
                        caas(r, &m, &sh, smems, hits);
                        if (sh.stop_be) break;
                    }
                }
                j -= 1;
            }
        }
    }
}

/* :3425 */
static void last(raux_t *r, mv_t *smems, int limit, hv_t *hits)
{
    int i = 0;
    const uint8_t minSeedLen = (uint8_t)(r->min_seed_len + 1);
    r->limit = limit;
    while (i < r->l_seq) {
        mem_t rm;
        memset(&rm, 0, sizeof rm);          /* the reference leaves fetch_leaves, end_correction, pt unset; forward = 1 decides */
        rm.start = i;
        rm.forward = 1;
        rm.hitbeg = (int32_t)hits->n;
        r->read_buf = r->fw;
        right_extend_last(r, &i, &rm, hits);
        if (rm.hitcount > 0 && !rm.skip_ref_fetch) {
            int64_t len;
            const int64_t h = (int64_t)hits->a[rm.hitbeg];
            const uint8_t *rseq = get_seq(r, h + i - rm.start, h + r->l_seq - rm.start, &len);
            int n = 0;
            for (int64_t m = 0; m < len; ++m) {
                const int seedLen = (int)(i + m) - rm.start;
                const int match_next_bp = (seedLen < minSeedLen || rm.hitcount >= r->limit) ? 1 : 0;
                if (!match_next_bp) break;
                if (rseq[m] == r->fw[i + m]) n++;
                else {
                    ++i;
                    hits->n -= rm.hitcount;
                    rm.hitcount = 0;
                    break;
                }
            }
            i += n;
        }
        rm.end = i;
        const int rmemLen = rm.end - rm.start;
        if (rmemLen >= minSeedLen) {
            if (rm.hitcount > 0 && rm.hitcount < r->limit) mv_push(smems, &rm);
            else hits->n -= rm.hitcount;
        } else {
            hits->n -= rm.hitcount;
        }
        if (i <= 0) { r->flags |= ORC_ERTW_ASSERT; break; }
        const int foundN = r->fw[i - 1] == 4;
        if (!foundN) {
            while (i < r->l_seq && (i - rm.start) < minSeedLen) {
                if (r->fw[i] == 4) { ++i; break; }
                ++i;
            }
        }
    }
}

/* the per-read block of mem_kernel1_core_ert (/root/reference/src/bwamem.cpp:1122-1191) up to, not including, the sort */
static int walk_read(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *seq, int len, mv_t *smems, hv_t *hits)
{
    uint8_t rcbuf[512];
    int hasN = 0;
    for (int i = 0; i < len; ++i) {
        hasN = seq[i] < 4 ? hasN : 1;
        rcbuf[len - i - 1] = seq[i] < 4 ? (uint8_t)(3 - seq[i]) : 4;
    }
    const int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
    raux_t r;
    memset(&r, 0, sizeof r);
    r.e = e; r.K = e->kmer; r.X = e->xmer; r.l_pac = e->ref_len / 2;
    r.min_seed_len = opt->min_seed_len;
    r.l_seq = len;
    r.fw = seq; r.rc = rcbuf; r.read_buf = seq;
    smems->n = 0;
    hits->n = 0;
    get_seeds(&r, smems, hits, !hasN);
    const int64_t old_n = smems->n;
    for (int64_t i = 0; i < old_n; ++i) {
        const int qbeg = smems->a[i].start, qend = smems->a[i].end;
        if (qend - qbeg < split_len || smems->a[i].hitcount > opt->split_width) continue;
        const mem_t pt = smems->a[i];       /* &smems->a[i].pt: copied, the vector may move */
        reseed(&r, smems, (qbeg + qend) >> 1, smems->a[i].hitcount + 1, &pt, hits, !hasN);
    }
    last(&r, smems, opt->max_mem_intv, hits);
    return r.flags;
}

/* Batch form.  Per read r (skip[r] reads excepted): mems[mem_off[r] .. mem_off[r+1]) in the order the walk pushed them
 * (mem_kernel1_core_ert sorts them afterwards: orc_chain_new_ert does), hitbeg relative to hits[hit_off[r]].
 * Returns the number of MEMs, -1 when a buffer is too small, -2 for reads the reference refuses (longer than READ_LEN).
 * *flags_out = OR of the ORC_ERTW_* conditions met. */
int64_t orc_ert_walk(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *enc, const int64_t *cum, const uint8_t *skip,
                     int32_t nseq, bwams_ert_mem_t *mems, int64_t mem_cap, int64_t *mem_off, uint64_t *hits, int64_t hit_cap,
                     int64_t *hit_off, int32_t *flags_out)
{
    mv_t sm = {0};
    hv_t hv = {0};
    int64_t nm = 0, nh = 0;
    int flags = 0;
    for (int32_t r = 0; r < nseq; ++r) {
        mem_off[r] = nm;
        hit_off[r] = nh;
        if (skip && skip[r]) continue;
        const int len = (int)(cum[r + 1] - cum[r]);
        if (len > e->read_len || len > 320) { free(sm.a); free(hv.a); return -2; }
        if (len < 1) continue;
        flags |= walk_read(e, opt, enc + cum[r], len, &sm, &hv);
        if (nm + sm.n > mem_cap || nh + hv.n > hit_cap) { free(sm.a); free(hv.a); return -1; }
        memcpy(mems + nm, sm.a, (size_t)sm.n * sizeof(mem_t));
        memcpy(hits + nh, hv.a, (size_t)hv.n * sizeof(uint64_t));
        nm += sm.n;
        nh += hv.n;
    }
    mem_off[nseq] = nm;
    hit_off[nseq] = nh;
    free(sm.a);
    free(hv.a);
    if (flags_out) *flags_out = flags;
    return nm;
}

static int cmp_mem(const void *pa, const void *pb)
{
    const mem_t *a = (const mem_t *)pa, *b = (const mem_t *)pb;
    if (a->start != b->start) return a->start < b->start ? -1 : 1;
    if (a->end != b->end) return a->end < b->end ? -1 : 1;
    return 0;
}

/* The walk's result in the layout of orc_collect_smem + orc_sa_lookup (and of orc_ert_collect): per read the MEMs sorted by
 * (start, end) — mem_smem_sort_lt, bwamem.cpp:73-74, :1193; MEMs with equal keys carry the same hit set — as SMEM records
 * with k = l = 0, s = hitcount, and per MEM the coordinates mem_chain_new (bwamem.cpp:993-1006) gives its seeds: every
 * step-th hit, at most max_occ, taken as it is for forward / fetch_leaves MEMs and mapped
 * 2 l_pac - (hit + slen - end_correction) for a backward-found one.  cls[t] = forward | fetch_leaves << 1 |
 * (end_correction != 0) << 2 for MEM t (may be NULL). */
int64_t orc_ert_walk_collect(const orc_ert_t *e, const bwams_seed_opt_t *opt, const uint8_t *enc, const int64_t *cum,
                             const uint8_t *skip, int32_t nseq, bwams_smem_t *out, int64_t cap, int64_t *sa_coord, int64_t sa_cap,
                             int64_t *sa_off, uint8_t *cls, int32_t *flags_out)
{
    mv_t sm = {0};
    hv_t hv = {0};
    int64_t n_out = 0, tot = 0;
    int flags = 0;
    const int64_t l_pac = e->ref_len / 2;
    for (int32_t r = 0; r < nseq; ++r) {
        if (skip && skip[r]) continue;
        const int len = (int)(cum[r + 1] - cum[r]);
        if (len > e->read_len || len > 320) { free(sm.a); free(hv.a); return -2; }
        if (len < 1) continue;
        flags |= walk_read(e, opt, enc + cum[r], len, &sm, &hv);
        qsort(sm.a, (size_t)sm.n, sizeof(mem_t), cmp_mem);
        for (int64_t t = 0; t < sm.n; ++t) {
            const mem_t *p = &sm.a[t];
            if (n_out >= cap) { free(sm.a); free(hv.a); return -1; }
            bwams_smem_t *o = &out[n_out];
            memset(o, 0, sizeof *o);
            o->rid = (uint32_t)r; o->m = (uint32_t)p->start; o->n = (uint32_t)(p->end - 1); o->s = p->hitcount;
            if (cls) cls[n_out] = (uint8_t)((p->forward ? 1 : 0) | (p->fetch_leaves ? 2 : 0) | (p->end_correction ? 4 : 0));
            sa_off[n_out] = tot;
            const int slen = p->end - p->start;
            const int step = p->hitcount > opt->max_occ ? p->hitcount / opt->max_occ : 1;
            int count = 0;
            for (int64_t k = 0; k < p->hitcount && count < opt->max_occ; k += step, ++count) {
                if (tot >= sa_cap) { free(sm.a); free(hv.a); return -1; }
                const uint64_t h = hv.a[p->hitbeg + k];
                sa_coord[tot++] = (p->forward || p->fetch_leaves) ? (int64_t)h
                                                                 : (l_pac << 1) - ((int64_t)h + slen - p->end_correction);
            }
            n_out++;
        }
    }
    sa_off[n_out] = tot;
    free(sm.a);
    free(hv.a);
    if (flags_out) *flags_out = flags;
    return n_out;
}
