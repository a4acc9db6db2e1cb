// ref_harness_chain.cpp — thin C entry points over the REFERENCE's own klib headers
// (TEST INFRASTRUCTURE ONLY).  kbtree.h and ksort.h are compiled from where they lie
// (/root/reference/src, -I on the command line of oracle/Makefile); nothing of them is
// copied here.  The instantiations mirror the two the chaining code makes:
//   KBTREE_INIT(chn, mem_chain_t, chain_cmp) + kb_init(chn, KB_DEFAULT_SIZE + 8)  (bwamem.cpp:63-64, :830)
//   KSORT_INIT(mem_flt, mem_chain_t, flt_lt)                                       (bwamem.cpp:89-90)
// with a 48-byte stand-in key of the same size as mem_chain_t (so that the B-tree order t is
// the reference's) carrying the two fields the comparators read: pos and w.
#include <assert.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "kbtree.h"
#include "ksort.h"
#include "utils.h"      /* hash_64 */
#include "bntseq.h"     /* bns_depos */

typedef struct {
    int32_t id, pad0[5];
    uint32_t w;
    float pad1;
    int64_t pos;
    void *pad2;
} key48_t;
static_assert(sizeof(key48_t) == 48, "same size as mem_chain_t");

#define chain_cmp(a, b) (((b).pos < (a).pos) - ((a).pos < (b).pos))
KBTREE_INIT(chn, key48_t, chain_cmp)
#define flt_lt(a, b) ((a).w > (b).w)
KSORT_INIT(mem_flt, key48_t, flt_lt)

// the two sorts of mem_sort_dedup_patch (bwamem.cpp:176-180) over the fields their comparators read
typedef struct { int64_t re, rb; int32_t score, qb, id; } reg5_t;
#define alnreg_slt2(a, b) ((a).re < (b).re)
KSORT_INIT(mem_ars2, reg5_t, alnreg_slt2)
#define alnreg_slt(a, b) ((a).score > (b).score || ((a).score == (b).score && ((a).rb < (b).rb || ((a).rb == (b).rb && (a).qb < (b).qb))))
KSORT_INIT(mem_ars, reg5_t, alnreg_slt)

extern "C" {

// which = 0: ks_introsort(mem_ars2) on k0 = re; which = 1: ks_introsort(mem_ars) on (k0, k1, k2) = (score, rb, qb)
void ref_ars_sort(int64_t n, int which, const int64_t *k0, const int64_t *k1, const int64_t *k2, int32_t *order)
{
    reg5_t *a = (reg5_t *)calloc(n ? n : 1, sizeof(reg5_t));
    for (int64_t i = 0; i < n; ++i) {
        a[i].id = (int32_t)i;
        if (which) { a[i].score = (int32_t)k0[i]; a[i].rb = k1[i]; a[i].qb = (int32_t)k2[i]; }
        else a[i].re = k0[i];
    }
    if (which) ks_introsort(mem_ars, n, a); else ks_introsort(mem_ars2, n, a);
    for (int64_t i = 0; i < n; ++i) order[i] = a[i].id;
    free(a);
}

// for every i: look up the closest key <= pos[i] (when the tree is not empty), then insert i if do_put[i]
int64_t ref_kbt_script(int64_t n, const int64_t *pos, const uint8_t *do_put, int32_t *lower, int32_t *order)
{
    kbtree_t(chn) *tree = kb_init(chn, KB_DEFAULT_SIZE + 8);
    for (int64_t i = 0; i < n; ++i) {
        key48_t tmp, *lo = 0, *up = 0;
        tmp.pos = pos[i]; tmp.id = (int32_t)i; tmp.w = 0;
        lower[i] = -1;
        if (kb_size(tree)) {
            kb_intervalp(chn, tree, &tmp, &lo, &up);
            if (lo) lower[i] = lo->id;
        }
        if (do_put[i]) kb_putp(chn, tree, &tmp);
    }
    int64_t m = 0;
#define trav(p_) (order[m++] = (p_)->id)
    __kb_traverse(key48_t, tree, trav);
#undef trav
    kb_destroy(chn, tree);
    return m;
}

void ref_flt_sort(int64_t n, const uint32_t *w, int32_t *order)
{
    key48_t *a = (key48_t *)calloc(n ? n : 1, sizeof(key48_t));
    for (int64_t i = 0; i < n; ++i) { a[i].w = w[i]; a[i].id = (int32_t)i; }
    ks_introsort(mem_flt, n, a);
    for (int64_t i = 0; i < n; ++i) order[i] = a[i].id;
    free(a);
}

/* hash_64 (utils.h:117-128) and bns_depos (bntseq.h:88-91): header-only inlines */
uint64_t ref_hash_64(uint64_t key) { return hash_64(key); }
int64_t ref_bns_depos(int64_t l_pac, int64_t pos, int *is_rev)
{
    bntseq_t b;
    memset(&b, 0, sizeof b);
    b.l_pac = l_pac;
    return bns_depos(&b, pos, is_rev);
}

}  // extern "C"
