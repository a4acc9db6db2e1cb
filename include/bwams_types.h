/*
 * bwams_types.h — plain-old-data records that cross the C-ABI boundary.
 *
 * Every record is byte-for-byte the layout the reference keeps in host memory,
 * so a reference-side caller can hand its own arrays to the library without a
 * conversion pass.  Reference definitions (file:line under /root/reference):
 *   CP_OCC     src/FMI_search.h:64-68     (64 B checkpoint block, 64 BWT rows)
 *   SMEM       src/FMI_search.h:85-93     (40 B without DEBUG)
 *   SeqPair    src/bandedSWA.h:90-99      (56 B)
 *   mem_opt_t  src/bwamem.h:89-124        (only the fields the hot path reads)
 */
#ifndef BWAMS_TYPES_H
#define BWAMS_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One checkpoint of the FM-index: occurrence counts of A,C,G,T in BWT[0, 64*blk)
 * followed by four one-hot bit strings; bit (63-j) of one_hot_bwt_str[c] is set
 * iff BWT[64*blk + j] == c.  The sentinel row and the padding rows set no bit. */
typedef struct bwams_cp_occ {
    int64_t  cp_count[4];
    uint64_t one_hot_bwt_str[4];
} bwams_cp_occ_t;

/* Bi-directional interval of the query span [m, n] (inclusive):
 * k = first BWT row of the forward pattern, l = first row of its reverse
 * complement, s = interval size. */
typedef struct bwams_smem {
    uint32_t rid;
    uint32_t m, n;
    uint32_t pad_;      /* the reference struct has 4 B of alignment padding here */
    int64_t  k, l, s;
} bwams_smem_t;

/* One extension task.  idr/idq are byte offsets into the flat reference/query
 * buffers (one base code 0..4 per byte); len1 = target length, len2 = query
 * length, h0 = initial score.  The last six fields are outputs. */
typedef struct bwams_seqpair {
    int32_t idr, idq, id;
    int32_t len1, len2;
    int32_t h0;
    int32_t seqid, regid;
    int32_t score, tle, gtle, qle;
    int32_t gscore, max_off;
} bwams_seqpair_t;

/* Seeding options: the subset of mem_opt_t read by mem_collect_smem
 * (src/bwamem.cpp:648-786) and mem_chain_seeds' SA step (src/bwamem.cpp:861-873).
 * Defaults: src/bwamem.cpp:135-171. */
typedef struct bwams_seed_opt {
    int32_t min_seed_len;   /* -k, default 19 */
    float   split_factor;   /* -r, default 1.5 */
    int32_t split_width;    /* default 10 */
    int32_t max_mem_intv;   /* default 20; 0 disables round 3 */
    int32_t max_occ;        /* -c, default 500 */
} bwams_seed_opt_t;

/* Extension options: the BandedPairWiseSW constructor arguments
 * (src/bandedSWA.cpp:48-52).  mat is the 5x5 substitution matrix filled by
 * bwa_fill_scmat (src/bwa.cpp) from a (match) and b (mismatch). */
typedef struct bwams_sw_opt {
    int32_t o_del, e_del, o_ins, e_ins;
    int32_t zdrop;
    int32_t end_bonus;
    int8_t  mat[25];
    int8_t  pad_[3];
} bwams_sw_opt_t;

/* EMF seed table entry and probe result: the reference's seed_entry_t (src/perfect.h:93-108)
 * and bseq1_perfect_t (src/perfect.h:153-160: flags bit0 valid, bit1 reverse-complement match,
 * bits 2.. index of the multi-location list). */
typedef struct bwams_seed_entry {
    uint32_t flags, location, left, right;
} bwams_seed_entry_t;
typedef struct bwams_perfect {
    uint32_t flags, location;
} bwams_perfect_t;

/* Result of the local Smith-Waterman of mate rescue: the reference's kswr_t
 * (src/ksw.h:43-48).  Unset values are -1. */
typedef struct bwams_kswr {
    int32_t score;          /* best score */
    int32_t te, qe;         /* target / query end (inclusive) */
    int32_t score2, te2;    /* second best score and its target end */
    int32_t tb, qb;         /* target / query start */
} bwams_kswr_t;

/* ---- chaining and chain-to-alignment (mem_chain_seeds .. mem_chain2aln_across_reads_V2) ---- */

/* One reference sequence: the fields of bntann1_t (src/bntseq.h) the hot path reads. */
typedef struct bwams_contig {
    int64_t offset;         /* start on the forward strand */
    int32_t len;
    int32_t is_alt;
} bwams_contig_t;

/* The subset of mem_opt_t (src/bwamem.h:89-124) read by chaining, chain filtering, task
 * construction and the post-extension bookkeeping.  Defaults: src/bwamem.cpp:135-171. */
typedef struct bwams_mem_opt {
    int32_t a;                          /* match score, default 1 */
    int32_t o_del, e_del, o_ins, e_ins; /* 6, 1, 6, 1 */
    int32_t pen_clip5, pen_clip3;       /* 5, 5 */
    int32_t w;                          /* band width, 100 */
    int32_t zdrop;                      /* 100 */
    int32_t min_seed_len;               /* 19 */
    int32_t min_chain_weight;           /* 0 */
    int32_t max_chain_extend;           /* 1<<30 */
    int32_t max_occ;                    /* 500 */
    int32_t max_chain_gap;              /* 10000 */
    float   mask_level;                 /* 0.50 */
    float   drop_ratio;                 /* 0.50 */
    int8_t  mat[25];
    int8_t  pad_[3];
    /* Not a mem_opt_t field.  0 (default): a seed that an earlier kept region already explains is
     * purged WITHOUT being extended — the reference extends it first and discards the result
     * (mem_kernel2_core drops purged regions, src/bwamem.cpp:1446-1456), so every kept region and
     * every purge decision is unchanged; only the contents of purged regions differ.
     * 1: extend every seed, purged regions hold the reference's dead values too. */
    int32_t extend_all;
    float   mask_level_redun;           /* mem_opt_t again: 0.95, read by mem_sort_dedup_patch */
    int32_t max_ins;                    /* 10000, read by mem_pestat */
    int32_t b;                          /* mismatch penalty, 4: mem_mark_primary_se and mem_pair read a + b */
    int32_t pen_unpaired;               /* 17: anchors of mate rescue score >= best - pen_unpaired */
    int32_t max_matesw;                 /* 50: anchors per end that mate rescue tries */
    int32_t mapq_coef_len;              /* mapQ_coef_len, 50 (mapQ_coef_fac = log(mapQ_coef_len)): read by mem_approx_mapq_se */
} bwams_mem_opt_t;

/* mem_pestat_t (src/bwamem.h:178-182), same layout. */
typedef struct bwams_pestat {
    int32_t low, high;      /* bounds within which a pair counts as properly paired */
    int32_t failed;         /* non-zero: orientation not supported by enough data */
    int32_t pad_;
    double  avg, std;
} bwams_pestat_t;

/* The pairing decision for one read pair: what mem_sam_pe_batch_post (src/bwamem_pair.cpp:981-1098) has in hand
 * after mate rescue, mem_mark_primary_se of both ends and mem_pair (:366-427).  z[] / sub / n_sub are those of
 * mem_pair and are meaningful only when score > 0 (otherwise the reference leaves them unset; here 0 / -1). */
typedef struct bwams_pair {
    int32_t score;          /* mem_pair's return value o; 0: no proper pair (or an end without a primary hit) */
    int32_t sub, n_sub;     /* second best pairing score, pairings within max(a+b, o_del+e_del, o_ins+e_ins) of it */
    int32_t z[2];           /* index of the paired region in each end's list, -1 when unset */
    int32_t n_pri[2];       /* mem_mark_primary_se's return value of each end (regions on the primary assembly) */
    int32_t n_matesw;       /* mem_matesw's return values summed: rescue alignments this pair consumed */
} bwams_pair_t;

/* mem_t (src/ertseeding.h:62-78), 56 B, same field offsets: one maximal exact match found by the reference's ERT walk
 * and the slice [hitbeg, hitbeg + hitcount) of the read's hit array that belongs to it. */
typedef struct bwams_ert_mem {
    uint8_t forward;        /* found by forward search (hits are reference coordinates as they are) */
    uint8_t pad_[3];
    int32_t start, end;     /* [start, end) in the read */
    int32_t rc_start, rc_end;
    int32_t skip_ref_fetch;
    int32_t fetch_leaves;   /* hits were gathered from the leaves: coordinates as they are, like forward */
    int32_t hitbeg, hitcount;
    int32_t end_correction; /* backward search: bases by which the match ran past its start position */
    int32_t is_multi_hit;
    int32_t c_pivot, p_pivot, pp_pivot;     /* pivot_t */
} bwams_ert_mem_t;

/* mem_seed_t (src/bwamem.h:129-140), 32 B, same field offsets. */
typedef struct bwams_chain_seed {
    int64_t rbeg;
    int32_t qbeg, len, score;
    int8_t  done;
    int8_t  pad0_[3];
    int32_t aln;            /* index of this seed's region within its read's region list */
    int32_t pad1_;
} bwams_chain_seed_t;

/* mem_chain_t (src/bwamem.h:142-149), 48 B, same field offsets; the seeds pointer is replaced
 * by the index of the chain's first seed in the flat seed array (seeds of a chain are contiguous,
 * in the order mem_chain_seeds appended them). */
typedef struct bwams_chain {
    int32_t  seqid, cseed;
    int32_t  n, m, first, rid;
    uint32_t w_kept_alt;    /* w: bits 0-28, kept: bits 29-30, is_alt: bit 31 */
    float    frac_rep;
    int64_t  pos;
    int64_t  seed_off;
} bwams_chain_t;
#define BWAMS_CHAIN_W(c)      ((c).w_kept_alt & 0x1fffffffu)
#define BWAMS_CHAIN_KEPT(c)   (((c).w_kept_alt >> 29) & 3u)
#define BWAMS_CHAIN_IS_ALT(c) ((c).w_kept_alt >> 31)

/* mem_alnreg_t (src/bwamem.h:153-174), 112 B, same field offsets; the chain pointer is replaced
 * by the index of the chain in the flat chain array.  Fields the extension stage does not set
 * are zero, as after the reference's memset. */
typedef struct bwams_alnreg {
    int64_t  rb, re;
    int32_t  qb, qe;
    int32_t  rid;
    int32_t  pad0_;
    int64_t  chain;
    int32_t  score, truesc, sub, alt_sc, csub, sub_n;
    int32_t  w, seedcov, secondary, secondary_all, seedlen0;
    int32_t  n_comp_is_alt;
    float    frac_rep;
    int32_t  pad1_;
    uint64_t hash;
    int32_t  flg;
    int32_t  pad2_;
} bwams_alnreg_t;

/* mem_aln_t (src/bwamem.h:184-194) as mem_reg2aln fills it, with the bit fields widened and the CIGAR / MD pointer replaced
 * by offsets into the flat CIGAR (uint32 opLen << 4 | op, op: MIDSH = 01234) and MD (NUL-terminated strings) pools. */
typedef struct bwams_aln {
    int64_t  pos;            /* forward-strand 5'-end position within the sequence */
    int32_t  rid;            /* sequence index; < 0 for the unmapped record */
    int32_t  flag;           /* 0x4 unmapped, 0x100 secondary */
    int32_t  is_rev, is_alt, mapq, NM;
    int32_t  n_cigar, md_len;        /* md_len counts the terminating NUL */
    int64_t  cigar_off, md_off;
    int32_t  score, sub, alt_sc;
    int32_t  pad_;
} bwams_aln_t;

/* What mem_reg2sam / mem_gen_alt / mem_aln2sam read beyond bwams_mem_opt_t: fields of mem_opt_t (src/bwamem.h:89-124,
 * defaults src/bwamem.cpp:135-171) and the read-group id (bwa_rg_id, src/bwa.cpp). */
#define BWAMS_MEM_F_NOPAIRING      0x4      /* bwams_pair_run_sam, mem_sam_pe's proper-pair flag of unpaired ends */
#define BWAMS_MEM_F_ALL            0x8
#define BWAMS_MEM_F_NO_MULTI       0x10
#define BWAMS_MEM_F_NO_RESCUE      0x20     /* bwams_pair_run_sam */
#define BWAMS_MEM_F_REF_HDR        0x100    /* XR:Z:<annotation> on every record (bwams_index_set_contig_annos first) */
#define BWAMS_MEM_F_SOFTCLIP       0x200
#define BWAMS_MEM_F_PRIMARY5       0x800    /* bwams_pair_run_sam (the text itself does not read it) */
#define BWAMS_MEM_F_KEEP_SUPP_MAPQ 0x1000
typedef struct bwams_sam_opt {
    int32_t T;                  /* minimum score to output, 30 */
    int32_t flag;               /* BWAMS_MEM_F_* (the reference's MEM_F_* values) */
    float   XA_drop_ratio;      /* 0.80 */
    int32_t max_XA_hits;        /* 5 */
    int32_t max_XA_hits_alt;    /* 200 */
    char    rg_id[256];         /* "" = no RG:Z tag */
} bwams_sam_opt_t;

#ifdef __cplusplus
}
#endif

#endif /* BWAMS_TYPES_H */
